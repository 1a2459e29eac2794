#!/usr/bin/env python3
"""The library's automatic path choices against the pinned alternatives (tuning knobs), over sizes away from the bench's: Mersenne61
matrix products (the "mfma" knob), the streaming share / reconstruct kernels' residency caps, the PRG-driven sharing's one or two
passes, the layout bridge's tile.  Prints auto / best per case: anything well above 1 is a threshold to re-measure."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402


def timed(fn, warm=10, reps=20):
    for _ in range(warm):
        fn()
    tm = scl.Timer()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


def sweep(label, fn, knob, values, default):
    row = []
    for v in list(values) + [None]:
        if v is not None:
            scl.set_tuning(knob, v)
        try:
            row.append((v, timed(fn)))
        except scl.SclError:
            row.append((v, float("inf")))
        finally:
            scl.set_tuning(knob, default)
    best = min(row[:-1], key=lambda r: r[1])
    print(f"{label:58s} {knob:14s} " + "  ".join(f"{'auto' if v is None else v}: {ms:.4f}" for v, ms in row) +
          f"   auto / best = {row[-1][1] / best[1]:.2f} (best {best[0]})", flush=True)


f = scl.M61
for (M, K, N) in [(64, 64, 64), (128, 128, 128), (256, 256, 256), (384, 384, 384), (512, 512, 512), (768, 768, 768), (1024, 1024, 1024),
                  (33, 4096, 33), (64, 8192, 64), (100, 100, 100000), (200, 65, 50000), (2048, 64, 2048), (40, 2000, 40), (300, 300, 3000)]:
    A = scl.vector_random(f, M * K, b"A").reshape(M, K, -1)
    B = scl.vector_random(f, K * N, b"B").reshape(K, N, -1)
    out = scl.empty(f, M, N)
    sweep(f"matmul {M} x {K} x {N}", lambda: scl.matmul(f, A, B, out=out), "mfma", (-1, 1, 2), 0)
for fld, name in ((scl.M61, "Mersenne61"), (scl.M127, "Mersenne127")):
    for N in (100_000, 1_000_000, 10_000_000):
        n, t = 10, 3
        secrets = scl.vector_random(fld, N, b"s")
        coeffs = scl.vector_random(fld, t * N, b"c").reshape(t, N, -1)
        shares = scl.empty(fld, n, N)
        lam = scl.lagrange_basis(fld, n)
        out = scl.empty(fld, N)
        sweep(f"share (10,3) {name} N {N}", lambda: scl.shamir_share(fld, secrets, coeffs, n, out=shares),
              "share_waves" if fld == scl.M61 else "share_waves128", (0, 6, 9, 12, 16), 9 if fld == scl.M61 else 12)
        sweep(f"reconstruct (10,3) {name} N {N}", lambda: scl.shamir_recover(fld, shares, lam, out=out), "stream_waves", (0, 6, 8, 10, 12, 16), -1)
        for (n2, t2) in ((10, 3), (10, 7), (20, 8), (40, 13)):
            sh2 = scl.empty(fld, n2, N)
            sweep(f"share_prg ({n2},{t2}) {name} N {N}", lambda: scl.shamir_share_prg(fld, secrets, t2, n2, b"seed", out=sh2), "prg_two_pass", (-1, 1), 0)
            del sh2
        del secrets, coeffs, shares, out
for fld, name in ((scl.M61, "Mersenne61"), (scl.M127, "Mersenne127"), (scl.SECP256K1_SCALAR, "secp256k1")):
    for (N, n) in ((100_000, 10), (1_000_000, 10), (1_000_000, 3), (1_000_000, 40), (200_000, 128)):
        soa = scl.vector_random(fld, N * n, b"l").reshape(n, N, -1)
        sweep(f"soa_to_aos {name} N {N} n {n}", lambda: scl.soa_to_aos(fld, soa), "transpose_tile", (64, 128, 256, 512), 0)
        del soa

# explicit-coefficient sharing and reconstruction at larger (n, t): matrix cores or vector ALU ("mfma"), the Horner forms ("force_table")
for (n, t) in ((16, 5), (20, 6), (32, 10), (40, 13), (64, 21), (100, 33), (128, 42)):
    for N in (200_000, 2_000_000):
        secrets = scl.vector_random(scl.M61, N, b"s")
        coeffs = scl.vector_random(scl.M61, t * N, b"c").reshape(t, N, -1)
        shares = scl.empty(scl.M61, n, N)
        lam = scl.lagrange_basis(scl.M61, n)
        out = scl.empty(scl.M61, N)
        sweep(f"share ({n},{t}) Mersenne61 N {N}", lambda: scl.shamir_share(scl.M61, secrets, coeffs, n, out=shares), "mfma", (-1, 1), 0)
        sweep(f"share ({n},{t}) Mersenne61 N {N}", lambda: scl.shamir_share(scl.M61, secrets, coeffs, n, out=shares), "force_table", (1, 2), 0)
        sweep(f"reconstruct ({n}) Mersenne61 N {N}", lambda: scl.shamir_recover(scl.M61, shares, lam, out=out), "stream_block", (64, 256), 64)
        del secrets, coeffs, shares, out
# the element-wise inverse at small sizes: one chain per lane or one Fermat chain per element ("inv_batch" -1)
for fld, name in ((scl.M61, "Mersenne61"), (scl.M127, "Mersenne127"), (scl.SECP256K1_SCALAR, "secp256k1"), (scl.GF2_128, "GF(2^128)")):
    for n in (1_000, 10_000, 50_000, 100_000, 300_000):
        a = scl.vector_random(fld, n, b"i")
        out = scl.empty(fld, n)

        def call():
            try:
                scl.ew(fld, scl.INV, a, out=out)
            except scl.SclError:
                pass
        sweep(f"inverse {name} n {n}", call, "inv_batch", (-1, 8, 16, 32), 0)
        del a, out

# PRG-driven sharing over GF(2^128) by threshold: fused or two passes
fld = scl.GF2_128
for N in (200_000, 2_000_000):
    secrets = scl.vector_random(fld, N, b"s")
    for n in (10, 20, 40):
        for t in (3, 4, 5, 6, 8, 11, 12):
            if t < n:
                sh = scl.empty(fld, n, N)
                sweep(f"share_prg ({n},{t}) GF(2^128) N {N}", lambda: scl.shamir_share_prg(fld, secrets, t, n, b"seed", out=sh), "prg_two_pass", (-1, 1), 0)
                del sh
# explicit-coefficient sharing over Mersenne61 around 48 coefficients: the matrix cores take up to 63
N = 2_000_000
secrets = scl.vector_random(scl.M61, N, b"s")
for n in (64, 128):
    for t in (42, 48, 49, 56, 63):
        if t < n:
            coeffs = scl.vector_random(scl.M61, t * N, b"c").reshape(t, N, -1)
            sh = scl.empty(scl.M61, n, N)
            sweep(f"share ({n},{t}) Mersenne61 N {N}", lambda: scl.shamir_share(scl.M61, secrets, coeffs, n, out=sh), "mfma", (-1, 1), 0)
            del coeffs, sh
# the Montgomery fields and GF(2^128): passes of the PRG-driven sharing, Horner forms, small-integer reconstruct, matrix-core variants
for fld, name in ((scl.MONT128, "Mont128"), (scl.SECP256K1_SCALAR, "secp256k1"), (scl.GF2_128, "GF(2^128)")):
    N = 2_000_000
    secrets = scl.vector_random(fld, N, b"s")
    for (n, t) in ((10, 3), (10, 7), (20, 11), (40, 13)):
        sh = scl.empty(fld, n, N)
        if fld != scl.GF2_128:
            sweep(f"share_prg ({n},{t}) {name} N {N}", lambda: scl.shamir_share_prg(fld, secrets, t, n, b"seed", out=sh), "prg_two_pass", (-1, 1), 0)
        coeffs = scl.vector_random(fld, t * N, b"c").reshape(t, N, -1)
        sweep(f"share ({n},{t}) {name} N {N}", lambda: scl.shamir_share(fld, secrets, coeffs, n, out=sh), "force_table", (1, 2), 0)
        lam = scl.lagrange_basis(fld, n)
        out = scl.empty(fld, N)
        sweep(f"reconstruct ({n}) {name} N {N}", lambda: scl.shamir_recover(fld, sh, lam, out=out), "force_table", (1,), 0)
        del sh, coeffs, out
for (n, t) in ((64, 21), (128, 42)):
    coeffs = scl.vector_random(scl.M61, t * N, b"c").reshape(t, N, -1)
    sh = scl.empty(scl.M61, n, N)
    secrets = scl.vector_random(scl.M61, N, b"s")
    sweep(f"share ({n},{t}) Mersenne61 N {N}", lambda: scl.shamir_share(scl.M61, secrets, coeffs, n, out=sh), "mfma_pipe", (0, 1, 2), 2)
    sweep(f"share ({n},{t}) Mersenne61 N {N}", lambda: scl.shamir_share(scl.M61, secrets, coeffs, n, out=sh), "mfma_areg", (0, 1), 1)
    del coeffs, sh

# the thread-per-column matrix kernels (thin inner dimension; left factor in LDS) against the tiled one ("matmul_lds_min": 1024 pins
# the former from 1024 columns on, 10^9 never takes them)
for fld, name in ((scl.M127, "Mersenne127"), (scl.SECP256K1_SCALAR, "secp256k1"), (scl.M61, "Mersenne61 (vector ALU)")):
    for (M, K) in ((10, 4), (40, 14), (20, 200)):
        for N in (1024, 10_000, 100_000, 1_000_000):
            A = scl.vector_random(fld, M * K, b"A").reshape(M, K, -1)
            B = scl.vector_random(fld, K * N, b"B").reshape(K, N, -1)
            out = scl.empty(fld, M, N)
            if fld == scl.M61:
                scl.set_tuning("mfma", -1)
            sweep(f"matmul {M} x {K} x {N} {name}", lambda: scl.matmul(fld, A, B, out=out), "matmul_lds_min", (1024, 10 ** 9), 0)
            scl.set_tuning("mfma", 0)
            del A, B, out
