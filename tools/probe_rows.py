#!/usr/bin/env python3
"""Throughput of every entry point DESIGN.md section 3 lists without a number (run on the GPU box):
error-detecting recovery, matmul, wire / frame pack + unpack, AoS<->SoA, scalar multiply, equals,
sum / dot, the Z2k ring kernels, additive sharing.  Algorithmic bytes per launch / HIP-event time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch, scl_amd as scl


def timeit(fn, reps=5):
    fn(); fn()
    tm = scl.Timer(); tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


def line(name, ms, nbytes, units=None, unit_name="elements"):
    extra = f"  {units / ms / 1e6:.2f} G{unit_name}/s" if units else ""
    print(f"{name:<58} {ms:8.3f} ms  {nbytes / ms / 1e6:7.0f} GB/s{extra}", flush=True)


only = set(sys.argv[1:])


def want(k):
    return not only or k in only


FIELDS = [(scl.M61, "M61"), (scl.M127, "M127"), (scl.SECP256K1_SCALAR, "secp256k1")]

if want("detect"):
    for f, fn in FIELDS:
        E = 8 * scl.limbs(f)
        for t, N in ((3, 50_000_000 * 8 // E), (13, 20_000_000 * 8 // E), (42, 4_000_000 * 8 // E)):
            n = 2 * t + 1
            if f == scl.SECP256K1_SCALAR and t > 13:
                continue
            secrets = scl.vector_random(f, N, b"s")
            sh = scl.shamir_share_prg(f, secrets, t, n, b"seed")
            ms = timeit(lambda: scl.shamir_recover_detect(f, sh, t))
            _, _, bad = scl.shamir_recover_detect(f, sh, t)
            assert bad == 0
            line(f"{fn} recover_detect n={n} t={t} N={N}", ms, (n + 1) * E * N + N, N, "secrets")
            del sh, secrets
            torch.cuda.empty_cache()

if want("layout"):
    for f, fn in FIELDS[:2]:
        E = 8 * scl.limbs(f)
        for n in (3, 10, 40, 128):
            N = 1_600_000_000 // (n * E)
            soa = scl.vector_random(f, n * N, b"x").reshape(n, N, -1)
            ms = timeit(lambda: scl.soa_to_aos(f, soa))
            line(f"{fn} soa_to_aos n={n} N={N}", ms, 2 * n * N * E)
            aos = scl.soa_to_aos(f, soa)
            ms = timeit(lambda: scl.aos_to_soa(f, aos))
            line(f"{fn} aos_to_soa n={n} N={N}", ms, 2 * n * N * E)
            del soa, aos
            torch.cuda.empty_cache()

if want("wire"):
    for f, fn in FIELDS:
        E = 8 * scl.limbs(f)
        N = 800_000_000 // E
        a = scl.vector_random(f, N, b"w")
        ms = timeit(lambda: scl.wire_pack(f, a))
        line(f"{fn} wire_pack N={N}", ms, 2 * N * E)
        raw = scl.wire_pack(f, a)
        ms = timeit(lambda: scl.wire_unpack(f, raw))
        line(f"{fn} wire_unpack N={N}", ms, 2 * N * E)
        ms = timeit(lambda: scl.frame_pack(f, a))
        line(f"{fn} frame_pack N={N}", ms, 2 * N * E)
        fr = scl.frame_pack(f, a)
        ms = timeit(lambda: scl.frame_unpack(f, fr))
        line(f"{fn} frame_unpack N={N}", ms, 2 * N * E)
        rows = 10
        m = a[: (N // rows) * rows].reshape(rows, N // rows, -1)
        ms = timeit(lambda: scl.wire_pack_matrix(f, m))
        line(f"{fn} wire_pack_matrix {rows}x{N // rows}", ms, 2 * N * E)
        rm = scl.wire_pack_matrix(f, m)
        ms = timeit(lambda: scl.wire_unpack_matrix(f, rm, (rows, N // rows)))
        line(f"{fn} wire_unpack_matrix {rows}x{N // rows}", ms, 2 * N * E)
        del a, raw, fr, m, rm
        torch.cuda.empty_cache()

if want("vector"):
    for f, fn in FIELDS + [(scl.Z2K(64), "Z2k<64>"), (scl.Z2K(128), "Z2k<128>"), (scl.Z2K(37), "Z2k<37>")]:
        E = 8 * scl.limbs(f)
        N = 800_000_000 // E
        a = scl.vector_random(f, N, b"a")
        b = scl.vector_random(f, N, b"b")
        o = torch.empty_like(a)
        for op, on in ((scl.ADD, "add"), (scl.MUL, "mul")):
            ms = timeit(lambda: scl.ew(f, op, a, b, out=o))
            line(f"{fn} ew {on} N={N}", ms, 3 * N * E)
        ms = timeit(lambda: scl.scalar_mul(f, a, scl.to_host(b[:1]), out=o))
        line(f"{fn} scalar_mul N={N}", ms, 2 * N * E)
        ms = timeit(lambda: scl.vsum(f, a))
        line(f"{fn} sum N={N}", ms, N * E)
        ms = timeit(lambda: scl.dot(f, a, b))
        line(f"{fn} dot N={N}", ms, 2 * N * E)
        ms = timeit(lambda: scl.equals(f, a, a))
        line(f"{fn} equals N={N}", ms, 2 * N * E)
        del a, b, o
        torch.cuda.empty_cache()

if want("additive"):
    for f, fn in FIELDS[:2] + [(scl.Z2K(64), "Z2k<64>")]:
        E = 8 * scl.limbs(f)
        for n in (3, 10):
            N = 1_600_000_000 // (n * E)
            s = scl.vector_random(f, N, b"s")
            rnd = scl.vector_random(f, (n - 1) * N, b"r").reshape(n - 1, N, -1)
            out = scl.empty(f, n, N)
            ms = timeit(lambda: scl.additive_share(f, s, rnd, n, out=out))
            line(f"{fn} additive_share n={n} N={N}", ms, 2 * n * N * E, N, "secrets")
            ms = timeit(lambda: scl.additive_share_prg(f, s, n, b"seed", out=out))
            line(f"{fn} additive_share_prg n={n} N={N}", ms, (n + 1) * N * E, N, "secrets")
            r = torch.empty_like(s)
            ms = timeit(lambda: scl.additive_recover(f, out, out=r))
            line(f"{fn} additive_recover n={n} N={N}", ms, (n + 1) * N * E, N, "secrets")
            del s, rnd, out, r
            torch.cuda.empty_cache()

if want("matmul"):
    for f, fn in FIELDS[:2]:
        E = 8 * scl.limbs(f)
        for M, K, N in ((10, 4, 20_000_000), (40, 14, 4_000_000), (256, 256, 65536), (1024, 1024, 1024)):
            A = scl.vector_random(f, M * K, b"A").reshape(M, K, -1)
            B = scl.vector_random(f, K * N, b"B").reshape(K, N, -1)
            o = scl.empty(f, M, N)
            ms = timeit(lambda: scl.matmul(f, A, B, out=o), reps=3)
            line(f"{fn} matmul {M}x{K}x{N}", ms, (K * N + M * N) * E, M * K * N, "mul-adds")
            del A, B, o
            torch.cuda.empty_cache()
