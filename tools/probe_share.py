#!/usr/bin/env python3
"""Kernel scaling probes (run on the GPU box): GB/s of the share kernel vs threshold t
(compute per byte), plus the other streaming kernels, for one field."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402,F401
import scl_amd as scl  # noqa: E402

f = int(os.environ.get("FIELD", "0"))
N = int(os.environ.get("SECRETS", "100000000"))
L = scl.limbs(f)
E = 8 * L
secrets = scl.vector_random(f, N, b"p")


def timeit(fn, reps=10):
    fn()
    tm = scl.Timer()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


n = int(os.environ.get("NPARTIES", "10"))
shares = scl.empty(f, n, N)
print(f"field={scl.field_name(f)} N={N}")
for t in (0, 1, 2, 3, 4):
    coeffs = scl.empty(f, max(t, 1), N)
    for k in range(t):
        coeffs[k].copy_(scl.vector_random(f, N, b"c", counter0=k * ((N * E + 15) // 16)))
    ms = timeit(lambda: scl.shamir_share(f, secrets, coeffs[:t] if t else None, n, out=shares))
    print(f"share n={n} t={t}: {ms:.3f} ms  {((1 + t) + n) * E * N / ms / 1e6:.0f} GB/s  {N / ms / 1e6:.1f} Gsecrets/s")
ms = timeit(lambda: scl.shamir_share_prg(f, secrets, 3, n, b"seed", out=shares), reps=3)
print(f"share_prg n={n} t=3: {ms:.3f} ms  {(1 + n) * E * N / ms / 1e6:.0f} GB/s  {N / ms / 1e6:.1f} Gsecrets/s")
out = scl.empty(f, N)
ms = timeit(lambda: scl.shamir_recover(f, shares, out=out))
print(f"recover n={n}: {(n + 1) * E * N / ms / 1e6:.0f} GB/s")
ms = timeit(lambda: scl.additive_recover(f, shares, out=out))
print(f"additive_recover n={n}: {(n + 1) * E * N / ms / 1e6:.0f} GB/s")
ms = timeit(lambda: scl.ew(f, 2, shares[0], shares[1], out=shares[2]))
print(f"ew mul: {3 * E * N / ms / 1e6:.0f} GB/s")
ms = timeit(lambda: scl.ew(f, 0, shares[0], shares[1], out=shares[2]))
print(f"ew add: {3 * E * N / ms / 1e6:.0f} GB/s")
ms = timeit(lambda: scl.ew(f, 4, shares[0], None, out=shares[2]), reps=3)
print(f"ew inv: {2 * E * N / ms / 1e6:.0f} GB/s  {N / ms / 1e6:.2f} Ginv/s")
ms = timeit(lambda: scl.vector_random(f, N, b"x"), reps=3)
print(f"vector_random: {N * E / 16 / ms / 1e6:.2f} G AES blocks/s")
ms = timeit(lambda: scl.prg_blocks(N // 2, b"x"), reps=3)
print(f"prg_blocks: {N / 2 / ms / 1e6:.2f} G AES blocks/s  {N * 8 / ms / 1e6:.0f} GB/s written")
ms = timeit(lambda: scl.additive_share_prg(f, secrets, 3, b"x", out=shares[:3]), reps=3)
print(f"additive_share_prg n=3: {N / ms / 1e6:.2f} Gsecrets/s")
ms = timeit(lambda: shares.zero_())
print(f"torch zero_ {shares.numel() * 8 / ms / 1e6:.0f} GB/s write-only")
ms = timeit(lambda: scl.vsum(f, shares))
print(f"vsum (read-only): {shares.numel() * 8 / ms / 1e6:.0f} GB/s")
