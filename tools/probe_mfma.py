#!/usr/bin/env python3
"""share kernel: matrix-core path vs VALU Horner, Mersenne61 (run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch, scl_amd as scl
f = 0
def timeit(fn, reps=5):
    fn(); tm = scl.Timer(); tm.start()
    for _ in range(reps): fn()
    tm.stop(); return tm.elapsed_ms() / reps
for n, t, N in ((128, 42, 10_000_000), (64, 21, 10_000_000), (40, 13, 20_000_000), (32, 10, 20_000_000), (16, 8, 40_000_000), (10, 3, 50_000_000)):
    secrets = scl.vector_random(f, N, b"s")
    coeffs = scl.empty(f, t, N)
    for k in range(t):
        coeffs[k].copy_(scl.vector_random(f, N, b"c", counter0=k * ((N * 8 + 15) // 16)))
    shares = scl.empty(f, n, N)
    res = {}
    for mode in (-1, 1):
        scl.set_tuning("mfma", mode)
        ms = timeit(lambda: scl.shamir_share(f, secrets, coeffs, n, out=shares))
        res[mode] = ms
    scl.set_tuning("mfma", 0)
    b = (1 + t + n) * 8 * N
    print(f"n={n:3d} t={t:2d} N={N}: VALU {res[-1]:8.3f} ms {b/res[-1]/1e6:6.0f} GB/s {N/res[-1]/1e6:6.2f} Gsec/s | MFMA {res[1]:8.3f} ms {b/res[1]/1e6:6.0f} GB/s {N/res[1]/1e6:6.2f} Gsec/s")
    del secrets, coeffs, shares
