#!/usr/bin/env python3
"""MFMA share kernel: LDS-fed (mfma_areg=0) vs register-resident V fragments (mfma_areg=1), run on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402,F401
import scl_amd as scl  # noqa: E402


def timeit(fn, reps=5):
    fn()
    tm = scl.Timer()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


f = 0
for n, t, N in ((128, 42, 10_000_000), (100, 33, 10_000_000), (128, 20, 10_000_000), (128, 48, 5_000_000)):
    secrets = scl.vector_random(f, N, b"p")
    coeffs = scl.empty(f, t, N)
    for k in range(t):
        coeffs[k].copy_(scl.vector_random(f, N, b"c", counter0=k * ((N * 8 + 15) // 16)))
    shares = scl.empty(f, n, N)
    ref = None
    scl.set_tuning("mfma", 1)
    for areg, tpb in ((0, 0), (1, 0), (1, 256)):
        scl.set_tuning("mfma_areg", areg)
        scl.set_tuning("mfma_tpb", tpb)
        ms = timeit(lambda: scl.shamir_share(f, secrets, coeffs, n, out=shares))
        chk = int(shares.view(-1)[:: max(1, shares.numel() // 1000003)].sum().item())
        ref = chk if ref is None else ref
        print(f"n={n} t={t} N={N} areg={areg} tpb={tpb or 512}: {ms:8.3f} ms  {((1 + t) + n) * 8 * N / ms / 1e6:6.0f} GB/s  {N / ms / 1e6:6.2f} Gsec/s  "
              f"{'same' if chk == ref else 'DIFFERENT'}", flush=True)
    scl.set_tuning("mfma_areg", 1)
    scl.set_tuning("mfma_tpb", 0)
    scl.set_tuning("mfma", 0)
