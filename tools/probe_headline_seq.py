#!/usr/bin/env python3
"""The headline step (Shamir (10,3) Mersenne61, 10^8 secrets: share + reconstruct) launch by launch from a cold process:
HIP-event times of the first 40 steps, no warm-up -- is there a clock ramp inside bench.py's warm-up + timed region?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

f, n, t, N = scl.M61, 10, 3, 100_000_000
secrets = scl.vector_random(f, N, b"hs-s")
coeffs = scl.vector_random(f, t * N, b"hs-c").reshape(t, N, -1)
shares, out = scl.empty(f, n, N), scl.empty(f, N)
lam = scl.lagrange_basis(f, n)
K = 40
ts = [(scl.Timer(), scl.Timer()) for _ in range(K)]
for a, b in ts:
    a.start()
    scl.shamir_share(f, secrets, coeffs, n, out=shares)
    a.stop()
    b.start()
    scl.shamir_recover(f, shares, lam, out=out)
    b.stop()
torch.cuda.synchronize()
print("share", " ".join(f"{a.elapsed_ms():.3f}" for a, _ in ts))
print("rec  ", " ".join(f"{b.elapsed_ms():.3f}" for _, b in ts))
