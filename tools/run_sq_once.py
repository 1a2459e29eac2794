#!/usr/bin/env python3
"""The kernels of the bench that HBM does not bound, three launches each at the bench's sizes -- the workload of
tools/sq_counters.sh: inverse over Mersenne61 (10^8), Mersenne127, Mont128, GF(2^128) (10^7), the GF(2^128) product (10^7),
C4's share and reconstruct kernels ((40,13), 1.25e7 secrets) and k_prg_blocks (2e8 blocks)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

status = scl.ew_status_buffer()
for f, N in ((scl.M61, 100_000_000), (scl.M127, 10_000_000), (scl.MONT128, 10_000_000), (scl.GF2_128, 10_000_000)):
    a = scl.vector_random(f, N, b"a")
    b = scl.vector_random(f, N, b"b")
    out = scl.empty(f, N)
    for _ in range(3):
        scl.ew_status(f, scl.INV, a, None, status, out=out)
        if f == scl.GF2_128:
            scl.ew(f, scl.MUL, a, b, out=out)
    del a, b, out
f, n, t, N = scl.GF2_128, 40, 13, 12_500_000
secrets = scl.vector_random(f, N, b"s")
coeffs = scl.vector_random(f, t * N, b"c").reshape(t, N, 2)
shares = scl.empty(f, n, N)
out = scl.empty(f, N)
lam = scl.lagrange_basis(f, n)
for _ in range(3):
    scl.shamir_share(f, secrets, coeffs, n, out=shares)
    scl.shamir_recover(f, shares, lam, out=out)
del secrets, coeffs, shares, out
blocks = scl.prg_blocks(200_000_000, b"seed")
for _ in range(3):
    scl.prg_blocks(200_000_000, b"seed", out=blocks)
torch.cuda.synchronize()
