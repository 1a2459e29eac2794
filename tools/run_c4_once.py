#!/usr/bin/env python3
"""C4's per-GPU shard once (GF(2^128), (40,13), 1.25e7 secrets: share from coefficients, reconstruct) and 2^27 AES blocks --
the workload of the PMC passes in profiles/r2_pmc_round_end.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

f, n, t, N = scl.GF2_128, 40, 13, 12_500_000
secrets = scl.vector_random(f, N, b"s")
coeffs = scl.vector_random(f, t * N, b"c").reshape(t, N, -1)
shares = scl.empty(f, n, N)
lam = scl.lagrange_basis(f, n)
for _ in range(2):
    scl.shamir_share(f, secrets, coeffs, n, out=shares)
    out = scl.shamir_recover(f, shares, lam)
assert scl.equals(f, out, secrets)
blocks = scl.prg_blocks(1 << 27, b"seed")
torch.cuda.synchronize()
