#!/usr/bin/env python3
"""Berlekamp-Welch batch (scl_hip_shamir_recover_correct) throughput on the GPU box: clean shares, 1 % and 100 % of
the secrets with one corrupted share.  (n, t) = (10, 3) and (40, 13)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

for f, name in ((0, "Mersenne61"), (1, "Mersenne127"), (4, "secp256k1_order")):
    for case, (n, t, N) in enumerate(((10, 3, 4_000_000), (40, 13, 200_000), (10, 3, 40_000_000), (40, 13, 4_000_000))):
        if f == 4:
            N //= 8
        big = case >= 2  # large clean batches only: the rate of the consistency pass
        secrets = scl.vector_random(f, N, b"bw")
        shares = scl.shamir_share_prg(f, secrets, t, n, b"bw-c")
        for frac in ((0.0,) if big else (0.0, 0.01, 1.0)):
            sh = shares.clone()
            k = int(N * frac)
            if k:
                idx = torch.randperm(N, device="cuda")[:k]
                sh[1, idx] = sh[2, idx]          # party 1's share replaced: one error per chosen secret
            if frac < 1.0:
                scl.shamir_recover_correct(f, sh)  # warm-up: code load, scratch allocation
            torch.cuda.synchronize()
            reps = 3 if frac < 1.0 else 1
            t0 = time.perf_counter()
            for _ in range(reps):
                r = scl.shamir_recover_correct(f, sh)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            ok = bool(scl.equals(f, r["f"][0], secrets)) and r["failed"] == 0
            print(f"{name:16s} n={n:2d} t={t:2d} N={N:8d} corrupted={frac:5.0%}: {dt * 1e3:9.2f} ms  {N / dt / 1e6:9.2f} M secrets/s  "
                  f"solver ran on {r['queued']:8d}  all corrected: {ok}", flush=True)
