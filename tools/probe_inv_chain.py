#!/usr/bin/env python3
"""The chained inversion of scl_hip_ew (k_ew_inv_rolled) by chain length ("inv_batch" knob) and batch size, per field: which
length the automatic choice should take where.  HIP events around repeated calls after a warm-up."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402


def timed(fn, warm=20, reps=20):
    for _ in range(warm):
        fn()
    tm = scl.Timer()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


for name, f in (("Mersenne127", scl.M127), ("Mont128", scl.MONT128), ("secp256k1", scl.SECP256K1_SCALAR), ("GF(2^128)", scl.GF2_128)):
    for n in (300_000, 700_000, 1_000_000, 2_000_000, 3_000_000, 5_000_000, 7_000_000, 10_000_000, 20_000_000, 30_000_000, 100_000_000):
        a = scl.vector_random(f, n, b"pic")
        out = scl.empty(f, n)
        row = []
        for L in (8, 16, 32, 64, 128, 0):
            scl.set_tuning("inv_batch", L)
            try:
                def call():
                    try:
                        scl.ew(f, scl.INV, a, out=out)
                    except scl.SclError:
                        pass
                ms = timed(call, 5 if n >= 10_000_000 else 20, 10 if n >= 10_000_000 else 20)
            finally:
                scl.set_tuning("inv_batch", 0)
            row.append((L, ms))
        best = min(row[:-1], key=lambda r: r[1])
        print(f"{name:12s} n {n:9d}  " + "  ".join(f"{'auto' if L == 0 else L}: {ms:.4f}" for L, ms in row) + f"   best {best[0]} (auto / best = {row[-1][1] / best[1]:.2f})", flush=True)
        del a, out
