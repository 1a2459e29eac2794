#!/usr/bin/env python3
"""The rolled inversion (k_ew_inv_rolled: every prefix product through scratch memory, 5 E bytes per element) against the
two-level form (k_ew_inv_blocked: checkpoints + recomputed blocks, 3 E bytes and one more product per element) over Mersenne127
and Mont128, inverse and divide, by batch size and chain length -- same outputs word for word, HIP events around every launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402


def timed(fn, reps=30, warm=60):
    tms = [scl.Timer() for _ in range(reps)]
    for k in range(-warm, reps):
        if k >= 0:
            tms[k].start()
        fn()
        if k >= 0:
            tms[k].stop()
    torch.cuda.synchronize()
    ms = sorted(t.elapsed_ms() for t in tms)
    return sum(ms) / len(ms), ms[0]


status = scl.ew_status_buffer()
for f, name in ((scl.M127, "Mersenne127"), (scl.MONT128, "Mont128")):
    for N in (1_000_000, 3_000_000, 10_000_000, 30_000_000, 100_000_000):
        a = scl.vector_random(f, N, b"two-level-a")
        b = scl.vector_random(f, N, b"two-level-b")
        ref, out = scl.empty(f, N), scl.empty(f, N)
        for op, opn in ((scl.INV, "inv"), (scl.DIV, "div")):
            for chain in (0, 32, 64, 128, 256):
                row = []
                for two in (-1, 4, 8):
                    scl.set_tuning("inv_batch", chain)
                    scl.set_tuning("inv_two_level", two)
                    try:
                        tgt = ref if two < 0 else out
                        if two == 8 and not bool(scl.equals(f, ref, out)) and len(row) == 2:
                            print("MISMATCH blocks of 4", flush=True)
                        mean, best = timed(lambda: scl.ew_status(f, op, a if op == scl.INV else b, a if op == scl.DIV else None, status, out=tgt),
                                           reps=20 if N >= 30_000_000 else 30, warm=20 if N >= 30_000_000 else 60)
                        row.append((mean, best))
                    finally:
                        scl.set_tuning("inv_batch", 0)
                        scl.set_tuning("inv_two_level", 0)
                same = bool(scl.equals(f, ref, out))
                print(f"{name:12s} {opn} N {N:>11d} chain {chain or 'auto':>4}: rolled {row[0][0]:.4f} ms  blocks of 4 {row[1][0]:.4f} ms (x{row[0][0] / row[1][0]:.3f})  "
                      f"blocks of 8 {row[2][0]:.4f} ms (x{row[0][0] / row[2][0]:.3f})  identical {same}", flush=True)
        del a, b, ref, out
        torch.cuda.empty_cache()
