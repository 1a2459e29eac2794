#!/usr/bin/env python3
"""Mersenne61 inverse over 10^8 elements, launch after launch: scl_hip_ew (host-synchronous: the stream drains between launches)
against scl_hip_ew_status (nothing waits: the launches queue back to back), HIP events around every launch.  Round 6's first
bench run showed the back-to-back kernel at 0.36-0.46 ms where the synchronous call's kernel takes 0.31: this prints the
series, so that one can see whether the time drifts (clocks under a sustained vector-ALU load) or steps (placement / the
previous launch's dirty lines)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

f, N = scl.M61, 100_000_000
a = scl.vector_random(f, N, b"probe-inv-a")
out = scl.empty(f, N)
out2 = scl.empty(f, N)
status = scl.ew_status_buffer()


def series(fn, reps, gap_s=0.0):
    tms = [scl.Timer() for _ in range(reps)]
    for k in range(reps):
        tms[k].start()
        fn(k)
        tms[k].stop()
        if gap_s:
            torch.cuda.synchronize()
            time.sleep(gap_s)
    torch.cuda.synchronize()
    return [t.elapsed_ms() for t in tms]


def show(name, ms):
    chunks = [sum(ms[i:i + 20]) / len(ms[i:i + 20]) for i in range(0, len(ms), 20)]
    print(f"{name:58s} first 8: {' '.join(f'{x:.3f}' for x in ms[:8])} | means of 20: {' '.join(f'{x:.3f}' for x in chunks)}", flush=True)


for rnd in range(2):
    show("scl_hip_ew INV (synchronous call)", series(lambda k: scl.ew(f, scl.INV, a, None, out=out), 100))
    show("scl_hip_ew_status INV (back to back)", series(lambda k: scl.ew_status(f, scl.INV, a, None, status, out=out), 100))
    show("scl_hip_ew_status INV, alternating output buffers", series(lambda k: scl.ew_status(f, scl.INV, a, None, status, out=out if k % 2 else out2), 100))
    show("scl_hip_ew_status INV, synchronize after each", series(lambda k: (scl.ew_status(f, scl.INV, a, None, status, out=out), torch.cuda.synchronize()), 100))
    show("scl_hip_ew_status INV, 1 ms idle after each", series(lambda k: scl.ew_status(f, scl.INV, a, None, status, out=out), 60, gap_s=0.001))
    show("scl_hip_ew MUL (back to back, the streaming kernel)", series(lambda k: scl.ew(f, scl.MUL, a, a, out=out), 100))
assert int(status.item()) in (0, 1)
