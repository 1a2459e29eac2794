#!/usr/bin/env python3
"""C3's share kernels launch by launch (HIP events around each of 60 back-to-back launches, after 10 warm-up launches): where the
0.34-0.47 ms spread of bench.py's configs.C3_* comes from.  Fresh allocations twice, to see whether the slow launches follow
the buffer or the time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

n, t, N = 10, 3, 10_000_000
for f in (scl.MONT128, scl.M127):
    for rep in range(2):
        secrets = scl.vector_random(f, N, b"seq-s")
        coeffs = scl.vector_random(f, t * N, b"seq-c").reshape(t, N, -1)
        shares, out = scl.empty(f, n, N), scl.empty(f, N)
        lam = scl.lagrange_basis(f, n)
        for which in ("share", "rec"):
            fn = (lambda: scl.shamir_share(f, secrets, coeffs, n, out=shares)) if which == "share" else (lambda: scl.shamir_recover(f, shares, lam, out=out))
            ts = [scl.Timer() for _ in range(60)]
            for _ in range(10):
                fn()
            for tm in ts:
                tm.start()
                fn()
                tm.stop()
            torch.cuda.synchronize()
            ms = [tm.elapsed_ms() for tm in ts]
            srt = sorted(ms)
            print(f"{scl.field_name(f):12s} alloc {rep} {which:5s} median {srt[30]:.3f} p10 {srt[6]:.3f} p90 {srt[54]:.3f} max {srt[-1]:.3f} | " + " ".join(f"{x:.3f}" for x in ms), flush=True)
        del secrets, coeffs, shares, out
        torch.cuda.empty_cache()
