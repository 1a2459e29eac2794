#!/usr/bin/env python3
"""C3 read literally (Shamir (10,3) over the 128-bit Montgomery prime, 10^7 secrets): the small-node share kernel as the
256-thread kernel with the threshold at run time ("share_waves128" 0, what Mont128 ran until round 4) against the
single-wave kernel with the threshold compiled in under residency caps of 8..24 waves per CU; Mersenne127 beside it.
Fifty launches between two events, three operand sets."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

n, t, N = 10, 3, 10_000_000
tm = scl.Timer()


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


for f in (scl.MONT128, scl.M127):
    E = 8 * scl.limbs(f)
    sets = []
    for a in range(3):
        secrets = scl.vector_random(f, N, b"c3m-s%d" % a)
        coeffs = scl.vector_random(f, t * N, b"c3m-c%d" % a).reshape(t, N, -1)
        sets.append((secrets, coeffs, scl.empty(f, n, N), scl.empty(f, N)))
    lam = scl.lagrange_basis(f, n)
    ref = None
    for sw in (0, 8, 10, 12, 16, 20, 24):
        scl.set_tuning("share_waves128", sw)
        row = []
        for secrets, coeffs, shares, out in sets:
            s_ms = timed(lambda: scl.shamir_share(f, secrets, coeffs, n, out=shares))
            scl.shamir_recover(f, shares, lam, out=out)
            assert scl.equals(f, out, secrets)
            row.append(f"{s_ms:6.3f} ms {(1 + t + n) * E * N / s_ms / 1e6 / 8000:5.3f}")
        print(f"{scl.field_name(f):12s} share_waves128 {sw:2d}:  " + "  |  ".join(row), flush=True)
    scl.set_tuning("share_waves128", 12)
