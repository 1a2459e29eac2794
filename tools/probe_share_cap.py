#!/usr/bin/env python3
"""The Mersenne61 (10,3) share and reconstruct kernels on two operand sets in plain allocations, with and without the
residency cap ("share_waves" / "stream_waves" 0 or 8): HIP events around 10 launches each."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

f, n, t, N = scl.M61, 10, 3, 100_000_000
sets = []
for a in range(3):
    secrets = scl.vector_random(f, N, b"cap-s%d" % a)
    coeffs = scl.empty(f, t, N)
    per_row = (N * 8 + 15) // 16
    for k in range(t):
        scl.vector_random(f, N, b"cap-c%d" % a, counter0=k * per_row, out=coeffs[k])
    sets.append((secrets, coeffs, scl.empty(f, n, N), scl.empty(f, N)))
lam = scl.lagrange_basis(f, n)
tm = scl.Timer()


def timed(fn):
    fn()
    torch.cuda.synchronize()
    tm.start()
    for _ in range(10):
        fn()
    tm.stop()
    return tm.elapsed_ms() / 10


caps = [int(v) for v in sys.argv[1:] if not v.startswith("-")] or [0, 8]
share_fixed = "--share-8" in sys.argv      # sweep the reconstruct kernel's cap only
rec_fixed = "--rec-10" in sys.argv         # sweep the share kernel's cap only
for rnd in range(2):
    for sw in caps:
        scl.set_tuning("share_waves", 8 if share_fixed else sw)
        scl.set_tuning("stream_waves", 10 if rec_fixed else sw)
        row = []
        for secrets, coeffs, shares, out in sets:
            s_ms = timed(lambda: scl.shamir_share(f, secrets, coeffs, n, out=shares))
            r_ms = timed(lambda: scl.shamir_recover(f, shares, lam, out=out))
            assert scl.equals(f, out, secrets)
            row.append(f"{s_ms:6.3f} + {r_ms:6.3f} = {s_ms + r_ms:6.3f}")
        print(f"round {rnd} waves-per-CU cap {sw}:  share + reconstruct ms on sets A | B | C:  " + "  |  ".join(row), flush=True)
