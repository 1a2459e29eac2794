#!/usr/bin/env python3
"""C3's shape, Shamir (10,3) over the 128-bit fields at 10^7 secrets: the share and reconstruct kernels under the residency
cap settings ("stream_waves" = resident waves per CU of the reconstruct kernel, 0 = no cap; "stream_block" = its workgroup
size): HIP events around 20 launches each, two operand sets."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

n, t, N = 10, 3, 10_000_000
tm = scl.Timer()


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


for f in (scl.M127, scl.MONT128, scl.M61):
    E = 8 * scl.limbs(f)
    sets = []
    for a in range(2):
        secrets = scl.vector_random(f, N, b"c3-s%d" % a)
        coeffs = scl.vector_random(f, t * N, b"c3-c%d" % a).reshape(t, N, -1)
        sets.append((secrets, coeffs, scl.empty(f, n, N), scl.empty(f, N)))
    lam = scl.lagrange_basis(f, n)
    for blk in (64, 256):
        for sw in (0, 8, 12, 16, 24):
            if blk == 256 and sw not in (0, 8):
                continue
            scl.set_tuning("stream_block", blk)
            scl.set_tuning("stream_waves", sw)
            row = []
            for secrets, coeffs, shares, out in sets:
                s_ms = timed(lambda: scl.shamir_share(f, secrets, coeffs, n, out=shares))
                r_ms = timed(lambda: scl.shamir_recover(f, shares, lam, out=out))
                assert scl.equals(f, out, secrets)
                row.append(f"share {s_ms:6.3f} ms {(1 + t + n) * E * N / s_ms / 1e6:5.0f} GB/s  rec {r_ms:6.3f} ms {(n + 1) * E * N / r_ms / 1e6:5.0f} GB/s")
            print(f"{scl.field_name(f):12s} block {blk:3d} waves/CU cap {sw:2d}:  " + "  |  ".join(row), flush=True)
    scl.set_tuning("stream_block", 64)
    scl.set_tuning("stream_waves", -1)
