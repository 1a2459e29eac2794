#!/usr/bin/env python3
"""secp256k1 (10,3) at 10^7 secrets: the lane-pair share kernel and the small-integer reconstruct kernel against their residency
caps ("share_waves128", "stream_waves": resident single-wave workgroups per CU; 0 = no cap).  HIP events, 100 warm-up launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import scl_amd as scl  # noqa: E402


def timed(fn, warm=100, reps=50):
    for _ in range(warm):
        fn()
    tm = scl.Timer()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


f, N, n, t = scl.SECP256K1_SCALAR, 10_000_000, 10, 3
E = 32
secrets = scl.vector_random(f, N, b"f3w-s")
coeffs = scl.vector_random(f, t * N, b"f3w-c").reshape(t, N, -1)
shares = scl.shamir_share(f, secrets, coeffs, n)
out = scl.empty(f, N)
lam = scl.lagrange_basis(f, n)
for sw in (6, 8, 10, 12, 14, 16, 20, 24, 32):
    scl.set_tuning("share_waves128", sw)
    ms = timed(lambda: scl.shamir_share(f, secrets, coeffs, n, out=shares))
    print(f"share   share_waves128={sw:2d}  {ms:7.3f} ms  {(1 + t + n) * E * N / ms / 8e9:.3f}", flush=True)
scl.set_tuning("share_waves128", 12)
for sw in (6, 8, 10, 12, 14, 16, 20, 24, 0):
    scl.set_tuning("stream_waves", sw)
    ms = timed(lambda: scl.shamir_recover(f, shares, lam, out=out))
    print(f"recover stream_waves={sw:2d}    {ms:7.3f} ms  {(n + 1) * E * N / ms / 8e9:.3f}", flush=True)
scl.set_tuning("stream_waves", -1)
assert scl.equals(f, out, secrets)
