#!/usr/bin/env python3
"""A/B of two builds of the library on one box: Mersenne127 / Mont128 (10,3) share and reconstruct at 10^7 secrets, 100 warm-up +
50 timed launches, three rounds.  usage: probe_ab_c3.py <directory that holds the scl_amd package to load>"""
import sys

sys.path.insert(0, sys.argv[1])
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


for f, name in ((scl.M127, "Mersenne127"), (scl.MONT128, "Mont128")):
    N, n, t, E = 10_000_000, 10, 3, 16
    secrets = scl.vector_random(f, N, b"ab-s")
    coeffs = scl.vector_random(f, t * N, b"ab-c").reshape(t, N, -1)
    shares = scl.shamir_share(f, secrets, coeffs, n)
    out = scl.empty(f, N)
    lam = scl.lagrange_basis(f, n)
    for rep in range(3):
        r = timed(lambda: scl.shamir_recover(f, shares, lam, out=out))
        s = timed(lambda: scl.shamir_share(f, secrets, coeffs, n, out=shares))
        print(f"{sys.argv[1][-12:]:12s} {name:12s} reconstruct {r:.4f} ms {(n + 1) * E * N / r / 8e9:.3f}   share {s:.4f} ms {(1 + t + n) * E * N / s / 8e9:.3f}", flush=True)
    assert scl.equals(f, out, secrets)
