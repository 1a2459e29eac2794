#!/usr/bin/env python3
"""Share kernel paths side by side (run on the GPU box): for each (field, n, t) the default dispatch, the
matrix-core path forced / disabled, and the plain Horner kernel (force_table), in ms, GB/s and G secrets/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402,F401
import scl_amd as scl  # noqa: E402


def timeit(fn, reps=5):
    fn()
    tm = scl.Timer()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


def run(f, n, t, N):
    E = 8 * scl.limbs(f)
    secrets = scl.vector_random(f, N, b"p")
    coeffs = scl.empty(f, t, N)
    for k in range(t):
        coeffs[k].copy_(scl.vector_random(f, N, b"c", counter0=k * ((N * E + 15) // 16)))
    shares = scl.empty(f, n, N)
    ref = None
    modes = [("default", {}), ("mfma off", {"mfma": -1}), ("horner", {"mfma": -1, "force_table": 1})]
    if f == 0 and n <= 128 and t <= 63:
        modes.insert(1, ("mfma on", {"mfma": 1}))
    for name, tune in modes:
        for k in ("mfma", "force_table"):
            scl.set_tuning(k, tune.get(k, 0))
        ms = timeit(lambda: scl.shamir_share(f, secrets, coeffs, n, out=shares))
        chk = int(shares.view(-1)[:: max(1, shares.numel() // 1000003)].sum().item())
        if ref is None:
            ref = chk
        print(f"{scl.field_name(f):12s} n={n:3d} t={t:2d} N={N:9d} {name:9s}: {ms:8.3f} ms {((1 + t) + n) * E * N / ms / 1e6:6.0f} GB/s "
              f"{N / ms / 1e6:6.2f} Gsec/s {'same' if chk == ref else 'DIFFERENT'}", flush=True)
    for k in ("mfma", "force_table"):
        scl.set_tuning(k, 0)


if __name__ == "__main__":
    shapes = ((0, 40, 13, 20_000_000), (0, 20, 9, 40_000_000), (0, 64, 16, 10_000_000), (0, 128, 16, 10_000_000),
              (0, 10, 8, 50_000_000), (1, 40, 13, 10_000_000), (1, 128, 16, 4_000_000), (1, 10, 8, 20_000_000), (3, 40, 13, 10_000_000), (2, 40, 13, 5_000_000))
    for f, n, t, N in shapes:
        run(f, n, t, N)
