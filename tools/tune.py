#!/usr/bin/env python3
"""Launch-geometry sweep for the streaming kernels (run on the GPU box).
Prints GB/s (algorithmic bytes) per (max_blocks, nontemporal) for recover, share and the copy probe."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

f = int(os.environ.get("FIELD", "0"))
n, t = int(os.environ.get("NPARTIES", "10")), int(os.environ.get("THRESH", "3"))
N = int(os.environ.get("SECRETS", "100000000"))
L = scl.limbs(f)
E = 8 * L
secrets = scl.vector_random(f, N, b"tune")
coeffs = scl.empty(f, t, N)
for k in range(t):
    coeffs[k].copy_(scl.vector_random(f, N, b"tune-c", counter0=k * ((N * E + 15) // 16)))
shares = scl.empty(f, n, N)
out = scl.empty(f, N)
lam = scl.lagrange_basis(f, n)


def timeit(fn, reps=10):
    fn()
    tm = scl.Timer()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


half = (shares.numel() // 2) & ~1
src, dst = shares.view(-1)[:half], shares.view(-1)[half:2 * half]
print(f"field={scl.field_name(f)} n={n} t={t} N={N}")
print(f"{'max_blocks':>10s} {'nt':>3s} {'recover GB/s':>13s} {'share GB/s':>11s} {'copy GB/s':>10s}")
for nt in (1, 0):
    scl.set_tuning("nontemporal", nt)
    for mb in (512, 1024, 2048, 4096, 8192, 16384, 65536, 1 << 30):
        scl.set_tuning("max_blocks", mb)
        r = timeit(lambda: scl.shamir_recover(f, shares, lam, out=out))
        s = timeit(lambda: scl.shamir_share(f, secrets, coeffs, n, out=shares))
        c = timeit(lambda: scl.stream_copy(dst, src))
        print(f"{mb:>10d} {nt:>3d} {(n + 1) * E * N / r / 1e6:13.0f} {((1 + t) + n) * E * N / s / 1e6:11.0f} "
              f"{2 * half * 8 / c / 1e6:10.0f}")
