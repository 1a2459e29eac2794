#!/usr/bin/env python3
"""secp256k1 (10,3) reconstruct / share and the AoS <-> SoA bridge at n = 10 on the GPU box: HIP events around repeated calls after
a 100-launch warm-up (the clock ramp), fractions of the 8 TB/s peak.  `transpose_tile` sweeps the LDS tile of
k_transpose16; `force_table` 1 = the Montgomery-product reconstruct kernel the small-integer one replaces."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
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


for f, name, N in ((scl.SECP256K1_SCALAR, "secp256k1_order", 10_000_000), (scl.MONT128, "Mont128", 10_000_000)):
    E, n, t = 8 * scl.limbs(f), 10, 3
    secrets = scl.vector_random(f, N, b"f3-s")
    coeffs = scl.vector_random(f, t * N, b"f3-c").reshape(t, N, -1)
    shares = scl.shamir_share(f, secrets, coeffs, n)
    out = scl.empty(f, N)
    lam = scl.lagrange_basis(f, n)
    for ft in (0, 1):
        scl.set_tuning("force_table", ft)
        ms = timed(lambda: scl.shamir_recover(f, shares, lam, out=out))
        assert scl.equals(f, out, secrets)
        print(f"{name:16s} reconstruct force_table={ft}  {ms:7.3f} ms  {(n + 1) * E * N / ms / 1e6:6.0f} GB/s  {(n + 1) * E * N / ms / 8e9:.3f}", flush=True)
    scl.set_tuning("force_table", 0)
    for sw in (12, 0):    # 0 = the lane-per-element kernel (secp256k1: instead of the lane pairs)
        scl.set_tuning("share_waves128", sw)
        ms = timed(lambda: scl.shamir_share(f, secrets, coeffs, n, out=shares))
        print(f"{name:16s} share share_waves128={sw:2d}    {ms:7.3f} ms  {(1 + t + n) * E * N / ms / 1e6:6.0f} GB/s  {(1 + t + n) * E * N / ms / 8e9:.3f}", flush=True)
    scl.set_tuning("share_waves128", 12)
    del secrets, coeffs, shares, out
    torch.cuda.empty_cache()

for f, name, N in ((scl.M61, "Mersenne61", 100_000_000), (scl.M127, "Mersenne127", 10_000_000), (scl.SECP256K1_SCALAR, "secp256k1_order", 10_000_000)):
    E, n = 8 * scl.limbs(f), 10
    soa = scl.empty(f, n, N)
    for i in range(n):
        scl.vector_random(f, N, b"lay-%d" % i, out=soa[i])
    aos = scl.soa_to_aos(f, soa)
    for tile in (0, 256, 128, 64):
        scl.set_tuning("transpose_tile", tile)
        for fs in ((0, 1) if tile == 0 else (0,)):
            scl.set_tuning("force_scalar", fs)
            warm, reps = (100, 50) if N <= 10_000_000 else (10, 10)
            a = timed(lambda: scl.lib.scl_hip_soa_to_aos(f, scl._dev(aos), scl._dev(soa), N, N, n, scl._stream()), warm, reps)
            b = timed(lambda: scl.lib.scl_hip_aos_to_soa(f, scl._dev(soa), N, scl._dev(aos), N, n, scl._stream()), warm, reps)
            nb = 2 * n * E * N
            print(f"{name:16s} tile={tile:3d} scalar={fs}  soa_to_aos {a:7.3f} ms {nb / a / 8e9:.3f}   aos_to_soa {b:7.3f} ms {nb / b / 8e9:.3f}", flush=True)
        scl.set_tuning("force_scalar", 0)
    scl.set_tuning("transpose_tile", 0)
    del soa, aos
    torch.cuda.empty_cache()
