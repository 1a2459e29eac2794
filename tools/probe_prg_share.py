#!/usr/bin/env python3
"""PRG-driven sharing (shamirSecretShare(secret, t, n, prg): the reference's own mode) against sharing from coefficients in
HBM, by shape and field: HIP events around 5 launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

tm = scl.Timer()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


shapes = [(10, 3, 20_000_000), (10, 7, 10_000_000), (20, 9, 10_000_000), (40, 13, 10_000_000), (64, 16, 5_000_000), (64, 21, 5_000_000),
          (128, 42, 2_000_000)]
for f in (scl.M61, scl.M127, scl.GF2_128, scl.MONT128, scl.SECP256K1_SCALAR):
    for n, t, N in shapes:
        if scl.limbs(f) == 4:
            N //= 4
        elif scl.limbs(f) == 2:
            N //= 2
        secrets = scl.vector_random(f, N, b"ps")
        coeffs = scl.vector_random(f, t * N, b"pc").reshape(t, N, -1)
        out = scl.empty(f, n, N)
        ms_c = timed(lambda: scl.shamir_share(f, secrets, coeffs, n, out=out))
        ms_p = timed(lambda: scl.shamir_share_prg(f, secrets, t, n, b"seed", out=out))
        blocks = scl.blocks_per_secret(f, t)
        print(f"{scl.field_name(f):22s} ({n:3d},{t:2d}) N={N:9d}: from coefficients {ms_c:7.3f} ms {N / ms_c / 1e6:7.2f} G/s   "
              f"from the PRG {ms_p:7.3f} ms {N / ms_p / 1e6:7.2f} G secrets/s  ({blocks * N / ms_p / 1e6:5.1f} G AES blocks/s)", flush=True)
        del secrets, coeffs, out
