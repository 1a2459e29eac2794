#!/usr/bin/env python3
"""PRG-driven (10,3) sharing -- the reference's own mode -- with the threshold compiled into the fused kernel ("prg_t3" 1,
k_share_prg_small_t) against the any-t fused kernel ("prg_t3" 0) and the two-pass form ("prg_two_pass" 1), Mersenne61 at 10^8
secrets and Mersenne127 at 10^7; the share matrices are compared word for word."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "secure-computation-library_amd"))
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


for f, name, N in ((scl.M61, "Mersenne61", 100_000_000), (scl.M127, "Mersenne127", 10_000_000)):
    n, t = 10, 3
    secrets = scl.vector_random(f, N, b"s")
    ref = None
    for label, knobs in (("any-t fused kernel", {"prg_t3": 0}), ("threshold compiled in", {"prg_t3": 1}),
                         ("two passes (rows, then the explicit-coefficient kernel)", {"prg_two_pass": 1})):
        for k, v in knobs.items():
            scl.set_tuning(k, v)
        shares = scl.empty(f, n, N)
        ms = timed(lambda: scl.shamir_share_prg(f, secrets, t, n, b"seed", out=shares))
        for k in knobs:
            scl.set_tuning(k, 1 if k == "prg_t3" else 0)
        same = True if ref is None else bool(torch.equal(ref, shares))
        if ref is None:
            ref = shares
        print(f"{name} (10,3) N={N} {label}: {ms:.3f} ms = {N / ms / 1e6:.2f} G secrets/s; identical to the first: {same}", flush=True)
    rec = scl.shamir_recover(f, ref)
    print("   reconstructs:", scl.equals(f, rec, secrets), flush=True)
    del ref, shares, secrets
