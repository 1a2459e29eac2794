#!/usr/bin/env python3
"""GF(2^128) sharing at the default nodes: eight nodes per Horner loop (k_share_gf_tiles, "gf_tiles" 1) against one node at a
time (k_share_gf_nodes, "gf_tiles" 0), at C4's shard size and two other shapes; the two share matrices are compared word for
word and reconstructed."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

f = scl.GF2_128
tm = scl.Timer()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


for n, t, N in ((40, 13, 12_500_000), (40, 13, 5_000_000), (20, 6, 10_000_000), (64, 16, 4_000_000)):
    secrets = scl.vector_random(f, N, b"s")
    coeffs = scl.vector_random(f, t * N, b"c").reshape(t, N, -1)
    got = {}
    for mode in (0, 1):
        scl.set_tuning("gf_tiles", mode)
        shares = scl.empty(f, n, N)
        ms = timed(lambda: scl.shamir_share(f, secrets, coeffs, n, out=shares))
        byts = (1 + t + n) * 16 * N
        print(f"({n},{t}) N={N} gf_tiles={mode}: share {ms:.3f} ms = {byts / ms / 1e9:.2f} TB/s-equivalent ({byts / ms / 1e9 / 8:.3f} of 8 TB/s)", flush=True)
        got[mode] = shares
    same = bool(torch.equal(got[0], got[1]))
    rec = scl.shamir_recover(f, got[1], scl.lagrange_basis(f, n))
    print(f"   identical share matrices: {same}; reconstructs: {scl.equals(f, rec, secrets)}", flush=True)
    del got, shares, secrets, coeffs
scl.set_tuning("gf_tiles", 1)
