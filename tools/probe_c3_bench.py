#!/usr/bin/env python3
"""Why C3's two kernels read 0.33-0.34 ms inside bench.py (share_recover_config: fresh allocations, one warm-up, five alternating
share / reconstruct launches, HIP events around every launch) and 0.28 ms in tools/probe_c3_waves.py (twenty launches of ONE
kernel between two events).  Per-launch event times, in launch order, for: the bench's own pattern at 5 and at 50 steps, the same
kernel repeated, the alternation with a device synchronisation before every launch, and the alternation after a 200 ms spin of
another kernel (clocks up).  VERDICT r3, weak #5."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

n, t, N = 10, 3, 10_000_000


def fmt(xs):
    return " ".join(f"{x:.3f}" for x in xs)


def run(f, label, steps, pattern, sync_each=False, spin=False, fresh=True, bufs=None):
    if fresh or bufs is None:
        secrets = scl.vector_random(f, N, b"c3b-s")
        coeffs = scl.vector_random(f, t * N, b"c3b-c").reshape(t, N, -1)
        shares, out = scl.empty(f, n, N), scl.empty(f, N)
    else:
        secrets, coeffs, shares, out = bufs
    lam = scl.lagrange_basis(f, n)
    if spin:
        big = torch.empty(1 << 28, dtype=torch.int64, device="cuda")
        for _ in range(40):
            big.add_(1)
        del big
    ts = [(scl.Timer(), scl.Timer()) for _ in range(steps)]
    scl.shamir_share(f, secrets, coeffs, n, out=shares)
    scl.shamir_recover(f, shares, lam, out=out)
    for k in range(steps):
        if pattern in ("alt", "share"):
            if sync_each:
                torch.cuda.synchronize()
            ts[k][0].start()
            scl.shamir_share(f, secrets, coeffs, n, out=shares)
            ts[k][0].stop()
        if pattern in ("alt", "rec"):
            if sync_each:
                torch.cuda.synchronize()
            ts[k][1].start()
            scl.shamir_recover(f, shares, lam, out=out)
            ts[k][1].stop()
    torch.cuda.synchronize()
    assert scl.equals(f, out, secrets)
    E = 8 * scl.limbs(f)
    for which, name, b in ((0, "share", (1 + t + n) * E), (1, "rec  ", (n + 1) * E)):
        if (which == 0 and pattern == "rec") or (which == 1 and pattern == "share"):
            continue
        ms = [x[which].elapsed_ms() for x in ts]
        mean = sum(ms) / len(ms)
        print(f"{scl.field_name(f):12s} {label:34s} {name} mean {mean:.3f} ms = {b * N / mean / 1e6 / 8000:.3f} of peak  min {min(ms):.3f} max {max(ms):.3f}"
              + (f"  [{fmt(ms)}]" if steps <= 10 else f"  first five [{fmt(ms[:5])}] last five [{fmt(ms[-5:])}]"), flush=True)
    return secrets, coeffs, shares, out


for f in (scl.M127, scl.MONT128):
    bufs = run(f, "bench pattern, 5 steps, fresh", 5, "alt")
    run(f, "bench pattern, 5 steps, same bufs", 5, "alt", fresh=False, bufs=bufs)
    run(f, "bench pattern, 50 steps", 50, "alt", fresh=False, bufs=bufs)
    run(f, "share only x50", 50, "share", fresh=False, bufs=bufs)
    run(f, "reconstruct only x50", 50, "rec", fresh=False, bufs=bufs)
    run(f, "alternating, sync before each", 10, "alt", sync_each=True, fresh=False, bufs=bufs)
    run(f, "bench pattern, 5 steps, after spin", 5, "alt", spin=True, fresh=False, bufs=bufs)
    del bufs
    torch.cuda.empty_cache()
    run(f, "bench pattern, 5 steps, fresh again", 5, "alt")
    torch.cuda.empty_cache()
