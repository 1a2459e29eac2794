#!/usr/bin/env python3
"""Launch-bound batches under a hipGraph: BASELINE configs[0] (additive sharing, n = 3, 10^6 Mersenne61 secrets: 24 MB of shares)
and the headline's (10,3) step at 10^6 secrets, issued call by call from Python against the same calls captured ONCE into a
graph (torch.cuda.CUDAGraph = hipStreamBeginCapture on the stream the library launches on) and replayed.  The batch entry points
that take their tables as kernel arguments capture cleanly: nothing in them synchronises, allocates or copies from pageable
memory."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

f = scl.M61


def bench(name, step, reps=200):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / reps
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step()                      # (warm: table caches, arenas)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            step()
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / reps
    print(f"{name:58s} call by call {1e6 * eager:8.1f} us per step   graph replay {1e6 * graph:8.1f} us per step   x{eager / graph:.2f}", flush=True)
    return g


for N in (10_000, 100_000, 1_000_000):
    secrets = scl.vector_random(f, N, b"g-s")
    # C1: additive sharing on the PRG + Vector::sum per secret
    sh3, out = scl.empty(f, 3, N), scl.empty(f, N)

    def c1():
        scl.additive_share_prg(f, secrets, 3, b"g-seed", out=sh3)
        scl.additive_recover(f, sh3, out=out)
    bench(f"C1 additive n=3, {N} secrets: share (PRG) + sum", c1)
    assert scl.equals(f, out, secrets)
    # the headline's step
    coeffs = scl.vector_random(f, 3 * N, b"g-c").reshape(3, N, -1)
    sh10, out2 = scl.empty(f, 10, N), scl.empty(f, N)
    lam = scl.lagrange_basis(f, 10)

    def c2():
        scl.shamir_share(f, secrets, coeffs, 10, out=sh10)
        scl.shamir_recover(f, sh10, lam, out=out2)
    g = bench(f"C2 Shamir (10,3), {N} secrets: share + reconstruct", c2)
    out2.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert scl.equals(f, out2, secrets)
