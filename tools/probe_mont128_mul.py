#!/usr/bin/env python3
"""Mont128 kernels whose time is the Montgomery product: element-wise multiply / inverse / divide at 10^7, dot, a 1024^3 matrix
product, reconstruction at explicit (full-width) nodes -- timed with HIP events, for the A/B of Mont128::mul's formulations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402


def timed(fn, reps=30, warm=60):
    tms = [scl.Timer() for _ in range(reps)]
    for k in range(-warm, reps):
        if k >= 0:
            tms[k].start()
        fn()
        if k >= 0:
            tms[k].stop()
    torch.cuda.synchronize()
    ms = [t.elapsed_ms() for t in tms]
    return sum(ms) / len(ms)


f, N = scl.MONT128, 10_000_000
a, b, out = scl.vector_random(f, N, b"mm-a"), scl.vector_random(f, N, b"mm-b"), scl.empty(f, N)
status = scl.ew_status_buffer()
print(f"ew mul 10^7      {timed(lambda: scl.ew(f, scl.MUL, a, b, out=out)):.4f} ms")
print(f"ew inv 10^7      {timed(lambda: scl.ew_status(f, scl.INV, a, None, status, out=out)):.4f} ms")
print(f"ew div 10^7      {timed(lambda: scl.ew_status(f, scl.DIV, b, a, status, out=out)):.4f} ms")
print(f"dot 10^7         {timed(lambda: scl.dot(f, a, b), 10, 10):.4f} ms")
M = 1024
A = scl.vector_random(f, M * M, b"mm-A").reshape(M, M, 2)
B = scl.vector_random(f, M * M, b"mm-B").reshape(M, M, 2)
C = scl.empty(f, M, M)
print(f"matmul 1024^3    {timed(lambda: scl.matmul(f, A, B, out=C), 10, 5):.4f} ms")
n, Ns = 10, 10_000_000
shares = scl.vector_random(f, n * Ns, b"mm-s").reshape(n, Ns, 2)
nodes = scl.to_host(scl.vector_random(f, n, b"mm-nodes"))
lam = scl.lagrange_basis(f, n, nodes)
rec = scl.empty(f, Ns)
print(f"recover (10 full-width nodes) 10^7  {timed(lambda: scl.shamir_recover(f, shares, lam, out=rec), 20, 20):.4f} ms")
