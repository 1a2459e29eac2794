#!/usr/bin/env python3
"""The element-wise kernels that are not stream-bound, three launches each at the bench line's sizes: inverse and divide over
Mersenne61 (10^8), inverse over Mersenne127, multiply and inverse over GF(2^128) (10^7) -- the workload of tools/ew_sq.sh."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

for f, N in ((scl.M61, 100_000_000), (scl.M127, 10_000_000), (scl.GF2_128, 10_000_000)):
    a = scl.vector_random(f, N, b"a")
    b = scl.vector_random(f, N, b"b")
    out = scl.empty(f, N)
    for _ in range(3):
        for op in (scl.INV, scl.DIV, scl.MUL):
            try:
                scl.ew(f, op, a, b, out=out)
            except scl.SclError:
                pass    # a zero among the random elements: reported after the kernel has run
    del a, b, out
torch.cuda.synchronize()
