#!/usr/bin/env python3
"""Does the share kernel's slow / fast mode belong to the buffer it writes, to the buffers it reads, or to the pair?
Mersenne61 (10,3), 1e8 secrets: inputs (secrets + coefficients) of set i, shares buffer of set j, all 16 combinations; then the
reconstruct kernel per shares buffer."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

f, n, t, N = scl.M61, 10, 3, 100_000_000
ins, outs = [], []
for a in range(4):
    secrets = scl.vector_random(f, N, b"cap-s%d" % a)
    coeffs = scl.empty(f, t, N)
    per_row = (N * 8 + 15) // 16
    for k in range(t):
        scl.vector_random(f, N, b"cap-c%d" % a, counter0=k * per_row, out=coeffs[k])
    ins.append((secrets, coeffs))
    outs.append(scl.empty(f, n, N))
rec = scl.empty(f, N)
lam = scl.lagrange_basis(f, n)
tm = scl.Timer()


def timed(fn):
    fn()
    torch.cuda.synchronize()
    tm.start()
    for _ in range(10):
        fn()
    tm.stop()
    return tm.elapsed_ms() / 10


print("share ms: rows = inputs of set i, columns = shares buffer of set j")
for i, (secrets, coeffs) in enumerate(ins):
    print(f"inputs {i}: " + "  ".join(f"{timed(lambda: scl.shamir_share(f, secrets, coeffs, n, out=outs[j])):6.3f}" for j in range(4)), flush=True)
print("reconstruct ms per shares buffer: " + "  ".join(f"{timed(lambda: scl.shamir_recover(f, outs[j], lam, out=rec)):6.3f}" for j in range(4)))
