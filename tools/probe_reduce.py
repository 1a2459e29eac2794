#!/usr/bin/env python3
"""Vector::sum / Vector::dot / element-wise rates on the GPU box (HIP events around repeated calls; the reductions include
their host fold of the per-workgroup partials, so sizes are large).  Algorithmic bytes: sum E, dot 2E, binary op 3E per element."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402

for f, name, N in ((scl.M61, "Mersenne61", 400_000_000), (scl.M127, "Mersenne127", 200_000_000), (scl.Z2K(64), "Z2k<64>", 400_000_000),
                   (scl.MONT128, "Mont128", 200_000_000), (scl.GF2_128, "GF(2^128)", 100_000_000)):
    E = 8 * scl.limbs(f)
    a = scl.empty(f, N)
    b = scl.empty(f, N)
    if f >= 0x100:
        a.copy_(scl.vector_random(scl.M61, N, b"pr-a"))
        b.copy_(scl.vector_random(scl.M61, N, b"pr-b"))
    else:
        scl.vector_random(f, N, b"pr-a", out=a)
        scl.vector_random(f, N, b"pr-b", out=b)
    out = torch.empty_like(a)
    tm = scl.Timer()
    for label, fn, bytes_per in (("sum", lambda: scl.vsum(f, a), E), ("dot", lambda: scl.dot(f, a, b), 2 * E),
                                 ("add", lambda: scl.ew(f, scl.ADD, a, b, out=out), 3 * E),
                                 ("mul", lambda: scl.ew(f, scl.MUL, a, b, out=out), 3 * E),
                                 ("smul", lambda: scl.scalar_mul(f, a, scl.to_host(b[7]), out=out), 2 * E)):
        fn()
        torch.cuda.synchronize()
        tm.start()
        for _ in range(5):
            fn()
        tm.stop()
        ms = tm.elapsed_ms() / 5
        print(f"{name:12s} {label:4s} N={N:10d}  {ms:8.3f} ms  {bytes_per * N / ms / 1e6:7.0f} GB/s", flush=True)
    del a, b, out
    torch.cuda.empty_cache()
