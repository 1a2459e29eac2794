#!/usr/bin/env python3
"""GPU box: the device-to-device copy rate of the runtime (torch's copy_ / hipMemcpyAsync) beside the library's stream_copy,
for the `measured_copy_GBps` figure bench.py prints (read + write bytes / time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch, scl_amd as scl
for gib in (1, 4, 16):
    n = gib << 27
    a = torch.empty(n, dtype=torch.int64, device="cuda"); b = torch.empty_like(a)
    a.random_()
    def timeit(fn, reps=10):
        fn(); torch.cuda.synchronize(); tm = scl.Timer(); tm.start()
        for _ in range(reps): fn()
        tm.stop(); return tm.elapsed_ms() / reps
    t1 = timeit(lambda: b.copy_(a))
    t2 = timeit(lambda: scl.stream_copy(b, a))
    t3 = timeit(lambda: b.fill_(7))
    t4 = timeit(lambda: torch.add(a, 1, out=b))
    print(f"{gib} GiB: torch copy_ {2*n*8/t1/1e6:7.0f} GB/s | scl.stream_copy {2*n*8/t2/1e6:7.0f} GB/s | torch fill_ (write only) {n*8/t3/1e6:7.0f} GB/s | torch add {2*n*8/t4/1e6:7.0f} GB/s")
    del a, b
