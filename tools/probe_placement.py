#!/usr/bin/env python3
"""Share / reconstruct (10, 3) Mersenne61, 10^8 secrets inside ONE 26 GB allocation: the share matrix at different byte
offsets and row strides.  Separates 'which physical pages' (fixed here) from 'which address bits' (varied)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch, scl_amd as scl
f, n, t, N = scl.M61, 10, 3, 100_000_000
secrets = scl.vector_random(f, N, b"s")
coeffs = scl.vector_random(f, t * N, b"c").reshape(t, N, 1)
lam = scl.lagrange_basis(f, n)
out = scl.empty(f, N)
big = torch.empty(26 * (1 << 30) // 8, dtype=torch.int64, device="cuda")
lib = scl.lib
print(f"big base % 1 GiB = {big.data_ptr() % (1 << 30)}")


def timeit(fn, reps=6):
    fn(); tm = scl.Timer(); tm.start()
    for _ in range(reps):
        fn()
    tm.stop(); return tm.elapsed_ms() / reps


for off in (0, 1 << 18, 1 << 21, 34 << 20, (1 << 30) + (1 << 20), 5 << 30, (12 << 30) + (6 << 20)):
    for pad in (0, 2048, 32768, 262144, 1 << 20, 12_500_000):
        stride = N + pad
        if off + n * stride * 8 > big.numel() * 8:
            continue
        base = C.c_void_p(big.data_ptr() + off)
        s = timeit(lambda: scl._chk(lib.scl_hip_shamir_share(f, base, C.c_size_t(stride), scl._dev(secrets), scl._dev(coeffs), C.c_size_t(N), C.c_size_t(N), C.c_size_t(t), C.c_size_t(n), None, scl._stream())))
        r = timeit(lambda: scl._chk(lib.scl_hip_shamir_recover(f, scl._dev(out), base, C.c_size_t(stride), scl._hp(lam), C.c_size_t(n), C.c_size_t(N), scl._stream())))
        print(f"offset {off:12d} B  stride N+{pad:<9d}: share {s:.3f} ms {112 * N / s / 1e6:6.0f} GB/s   recover {r:.3f} ms {88 * N / r / 1e6:6.0f} GB/s   sum {s + r:.3f}", flush=True)
