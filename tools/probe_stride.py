#!/usr/bin/env python3
"""Does the share matrix's row stride matter for the headline share / reconstruct kernels?  (n, t) = (10, 3),
Mersenne61, 10^8 secrets, row stride N + pad elements; each case timed 3 times, interleaved, on one box."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch, scl_amd as scl
f, n, t, N = scl.M61, 10, 3, 100_000_000
secrets = scl.vector_random(f, N, b"s")
coeffs = scl.vector_random(f, t * N, b"c").reshape(t, N, 1)
lam = scl.lagrange_basis(f, n)
out = scl.empty(f, N)
pads = [0, 64, 512, 2048, 4096 + 64, 65536 + 512, 1 << 20, 2304, 1024, 1536, 3072, 6144]
order = list(reversed(pads)) if len(sys.argv) > 1 else pads          # allocation order: does speed follow the stride or the buffer?
bufs = {}
for p in order:
    bufs[p] = torch.empty(n * (N + p), dtype=torch.int64, device="cuda")
    print(f"pad {p:8d}: base % 2 MiB = {bufs[p].data_ptr() % (1 << 21):8d}  row stride % 64 KiB = {(N + p) * 8 % 65536:6d}", flush=True)
lib = scl.lib


def share(p):
    scl._chk(lib.scl_hip_shamir_share(f, C.c_void_p(bufs[p].data_ptr()), C.c_size_t(N + p), scl._dev(secrets), scl._dev(coeffs),
                                      C.c_size_t(N), C.c_size_t(N), C.c_size_t(t), C.c_size_t(n), None, scl._stream()))


def recover(p):
    scl._chk(lib.scl_hip_shamir_recover(f, scl._dev(out), C.c_void_p(bufs[p].data_ptr()), C.c_size_t(N + p), scl._hp(lam),
                                        C.c_size_t(n), C.c_size_t(N), scl._stream()))


def timeit(fn, reps=10):
    fn(); tm = scl.Timer(); tm.start()
    for _ in range(reps):
        fn()
    tm.stop(); return tm.elapsed_ms() / reps


for rnd in range(2):
    for p in pads:
        s = timeit(lambda: share(p)); r = timeit(lambda: recover(p))
        print(f"round {rnd} pad {p:8d} elements: share {s:.3f} ms {112 * N / s / 1e6:6.0f} GB/s   recover {r:.3f} ms {88 * N / r / 1e6:6.0f} GB/s", flush=True)
assert scl.equals(f, out, secrets)
