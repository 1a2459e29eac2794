#!/usr/bin/env python3
"""Where should the inputs and outputs sit relative to the share matrix?  One 44 GB arena; matrix M (8 GB), inputs I
(secret + 3 coefficient rows, 3.2 GB) and the reconstruct output O (0.8 GB) at chosen GiB offsets."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch, scl_amd as scl
f, n, t, N = scl.M61, 10, 3, 100_000_000
lam = scl.lagrange_basis(f, n)
G = 1 << 30
arena = torch.empty(44 * G // 8, dtype=torch.int64, device="cuda")
lib = scl.lib
src = scl.vector_random(f, 4 * N, b"s")            # secrets || c1 || c2 || c3


def view(off_bytes, elems):
    return arena[off_bytes // 8: off_bytes // 8 + elems]


def timeit(fn, reps=6):
    fn(); tm = scl.Timer(); tm.start()
    for _ in range(reps):
        fn()
    tm.stop(); return tm.elapsed_ms() / reps


for (m, i, o) in ((0, 10, 14), (0, 30, 14), (0, 30, 40), (0, 10, 40), (12, 0, 4), (12, 30, 4), (12, 30, 40), (24, 0, 4), (24, 0, 40),
                  (24, 36, 40), (34, 0, 4), (34, 0, 10), (34, 20, 10), (34, 20, 30), (16, 0, 30), (8, 0, 30), (8, 20, 30), (4, 14, 20)):
    M = view(m * G, n * N); I = view(i * G, 4 * N); O = view(o * G, N)
    I.copy_(src.view(-1))
    ip = I.data_ptr()
    s = timeit(lambda: scl._chk(lib.scl_hip_shamir_share(f, C.c_void_p(M.data_ptr()), C.c_size_t(N), C.c_void_p(ip), C.c_void_p(ip + 8 * N), C.c_size_t(N), C.c_size_t(N), C.c_size_t(t), C.c_size_t(n), None, scl._stream())))
    r = timeit(lambda: scl._chk(lib.scl_hip_shamir_recover(f, C.c_void_p(O.data_ptr()), C.c_void_p(M.data_ptr()), C.c_size_t(N), scl._hp(lam), C.c_size_t(n), C.c_size_t(N), scl._stream())))
    ok = bool(torch.equal(O, I[:N]))
    print(f"matrix @{m:2d} GiB  inputs @{i:2d} GiB  out @{o:2d} GiB: share {s:.3f} ms {112 * N / s / 1e6:6.0f} GB/s   recover {r:.3f} ms {88 * N / r / 1e6:6.0f} GB/s   sum {s + r:.3f}  ok={ok}", flush=True)
