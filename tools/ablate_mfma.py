"""time the MFMA share kernel from an alternate (diagnostic) build of the library: SCL_SO=path"""
import ctypes as C, os, sys, torch
so = os.environ["SCL_SO"]
lib = C.CDLL(so)
n, t, N = 128, 42, 10_000_000
dev = "cuda"
sec = torch.randint(0, 2**60, (N,), dtype=torch.int64, device=dev)
co = torch.randint(0, 2**60, (t, N), dtype=torch.int64, device=dev)
sh = torch.empty((n, N), dtype=torch.int64, device=dev)
lib.scl_hip_set_tuning(b"mfma", C.c_long(1))
def run():
    st = lib.scl_hip_shamir_share(0, C.c_void_p(sh.data_ptr()), C.c_size_t(N), C.c_void_p(sec.data_ptr()), C.c_void_p(co.data_ptr()),
                                  C.c_size_t(N), C.c_size_t(N), C.c_size_t(t), C.c_size_t(n), None, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0, st
run(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): run()
b.record(); torch.cuda.synchronize()
print(os.path.basename(so), f"{a.elapsed_time(b)/5:.3f} ms")
