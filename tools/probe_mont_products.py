import os, sys
sys.path.insert(0, "/root/repo/secure-computation-library_amd")
import torch, scl_amd as scl
def timed(fn, reps=30, warm=60):
    tms = [scl.Timer() for _ in range(reps)]
    for k in range(-warm, reps):
        if k >= 0: tms[k].start()
        fn()
        if k >= 0: tms[k].stop()
    torch.cuda.synchronize()
    ms = [t.elapsed_ms() for t in tms]; return sum(ms)/len(ms)
status = scl.ew_status_buffer()
for f, name in ((scl.SECP256K1_SCALAR, "secp256k1_order"), (scl.SECP256K1_FIELD, "secp256k1_field"), (scl.MONT128, "Mont128")):
    for N in (10_000_000, 100_000_000 if f == scl.MONT128 else 30_000_000):
        a, b, out = scl.vector_random(f, N, b"a"), scl.vector_random(f, N, b"b"), scl.empty(f, N)
        print(name, N, "mul %.4f ms  inv %.4f ms  div %.4f ms" % (timed(lambda: scl.ew(f, scl.MUL, a, b, out=out), 20, 30), timed(lambda: scl.ew_status(f, scl.INV, a, None, status, out=out), 20, 30), timed(lambda: scl.ew_status(f, scl.DIV, b, a, status, out=out), 20, 30)), flush=True)
        del a, b, out; torch.cuda.empty_cache()
