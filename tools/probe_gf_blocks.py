import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "secure-computation-library_amd"))
import torch, scl_amd as scl
f, n, t, N = scl.GF2_128, 40, 13, 12_500_000
secrets = scl.vector_random(f, N, b"s")
coeffs = scl.vector_random(f, t * N, b"c").reshape(t, N, -1)
shares = scl.empty(f, n, N)
tm = scl.Timer()
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); tm.start()
    for _ in range(reps): fn()
    tm.stop(); return tm.elapsed_ms() / reps
for mb in (0, 256, 512, 768, 1024, 2048, 4096):
    scl.set_tuning("max_blocks", mb)
    print(f"max_blocks {mb:5d}: share {timed(lambda: scl.shamir_share(f, secrets, coeffs, n, out=shares)):.3f} ms", flush=True)
