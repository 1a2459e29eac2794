import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch, scl_amd as scl
n, t, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
f = 0
secrets = scl.vector_random(f, N, b"s")
coeffs = scl.empty(f, t, N)
for k in range(t):
    coeffs[k].copy_(scl.vector_random(f, N, b"c", counter0=k * ((N * 8 + 15) // 16)))
shares = scl.empty(f, n, N)
scl.set_tuning("mfma", 1)
for _ in range(3):
    scl.shamir_share(f, secrets, coeffs, n, out=shares)
torch.cuda.synchronize()
