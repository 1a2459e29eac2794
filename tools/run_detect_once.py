"""One configuration of shamir_recover_detect, a few launches (profiling target): field t N"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch, scl_amd as scl
f, t, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = 2 * t + 1
secrets = scl.vector_random(f, N, b"s")
sh = scl.shamir_share_prg(f, secrets, t, n, b"seed")
for _ in range(3):
    scl.shamir_recover_detect(f, sh, t)
torch.cuda.synchronize()
