import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "secure-computation-library_amd"))
import torch, scl_amd as scl
f = scl.M61
for n, t, N in ((128, 42, 2_000_000), (40, 13, 10_000_000)):
    secrets = scl.vector_random(f, N, b"ps")
    out = scl.empty(f, n, N)
    for _ in range(3):
        scl.shamir_share_prg(f, secrets, t, n, b"seed", out=out)
    torch.cuda.synchronize()
