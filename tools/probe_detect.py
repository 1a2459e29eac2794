#!/usr/bin/env python3
"""error-detecting recovery (shamirRecoverD) throughput (run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch, scl_amd as scl
def timeit(fn, reps=5):
    fn(); tm = scl.Timer(); tm.start()
    for _ in range(reps): fn()
    tm.stop(); return tm.elapsed_ms() / reps
for f in (0, 1):
    for t, N in ((3, 50_000_000), (13, 20_000_000), (42, 5_000_000)):
        n = 2 * t + 1
        secrets = scl.vector_random(f, N, b"s")
        sh = scl.shamir_share_prg(f, secrets, t, n, b"seed") if t <= 13 else scl.shamir_share(f, secrets, scl.vector_random(f, t * N, b"c").reshape(t, N, -1), n)
        ms = timeit(lambda: scl.shamir_recover_detect(f, sh, t))
        E = 8 * scl.limbs(f)
        print(f"{scl.field_name(f)} recover_detect n={n} t={t}: {ms:.3f} ms  {N/ms/1e6:.2f} Gsecrets/s  {(n+1)*E*N/ms/1e6:.0f} GB/s")
        del sh, secrets
