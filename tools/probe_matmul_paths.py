#!/usr/bin/env python3
"""scl_hip_matmul over Mersenne61 by path (the "mfma" knob: -1 the vector-ALU kernels, 1 the general matrix-core kernel, 2 the
(row block, k-chunk) form on the sharing kernels) against the automatic choice, over small, thin and long shapes: where the
dispatch thresholds of capi.hip come from (profiles/r5_probe_matmul_paths.txt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch, scl_amd as scl
def timed(fn, warm=30, reps=30):
    for _ in range(warm): fn()
    tm = scl.Timer(); tm.start()
    for _ in range(reps): fn()
    tm.stop(); return tm.elapsed_ms() / reps
f = scl.M61
shapes = [(33,65,33),(40,70,40),(48,96,48),(64,65,64),(64,128,64),(96,96,96),(100,200,100),(33,1000,33),(33,300,33),(200,65,200),(33,65,1000),(1000,65,33),(64,65,4000),(33,65,10000),(500,70,500),
          (100,100,8000),(100,100,16000),(100,100,32000),(100,100,65000),(128,256,16000),(128,256,65000),(64,70,65000),(64,70,200000),(128,128,200000),(100,1000,20000),(100,1000,100000),
          (129,65,65000),(200,100,100000),(2048,64,2048),(1000,64,1000),(300,40,300),(300,64,3000)]
for (M,K,N) in shapes:
    A = scl.vector_random(f, M*K, b"A").reshape(M,K,-1); B = scl.vector_random(f, K*N, b"B").reshape(K,N,-1); out = scl.empty(f, M, N)
    row=[]
    for v in (-1,1,2,0):
        scl.set_tuning("mfma", v)
        try: ms = timed(lambda: scl.matmul(f, A, B, out=out))
        except scl.SclError: ms = float("inf")
        scl.set_tuning("mfma", 0)
        row.append(ms)
    best=min(row[:3]); print(f"{M:5d} x {K:5d} x {N:7d}  vector {row[0]:.4f}  gemm(1) {row[1]:.4f}  blocks(2) {row[2]:.4f}  auto {row[3]:.4f}  auto/best {row[3]/best:.2f} best {['vector','gemm','blocks'][row[:3].index(best)]}", flush=True)
