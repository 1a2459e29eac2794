#!/usr/bin/env python3
"""scl_hip_matmul's paths on the GPU box, in multiply-adds per second: the matrix cores (Mersenne61: one tile, row blocks, k-chunks),
k_matmul (left factor in LDS, a thread per column), k_matmul_tiled (both factors tiled, any shape) and k_matvec (one column).
HIP events around repeated calls after a warm-up."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402


def timed(fn, warm=3, reps=5):
    for _ in range(warm):
        fn()
    tm = scl.Timer()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps


def run(f, name, M, K, N, mfma, label):
    A = scl.vector_random(f, M * K, b"pm-A").reshape(M, K, -1)
    B = scl.vector_random(f, K * N, b"pm-B").reshape(K, N, -1)
    out = scl.empty(f, M, N)
    scl.set_tuning("mfma", mfma)
    try:
        ms = timed(lambda: scl.matmul(f, A, B, out=out))
    finally:
        scl.set_tuning("mfma", 0)
    print(f"{name:12s} {M:6d} x {K:6d} x {N:9d}  {label:34s} {ms:9.3f} ms  {M * K * N / ms / 1e9:8.3f} T multiply-adds/s", flush=True)


M61, M127, SECP, MONT, GF = scl.M61, scl.M127, scl.SECP256K1_SCALAR, scl.MONT128, scl.GF2_128
run(M61, "Mersenne61", 128, 43, 10_000_000, 1, "matrix cores, one tile")
run(M61, "Mersenne61", 128, 64, 10_000_000, 1, "matrix cores, one tile (K = 64)")
run(M61, "Mersenne61", 512, 64, 4_000_000, 1, "matrix cores, 4 row blocks")
run(M61, "Mersenne61", 128, 256, 4_000_000, 2, "matrix cores, 4 k-chunks")
run(M61, "Mersenne61", 512, 512, 1_000_000, 2, "matrix cores, 4 x 8 blocks")
run(M61, "Mersenne61", 512, 512, 1_000_000, 1, "matrix cores, general kernel")
run(M61, "Mersenne61", 128, 43, 10_000_000, -1, "k_matmul (left factor in LDS)")
run(M61, "Mersenne61", 10, 4, 100_000_000, -1, "k_matmul_thin (10 x 4: HBM-bound)")
run(M61, "Mersenne61", 40, 14, 12_500_000, -1, "k_matmul_thin (40 x 14)")
run(M127, "Mersenne127", 10, 4, 10_000_000, 0, "k_matmul_thin (10 x 4)")
run(SECP, "secp256k1", 10, 4, 10_000_000, 0, "k_matmul_thin (10 x 4)")
run(M61, "Mersenne61", 4096, 4096, 4096, -1, "k_matmul_tiled")
run(M61, "Mersenne61", 4096, 4096, 4096, 2, "matrix cores, 32 x 64 blocks")
run(M61, "Mersenne61", 4096, 4096, 4096, 1, "matrix cores, general kernel")
run(M61, "Mersenne61", 8192, 8192, 8192, 1, "matrix cores, general kernel")
run(M61, "Mersenne61", 1024, 1024, 1024, 1, "matrix cores, general kernel")
run(M61, "Mersenne61", 512, 16384, 512, 1, "matrix cores, general kernel")
run(M61, "Mersenne61", 128, 256, 4_000_000, 1, "matrix cores, general kernel")
run(M61, "Mersenne61", 200, 7000, 300, 0, "k_matmul_tiled (the verdict's shape)")
run(M61, "Mersenne61", 300, 300, 300, 0, "k_matmul_tiled")
run(M61, "Mersenne61", 20000, 10000, 1, 0, "k_matvec")
run(M127, "Mersenne127", 2048, 2048, 2048, 0, "k_matmul_tiled")
run(M127, "Mersenne127", 128, 43, 4_000_000, 0, "k_matmul (left factor in LDS)")
run(M127, "Mersenne127", 20000, 10000, 1, 0, "k_matvec")
run(MONT, "Mont128", 2048, 2048, 2048, 0, "k_matmul_tiled")
run(SECP, "secp256k1", 1024, 1024, 1024, 0, "k_matmul_tiled")
run(SECP, "secp256k1", 10000, 10000, 1, 0, "k_matvec")
run(GF, "GF(2^128)", 1024, 1024, 1024, 0, "k_matmul_tiled")
