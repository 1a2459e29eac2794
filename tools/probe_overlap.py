#!/usr/bin/env python3
"""GPU box: do the AES kernel (LDS + vector ALU bound, hardly any HBM traffic) and the explicit-coefficient share kernel (HBM
bound) overlap when they run from two streams?  The question behind a two-stream form of PRG-seeded sharing: chunk k's
coefficients are drawn on one stream while chunk k-1 is shared on the other."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
import torch, scl_amd as scl
f, n, t, N, CH = scl.M61, 10, 3, 100_000_000, 8
per = N // CH
secrets = scl.vector_random(f, N, b"ov-s")
coeffs = scl.empty(f, t, N)
for k in range(t):
    coeffs[k].copy_(scl.vector_random(f, N, b"ov-c", counter0=k * ((N * 8 + 15) // 16)))
shares = scl.empty(f, n, N)
nb = 2 * per                                   # AES blocks the PRG mode draws for `per` secrets at t = 3
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
blocks = [None, None]
def aes(k):
    blocks[k & 1] = scl.prg_blocks(nb, b"overlap", counter0=k * nb)
def share(k):
    sl = slice(k * per, (k + 1) * per)
    scl.shamir_share(f, secrets[sl], coeffs[:, sl].contiguous() if False else coeffs[:, sl], n, out=shares[:, sl])
def wall(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
def only_aes():
    with torch.cuda.stream(s1):
        for k in range(CH): aes(k)
def only_share():
    with torch.cuda.stream(s2):
        for k in range(CH): share_chunk(k)
# the share of a chunk needs strided views: use the raw ABI through the wrapper's row-stride support
import ctypes as C
def share_chunk(k):
    off = k * per
    st = scl.lib.scl_hip_shamir_share(f, C.c_void_p(shares.data_ptr() + 8 * off), C.c_size_t(N), C.c_void_p(secrets.data_ptr() + 8 * off),
                                      C.c_void_p(coeffs.data_ptr() + 8 * off), C.c_size_t(N), C.c_size_t(per), C.c_size_t(t), C.c_size_t(n),
                                      None, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
def both():
    for k in range(CH):
        with torch.cuda.stream(s1): aes(k)
        with torch.cuda.stream(s2): share_chunk(k)
for knobs in ({}, {"stream_waves": 0}, {"stream_waves": 0, "stream_block": 256}, {"stream_waves": 4}, {"stream_waves": 6}):
    for k_, v_ in knobs.items(): scl.set_tuning(k_, v_)
    a, b, c = wall(only_aes), wall(only_share), wall(both)
    print(knobs, f"AES only {a:.3f} | share only {b:.3f} | both {c:.3f} (sum {a + b:.3f})")
scl.set_tuning("stream_waves", -1); scl.set_tuning("stream_block", 64)
a, b, c = wall(only_aes), wall(only_share), wall(both)
print(f"(10,3) Mersenne61, 10^8 secrets in {CH} chunks: AES only ({CH * nb} blocks) {a:.3f} ms | share only {b:.3f} ms | both, two streams {c:.3f} ms (sum {a + b:.3f}, max {max(a, b):.3f})")
whole = wall(lambda: scl.shamir_share_prg(f, secrets, t, n, b"overlap", out=shares))
print(f"scl_hip_shamir_share_prg (fused kernel): {whole:.3f} ms")
