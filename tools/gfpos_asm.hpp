// tools/gfpos_asm.hpp -- the hand-issued LDS pipelines of k_recover_gf128_pos from rounds 2-3, kept for A/B runs only
// (tools/gfpos_bench.hip, tools/gf128_bench.hip).  The reads are inline-assembly ds_read_b128 whose destination registers the
// compiler believes valid from the asm statement on, while the s_waitcnt that covers them sits in a LATER statement: correct
// only as long as the register allocator leaves those registers alone in between.  The library ships the compiler-visible
// form (csrc/kernels.hpp, GfposPipeCV); these run within 1.5 % of it (profiles/r4_gfpos_bench.txt).
#pragma once
#include "../secure-computation-library_amd/csrc/kernels.hpp"

namespace sclhip {

template <int OFF>
__device__ __forceinline__ void lds_read128(u32x4& d, u32 addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// batch IDX of a group of G parties: word wd = IDX / (2G), party j = (IDX / 2) % G, half = IDX & 1 (0: the odd nibbles)
template <int G, int IDX>
__device__ __forceinline__ void gfpos_issue4(u32x4 (&buf)[4], u32 gbase, const u32 (&w)[G][4]) {
  constexpr int wd = IDX / (2 * G), j = (IDX / 2) % G, half = IDX & 1;
  const u32 m = half ? ((w[j][wd] << 4) & 0xF0F0F0F0u) : (w[j][wd] & 0xF0F0F0F0u);
  lds_read128<j * 2048 + (0 + (1 - half)) * 256>(buf[0], add_byte<0>(gbase, m));
  lds_read128<j * 2048 + (2 + (1 - half)) * 256>(buf[1], add_byte<1>(gbase, m));
  lds_read128<j * 2048 + (4 + (1 - half)) * 256>(buf[2], add_byte<2>(gbase, m));
  lds_read128<j * 2048 + (6 + (1 - half)) * 256>(buf[3], add_byte<3>(gbase, m));
}
template <int G, int IDX>
__device__ __forceinline__ void gfpos_pipe(u32x4 (&A)[4], u32x4 (&B)[4], u32x4 (&acc)[4], u32 gbase, const u32 (&w)[G][4]) {
  constexpr int NB = 8 * G;
  if constexpr (IDX < NB) {
    if constexpr (IDX + 1 < NB) gfpos_issue4<G, IDX + 1>((IDX & 1) ? A : B, gbase, w);
    u32x4(&cur)[4] = (IDX & 1) ? B : A;
    if constexpr (IDX + 1 < NB) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
    gfpos_fold(acc[IDX / (2 * G)], cur);
    gfpos_pipe<G, IDX + 1>(A, B, acc, gbase, w);
  }
}
// the same pipeline NBUF - 1 batches deep: batch IDX + NBUF - 1 is issued before batch IDX is waited for (all but the
// 4 (NBUF - 1) newest reads: LDS returns in order)
template <int I>
__device__ __forceinline__ void gfpos_wait(u32x4 (&cur)[4]) {
  static_assert(I >= 0 && I <= 12 && I % 4 == 0, "lgkmcnt is a 4-bit counter");
  if constexpr (I == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
  if constexpr (I == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
  if constexpr (I == 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
  if constexpr (I == 12) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
}
template <int G, int IDX, int NBUF>
__device__ __forceinline__ void gfpos_pipe_n(u32x4 (&buf)[NBUF][4], u32x4 (&acc)[4], u32 gbase, const u32 (&w)[G][4]) {
  constexpr int NB = 8 * G;
  if constexpr (IDX < NB) {
    if constexpr (IDX + NBUF - 1 < NB) gfpos_issue4<G, IDX + NBUF - 1>(buf[(IDX + NBUF - 1) % NBUF], gbase, w);
    constexpr int younger = (NB - 1 - IDX) < (NBUF - 1) ? (NB - 1 - IDX) : (NBUF - 1);
    u32x4(&cur)[4] = buf[IDX % NBUF];
    gfpos_wait<4 * younger>(cur);
    gfpos_fold(acc[IDX / (2 * G)], cur);
    gfpos_pipe_n<G, IDX + 1, NBUF>(buf, acc, gbase, w);
  }
}
template <int G, int IDX, int NBUF>
__device__ __forceinline__ void gfpos_prologue_n(u32x4 (&buf)[NBUF][4], u32 gbase, const u32 (&w)[G][4]) {
  if constexpr (IDX < NBUF - 1) {
    gfpos_issue4<G, IDX>(buf[IDX], gbase, w);
    gfpos_prologue_n<G, IDX + 1, NBUF>(buf, gbase, w);
  }
}

template <int NBUF>
struct GfposPipeAsm {
  template <int G>
  static __device__ __forceinline__ void group(u32x4 (&acc)[4], u32 gbase, const u32 (&w)[G][4]) {
    if constexpr (NBUF == 2) {
      u32x4 A[4], B[4];
      gfpos_issue4<G, 0>(A, gbase, w);
      gfpos_pipe<G, 0>(A, B, acc, gbase, w);
    } else {
      u32x4 buf[NBUF][4];
      gfpos_prologue_n<G, 0, NBUF>(buf, gbase, w);
      gfpos_pipe_n<G, 0, NBUF>(buf, acc, gbase, w);
    }
  }
};

// ---- vector-ALU slack probe (round 4): the library's compiler-visible pipeline with NF extra, independent three-input
// operations per batch stage -- how much vector work fits under the table reads before the kernel slows down?  (What a hybrid
// that multiplies a few parties' shares by their coefficients on the vector ALU, 5 instructions per share bit, would add.)
template <int G, int IDX, int NF>
__device__ __forceinline__ void gfpos_pipe_fill(u32x4 (&A)[4], u32x4 (&B)[4], u32x4 (&acc)[4], u32 gbase, const u32 (&w)[G][4],
                                                u32 (&f)[4]) {
  constexpr int NB = 8 * G;
  if constexpr (IDX < NB) {
    if constexpr (IDX + 1 < NB) gfpos_issue4_cv<G, IDX + 1>((IDX & 1) ? A : B, gbase, w);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NF; ++q) f[q & 3] = xor3(f[q & 3], f[(q + 1) & 3], w[q % G][q & 3]);
    gfpos_fold(acc[IDX / (2 * G)], (IDX & 1) ? B : A);
    __builtin_amdgcn_sched_barrier(0);
    gfpos_pipe_fill<G, IDX + 1, NF>(A, B, acc, gbase, w, f);
  }
}
template <int NF>
struct GfposPipeFill {
  template <int G>
  static __device__ __forceinline__ void group(u32x4 (&acc)[4], u32 gbase, const u32 (&w)[G][4]) {
    u32x4 A[4], B[4];
    u32 f[4] = {w[0][0], w[0][1], w[0][2], w[0][3]};
    gfpos_issue4_cv<G, 0>(A, gbase, w);
    gfpos_pipe_fill<G, 0, NF>(A, B, acc, gbase, w, f);
    asm volatile("" ::"v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]));  // (kept alive; the results are not used)
  }
};

}  // namespace sclhip
