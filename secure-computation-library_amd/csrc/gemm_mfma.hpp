// csrc/gemm_mfma.hpp -- Matrix::multiply over Mersenne61 for GENERAL shapes on the matrix cores (reference: the unbounded
// i-k-j loop of include/scl/math/matrix.h:477-495).
//
// share_mfma.hpp's kernels are built for sharing: a left factor of at most 128 x 64 resident in LDS or registers, the right
// factor streamed and recoded once -- its inner dimension is one launch.  Here K is a loop and BOTH factors are big, so the
// digit recoding is a pass of its own and the inner loop is matrix instructions only:
//
//   * every 61-bit value is recoded into 8 signed base-256 digits (mf_recode, share_mfma.hpp); with A = sum_l A_l 2^(8l) and
//     B = sum_m B_m 2^(8m),  A B = sum_{d=0..14} 2^(8d) E_d,  E_d = sum_{l+m=d} A_l B_m  (i8 products, exact in int32);
//   * k_gemm_planes_{a,b} write the digit planes in FRAGMENT order: for a 32 x 32 (rows x k) tile of A and digit l, lane
//     r + 32 h of a wave gets the 16 bytes A_l[row r][k = 16 h .. 16 h + 15]; for B, lane c + 32 h gets B_m[k = 16 h ..][col c].
//     Both operands use the same k-slot order, so the contraction does not depend on the instruction's internal k order.  A
//     wave's fragment is 1 KiB of consecutive bytes: the main kernel loads operands straight from global memory (L2 / L1),
//     coalesced, with no LDS at all;
//   * k_gemm_mfma_m61: a wave owns a 32 x 32 tile of C and ALL FIFTEEN diagonal accumulators (240 registers); per k-step of 32
//     it loads 8 + 8 fragments and issues the 64 digit-pair v_mfma_i32_32x32x32_i8.  |E_d| <= 8 K 2^14, so 8192 inner columns
//     (256 k-steps) fit an int32; then the diagonals are recombined into a 61-bit residue -- four diagonals to a 32-bit-aligned
//     word by three v_mad_i64_i32, the four words placed by rotations in the 61-bit ring, a bias keeping the words positive, as
//     in share_mfma.hpp with wider bounds -- and added to the running sum ("super-steps": K has no bound);
//   * a workgroup is four waves = a 64 x 64 tile of C, so that the A tile of a wave row and the B tile of a wave column are
//     fetched by two waves of the same CU back to back (L1 hits).
//
// Work per operand byte: a k-step moves 16 KiB per wave for 64 matrix instructions (>= 2048 cycles of the matrix pipe), so the
// kernel is bound by the matrix pipe, not by L2.  Measured: profiles/r5_probe_matmul.txt (19.2 T multiply-adds/s at 4096^3: 0.55 of
// the matrix instruction's own measured issue rate, profiles/r1_mfma_chain_issue.txt).  Compiled as a unit of its own without
// -amdgpu-mfma-vgpr-form (gemm_unit.hip): the accumulators belong in the accumulation registers here.
#pragma once

#include <hip/hip_runtime.h>

#include "share_mfma.hpp"

namespace sclhip {

constexpr int GEMM_SUPER = 256;  // k-steps of 32 per super-step: 8 digit pairs x 8192 x 2^14 < 2^31

// Word bias for K up to 8192 per super-step: |T_a| < 2^30 (1 + 2^8 + 2^16 + 2^24) < 2^55; every word gets 2^57 added, and
//   2^57 (1 + 2^32 + 2^64 + 2^96) = 2^57 + 2^28 + 2^60 + 2^31   (mod 2^61 - 1)
// is taken off once per super-step.
constexpr u64 GEMM_WORD_BIAS = 1ull << 57;
constexpr u64 GEMM_TOTAL_BIAS = (1ull << 57) + (1ull << 28) + (1ull << 60) + (1ull << 31);

// digit planes of A[M x K] (row-major, lda) in fragment order: [row tile][k tile][digit][lane] 16-byte units
template <int = 0>
__global__ __launch_bounds__(256) void k_gemm_planes_a(u64x2* planes, const u64* A, size_t lda, size_t M, size_t K, size_t ktiles) {
  const size_t mtiles = (M + 31) / 32;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < mtiles * ktiles * 64; idx += (size_t)gridDim.x * 256) {
    const size_t lane = idx & 63, tile = idx >> 6, kt = tile % ktiles, mt = tile / ktiles;
    const size_t row = mt * 32 + (lane & 31), k0 = kt * 32 + 16 * (lane >> 5);
    u64 dig[MF_LIMBS][2] = {};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const u64 x = (row < M && k0 + j < K) ? A[row * lda + k0 + j] : 0;
      const u64 d = mf_recode(x);
#pragma unroll
      for (int l = 0; l < MF_LIMBS; ++l) dig[l][j >> 3] |= ((d >> (8 * l)) & 0xFFull) << (8 * (j & 7));
    }
#pragma unroll
    for (int l = 0; l < MF_LIMBS; ++l) {
      u64x2 v;
      v.x = dig[l][0];
      v.y = dig[l][1];
      planes[(tile * MF_LIMBS + l) * 64 + lane] = v;
    }
  }
}

// digit planes of B[K x N] (row-major, ldb): [column tile][k tile][digit][lane]; lane c + 32 h holds k = 16 h .. 16 h + 15 of column c
template <int = 0>
__global__ __launch_bounds__(256) void k_gemm_planes_b(u64x2* planes, const u64* B, size_t ldb, size_t K, size_t N, size_t ktiles) {
  const size_t ntiles = (N + 31) / 32;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < ntiles * ktiles * 64; idx += (size_t)gridDim.x * 256) {
    const size_t lane = idx & 63, tile = idx >> 6, kt = tile % ktiles, nt = tile / ktiles;
    const size_t col = nt * 32 + (lane & 31), k0 = kt * 32 + 16 * (lane >> 5);
    u64 dig[MF_LIMBS][2] = {};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const u64 x = (col < N && k0 + j < K) ? B[(k0 + j) * ldb + col] : 0;  // (consecutive lanes: consecutive columns)
      const u64 d = mf_recode(x);
#pragma unroll
      for (int l = 0; l < MF_LIMBS; ++l) dig[l][j >> 3] |= ((d >> (8 * l)) & 0xFFull) << (8 * (j & 7));
    }
#pragma unroll
    for (int l = 0; l < MF_LIMBS; ++l) {
      u64x2 v;
      v.x = dig[l][0];
      v.y = dig[l][1];
      planes[(tile * MF_LIMBS + l) * 64 + lane] = v;
    }
  }
}

// the fifteen diagonals of one super-step -> a canonical residue per element, added to run[]
__device__ __forceinline__ void gemm_recombine(const v16i (&acc)[15], u64 (&run)[16]) {
  int m8 = 256, m16 = 65536, m24 = 16777216, m0 = 1;
  asm volatile("" : "+s"(m8), "+s"(m16), "+s"(m24), "+s"(m0));  // keep the products v_mad_i64_i32 (share_mfma.hpp, mf_word)
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    u64 S = 0;
#pragma unroll
    for (int a = 3; a >= 0; --a) {
      long long T = (long long)acc[4 * a][e] * m0 + (long long)GEMM_WORD_BIAS;
      T = (long long)acc[4 * a + 1][e] * m8 + T;
      T = (long long)acc[4 * a + 2][e] * m16 + T;
      if (a < 3) T = (long long)acc[4 * a + 3][e] * m24 + T;
      const u64 w = (u64)T;  // 0 < w < 2^58
      if (a == 3) S = rotl61(w, 35);    // 2^96
      if (a == 2) S += w << 3;          // 2^64
      if (a == 1) S += rotl61(w, 32);   // 2^32
      if (a == 0) S += w;
    }
    const u64 sum = S + (M61::P - GEMM_TOTAL_BIAS);  // < 2^64
    const u64 f = (sum & M61::P) + (sum >> 61);
    run[e] = M61::add(M61::Ctx{}, run[e], f >= M61::P ? f - M61::P : f);
  }
}

// gridDim.y > 1: split-K -- slice y of the k-steps ([y * kslice, (y + 1) * kslice)) writes its canonical partial product to
// C + y * cslice; the host adds the slices (few output tiles and a long inner dimension would otherwise leave most CUs idle)
template <int = 0>
__global__ __launch_bounds__(256, 1) void k_gemm_mfma_m61(u64* C, size_t ldc, const u64x2* Ap, const u64x2* Bp, size_t M, size_t N,
                                                          size_t ktiles, size_t kslice, size_t cslice) {
  const size_t mtiles = (M + 31) / 32, ntiles = (N + 31) / 32;
  const size_t wg_n = (ntiles + 1) / 2, wg_m = (mtiles + 1) / 2;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (size_t wg = blockIdx.x; wg < wg_m * wg_n; wg += gridDim.x) {
    const size_t mt = (wg / wg_n) * 2 + (w >> 1), nt = (wg % wg_n) * 2 + (w & 1);
    if (mt >= mtiles || nt >= ntiles) continue;  // (a wave of an edge workgroup without a tile; no barrier in this kernel)
    const v4i* a_tile = reinterpret_cast<const v4i*>(Ap) + (mt * ktiles) * MF_LIMBS * 64 + lane;
    const v4i* b_tile = reinterpret_cast<const v4i*>(Bp) + (nt * ktiles) * MF_LIMBS * 64 + lane;
    u64 run[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) run[e] = 0;
    const size_t kbeg = blockIdx.y * kslice, kend = kbeg + kslice < ktiles ? kbeg + kslice : ktiles;
    for (size_t k0 = kbeg; k0 < kend; k0 += GEMM_SUPER) {
      const size_t k1 = k0 + GEMM_SUPER < kend ? k0 + GEMM_SUPER : kend;
      v16i acc[15];
      const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int d = 0; d < 15; ++d) acc[d] = zero;
      // software pipeline over the k-steps, two register images of the sixteen fragments: with ONE wave per SIMD (the accumulators
      // take half the register file) nothing else hides the loads' latency, so step kt + 1 is requested before step kt's 64 matrix
      // instructions are issued (2048+ cycles of matrix pipe per step against ~1-2 us of L2 latency).  A third image -- two steps
      // ahead -- does not fit: 144 spills (round 5).
      auto fetch = [&](v4i (&af)[MF_LIMBS], v4i (&bf)[MF_LIMBS], size_t kt) {
#if defined(GEMM_PROBE_NO_LOADS)  // tools/gemm_bench.hip: every step reads step 0's fragments (L1 hits): the matrix pipe's own ceiling
        kt = 0;
#endif
#pragma unroll
        for (int l = 0; l < MF_LIMBS; ++l) {
          af[l] = a_tile[(kt * MF_LIMBS + l) * 64];
          bf[l] = b_tile[(kt * MF_LIMBS + l) * 64];
        }
      };
      auto contract = [&](const v4i (&af)[MF_LIMBS], const v4i (&bf)[MF_LIMBS]) {
#pragma unroll
        for (int l = 0; l < MF_LIMBS; ++l)
#pragma unroll
          for (int m = 0; m < MF_LIMBS; ++m) acc[l + m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[l], bf[m], acc[l + m], 0, 0, 0);
      };
      v4i a0[MF_LIMBS], b0[MF_LIMBS], a1[MF_LIMBS], b1[MF_LIMBS];
      fetch(a0, b0, k0);
      size_t kt = k0;
      for (; kt + 2 <= k1; kt += 2) {
        fetch(a1, b1, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        contract(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < k1) fetch(a0, b0, kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        contract(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kt < k1) contract(a0, b0);  // an odd last step (its fragments were requested in the loop, or before it)
      gemm_recombine(acc, run);
    }
    // element e of the lane: row (e & 3) + 8 (e >> 2) + 4 (lane >> 5), column lane & 31 of the tile (share_mfma.hpp)
    const size_t col = nt * 32 + (lane & 31);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const size_t row = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      if (row < M && col < N) C[blockIdx.y * cslice + row * ldc + col] = run[e];
    }
  }
}

}  // namespace sclhip
