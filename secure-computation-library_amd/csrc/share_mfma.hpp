// csrc/share_mfma.hpp -- Shamir sharing over Mersenne61 as a dense integer contraction on the matrix cores.
//
// shares[n x N] = V[n x (t+1)] * C[(t+1) x N] mod p with V[i][k] = alpha_i^k (Matrix::vandermonde times
// the coefficient matrix, reference include/scl/math/matrix.h:444-460,477-495 and
// test/scl/math/test_matrix.cc:342-365).  For large (n, t) -- BASELINE configs[4]: n=128, t=42, i.e.
// 5 504 modular multiply-adds per secret -- the VALU Horner kernel is compute-bound far below HBM.
//
// Formulation: every 61-bit value is split into 9 limbs of 7 bits (signed-i8-safe).  With
// V = sum_l V_l 2^(7l) and C = sum_m C_m 2^(7m),
//     V*C = sum_{d=0..16} 2^(7d) E_d,   E_d = sum_{l+m=d} V_l * C_m      (exact in int32: < 9*64*127^2 < 2^24)
// Each V_l * C_m is an i8 GEMM on v_mfma_i32_32x32x32_i8; all (l, m) pairs of one diagonal d accumulate
// into the SAME accumulator tile, so the VALU only sees 17 int32 tiles per 32x32 output tile.  They are
// recombined by Horner in d: adjacent diagonals pair up in 32 bits, three pairs make a 64-bit word of 6
// diagonals, and the three words are placed by rotations in the 61-bit ring (2^84 = 2^23 mod 2^61 - 1).
//
// Work split: M = evaluation points (tiles of 32, MT = 1, 2 or 4 tiles), N = secrets (tiles of 32),
// K = coefficients (KS = 1 or 2 steps of 32).  A workgroup is two groups of 4 waves; within a group wave
// w owns the output tile (N-tile w / MT, M-tile w % MT) of the group's 4/MT N-tiles.  V's limb planes live in LDS for the whole
// kernel (rows padded to a stride that makes ds_read_b128 fragment loads conflict-free); the block's
// coefficient limbs are staged into LDS once per block.  k-slot order is the byte order of the LDS rows for both operands,
// which makes the contraction independent of the instruction's internal k permutation.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/scl_hip/detail/field.hpp"

namespace sclhip {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int MF_LIMBS = 9;  // ceil(61 / 7)
constexpr int MF_PAD = 16;   // row padding (bytes): row stride/4 is then 4 * odd -> conflict-free b128 reads

__host__ __device__ constexpr int mf_rowb(int KS) { return KS * 32 + MF_PAD; }
__host__ __device__ constexpr size_t mf_a_bytes(int KS, int MT) { return (size_t)MF_LIMBS * MT * 32 * mf_rowb(KS); }
__host__ __device__ constexpr size_t mf_b_bytes(int KS, int MT) { return (size_t)MF_LIMBS * 2 * (4 / MT) * 32 * mf_rowb(KS); }

// rotate left by s in the 61-bit ring: x * 2^s mod (2^61 - 1) for x < 2^61 (the all-ones pattern maps to itself = 0)
__device__ __forceinline__ u64 rotl61(u64 x, int s) { return ((x << s) & M61::P) | (x >> (61 - s)); }

// One loop trip of the diagonal recombination: the two diagonals d = 2Q+1 and d = 2Q (Q = 8: only d = 16).
// Everything here is straight-line with compile-time offsets and shift counts: MFMA chains into two
// accumulator tiles, then  pairword = E_{2Q+1} * 2^7 + E_{2Q}  (fits 32 bits),  W = W * 2^14 + pairword,
// and when a 6-diagonal chunk is complete (Q = 6, 3, 0) it is rotated into place in the 61-bit ring.
template <int Q, int KS, int MT, int COLS>
__device__ __forceinline__ void mf_pair(const unsigned char* arow, const unsigned char* brow, u64 (&S)[16], u64 (&W)[16]) {
  constexpr int ROWB = mf_rowb(KS);
  constexpr int L = MF_LIMBS;
  v16i acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    acc0[e] = 0;
    acc1[e] = 0;
  }
  constexpr int D0 = 2 * Q, D1 = 2 * Q + 1;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    constexpr int dummy = 0;
    (void)dummy;
    const int m0 = D0 - l, m1 = D1 - l;
    if (m0 >= 0 && m0 < L) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const v4i a = *reinterpret_cast<const v4i*>(arow + (size_t)l * MT * 32 * ROWB + ks * 32);
        const v4i b = *reinterpret_cast<const v4i*>(brow + (size_t)m0 * COLS * ROWB + ks * 32);
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
      }
    }
    if (D1 <= 2 * (L - 1) && m1 >= 0 && m1 < L) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const v4i a = *reinterpret_cast<const v4i*>(arow + (size_t)l * MT * 32 * ROWB + ks * 32);
        const v4i b = *reinterpret_cast<const v4i*>(brow + (size_t)m1 * COLS * ROWB + ks * 32);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc1, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const u32 pw = ((u32)acc1[e] << 7) + (u32)acc0[e];
    if constexpr (Q % 3 == 2) W[e] = pw;  // first pair-word of a chunk (Q = 8, 5, 2)
    else W[e] = (W[e] << 14) + pw;
    if constexpr (Q == 6) S[e] = rotl61(W[e], 23);        // 2^84 = 2^23 (mod p)
    if constexpr (Q == 3) S[e] += rotl61(W[e], 42);
    if constexpr (Q == 0) S[e] += W[e];
  }
}

// A table layout (host-built, see mfma_table in capi.hip): [limb l][m-tile][row 0..31][mf_rowb] bytes,
// byte k of a row = limb l of V[mtile*32 + row][k] (0 for k > t or row >= n).
//
// 8 waves per workgroup: waves 0-3 and waves 4-7 work on two different groups of N-tiles against the same
// V planes, so that on every SIMD one wave's recombination (VALU) runs under the other wave's MFMAs.
template <int KS, int MT>
__global__ __launch_bounds__(512) void k_share_mfma_m61(u64* shares, size_t stride, const u64* secrets,
                                                        const u64* coeffs, size_t cstride,
                                                        const unsigned char* Atab, int t, int n, size_t N) {
  constexpr int ROWB = mf_rowb(KS);
  constexpr int NBLK = 4 / MT;            // N-tiles per 4-wave group
  constexpr int COLS = 2 * NBLK * 32;     // secrets per workgroup iteration (two groups)
  constexpr int KG = KS * 8;              // groups of four k-slots
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;
  unsigned char* Bs = smem + mf_a_bytes(KS, MT);  // [limb][2*NBLK n-tiles][32 cols][ROWB]

  {
    const uint4* src = reinterpret_cast<const uint4*>(Atab);
    uint4* dst = reinterpret_cast<uint4*>(As);
    for (int i = threadIdx.x; i < (int)(mf_a_bytes(KS, MT) / 16); i += 512) dst[i] = src[i];
  }
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int nt = (w >> 2) * NBLK + (w & 3) / MT, mt = (w & 3) % MT;
  const size_t nblocks = (N + COLS - 1) / COLS;
  const u64 P = M61::P;

  // coefficient words of the NEXT block travel in registers while the current block is on the matrix cores
  constexpr int ITEMS = (COLS * KG + 511) / 512;  // (column, k-group) items per thread
  u64 creg[ITEMS][4];
  auto fetch = [&](size_t blk) {
    const size_t s_base = blk * COLS;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int e = threadIdx.x + it * 512;
      const int col = e % COLS, kg = e / COLS;
      const size_t s = s_base + col;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 4 * kg + j;
        creg[it][j] = 0;
        if (e < COLS * KG && s < N && k <= t)
          creg[it][j] = (k == 0) ? __builtin_nontemporal_load(secrets + s)
                                 : __builtin_nontemporal_load(coeffs + (size_t)(k - 1) * cstride + s);
      }
    }
  };
  if (blockIdx.x < nblocks) fetch(blockIdx.x);

  for (size_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const size_t s_base = blk * COLS;
    __syncthreads();  // everyone is done reading the previous iteration's Bs (and As is in place)
    // ---- split this block's coefficients into 7-bit limbs: Bs[m][ntile*32 + col][k] ----
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int e = threadIdx.x + it * 512;
      if (e < COLS * KG) {
        const int col = e % COLS, kg = e / COLS;
        unsigned char* dst = Bs + (size_t)col * ROWB + 4 * kg;
#pragma unroll
        for (int m = 0; m < MF_LIMBS; ++m) {
          const u32 word = (u32)((creg[it][0] >> (7 * m)) & 127) | ((u32)((creg[it][1] >> (7 * m)) & 127) << 8) |
                           ((u32)((creg[it][2] >> (7 * m)) & 127) << 16) | ((u32)((creg[it][3] >> (7 * m)) & 127) << 24);
          *reinterpret_cast<u32*>(dst + (size_t)m * COLS * ROWB) = word;
        }
      }
    }
    __syncthreads();
    if (blk + gridDim.x < nblocks) fetch(blk + gridDim.x);  // in flight during the MFMA loop below

    const unsigned char* brow = Bs + ((size_t)nt * 32 + r) * ROWB + 16 * h;
    const unsigned char* arow = As + ((size_t)mt * 32 + r) * ROWB + 16 * h;

    // Nine trips, one per pair of diagonals (high to low).  The loop is kept rolled -- with an opaque trip
    // variable so that it is not unrolled and constant-folded back -- because fully unrolled the compiler
    // interleaves many diagonals, keeps their accumulator tiles live and spills.
    u64 S[16], W[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      S[e] = 0;
      W[e] = 0;
    }
    int q = MF_LIMBS - 1;
    asm volatile("" : "+s"(q));
#pragma unroll 1
    for (; q >= 0; --q) {
      switch (q) {
        case 8: mf_pair<8, KS, MT, COLS>(arow, brow, S, W); break;
        case 7: mf_pair<7, KS, MT, COLS>(arow, brow, S, W); break;
        case 6: mf_pair<6, KS, MT, COLS>(arow, brow, S, W); break;
        case 5: mf_pair<5, KS, MT, COLS>(arow, brow, S, W); break;
        case 4: mf_pair<4, KS, MT, COLS>(arow, brow, S, W); break;
        case 3: mf_pair<3, KS, MT, COLS>(arow, brow, S, W); break;
        case 2: mf_pair<2, KS, MT, COLS>(arow, brow, S, W); break;
        case 1: mf_pair<1, KS, MT, COLS>(arow, brow, S, W); break;
        default: mf_pair<0, KS, MT, COLS>(arow, brow, S, W); break;
      }
    }
    // ---- fold and store: element e of the lane is row (e&3) + 8*(e>>2) + 4*h, column r of the tile ----
    const size_t s = s_base + (size_t)nt * 32 + r;
    size_t row_stride = stride;
    asm volatile("" : "+s"(row_stride));  // keep the 16 row addresses out of the loop-invariant set (register pressure)
    if (s < N) {
      u64* colp = shares + s;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (i < n) {
          u64 v = (S[e] & P) + (S[e] >> 61);  // S < 3 * 2^61
          v = v >= P ? v - P : v;
          __builtin_nontemporal_store(v, colp + (size_t)i * row_stride);
        }
      }
    }
  }
}

}  // namespace sclhip
