// csrc/share_mfma.hpp -- Shamir sharing over Mersenne61 as a dense integer contraction on the matrix cores.
//
// shares[n x N] = V[n x (t+1)] * C[(t+1) x N] mod p with V[i][k] = alpha_i^k (Matrix::vandermonde times
// the coefficient matrix, reference include/scl/math/matrix.h:444-460,477-495 and
// test/scl/math/test_matrix.cc:342-365).  For large (n, t) -- BASELINE configs[4]: n=128, t=42, i.e.
// 5 504 modular multiply-adds per secret -- the VALU Horner kernel is compute-bound far below HBM.
//
// Formulation: every 61-bit value x is recoded into 8 SIGNED base-256 digits, x = sum_i d_i 2^(8i) with
// d_i in [-128, 127]: the digits are the bytes of (x + 0x8080808080808080) with their top bits flipped
// (the +0x80 per byte makes the natural carries do the recoding).  With V = sum_l V_l 2^(8l) and
// C = sum_m C_m 2^(8m),
//     V*C = sum_{d=0..14} 2^(8d) E_d,   E_d = sum_{l+m=d} V_l * C_m      (|E_d| <= 8*64*128^2 < 2^24, exact in int32)
// Each V_l * C_m is an i8 GEMM on v_mfma_i32_32x32x32_i8; all (l, m) pairs of one diagonal d accumulate
// into the SAME accumulator tile, so the VALU only sees 15 int32 tiles per 32x32 output tile.  Four
// diagonals form one 32-bit-aligned word T_a = sum_b E_{4a+b} 2^(8b) (three v_mad_i64_i32), and the four
// words are placed by rotations in the 61-bit ring (2^64 = 8, 2^96 = 2^35 mod 2^61 - 1).  A bias keeps the
// words positive and is removed as one precomputed constant at the end.
//
// Work split: M = evaluation points (tiles of 32, MT = 1, 2 or 4 tiles), N = secrets (tiles of 32),
// K = coefficients (KS = 1 or 2 steps of 32).  A workgroup is two groups of 4 waves; within a group wave
// w owns the output tile (N-tile w / MT, M-tile w % MT) of the group's 4/MT N-tiles, so that on every
// SIMD one wave's recombination (VALU) runs under the other wave's MFMAs.  V's digit planes live in LDS
// for the whole kernel (rows padded to a stride that makes ds_read_b128 fragment loads conflict-free);
// the block's coefficient digits are staged into LDS once per block while the next block's coefficients
// are already in flight from HBM.  k-slot order is the byte order of the LDS rows for both operands, which
// makes the contraction independent of the instruction's internal k permutation.
#pragma once

#include <hip/hip_runtime.h>

#include <utility>

#include "../../include/scl_hip/detail/field.hpp"

namespace sclhip {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

constexpr int MF_LIMBS = 8;  // signed base-256 digits of a 61-bit value
constexpr int MF_PAD = 16;   // row padding (bytes): row stride/4 is then 4 * odd -> conflict-free b128 reads
constexpr u64 MF_RECODE = 0x8080808080808080ull;

__host__ __device__ constexpr int mf_rowb(int KS) { return KS * 32 + MF_PAD; }
__host__ __device__ constexpr size_t mf_a_bytes(int KS, int MT) { return (size_t)MF_LIMBS * MT * 32 * mf_rowb(KS); }
__host__ __device__ constexpr size_t mf_b_bytes(int KS, int MT, int groups = 2) {
  return (size_t)MF_LIMBS * groups * (4 / MT) * 32 * mf_rowb(KS);
}

// signed digits of x (x < 2^61) as the 8 bytes of the result (two's complement i8 each)
__host__ __device__ inline u64 mf_recode(u64 x) { return (x + MF_RECODE) ^ MF_RECODE; }

// rotate left by s in the 61-bit ring: x * 2^s mod (2^61 - 1) for x < 2^61
__device__ __forceinline__ u64 rotl61(u64 x, int s) { return ((x << s) & M61::P) | (x >> (61 - s)); }

// Word bias: every T_a gets 2^50 added (|T_a| < 2^49); the total bias
//   2^50 (1 + 2^32 + 2^64 + 2^96) = 2^50 + 2^21 + 2^53 + 2^24   (mod 2^61 - 1)
// is subtracted once at the end.
constexpr u64 MF_WORD_BIAS = 1ull << 50;
constexpr u64 MF_TOTAL_BIAS = (1ull << 50) + (1ull << 21) + (1ull << 53) + (1ull << 24);

// One loop trip: the four diagonals d = 4A .. 4A+3 (A = 3: d = 12, 13, 14) -> the word T_A, rotated into S.
// Everything is straight-line with compile-time offsets: MFMA chains into up to four accumulator tiles, then
//   T = E_{4A} + E_{4A+1} 2^8 + E_{4A+2} 2^16 + E_{4A+3} 2^24 + 2^50      (signed 64-bit multiply-adds)
template <int A, int KS, int MT, int COLS>
__device__ __forceinline__ void mf_word(const unsigned char* arow, const unsigned char* brow, u64 (&S)[16]) {
  constexpr int ROWB = mf_rowb(KS);
  constexpr int L = MF_LIMBS;
  constexpr int DMAX = 2 * (L - 1);
  v16i acc[4];
  const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int l = 0; l < L; ++l) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int d = 4 * A + j, m = d - l;
      if (d <= DMAX && m >= 0 && m < L) {
        const int l_first = d > L - 1 ? d - (L - 1) : 0;  // the chain's first MFMA takes the inline-constant 0 as C
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const v4i a = *reinterpret_cast<const v4i*>(arow + (size_t)l * MT * 32 * ROWB + ks * 32);
          const v4i b = *reinterpret_cast<const v4i*>(brow + (size_t)m * COLS * ROWB + ks * 32);
          acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, (l == l_first && ks == 0) ? zero : acc[j], 0, 0, 0);
        }
      }
    }
  }
  // multipliers kept opaque so that the products stay v_mad_i64_i32 (one instruction each) instead of being
  // strength-reduced into sign-extend + 64-bit shift + 64-bit add
  int m8 = 256, m16 = 65536, m24 = 16777216, m0 = 1;
  asm volatile("" : "+s"(m8), "+s"(m16), "+s"(m24), "+s"(m0));
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    long long T = (long long)acc[0][e] * m0 + (long long)MF_WORD_BIAS;
    T = (long long)acc[1][e] * m8 + T;
    T = (long long)acc[2][e] * m16 + T;
    if (4 * A + 3 <= DMAX) T = (long long)acc[3][e] * m24 + T;
    const u64 w = (u64)T;  // 0 < w < 2^51
    if constexpr (A == 3) S[e] = rotl61(w, 35);   // 2^96
    if constexpr (A == 2) S[e] += w << 3;         // 2^64
    if constexpr (A == 1) S[e] += rotl61(w, 32);  // 2^32
    if constexpr (A == 0) S[e] += w;
  }
}

// Register-resident variant.  V's digit fragments of this wave's row tile never change, so they are loaded once
// per kernel (KS * 8 fragments, 4 VGPRs each) and the LDS only feeds the coefficient digits: one ds_read_b128 per
// (digit plane m, k-step), reused by the up to four diagonals of the word that pair it with a V digit -- 16 LDS
// fragment loads per word instead of the 48 of mf_word.  The four MFMAs that share a B fragment go to four
// different accumulator tiles, so consecutive matrix instructions are independent.
template <int A, int KS, int MT, int COLS>
__device__ __forceinline__ void mf_word_r(const v4i (&afrag)[MF_LIMBS][KS], const unsigned char* brow, u64 (&S)[16]) {
  constexpr int ROWB = mf_rowb(KS);
  constexpr int L = MF_LIMBS;
  constexpr int DMAX = 2 * (L - 1);
  // digit planes of C this word touches: m = d - l over its diagonals d = 4A .. min(4A+3, DMAX), 0 <= l < L
  constexpr int M_LO = 4 * A - (L - 1) > 0 ? 4 * A - (L - 1) : 0;
  constexpr int M_HI = (4 * A + 3 < L - 1 ? 4 * A + 3 : L - 1);
  v16i acc[4];
  const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // software pipeline: the fragments of plane m+1 are requested before the matrix instructions of plane m are
  // issued (sched_barrier pins that order), so the LDS latency hides under up to 4*KS MFMAs
  v4i bcur[KS], bnxt[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) bcur[ks] = *reinterpret_cast<const v4i*>(brow + (size_t)M_LO * COLS * ROWB + ks * 32);
#pragma unroll
  for (int m = M_LO; m <= M_HI; ++m) {
    if (m < M_HI) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        bnxt[ks] = *reinterpret_cast<const v4i*>(brow + (size_t)(m + 1) * COLS * ROWB + ks * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d = 4 * A + j, l = d - m;
        if (d <= DMAX && l >= 0 && l < L) {
          const int m_first = d > L - 1 ? d - (L - 1) : 0;  // the chain's first MFMA takes the inline-constant 0 as C
          acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[l][ks], bcur[ks], (m == m_first && ks == 0) ? zero : acc[j], 0,
                                                         0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bcur[ks] = bnxt[ks];
  }
  int m8 = 256, m16 = 65536, m24 = 16777216, m0 = 1;
  asm volatile("" : "+s"(m8), "+s"(m16), "+s"(m24), "+s"(m0));
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    long long T = (long long)acc[0][e] * m0 + (long long)MF_WORD_BIAS;
    T = (long long)acc[1][e] * m8 + T;
    T = (long long)acc[2][e] * m16 + T;
    if (4 * A + 3 <= DMAX) T = (long long)acc[3][e] * m24 + T;
    const u64 w = (u64)T;  // 0 < w < 2^51
    if constexpr (A == 3) S[e] = rotl61(w, 35);   // 2^96
    if constexpr (A == 2) S[e] += w << 3;         // 2^64
    if constexpr (A == 1) S[e] += rotl61(w, 32);  // 2^32
    if constexpr (A == 0) S[e] += w;
  }
}

// A table layout (host-built, see mfma_table in capi.hip): [digit l][m-tile][row 0..31][mf_rowb] bytes,
// byte k of a row = signed digit l of V[mtile*32 + row][k] (0 for k > t or row >= n).
// TPB = 512: one workgroup of two 4-wave groups per CU.  TPB = 256: one 4-wave group per workgroup, two workgroups
// per CU -- their barriers are independent, so one workgroup's matrix phase runs under the other's VALU / store
// phase instead of all eight waves of the CU moving in step.
// ACC: the product is ADDED (mod p) to what `shares` holds -- the k-chunks after the first of a Matrix::multiply whose inner
// dimension is longer than one launch's 64 (canonical partial products add exactly); sharing itself never sets it.
template <int KS, int MT, bool AREG = false, int TPB = 512, bool ACC = false>
__global__ __launch_bounds__(TPB, 512 / TPB) void k_share_mfma_m61(u64* shares, size_t stride, const u64* secrets,
                                                        const u64* coeffs, size_t cstride,
                                                        const unsigned char* Atab, int t, int n, size_t N) {
  constexpr int ROWB = mf_rowb(KS);
  constexpr int NBLK = 4 / MT;         // N-tiles per 4-wave group
  constexpr int GROUPS = TPB / 256;
  constexpr int COLS = GROUPS * NBLK * 32;  // secrets per workgroup iteration
  constexpr int KG = KS * 8;           // groups of four k-slots
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // AREG: V's digit planes go from the table in global memory straight into registers and take no LDS
  unsigned char* As = smem;
  unsigned char* Bs = smem + (AREG ? 0 : mf_a_bytes(KS, MT));  // [digit][GROUPS*NBLK n-tiles * 32 cols][ROWB]

  if constexpr (!AREG) {
    const uint4* src = reinterpret_cast<const uint4*>(Atab);
    uint4* dst = reinterpret_cast<uint4*>(As);
    for (int i = threadIdx.x; i < (int)(mf_a_bytes(KS, MT) / 16); i += TPB) dst[i] = src[i];
  }
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int nt = (w >> 2) * NBLK + (w & 3) / MT, mt = (w & 3) % MT;
  const size_t nblocks = (N + COLS - 1) / COLS;
  const u64 P = M61::P;

  // coefficient words of the NEXT block travel in registers while the current block is on the matrix cores
  constexpr int ITEMS = (COLS * KG + TPB - 1) / TPB;  // (column, k-group) items per thread
  u64 creg[ITEMS][4];
  auto fetch = [&](size_t blk) {
    const size_t s_base = blk * COLS;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int e = threadIdx.x + it * TPB;
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
  // recode the coefficients held in creg into signed digits: Bs[digit][ntile*32 + col][k]
  auto recode = [&]() {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int e = threadIdx.x + it * TPB;
      if (e < COLS * KG) {
        const int col = e % COLS, kg = e / COLS;
        unsigned char* dst = Bs + (size_t)col * ROWB + 4 * kg;
        u32 lo[4], hi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u64 u = mf_recode(creg[it][j]);  // 0 stays 0: padded k-slots contribute nothing
          lo[j] = (u32)u;
          hi[j] = (u32)(u >> 32);
        }
#pragma unroll
        for (int m = 0; m < MF_LIMBS; ++m) {
          // byte (m & 3) of the four values -> one 32-bit word, gathered with byte permutes:
          // v_perm_b32(s0, s1, sel): selector bytes 0-3 pick from s1, 4-7 from s0, 0x0c gives zero
          const u32* src = m < 4 ? lo : hi;
          const u32 b = m & 3;
          const u32 sel = 0x0c0c0000u | ((4 + b) << 8) | b;  // result byte0 = s1[b], byte1 = s0[b]
          const u32 p01 = __builtin_amdgcn_perm(src[1], src[0], sel);
          const u32 p23 = __builtin_amdgcn_perm(src[3], src[2], sel);
          *reinterpret_cast<u32*>(dst + (size_t)m * COLS * ROWB) = p01 | (p23 << 16);
        }
      }
    }
  };

  const unsigned char* brow = Bs + ((size_t)nt * 32 + r) * ROWB + 16 * h;
  const unsigned char* arow = (AREG ? Atab : As) + ((size_t)mt * 32 + r) * ROWB + 16 * h;

  // Software pipeline over blocks (b0 = this workgroup's first block, step = gridDim.x):
  //   prologue: fetch(b0); recode -> Bs; fetch(b0 + step)
  //   loop:     MFMA + recombination of block i (reads Bs)           -> S in registers
  //             barrier; recode block i+1 (its loads landed during the MFMA loop) -> Bs; barrier
  //             store block i's results; fetch block i+2
  // so that neither the HBM latency of the coefficient loads nor the completion of the result stores sits
  // between two MFMA loops (vmcnt is in-order: the loads are waited for a whole MFMA loop after the stores
  // that precede them were issued).
  size_t blk = blockIdx.x;
  if (blk < nblocks) fetch(blk);
  __syncthreads();  // As is in place
  v4i afrag[MF_LIMBS][KS];
  if constexpr (AREG) {
#pragma unroll
    for (int l = 0; l < MF_LIMBS; ++l)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        afrag[l][ks] = *reinterpret_cast<const v4i*>(arow + (size_t)l * MT * 32 * ROWB + ks * 32);
  }
  if (blk < nblocks) recode();
  __syncthreads();
  if (blk + gridDim.x < nblocks) fetch(blk + gridDim.x);

  for (; blk < nblocks; blk += gridDim.x) {
    const size_t s_base = blk * COLS;
    // Four trips, one per 32-bit word of diagonals (high to low).  The loop is kept rolled -- with an opaque
    // trip variable so that it is not unrolled and constant-folded back -- because fully unrolled the compiler
    // interleaves all diagonals, keeps their accumulator tiles live and spills.
    u64 S[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) S[e] = 0;
    int a = 3;
    asm volatile("" : "+s"(a));
#pragma unroll 1
    for (; a >= 0; --a) {
      if constexpr (AREG) {
        switch (a) {
          case 3: mf_word_r<3, KS, MT, COLS>(afrag, brow, S); break;
          case 2: mf_word_r<2, KS, MT, COLS>(afrag, brow, S); break;
          case 1: mf_word_r<1, KS, MT, COLS>(afrag, brow, S); break;
          default: mf_word_r<0, KS, MT, COLS>(afrag, brow, S); break;
        }
      } else {
        switch (a) {
          case 3: mf_word<3, KS, MT, COLS>(arow, brow, S); break;
          case 2: mf_word<2, KS, MT, COLS>(arow, brow, S); break;
          case 1: mf_word<1, KS, MT, COLS>(arow, brow, S); break;
          default: mf_word<0, KS, MT, COLS>(arow, brow, S); break;
        }
      }
    }
    __syncthreads();  // every wave is done reading this block's digits
    if (blk + gridDim.x < nblocks) recode();
    __syncthreads();
    // ---- fold and store.  Element e of the lane is row (e&3) + 8*(e>>2) + 4*h, column r of the tile.  A plain
    // epilogue is 16 dwordx2 stores per lane and is store-ISSUE bound; adjacent lanes (adjacent columns)
    // therefore trade one value per element pair over DPP (quad_perm 1,0,3,2) so that even lanes hold two
    // consecutive columns of row(e) and odd lanes two consecutive columns of row(e+1): 8 dwordx4 stores.
    const size_t s_even = s_base + (size_t)nt * 32 + (r & ~1);  // first of this lane's two columns
    size_t row_stride = stride;
    asm volatile("" : "+s"(row_stride));  // keep the row addresses out of the loop-invariant set (register pressure)
    const bool odd = lane & 1;
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      u64 v[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u64 sum = S[e + j] + (P - MF_TOTAL_BIAS);  // < 2^63
        const u64 f = (sum & P) + (sum >> 61);
        v[j] = f >= P ? f - P : f;
      }
      const u64 give = odd ? v[0] : v[1];  // what the neighbour needs: its row's value from my column
      const u32 glo = (u32)give, ghi = (u32)(give >> 32);
      const u32 tlo = (u32)__builtin_amdgcn_update_dpp(0, (int)glo, 0xB1, 0xF, 0xF, false);
      const u32 thi = (u32)__builtin_amdgcn_update_dpp(0, (int)ghi, 0xB1, 0xF, 0xF, false);
      const u64 take = ((u64)thi << 32) | tlo;
      const u64 keep = odd ? v[1] : v[0];
      u64x2 out;
      out.x = odd ? take : keep;  // column r & ~1
      out.y = odd ? keep : take;  // column (r & ~1) + 1
      const int i = mt * 32 + ((e + (odd ? 1 : 0)) & 3) + 8 * (e >> 2) + 4 * h;
      if (i < n) {
        u64* dst = shares + (size_t)i * row_stride + s_even;
        if constexpr (ACC) {
          if (s_even < N) out.x = M61::add(M61::Ctx{}, out.x, dst[0]);
          if (s_even + 1 < N) out.y = M61::add(M61::Ctx{}, out.y, dst[1]);
        }
        if (s_even + 1 < N && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
          __builtin_nontemporal_store(out, reinterpret_cast<u64x2*>(dst));
        } else {  // ragged tail or an 8-byte-aligned row: two scalar stores
          if (s_even < N) dst[0] = out.x;
          if (s_even + 1 < N) dst[1] = out.y;
        }
      }
    }
    if (blk + 2 * (size_t)gridDim.x < nblocks) fetch(blk + 2 * (size_t)gridDim.x);
  }
}

// ---- software-pipelined variant (n > 96: four row tiles) -----------------------------------------------------
// Measured on MI355X (tools/coexec_bench.hip): one wave that issues {1 matrix instruction, ~7 integer VALU
// instructions} in turn keeps both pipes busy -- 33 ns per group against 16 ns for the matrix instruction and
// 31 ns for the VALU group alone -- and so do two waves of one SIMD.  The kernel above does not reach that: each
// wave runs a word's 12-52 matrix instructions as one burst and then that word's ~150 VALU instructions, and the
// two waves of a SIMD do so in step (matrix pipe 39 % busy).  Here ONE wave per SIMD (a 4-wave workgroup per CU,
// up to 512 registers) carries two accumulator sets and issues, in source order pinned by sched_barrier, the
// matrix instructions of word a-1 with the recombination of word a spread between them.  Both operands' digit
// fragments sit in registers for the whole block (V's for the whole kernel), so the LDS is only the transposing
// buffer for the recoded coefficients: the next block is recoded into it while this block is on the matrix cores.
struct MfOp {
  int l, m, ks, j;
};
// the i-th matrix instruction of word A: digit planes (l of V, m of C), k-step ks, diagonal j of the word.  The four
// diagonals' chains alternate, so that consecutive matrix instructions never depend on each other.
template <int A, int KS>
__host__ __device__ constexpr MfOp mf_word_op(int i) {
  int c = 0;
  for (int m = 0; m < MF_LIMBS; ++m)
    for (int ks = 0; ks < KS; ++ks)
      for (int j = 0; j < 4; ++j) {
        const int d = 4 * A + j, l = d - m;
        if (d <= 2 * (MF_LIMBS - 1) && l >= 0 && l < MF_LIMBS) {
          if (c == i) return MfOp{l, m, ks, j};
          ++c;
        }
      }
  return MfOp{-1, -1, -1, c};  // j = the number of instructions of the word
}
template <int A, int KS>
__host__ __device__ constexpr int mf_word_nops() { return mf_word_op<A, KS>(1 << 20).j; }
template <int A, int KS>
__host__ __device__ constexpr bool mf_word_first(int i) {  // no earlier instruction of the word feeds the same tile
  for (int k = 0; k < i; ++k)
    if (mf_word_op<A, KS>(k).j == mf_word_op<A, KS>(i).j) return false;
  return true;
}

struct MfMul {  // the recombination multipliers, kept in scalar registers and opaque (see mf_word)
  int m0, m8, m16, m24;
};

// Recombination of word A as 80 unit operations, ordered so that eight independent elements are in flight (a
// dependent v_mad_i64_i32 issues ~20 cycles after its producer, measured in tools/coexec_bench.hip, and there is no
// second wave on the SIMD to fill that gap): for each half h of the 16 elements, stage k = 0..3 adds diagonal
// 4A+k of the eight elements into T (T = sum_j E_{4A+j}[e] 2^(8j) + bias), stage 4 rotates the finished words into S.
constexpr int MF_UNITS = 80;
template <int A, int U>
__device__ __forceinline__ void mf_recombine_unit(const v16i (&acc)[4], const MfMul& mm, u64 (&T)[8], u64 (&S)[16]) {
  constexpr int h = U / 40, k = (U % 40) / 8, i = U % 8, e = 8 * h + i;
  if constexpr (k == 0) T[i] = (u64)((long long)acc[0][e] * mm.m0 + (long long)MF_WORD_BIAS);
  if constexpr (k == 1) T[i] = (u64)((long long)acc[1][e] * mm.m8 + (long long)T[i]);
  if constexpr (k == 2) T[i] = (u64)((long long)acc[2][e] * mm.m16 + (long long)T[i]);
  if constexpr (k == 3 && 4 * A + 3 <= 2 * (MF_LIMBS - 1)) T[i] = (u64)((long long)acc[3][e] * mm.m24 + (long long)T[i]);
  if constexpr (k == 4) {
    const u64 w = T[i];  // 0 < w < 2^51
    if constexpr (A == 3) S[e] = rotl61(w, 35);   // 2^96
    if constexpr (A == 2) S[e] += w << 3;         // 2^64
    if constexpr (A == 1) S[e] += rotl61(w, 32);  // 2^32
    if constexpr (A == 0) S[e] += w;
    asm volatile("" : "+v"(S[e]));
  }
  // pin the work here: without a use the optimiser sinks it below the last matrix instruction of the block,
  // which keeps every accumulator tile alive
  asm volatile("" : "+v"(T[i]));
}
template <int A, int U0, int U1>
__device__ __forceinline__ void mf_recombine_units(const v16i (&acc)[4], const MfMul& mm, u64 (&T)[8], u64 (&S)[16]) {
  if constexpr (U0 < U1) {
    mf_recombine_unit<A, U0>(acc, mm, T, S);
    mf_recombine_units<A, U0 + 1, U1>(acc, mm, T, S);
  }
}

// instruction I of word A into accN; after it, the share of word A+1's recombination (accP) that spreads the
// VALU work evenly between the word's matrix instructions
template <int A, int KS, int I, int NM, bool PREV, int ABL, class Side>
__device__ __forceinline__ void mf_pipe_op(const v4i (&afrag)[MF_LIMBS][KS], const v4i (&bfrag)[MF_LIMBS][KS],
                                           v16i (&accN)[4], const v16i (&accP)[4], const MfMul& mm, u64 (&T)[8],
                                           u64 (&S)[16], Side& side) {
  constexpr MfOp op = mf_word_op<A, KS>(I);
  constexpr bool first = mf_word_first<A, KS>(I);
  const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (!(ABL & 1))
    accN[op.j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[op.l][op.ks], bfrag[op.m][op.ks], first ? zero : accN[op.j], 0, 0, 0);
  else if (first) {
    accN[op.j] = zero;
    asm volatile("" : "+v"(accN[op.j]));  // opaque: the recombination is not folded away
  }
  if constexpr (PREV && !(ABL & 2)) mf_recombine_units<A + 1, MF_UNITS * I / NM, MF_UNITS * (I + 1) / NM>(accP, mm, T, S);
  side(std::integral_constant<int, A>{}, std::integral_constant<int, I>{}, std::integral_constant<int, NM>{});
  __builtin_amdgcn_sched_barrier(0);
}
template <int A, int KS, bool PREV, int ABL, class Side, int... Is>
__device__ __forceinline__ void mf_pipe_word(const v4i (&afrag)[MF_LIMBS][KS], const v4i (&bfrag)[MF_LIMBS][KS],
                                             v16i (&accN)[4], const v16i (&accP)[4], const MfMul& mm, u64 (&T)[8],
                                             u64 (&S)[16], Side& side, std::integer_sequence<int, Is...>) {
  (mf_pipe_op<A, KS, Is, (int)sizeof...(Is), PREV, ABL>(afrag, bfrag, accN, accP, mm, T, S, side), ...);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits for the result
// stores just issued and for the coefficient loads of the block after next -- with one wave per SIMD that wait is
// fully exposed.  Nothing here communicates through global memory inside the kernel.
__device__ __forceinline__ void mf_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ABL: ablation switches of tools/mfma_bench.hip (1 no matrix instructions, 2 no recombination, 4 no stores,
// 8 no recode / fetch); the library instantiates ABL = 0 only
template <int KS, int ABL = 0>
__global__ __launch_bounds__(256, 1) void k_share_mfma_m61_pipe(u64* shares, size_t stride, const u64* secrets,
                                                                const u64* coeffs, size_t cstride,
                                                                const unsigned char* Atab, int t, int n, size_t N) {
  constexpr int MT = 4, TPB = 256, COLS = 32;
  constexpr int ROWB = mf_rowb(KS);
  constexpr int KG = KS * 8;  // groups of four k-slots
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Bs = smem;  // [digit][32 cols][ROWB]
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int mt = w;
  const size_t nblocks = (N + COLS - 1) / COLS;
  const u64 P = M61::P;

  constexpr int ITEMS = (COLS * KG + TPB - 1) / TPB;  // (column, k-group) items per thread
  static_assert(COLS * KG == ITEMS * TPB, "every thread owns ITEMS whole items");
  u64 creg[ITEMS][4];
  // Item `it` of a thread is column lane & 31 of k-group kg = 2 wave + (lane >> 5) + 8 it: the two halves of a wave
  // hold neighbouring k-groups, so whether a coefficient row exists (k <= t) is decided per wave with scalar
  // branches -- with one wave per SIMD every divergent region's exec-mask bookkeeping is exposed latency.
  const int wu = __builtin_amdgcn_readfirstlane(w);
  auto fetch = [&](size_t blk) {
    const size_t s_base = blk * COLS;
    const size_t s = s_base + r;
    const bool full = s_base + COLS <= N;  // block-uniform
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int kgw = 2 * wu + 8 * it;  // wave-uniform: the k-group of lanes 0..31, lanes 32..63 hold kgw + 1
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k_lo = 4 * kgw + j, k_hi = k_lo + 4;
        const u64* p_lo = k_lo == 0 ? secrets : coeffs + (size_t)(k_lo - 1) * cstride;
        const u64* p_hi = coeffs + (size_t)(k_hi - 1) * cstride;
        const u64* p = (h ? p_hi : p_lo) + s;
        creg[it][j] = 0;
        if (full && k_hi <= t) {
          creg[it][j] = __builtin_nontemporal_load(p);
        } else if (k_lo <= t) {  // the last rows of the polynomial, or the ragged last block
          if ((h ? k_hi : k_lo) <= t && s < N) creg[it][j] = __builtin_nontemporal_load(p);
        }
      }
    }
  };
  auto recode_item = [&](int it) {
    {
      {
        const int col = r, kg = 2 * w + h + 8 * it;
        unsigned char* dst = Bs + (size_t)col * ROWB + 4 * kg;
        u32 lo[4], hi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u64 u = mf_recode(creg[it][j]);
          lo[j] = (u32)u;
          hi[j] = (u32)(u >> 32);
        }
#pragma unroll
        for (int m = 0; m < MF_LIMBS; ++m) {
          const u32* src = m < 4 ? lo : hi;
          const u32 b = m & 3;
          const u32 sel = 0x0c0c0000u | ((4 + b) << 8) | b;
          const u32 p01 = __builtin_amdgcn_perm(src[1], src[0], sel);
          const u32 p23 = __builtin_amdgcn_perm(src[3], src[2], sel);
          *reinterpret_cast<u32*>(dst + (size_t)m * COLS * ROWB) = p01 | (p23 << 16);
        }
      }
    }
  };
  auto recode = [&]() {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) recode_item(it);
  };

  const unsigned char* brow = Bs + (size_t)r * ROWB + 16 * h;
  const u32 brow_lds = (u32)reinterpret_cast<uintptr_t>(brow);  // LDS byte address of this lane's fragment row
  const unsigned char* arow = Atab + ((size_t)mt * 32 + r) * ROWB + 16 * h;
  v4i afrag[MF_LIMBS][KS], bfrag[MF_LIMBS][KS];
#pragma unroll
  for (int l = 0; l < MF_LIMBS; ++l)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      afrag[l][ks] = *reinterpret_cast<const v4i*>(arow + (size_t)l * MT * 32 * ROWB + ks * 32);
      // operand fragments live in the accumulation registers (the matrix instructions read them there directly), which
      // leaves the architectural registers to the accumulator tiles the VALU recombines
      asm volatile("" : "=a"(afrag[l][ks]) : "0"(afrag[l][ks]));
    }
  MfMul mm{1, 256, 65536, 16777216};
  asm volatile("" : "+s"(mm.m0), "+s"(mm.m8), "+s"(mm.m16), "+s"(mm.m24));
  const bool rows_full = __builtin_amdgcn_readfirstlane(mt * 32 + 32 <= n);
  const bool aligned_rows = (reinterpret_cast<uintptr_t>(shares) & 15) == 0 && (stride & 1) == 0;

  size_t blk = blockIdx.x;
  if (blk < nblocks) {
    fetch(blk);
    recode();
  }
  mf_lds_barrier();
  if (blk + gridDim.x < nblocks) fetch(blk + gridDim.x);

  for (; blk < nblocks; blk += gridDim.x) {
    const size_t s_base = blk * COLS;
    // this block's coefficient digits: LDS -> registers, then the LDS is free for the next block's
#pragma unroll
    for (int m = 0; m < MF_LIMBS; ++m)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        // load and wait in one asm statement each (nothing may touch the register before the data is in); straight
        // into the accumulation registers -- the compiler would go through a VGPR and four v_accvgpr_write
        asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)"
                     : "=a"(bfrag[m][ks])
                     : "v"(brow_lds), "n"(m * COLS * ROWB + ks * 32)
                     : "memory");
      }
    mf_lds_barrier();
    // The next block's recode (its coefficients arrived during the previous block) and the fetch of the block after it
    // ride under the matrix instructions of words 2 and 1, whose own recombination work leaves VALU slots free.
    const bool have_next = blk + gridDim.x < nblocks, have_next2 = blk + 2 * (size_t)gridDim.x < nblocks;
    auto side = [&](auto Ac, auto Ic, auto NMc) {
      constexpr int A_ = decltype(Ac)::value, I_ = decltype(Ic)::value, NM_ = decltype(NMc)::value;
      if constexpr (!(ABL & 8)) {
        if constexpr (A_ == 2) {
#pragma unroll
          for (int it = 0; it < ITEMS; ++it)
            if (I_ == (2 * it + 1) * NM_ / (2 * ITEMS)) {
              if (have_next) recode_item(it);
            }
        }
        if constexpr (A_ == 1 && I_ == NM_ / 4) {
          if (have_next2) fetch(blk + 2 * (size_t)gridDim.x);
        }
      }
    };

    u64 S[16], T[8];
    v16i accX[4], accY[4];
    mf_pipe_word<3, KS, false, ABL>(afrag, bfrag, accX, accY, mm, T, S, side, std::make_integer_sequence<int, mf_word_nops<3, KS>()>{});
    mf_pipe_word<2, KS, true, ABL>(afrag, bfrag, accY, accX, mm, T, S, side, std::make_integer_sequence<int, mf_word_nops<2, KS>()>{});
    mf_pipe_word<1, KS, true, ABL>(afrag, bfrag, accX, accY, mm, T, S, side, std::make_integer_sequence<int, mf_word_nops<1, KS>()>{});
    mf_pipe_word<0, KS, true, ABL>(afrag, bfrag, accY, accX, mm, T, S, side, std::make_integer_sequence<int, mf_word_nops<0, KS>()>{});
    if constexpr (!(ABL & 2)) mf_recombine_units<0, 0, MF_UNITS>(accY, mm, T, S);
    else
#pragma unroll
      for (int e = 0; e < 16; ++e) S[e] = (u64)(accY[0][e] ^ accX[1][e] ^ accY[2][e] ^ accX[3][e]);

    // ---- fold and store (as in k_share_mfma_m61: neighbouring lanes trade values for 16-byte stores).  Whole tiles
    // of 16-byte aligned rows -- everything but the ragged edges -- store unconditionally from a running row pointer.
    const size_t s_even = s_base + (size_t)(r & ~1);
    const bool odd = lane & 1;
    const bool fast = rows_full && aligned_rows && s_base + COLS <= N;  // wave-uniform
    u64* rowp = shares + (size_t)(mt * 32 + 4 * h + (odd ? 1 : 0)) * stride + s_even;  // row of element pair 0
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      u64 v[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u64 sum = S[e + j] + (P - MF_TOTAL_BIAS);  // < 2^63
        const u64 f = (sum & P) + (sum >> 61);
        v[j] = f >= P ? f - P : f;
      }
      const u64 give = odd ? v[0] : v[1];
      const u32 glo = (u32)give, ghi = (u32)(give >> 32);
      const u32 tlo = (u32)__builtin_amdgcn_update_dpp(0, (int)glo, 0xB1, 0xF, 0xF, false);
      const u32 thi = (u32)__builtin_amdgcn_update_dpp(0, (int)ghi, 0xB1, 0xF, 0xF, false);
      const u64 take = ((u64)thi << 32) | tlo;
      const u64 keep = odd ? v[1] : v[0];
      u64x2 out;
      out.x = odd ? take : keep;
      out.y = odd ? keep : take;
      // element pair e covers rows (e & 3) + 8 (e >> 2) + {0, 1} of the tile half: +0, +2, +8, +10, +16, ...
      u64* dst = rowp + (size_t)((e & 3) + 8 * (e >> 2)) * stride;
      if (ABL & 4) {
        if (out.x == 0x123456789ull) __builtin_nontemporal_store(out, reinterpret_cast<u64x2*>(dst));
      } else if (fast) {
        __builtin_nontemporal_store(out, reinterpret_cast<u64x2*>(dst));
      } else {
        const int i = mt * 32 + ((e + (odd ? 1 : 0)) & 3) + 8 * (e >> 2) + 4 * h;
        if (i < n) {
          if (s_even + 1 < N && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
            __builtin_nontemporal_store(out, reinterpret_cast<u64x2*>(dst));
          } else {  // ragged tail or an 8-byte-aligned row: two scalar stores
            if (s_even < N) dst[0] = out.x;
            if (s_even + 1 < N) dst[1] = out.y;
          }
        }
      }
    }
    mf_lds_barrier();  // the next block's digits are complete in LDS
  }
}

// ---- two pipelined waves per SIMD on 16x16x64 tiles ---------------------------------------------------------------
// k_share_mfma_m61_pipe is bound by what a lone wave can issue (its VALU side alone takes longer than the matrix
// instructions); a second wave on the SIMD needs the whole working set under 256 registers.  v_mfma_i32_16x16x64_i8
// does that: the same multiply-accumulate rate, all 64 k-slots in one instruction, and a 16x16 tile is 4 accumulator
// registers instead of 16.  Eight waves per workgroup; wave w owns parties 16w .. 16w+15 for the block's 32 secrets
// (two tiles): accumulators 2 sets x 4 diagonals x 2 tiles x 4 = 64 VGPRs, V's digit fragments 32 and the block's
// coefficient fragments 64 accumulation registers.  Rows of a tile are parties (A operand: V's digits), columns are secrets
// (B operand: the coefficient digits), and column c of tile ct is secret 2 c + ct of the block (the recode writes secret s to
// image row 16 (s & 1) + (s >> 1)): a lane ends up with two CONSECUTIVE secrets of each of four parties, one 16-byte store
// per party, and sixteen adjacent lanes write 256 contiguous bytes of a party's row -- four store instructions per trip and
// wave, each two whole 128-byte lines of four rows.  (Round 2: the product transposed, a lane held four consecutive secrets
// of one party, four stores that each touched 64 bytes of 16 rows.  First half of round 3: one secret per lane, eight 8-byte
// stores of whole lines.  Per store instruction the kernel pays about 80 cycles of issue whatever it carries: 8 -> 4
// instructions was 4.8 % of the kernel, profiles/r3_p16_stamps.txt.)  Same digit tables, same LDS image of the recoded
// coefficients ([digit][32 columns][row of 64 k-bytes + pad]) as the kernels above.
//
// What bounds it (profiles/r3_p16_ablation.txt, r3_p16_sq.txt, r3_p16_stamps.txt; 128 parties, t = 42, 2 10^7 secrets):
// per trip and wave 128 matrix instructions, ~290 other vector instructions, 71 scalar, 20 LDS, 8 vector-memory.  The two
// waves of a SIMD are WORK-CONSERVING: in-kernel stamps show a wave parked at a barrier for 10-25 % of its trip, yet moving
// that wait around -- waves in step (the older wave of the SIMD wins every arbitration and waits 1 900 cycles per trip for
// the younger), staggered (1 650), staggered with s_setprio by half-trip (both wait 700), the mid-trip barrier 14 to 26
// instructions earlier -- leaves the trip at 7 200-7 500 cycles: while one wave waits the other has the SIMD to itself.
// What the trip costs is the SUM of what both waves issue: 256 matrix instructions of 16 cycles, ~580 vector instructions of
// 4.5, and every vector-memory instruction at ~80 -- so the levers are instruction COUNTS (stores 8 -> 4: -4.8 %; eight
// stores issued back to back at the trip's end instead of between the folds: +8 %; a branch per stored value: +1 %), not
// schedules.  Taking things OUT of the kernel shortens it by: the stores 2.0 ms of 9.2 (by cycles only 1.0: without them the
// chip holds 2.23 GHz instead of 2.0), the matrix instructions 2.4, the recombination 1.1, recode + fetch 1.1, the fragment
// loads 0.5, the barriers 0 -- the parts add, and the clock gives back about half of every saving.  Tried and measured
// equal within +-2 % (tools/mfma_bench.hip switches, not kept): stores deferred into the next trip's matrix instructions one
// at a time; recode + fetch at the end of the trip instead of word 2; a wave tile of 32 parties x 16 secrets (half the LDS
// fragment traffic); s_setprio by half-trip; the mid-trip barrier at other places.
typedef int v4acc __attribute__((ext_vector_type(4)));
struct MfOp16 {
  int l, m, j, ct;
};
template <int A>
__host__ __device__ constexpr MfOp16 mf16_word_op(int i) {
  int c = 0;
  for (int m = 0; m < MF_LIMBS; ++m)
    for (int j = 0; j < 4; ++j) {
      const int d = 4 * A + j, l = d - m;
      if (d <= 2 * (MF_LIMBS - 1) && l >= 0 && l < MF_LIMBS) {
        for (int ct = 0; ct < 2; ++ct) {
          if (c == i) return MfOp16{l, m, j, ct};
          ++c;
        }
      }
    }
  return MfOp16{-1, -1, c, -1};  // j = the number of instructions of the word
}
template <int A>
__host__ __device__ constexpr int mf16_word_nops() { return mf16_word_op<A>(1 << 20).j; }
template <int A>
__host__ __device__ constexpr bool mf16_word_first(int i) {
  for (int k = 0; k < i; ++k)
    if (mf16_word_op<A>(k).j == mf16_word_op<A>(i).j && mf16_word_op<A>(k).ct == mf16_word_op<A>(i).ct) return false;
  return true;
}
// Recombination with TWO running 64-bit sums instead of one word per four diagonals: 2^64 = 2^3 mod p, so word 2's diagonals
// enter the SAME sum as word 0's with multipliers 2^3 times as large (2^3 .. 2^27, still int32), and word 3's the same sum as
// word 1's:   T = sum_j E_j 2^(8j) + E_(8+j) 2^(8j+3),   S = sum_j E_(4+j) 2^(8j) + E_(12+j) 2^(8j+3),   result = T + 2^32 S.
// |T|, |S| < 9 * 2^24 * 2^24.01 < 2^52; each starts from the bias 2^53, removed at the end as 2^53 (1 + 2^32) = 2^53 + 2^24.
// One rotation per element instead of three and no per-word placement step: 32 unit operations per word (stage k = 0..3 adds
// diagonal 4A+k of the lane's eight elements, tile ct = e / 4, register e % 4), 15 v_mad_i64_i32 + one rotation per element
// where the four-word form of the kernels above spends 15 + three rotations + three 64-bit adds.
constexpr int MF16_UNITS = 32;
constexpr u64 MF16_BIAS = 1ull << 53;
constexpr u64 MF16_TOTAL_BIAS = (1ull << 53) + (1ull << 24);
struct MfMul16 {  // scalar registers, opaque (see mf_word)
  int m0, m8, m16, m24, h0, h8, h16, h24;
};
template <int A, int U>
__device__ __forceinline__ void mf16_recombine_unit(const v4acc (&acc)[4][2], const MfMul16& mm, u64 (&T)[8], u64 (&S)[8]) {
  constexpr int k = U / 8, e = U % 8, ct = e / 4, i = e % 4;
  u64& sum = (A & 1) ? S[e] : T[e];
  if constexpr (4 * A + k <= 2 * (MF_LIMBS - 1)) {
    const int mult = A >= 2 ? (k == 0 ? mm.h0 : k == 1 ? mm.h8 : k == 2 ? mm.h16 : mm.h24)
                            : (k == 0 ? mm.m0 : k == 1 ? mm.m8 : k == 2 ? mm.m16 : mm.m24);
    // words 3 and 2 come first and start their sums from the bias; T's also carries the constant of the final fold,
    // P - MF16_TOTAL_BIAS (T stays below 2^62)
    const long long add = (A == 3 && k == 0)   ? (long long)MF16_BIAS
                          : (A == 2 && k == 0) ? (long long)(MF16_BIAS + (M61::P - MF16_TOTAL_BIAS))
                                               : (long long)sum;
    sum = (u64)((long long)acc[k][ct][i] * mult + add);
  }
  asm volatile("" : "+v"(sum));  // pinned in place (see mf_recombine_unit)
}
template <int A, int U0, int U1>
__device__ __forceinline__ void mf16_recombine_units(const v4acc (&acc)[4][2], const MfMul16& mm, u64 (&T)[8], u64 (&S)[8]) {
  if constexpr (U0 < U1) {
    mf16_recombine_unit<A, U0>(acc, mm, T, S);
    mf16_recombine_units<A, U0 + 1, U1>(acc, mm, T, S);
  }
}
// MF16_ABL: ablation switches for tools/mfma_bench.hip (1 no matrix instructions, 2 no recombination, 4 no stores, 8 no recode /
// fetch, 16 no fragment loads, 32 no barriers, 64 twenty of the trip's 128 matrix instructions left out; results are then wrong by
// construction); the library builds with 0
#ifndef MF16_ABL
#define MF16_ABL 0
#endif
// MF16_HOOK(p): empty in the library.  tools/mfma_bench.hip -DMF16_STAMP defines the three hooks to accumulate s_memtime deltas
// between the points of a trip: 0 before / 1 after the trip-start barrier, 2 fragments loaded, 3 before / 4 after the recode,
// 5 fetch issued, 6 before / 7 after the mid-trip barrier, 8 matrix instructions and recombination done, 9 stores issued
#ifndef MF16_HOOK
#define MF16_HOOK(p)
#define MF16_HOOK_DECL
#define MF16_HOOK_END
#endif
template <int A, int I, int NM, bool PREV, class Side>
__device__ __forceinline__ void mf16_pipe_op(const v4i (&vfrag)[MF_LIMBS], const v4i (&cfrag)[2][MF_LIMBS], v4acc (&accN)[4][2],
                                             const v4acc (&accP)[4][2], const MfMul16& mm, u64 (&T)[8], u64 (&S)[8], Side& side) {
  constexpr MfOp16 op = mf16_word_op<A>(I);
  constexpr bool first = mf16_word_first<A>(I);
  const v4acc zero = {0, 0, 0, 0};
  // (MF16_ABL & 64: 5 of every 32 matrix instructions left out -- 20 of a trip's 128, the 54-of-64 an un-padded K would issue)
  constexpr bool dropped = (MF16_ABL & 64) != 0 && (I * 5) % 32 < 5;
  if constexpr (!(MF16_ABL & 1) && !dropped)
    accN[op.j][op.ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(vfrag[op.l], cfrag[op.ct][op.m], first ? zero : accN[op.j][op.ct], 0, 0, 0);
  else if constexpr (first)
    accN[op.j][op.ct] = cfrag[op.ct][op.m] ^ vfrag[op.l];
  if constexpr (PREV && !(MF16_ABL & 2)) mf16_recombine_units<A + 1, MF16_UNITS * I / NM, MF16_UNITS * (I + 1) / NM>(accP, mm, T, S);
  side(std::integral_constant<int, A>{}, std::integral_constant<int, I>{}, std::integral_constant<int, NM>{});
  __builtin_amdgcn_sched_barrier(0);
}
template <int A, bool PREV, class Side, int... Is>
__device__ __forceinline__ void mf16_pipe_word(const v4i (&vfrag)[MF_LIMBS], const v4i (&cfrag)[2][MF_LIMBS], v4acc (&accN)[4][2],
                                               const v4acc (&accP)[4][2], const MfMul16& mm, u64 (&T)[8], u64 (&S)[8], Side& side,
                                               std::integer_sequence<int, Is...>) {
  (mf16_pipe_op<A, Is, (int)sizeof...(Is), PREV>(vfrag, cfrag, accN, accP, mm, T, S, side), ...);
  if constexpr ((MF16_ABL & 2) != 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(accN[j][0]), "v"(accN[j][1]));
  }
}

// Atab: the KS = 2, MT = 4 table of mfma_table (rows = parties, 64 k-bytes + pad per row)
// ACC (Matrix::multiply's k-chunks, see k_share_mfma_m61): the stores add to what the rows hold; the sharing instantiation
// (ACC = false) is the code it was.
template <class FieldG = M61, bool ACC = false>  // (a template so that only the unit that launches it compiles it)
__global__ __launch_bounds__(512, 1) void k_share_mfma_m61_p16(u64* shares, size_t stride, const u64* secrets,
                                                               const u64* coeffs, size_t cstride, const unsigned char* Atab,
                                                               int t, int n, size_t N) {
  constexpr int KS = 2, MT = 4, COLS = 32;
  constexpr int ROWB = mf_rowb(KS);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // two images [digit][32 cols][ROWB] of recoded coefficients: block i is read from image i & 1 while block i + 1 is
  // recoded into the other one (the barriers that order the two: below, at the loop)
  constexpr int IMG = MF_LIMBS * COLS * ROWB;
  unsigned char* Bs = smem;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, kb = lane >> 4, r32 = lane & 31, h = lane >> 5;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const size_t nblocks = (N + COLS - 1) / COLS;
  const u64 P = M61::P;
  const int pbase = 16 * wu;  // the wave's first party

  // one (column, k-group) item per thread: column lane & 31 of k-group 2 wave + (lane >> 5)
  u64 creg[4];
  auto fetch = [&](size_t blk) {
    const size_t s = blk * COLS + r32;
    const bool full = blk * COLS + COLS <= N;  // block-uniform
    const int kgw = 2 * wu;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k_lo = 4 * kgw + j, k_hi = k_lo + 4;
      const u64* p_lo = k_lo == 0 ? secrets : coeffs + (size_t)(k_lo - 1) * cstride;
      const u64* p_hi = coeffs + (size_t)(k_hi - 1) * cstride;
      const u64* p = (h ? p_hi : p_lo) + s;
      creg[j] = 0;
      if (full && k_hi <= t) {
        creg[j] = __builtin_nontemporal_load(p);
      } else if (k_lo <= t) {
        if ((h ? k_hi : k_lo) <= t && s < N) creg[j] = __builtin_nontemporal_load(p);
      }
    }
  };
  auto recode = [&](int img) {
    const int kg = 2 * w + h;
    // secret r32 of the block goes to row 16 (r32 & 1) + (r32 >> 1) of the image: column c of tile ct is then secret 2 c + ct,
    // so a lane's two tiles hold two CONSECUTIVE secrets of each of its four parties -- one 16-byte store per party
    unsigned char* dst = Bs + (size_t)img * IMG + (size_t)(16 * (r32 & 1) + (r32 >> 1)) * ROWB + 4 * kg;
    u32 lo[4], hi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u64 u = mf_recode(creg[j]);
      lo[j] = (u32)u;
      hi[j] = (u32)(u >> 32);
    }
#pragma unroll
    for (int m = 0; m < MF_LIMBS; ++m) {
      const u32* src = m < 4 ? lo : hi;
      const u32 b = m & 3;
      const u32 sel = 0x0c0c0000u | ((4 + b) << 8) | b;
      const u32 p01 = __builtin_amdgcn_perm(src[1], src[0], sel);
      const u32 p23 = __builtin_amdgcn_perm(src[3], src[2], sel);
      *reinterpret_cast<u32*>(dst + (size_t)m * COLS * ROWB) = p01 | (p23 << 16);
    }
  };

  // V's digit fragments of this wave's parties (A operand: k-block lane >> 4 of party lane & 15 of the tile), kernel lifetime
  v4i vfrag[MF_LIMBS], cfrag[2][MF_LIMBS];
  {
    const unsigned char* vrow = Atab + (size_t)(pbase + r16) * ROWB + 16 * kb;
#pragma unroll
    for (int l = 0; l < MF_LIMBS; ++l) {
      vfrag[l] = *reinterpret_cast<const v4i*>(vrow + (size_t)l * MT * 32 * ROWB);
      asm volatile("" : "=a"(vfrag[l]) : "0"(vfrag[l]));
    }
  }
  // rows of a tile are parties (A operand: V's digits), columns are secrets (B operand: the coefficients' digits of secret
  // lane & 15 of the tile): a lane ends up with ONE secret (lane & 15) of four parties (4 (lane >> 4) + i), and a store
  // instruction writes whole 128-byte lines -- 16 adjacent lanes, 16 adjacent secrets of one party
  const u32 crow_lds = (u32)reinterpret_cast<uintptr_t>(Bs + (size_t)r16 * ROWB + 16 * kb);  // that secret's digits, tile 0
  MfMul16 mm{1, 1 << 8, 1 << 16, 1 << 24, 1 << 3, 1 << 11, 1 << 19, 1 << 27};
  asm volatile("" : "+s"(mm.m0), "+s"(mm.m8), "+s"(mm.m16), "+s"(mm.m24), "+s"(mm.h0), "+s"(mm.h8), "+s"(mm.h16), "+s"(mm.h24));
  const bool rows_full = __builtin_amdgcn_readfirstlane(pbase + 16 <= n);
  const bool aligned_rows = (reinterpret_cast<uintptr_t>(shares) & 15) == 0 && (stride & 1) == 0;  // 16-byte stores allowed

  // the order in which a workgroup takes its blocks: MF16_CHUNK consecutive blocks, then on by gridDim.x chunks (1: block
  // b, b + gridDim.x, ..: every workgroup of the grid writes into the same 64 KiB window of a share row at about the same time)
#ifndef MF16_CHUNK
#define MF16_CHUNK 1
#endif
  auto block_of = [&](size_t it) { return ((it / MF16_CHUNK) * gridDim.x + blockIdx.x) * MF16_CHUNK + it % MF16_CHUNK; };
  size_t it = 0;
  size_t blk = block_of(0);
  if (blk < nblocks) {
    fetch(blk);
    recode(0);
  }
  if (block_of(1) < nblocks) fetch(block_of(1));
  // Two barriers per trip, at its start and after the last matrix instruction of word 2, and waves 4-7 (the second wave of
  // each SIMD) run HALF A TRIP BEHIND waves 0-3: a late wave takes one extra barrier here, an early one after the loop, so a
  // late wave's trip start is an early wave's mid-trip.  On every SIMD one wave is then in its first half (fragment loads from
  // the LDS, words 3 and 2, the recode) while the other is in its second (words 1 and 0, most of the recombination, the fold
  // and the stores), instead of both waiting on the LDS and both folding at the same time.  Image protocol: every wave
  // recodes block i+1 in the FIRST half of its trip i (before its mid-trip barrier), so the image is complete at the barrier
  // that starts the early waves' trip i+1; the buffer it goes to was last read by the late waves at the start of their trip
  // i-1, one barrier before the early waves' trip i begins.  (MF16_STAGGER 0: all eight in step, one barrier per trip; 3 %
  // slower, tools/mfma_bench.hip.)
#ifndef MF16_STAGGER
#define MF16_STAGGER 1
#endif
  const bool late = MF16_STAGGER && (wu >> 2);  // wave-uniform
  if (late) mf_lds_barrier();
  MF16_HOOK_DECL

  for (int img = 0; blk < nblocks; ++it, blk = block_of(it), img ^= 1) {
    const size_t s_base = blk * COLS;
    MF16_HOOK(0);
    if constexpr (!(MF16_ABL & 32)) mf_lds_barrier();  // this block's digits are complete in image img; everyone is done reading the other image
    MF16_HOOK(1);
    const u32 crow = crow_lds + (u32)img * IMG;
    // the sixteen loads and their wait are ONE asm statement: nothing the compiler might insert (a copy, a spill)
    // can touch a fragment register before its data has arrived
    static_assert(MF_LIMBS == 8, "eight digit planes per tile below");
#define MF16_LD(ct)                                                                                                    \
  asm volatile("ds_read_b128 %0, %8 offset:%9\n\tds_read_b128 %1, %8 offset:%10\n\tds_read_b128 %2, %8 offset:%11\n\t"      \
               "ds_read_b128 %3, %8 offset:%12\n\tds_read_b128 %4, %8 offset:%13\n\tds_read_b128 %5, %8 offset:%14\n\t"     \
               "ds_read_b128 %6, %8 offset:%15\n\tds_read_b128 %7, %8 offset:%16\n\ts_waitcnt lgkmcnt(0)"                  \
               : "=&a"(cfrag[ct][0]), "=&a"(cfrag[ct][1]), "=&a"(cfrag[ct][2]), "=&a"(cfrag[ct][3]), "=&a"(cfrag[ct][4]),     \
                 "=&a"(cfrag[ct][5]), "=&a"(cfrag[ct][6]), "=&a"(cfrag[ct][7])                                                \
               : "v"(crow), "n"(0 * COLS * ROWB + ct * 16 * ROWB), "n"(1 * COLS * ROWB + ct * 16 * ROWB),                    \
                 "n"(2 * COLS * ROWB + ct * 16 * ROWB), "n"(3 * COLS * ROWB + ct * 16 * ROWB),                               \
                 "n"(4 * COLS * ROWB + ct * 16 * ROWB), "n"(5 * COLS * ROWB + ct * 16 * ROWB),                               \
                 "n"(6 * COLS * ROWB + ct * 16 * ROWB), "n"(7 * COLS * ROWB + ct * 16 * ROWB)                                \
               : "memory")
    if constexpr (!(MF16_ABL & 16)) {
      MF16_LD(0);
      MF16_LD(1);
    } else {
#pragma unroll
      for (int m = 0; m < MF_LIMBS; ++m) asm volatile("" : "=a"(cfrag[0][m]), "=a"(cfrag[1][m]) : "v"(crow));
    }
#undef MF16_LD
    MF16_HOOK(2);
    const bool have_next = block_of(it + 1) < nblocks, have_next2 = block_of(it + 2) < nblocks;
    auto side = [&](auto Ac, auto Ic, auto NMc) {
      constexpr int A_ = decltype(Ac)::value, I_ = decltype(Ic)::value, NM_ = decltype(NMc)::value;
      // the next block's recode (first half of the trip: see the barriers above) and the fetch of the block after it
      if constexpr (A_ == 2 && I_ == NM_ / 4 && !(MF16_ABL & 8)) {
        MF16_HOOK(3);
        if (have_next) recode(img ^ 1);
        MF16_HOOK(4);
      }
      if constexpr (A_ == 2 && I_ == NM_ / 2 && !(MF16_ABL & 8)) {
        if (have_next2) fetch(block_of(it + 2));
        MF16_HOOK(5);
      }
      if constexpr (MF16_STAGGER && A_ == 1 && I_ == 0 && !(MF16_ABL & 32)) {
        MF16_HOOK(6);
        mf_lds_barrier();  // mid-trip
        MF16_HOOK(7);
      }
    };
    u64 S[8], T[8];
    v4acc accX[4][2], accY[4][2];
    mf16_pipe_word<3, false>(vfrag, cfrag, accX, accY, mm, T, S, side, std::make_integer_sequence<int, mf16_word_nops<3>()>{});
    mf16_pipe_word<2, true>(vfrag, cfrag, accY, accX, mm, T, S, side, std::make_integer_sequence<int, mf16_word_nops<2>()>{});
    mf16_pipe_word<1, true>(vfrag, cfrag, accX, accY, mm, T, S, side, std::make_integer_sequence<int, mf16_word_nops<1>()>{});
    mf16_pipe_word<0, true>(vfrag, cfrag, accY, accX, mm, T, S, side, std::make_integer_sequence<int, mf16_word_nops<0>()>{});
    if constexpr (!(MF16_ABL & 2)) {
      mf16_recombine_units<0, 0, MF16_UNITS>(accY, mm, T, S);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        T[e] = (u64)(u32)accY[e & 3][e >> 2][0];
        S[e] = (u64)(u32)accX[e & 3][e >> 2][1];
      }
    }

    MF16_HOOK(8);
    // ---- fold and store: element e = 4 ct + i is party pbase + 4 (lane >> 4) + i, secret s_base + 16 ct + (lane & 15).
    // ONE wave-uniform branch around the whole part, and inside it a party's two values are folded and stored before the next
    // party's are touched: eight stores issued back to back at the end cost 8 % of the kernel (they queue at the memory pipe and
    // the next trip's wait for its coefficients then waits for all of them), a branch per value as much again in scalar work.
    // T already carries P - MF16_TOTAL_BIAS (mf16_recombine_unit).
    auto finish = [&](int e) -> u64 {
      const u32 slo = (u32)S[e], shi = (u32)(S[e] >> 32);                                       // S < 2^55
      const u64 x = ((u64)(slo & 0x1FFFFFFFu) << 32) | __builtin_amdgcn_alignbit(shi, slo, 29);  // 2^32 S mod p: rotl61(S, 32)
      const u64 sum = T[e] + x;                                                                 // < 2^62 + 2^61
      const u64 f = (sum & P) + (sum >> 61);                                                    // <= P + 2
      return (f + ((f + 1) >> 61)) & P;                                                         // canonical, no compare / select
    };
    const bool fast = rows_full && s_base + COLS <= N;  // wave-uniform
    const int p0 = pbase + 4 * kb;
    const size_t s0 = s_base + 2 * r16;  // element e = 4 ct + i: party p0 + i, secret s0 + ct
    u64* rowp = shares + (size_t)p0 * stride + s0;
    // cached stores: an instruction writes 256 contiguous bytes (two whole lines on rows that start on a line) of each of four
    // parties' rows; rows that are only 8-byte aligned take two 8-byte stores per party
    if constexpr ((MF16_ABL & 4) != 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const u64 v = finish(e);
        asm volatile("" ::"v"(v), "v"(rowp));
      }
    } else if (fast && aligned_rows) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u64x2 o;
        o.x = finish(i);
        o.y = finish(4 + i);
        if constexpr (ACC) {
          const u64x2 was = *reinterpret_cast<const u64x2*>(rowp + (size_t)i * stride);
          o.x = M61::add(M61::Ctx{}, o.x, was.x);
          o.y = M61::add(M61::Ctx{}, o.y, was.y);
        }
        *reinterpret_cast<u64x2*>(rowp + (size_t)i * stride) = o;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (fast) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u64* dst = rowp + (size_t)i * stride;
        u64 v0 = finish(i), v1 = finish(4 + i);
        if constexpr (ACC) {
          v0 = M61::add(M61::Ctx{}, v0, dst[0]);
          v1 = M61::add(M61::Ctx{}, v1, dst[1]);
        }
        dst[0] = v0;
        dst[1] = v1;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u64* dst = rowp + (size_t)i * stride;
        u64 v0 = finish(i), v1 = finish(4 + i);
        if constexpr (ACC) {
          if (p0 + i < n && s0 < N) v0 = M61::add(M61::Ctx{}, v0, dst[0]);
          if (p0 + i < n && s0 + 1 < N) v1 = M61::add(M61::Ctx{}, v1, dst[1]);
        }
        if (p0 + i < n && s0 < N) dst[0] = v0;
        if (p0 + i < n && s0 + 1 < N) dst[1] = v1;
      }
    }
    MF16_HOOK(9);
  }
  MF16_HOOK_END
  if (MF16_STAGGER && !late) mf_lds_barrier();  // (the late waves' extra barrier before the loop)
}

// Digit planes of an arbitrary row-major matrix A[M x K] (device memory) in the A-table layout above, so
// that Matrix::multiply (matrix.h:477-495) with a small left factor runs on the same kernel.  `tab` must be
// zero-filled beforehand (padding rows / k-slots).
template <int KS, int MT>
__global__ __launch_bounds__(256) void k_mfma_planes_from_matrix(unsigned char* tab, const u64* A, size_t lda, int M,
                                                                 int K) {
  constexpr int ROWB = mf_rowb(KS);
  for (int e = blockIdx.x * 256 + threadIdx.x; e < M * K; e += gridDim.x * 256) {
    const int i = e / K, k = e % K;
    const u64 digits = mf_recode(A[(size_t)i * lda + k]);
#pragma unroll
    for (int l = 0; l < MF_LIMBS; ++l)
      tab[((size_t)(l * MT + i / 32) * 32 + (i % 32)) * ROWB + k] = (unsigned char)(digits >> (8 * l));
  }
}

}  // namespace sclhip
