// tools/recover_gf_mfma.hpp -- GF(2^128) reconstruct as a GF(2) matrix product on the matrix cores (an experiment: measured,
// correct, slower than the nibble tables -- see the end of this comment; the library does not include it).
//
// shamirRecoverP over the plug-in GF(2^128) (shamir.h:81-104 with innerProd, vector.h:45-52) is  out = sum_i lambda_i * s_i.
// Multiplication by the constant lambda_i is a GF(2)-linear map: a 128 x 128 bit matrix M_i whose column c is
// lambda_i * x^c, and the whole reconstruction is  out = [M_0 | M_1 | ..] . (bits of s_0, s_1, ..)  over GF(2): 128 x 128 m
// bit multiply-adds per secret (655 360 at m = 40).  The nibble tables of k_recover_gf128_pos do those 512 at a time
// per 16-byte LDS lookup and end at the LDS array's rate: 32 lookups per party and secret, 4 LDS cycles per wave each, is
// 4.4 TB/s-equivalent at 100 % of the array, 3.0 measured (profiles/r2_pmc_gf128.txt).  The matrix cores do the same
// 2048 bit multiply-adds per clock and SIMD with v_mfma_scale_f32_32x32x64_f8f6f4 on fp4 operands -- one bit per 4-bit
// element, the densest the hardware has (measured: 16.0-17.2 ns per instruction per SIMD, tools/gf2_mfma_probe.hip) -- and
// leave the vector ALU and most of the LDS array idle:
//   A (32 x 64, from LDS)   = 32 output bits x 64 share bits of M, entries 0 / 2.0 (fp4 0x4)
//   B (64 x 32, registers)  = 64 bits of one party's share for 32 secrets, entries 0 / 0.5 (fp4 0x1): a share word w expands
//                             into the four operand registers as (w >> j) & 0x11111111, j = 0..3 -- seven VALU
//                             instructions per four matrix instructions
//   C (32 x 32 f32)         = exact counts (<= 5120 < 2^24); the result bit is the parity of the count.
// The columns of M are permuted on the host side of the table so that the registers above ARE the operand (element
// e = 8 j + q of lane (r, h) in k-step ks is share bit 32 (2 h + ks) + 4 q + j), and its rows so that lane (r, h) ends up
// with bits 64 h .. 64 h + 63 of secret r's result (row rho of row tile mt is bit 64 ((rho >> 2) & 1) + 16 mt + (rho & 3) +
// 4 (rho >> 3)): the lanes store 8 contiguous bytes each, no shuffles.
// One launch holds the matrices of up to 20 parties in LDS (8 KiB each: all 160 KiB); more parties take further launches
// that xor into the partial result (`prev`), as the table kernels do.
// Operand and result maps of the fp4 form: confirmed with exact 0/1 data (profiles/r3_gf2_mfma_probe.txt).
// MEASURED (profiles/r3_gf2_mfma_recover.txt): bit-exact against the host's Gf128::mul; 25-26 ns per matrix instruction per
// SIMD whatever the prefetch scheme (share words one party, four parties or a group of four ahead; fragment reads one pair
// ahead; per-fragment scalar selects removed), against 16.0-17.2 ns for back-to-back issue on idle operands: on random
// operands the chip holds a lower clock under the matrix load (MI355X_MICROARCH.md, DVFS give-back).  C4's shard: 3.06 ms =
// 0.335 of HBM peak, where k_recover_gf128_pos (nibble tables in LDS) takes 2.7 ms = 0.38.
#pragma once

#include <hip/hip_runtime.h>


#include "../include/scl_hip/detail/field.hpp"

namespace sclhip {

constexpr int GFM_PARTIES = 20;              // parties per launch: 20 x 8 KiB of matrix fragments = the whole LDS
constexpr int GFM_FRAG_WORDS = 2 * 4 * 64 * 4;  // per party: 2 k-steps x 4 row tiles x 64 lanes x 4 words
constexpr int GFM_BLOCK = 512;               // 8 waves, two per SIMD

typedef int gfm_v8i __attribute__((ext_vector_type(8)));
typedef float gfm_v16f __attribute__((ext_vector_type(16)));

// the bit of the result that row `rho` of row tile `mt` produces
__host__ __device__ constexpr int gfm_out_bit(int mt, int rho) { return 64 * ((rho >> 2) & 1) + 16 * mt + (rho & 3) + 4 * (rho >> 3); }
// the bit of the share that element e = 8 j + q of lane half h holds in k-step ks
__host__ __device__ constexpr int gfm_share_bit(int ks, int h, int j, int q) { return 32 * (2 * h + ks) + 4 * q + j; }

struct GfmLambdas {
  u128 v[GFM_PARTIES];
};

// The matrix fragments of m <= 20 parties: tab[((i * 2 + ks) * 4 + mt) * 64 + lane] = four words, nibble q of word j is
// 0x4 where bit gfm_out_bit(mt, lane & 31) of lambda_i * x^gfm_share_bit(ks, lane >> 5, j, q) is set.  One workgroup per
// party: the 128 products lambda_i * x^c by repeated multiplication by x, staged in LDS, then 2048 words by 256 threads.
__global__ __launch_bounds__(256) void k_gf_mfma_table(uint4* tab, GfmLambdas lam) {
  __shared__ u128 col[128];
  const int i = blockIdx.x;
  if (threadIdx.x < 128) {
    // thread c: lambda * x^c (c steps of mulx; the longest chain, 127 steps, is a few hundred instructions)
    u128 v = lam.v[i];
    for (int c = 0; c < (int)threadIdx.x; ++c) v = Gf128::mulx(v);
    col[threadIdx.x] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * 4 * 64; e += 256) {
    const int lane = e & 63, mt = (e >> 6) & 3, ks = e >> 8, r = lane & 31, h = lane >> 5;
    const int beta = gfm_out_bit(mt, r);
    u32 w[4];
    for (int j = 0; j < 4; ++j) {
      u32 word = 0;
      for (int q = 0; q < 8; ++q) {
        const u128 c = col[gfm_share_bit(ks, h, j, q)];
        if ((u32)(c >> beta) & 1u) word |= 0x4u << (4 * q);
      }
      w[j] = word;
    }
    tab[(size_t)i * (GFM_FRAG_WORDS / 4) + e] = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

__device__ __forceinline__ gfm_v16f gfm_mfma(const gfm_v8i& a, const gfm_v8i& b, const gfm_v16f& c) {
  // cbsz = blgp = 4: both operands fp4 (e2m1); scales 2^0 (E8M0 127) in every byte
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}
__device__ __forceinline__ gfm_v8i gfm_expand(u32 w) {
  gfm_v8i b = {(int)(w & 0x11111111u), (int)((w >> 1) & 0x11111111u), (int)((w >> 2) & 0x11111111u),
               (int)((w >> 3) & 0x11111111u), 0, 0, 0, 0};
  return b;
}

// out[s] = (prev ? prev[s] : 0) + sum_{i < m} lambda_i * shares[i][s]; tab = k_gf_mfma_table's fragments of the lambdas,
// m4 = m rounded up to a multiple of four of them (the padding parties have lambda = 0: all-zero fragments).
// A wave works on 64 secrets at a time (two column tiles that share every A fragment read from LDS); lane (r, h) loads
// bytes 8 h .. 8 h + 7 of secret r's share: the two words it expands in the party's two k-steps.
// Everything the matrix instructions wait for is fetched ahead, with no branch in between (a conditional load makes the
// compiler drain every outstanding load at the next use; party and secret indices are clamped instead: a clamped load meets
// a zero matrix or is never stored):
//   * the A fragment of the NEXT pair of matrix instructions;
//   * the share words of the NEXT GROUP of four parties (of the wave's next tile after a tile's last group), all eight loads
//     issued at the head of the group that runs 64 matrix instructions on the words loaded one group earlier.  The
//     compiler's wait in a loop lets the loads of the current iteration stay in flight and drains those of earlier ones:
//     this order makes that exact (s_waitcnt vmcnt(8)).  (A four-slot ring refilled party by party gets vmcnt(2) -- every
//     load but the newest pair -- and 26 ns per matrix instruction; loads and their wait written in inline assembly let
//     the register allocator reuse a register whose load was still in flight: a memory fault.)
__global__ __launch_bounds__(GFM_BLOCK, 2) void k_recover_gf128_mfma(u64* out, const u64* shares, size_t stride, const uint4* tab,
                                                                     int m, int m4, size_t N, const u64* prev) {
  extern __shared__ __align__(16) uint4 gfm_lds[];
  for (int e = threadIdx.x; e < m4 * (GFM_FRAG_WORDS / 4); e += GFM_BLOCK) gfm_lds[e] = tab[e];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const size_t ntiles = (N + 63) / 64, tstep = (size_t)gridDim.x * (GFM_BLOCK / 64);
  typedef u32 gfm_u32x2 __attribute__((ext_vector_type(2)));
  struct Group {
    gfm_u32x2 w[4][2];
  };
  auto load_group = [&](int i0, size_t s0) {
    Group g;
    const size_t a0 = s0 < N ? s0 : N - 1, a1 = s0 + 32 < N ? s0 + 32 : N - 1;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int ic = i0 + d < m ? i0 + d : m - 1;
      const u64* row = shares + (size_t)ic * stride * 2 + h;
      g.w[d][0] = __builtin_nontemporal_load(reinterpret_cast<const gfm_u32x2*>(row + a0 * 2));
      g.w[d][1] = __builtin_nontemporal_load(reinterpret_cast<const gfm_u32x2*>(row + a1 * 2));
    }
    return g;
  };
  size_t tile = (size_t)blockIdx.x * (GFM_BLOCK / 64) + wave;
  Group cur = load_group(0, tile * 64 + r);
  const uint4* frag = gfm_lds + lane;
  uint4 fa = frag[0];
  for (; tile < ntiles; tile += tstep) {
    const size_t s0 = tile * 64 + r, s1 = s0 + 32;
    gfm_v16f acc[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[ct][mt] = gfm_v16f{};
    const uint4* gfrag = frag;  // the group's 32 fragments: 32 KiB apart, the reads below at immediate offsets
    for (int i0 = 0; i0 < m4; i0 += 4) {
      const bool last = i0 + 4 >= m4;
      const Group nxt = load_group(last ? 0 : i0 + 4, last ? s0 + tstep * 64 : s0);
      const uint4* nfrag = last ? frag : gfrag + 32 * 64;  // (the tile's last fragment is followed by the next tile's first)
      __builtin_amdgcn_sched_barrier(0);  // the eight loads stay at the head of the group
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const gfm_u32x2 x0 = cur.w[d][0], x1 = cur.w[d][1];
#pragma unroll
        for (int f = 0; f < 8; ++f) {  // f = 4 ks + mt
          const int nf = d * 8 + f + 1;
          const uint4 fb = nf < 32 ? gfrag[nf * 64] : nfrag[0];
          const gfm_v8i a = {(int)fa.x, (int)fa.y, (int)fa.z, (int)fa.w, 0, 0, 0, 0};
          const gfm_v8i b0 = gfm_expand(f < 4 ? x0.x : x0.y), b1 = gfm_expand(f < 4 ? x1.x : x1.y);
          acc[0][f & 3] = gfm_mfma(a, b0, acc[0][f & 3]);
          acc[1][f & 3] = gfm_mfma(a, b1, acc[1][f & 3]);
          fa = fb;
        }
      }
      cur = nxt;
      gfrag = nfrag;
    }
    // parity of each count -> bit 16 (mt & 1) + reg of word mt >> 1 of this lane's half of the result
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      u32 word[2] = {0u, 0u};
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          word[mt >> 1] |= ((u32)acc[ct][mt][reg] & 1u) << (16 * (mt & 1) + reg);
      const size_t s = ct ? s1 : s0;
      if (s < N) {
        gfm_u32x2 o = {word[0], word[1]};
        if (prev) {
          const gfm_u32x2 p = *reinterpret_cast<const gfm_u32x2*>(prev + s * 2 + h);
          o.x ^= p.x;
          o.y ^= p.y;
        }
        *reinterpret_cast<gfm_u32x2*>(out + s * 2 + h) = o;
      }
    }
  }
}

}  // namespace sclhip
