// tools/gf128_bench.hip -- GF(2^128) reconstruct / share kernels (BASELINE config C4: n = 40, t = 13): the library's
// kernels against restructured variants, plus a probe of whether LDS reads and vector ALU work of one CU overlap.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/_build/gf128_bench tools/gf128_bench.hip
// run:   tools/_build/gf128_bench [N secrets, default 5e6]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../secure-computation-library_amd/csrc/kernels.hpp"
#include "gfpos_asm.hpp"  // lds_read128 and the hand-issued read pipelines (A/B only)
using namespace sclhip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1);} } while (0)

// ---- does a CU overlap ds_read_b128 with vector ALU instructions? -------------------------------------------------
// every wave issues groups of 16 units; a unit is NL conflict-free ds_read_b128 (1 KiB per wave) and NV v_xor_b32
template <int NL, int NV>
__global__ __launch_bounds__(256) void k_coissue(u32* out, int iters) {
  __shared__ uint4 T[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) T[i] = make_uint4(i, i * 3, i * 5, i * 7);
  __syncthreads();
  const u32 lane = threadIdx.x & 63;
  u32 addr = (lane & 15) * 16 + (lane >> 4) * 256;
  u32 x0 = lane, x1 = lane * 3, x2 = lane * 5, x3 = lane * 7, y = lane ^ 0x55;
  uint4 v0, v1, v2, v3;
  v0 = v1 = v2 = v3 = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if constexpr (NL >= 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v0) : "v"(addr), "n"((g & 7) * 1024));
      if constexpr (NL >= 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v1) : "v"(addr), "n"((g & 7) * 1024 + 4096));
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        if ((k & 3) == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x0) : "v"(y));
        if ((k & 3) == 1) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x1) : "v"(y));
        if ((k & 3) == 2) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x2) : "v"(y));
        if ((k & 3) == 3) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x3) : "v"(y));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ v0.x ^ v1.y ^ v2.z ^ v3.w;
}

// ---- reconstruct variant 1 ---------------------------------------------------------------------------------------
// Same nibble tables T_i[j] = j(x) * lambda_i (256 B per party, conflict-free ds_read_b128), but
//  * M known at compile time and the party loop fully unrolled: a lookup's table base is the instruction's offset
//    field, the address register is just (nibble << 4), one v_bfe / v_and per lookup out of a word masked once;
//  * odd and even nibbles run as two Horner chains in x^8 (the odd one is multiplied by x^4 at the end), so a
//    masked word serves four lookups without shifting;
//  * G parties share the x^8 steps of one pass.
__device__ __forceinline__ void gf_mulx8(u32 (&r)[4]) {
  const u32 t = r[3] >> 24;
  r[3] = __builtin_amdgcn_alignbit(r[3], r[2], 24);
  r[2] = __builtin_amdgcn_alignbit(r[2], r[1], 24);
  r[1] = __builtin_amdgcn_alignbit(r[1], r[0], 24);
  r[0] = (r[0] << 8) ^ t ^ (t << 1) ^ (t << 2) ^ (t << 7);
}
__device__ __forceinline__ void gf_mulx4(u32 (&r)[4]) {
  const u32 t = r[3] >> 28;
  r[3] = __builtin_amdgcn_alignbit(r[3], r[2], 28);
  r[2] = __builtin_amdgcn_alignbit(r[2], r[1], 28);
  r[1] = __builtin_amdgcn_alignbit(r[1], r[0], 28);
  r[0] = (r[0] << 4) ^ t ^ (t << 1) ^ (t << 2) ^ (t << 7);
}

template <int M, int G, int BLK, int WPS>
__global__ __launch_bounds__(BLK, WPS) void k_rec_gf_v1(u64* out, const u64* shares, size_t stride, const u128* lam, size_t N) {
  __shared__ uint4 T[M * 16];
  for (int e = threadIdx.x; e < M * 16; e += BLK) {
    const u128 l0 = lam[e >> 4];
    const u128 l1 = Gf128::mulx(l0), l2 = Gf128::mulx(l1), l3 = Gf128::mulx(l2);
    const int j = e & 15;
    const u128 v = (j & 1 ? l0 : (u128)0) ^ (j & 2 ? l1 : (u128)0) ^ (j & 4 ? l2 : (u128)0) ^ (j & 8 ? l3 : (u128)0);
    T[e] = make_uint4((u32)v, (u32)(v >> 32), (u32)(v >> 64), (u32)(v >> 96));
  }
  __syncthreads();
  const unsigned char* Tb = reinterpret_cast<const unsigned char*>(T);
  for (size_t s = (size_t)blockIdx.x * BLK + threadIdx.x; s < N; s += (size_t)gridDim.x * BLK) {
    u32 tot[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i0 = 0; i0 < M; i0 += G) {
      constexpr int GG = G;
      u32 w[GG][4];
#pragma unroll
      for (int j = 0; j < GG; ++j) {
        if (i0 + j < M) {
          const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(shares + ((size_t)(i0 + j) * stride + s) * 2));
          w[j][0] = (u32)v.x;
          w[j][1] = (u32)(v.x >> 32);
          w[j][2] = (u32)v.y;
          w[j][3] = (u32)(v.y >> 32);
        }
      }
      u32 ro[4] = {0, 0, 0, 0}, re[4] = {0, 0, 0, 0};
#pragma unroll
      for (int wd = 3; wd >= 0; --wd) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // 0: odd nibbles (high halves of the bytes), 1: even nibbles
          u32 m[GG];
#pragma unroll
          for (int j = 0; j < GG; ++j)
            if (i0 + j < M) m[j] = half == 0 ? (w[j][wd] & 0xF0F0F0F0u) : ((w[j][wd] << 4) & 0xF0F0F0F0u);
          u32(&r)[4] = half == 0 ? ro : re;
#pragma unroll
          for (int b = 3; b >= 0; --b) {
            if (!(wd == 3 && b == 3)) gf_mulx8(r);
#pragma unroll
            for (int j = 0; j < GG; ++j) {
              if (i0 + j < M) {
                const u32 a = b == 3 ? (m[j] >> 24) : b == 0 ? (m[j] & 0xFFu) : ((m[j] >> (8 * b)) & 0xFFu);
                const uint4 tv = *reinterpret_cast<const uint4*>(Tb + (i0 + j) * 256 + a);
                r[0] ^= tv.x;
                r[1] ^= tv.y;
                r[2] ^= tv.z;
                r[3] ^= tv.w;
              }
            }
          }
        }
      }
      gf_mulx4(ro);
#pragma unroll
      for (int c = 0; c < 4; ++c) tot[c] ^= ro[c] ^ re[c];
    }
    u64x2 o;
    o.x = (u64)tot[0] | ((u64)tot[1] << 32);
    o.y = (u64)tot[2] | ((u64)tot[3] << 32);
    __builtin_nontemporal_store(o, reinterpret_cast<u64x2*>(out + s * 2));
  }
}


// ---- reconstruct variant 2: one table per (party, nibble position within a 32-bit word) ---------------------------------
// T[i][p][j] = j(x) * lambda_i * x^(4p), p < 8: 2 KiB per party (80 KiB at n = 40).  The 8 nibbles of word wd of every share
// then add up with NO shifting of the accumulator (r_wd), and the four word sums combine by three x^32 steps at the very
// end -- against 32 x^4 steps per group of parties in the library kernel -- so the group can be small (few registers, many
// waves) at no cost.  A lookup is one v_add_u32_sdwa (table base + byte of the masked word), one ds_read_b128, four xors.
template <int B>
__device__ __forceinline__ void gf_lookup2(u32 (&a)[4], const unsigned char* T, u32 gbase, u32 o, u32 e, int j) {
  const uint4 to = *reinterpret_cast<const uint4*>(T + add_byte<B>(gbase, o) + (j * 2048 + (2 * B + 1) * 256));
  const uint4 te = *reinterpret_cast<const uint4*>(T + add_byte<B>(gbase, e) + (j * 2048 + (2 * B) * 256));
  a[0] ^= to.x ^ te.x;
  a[1] ^= to.y ^ te.y;
  a[2] ^= to.z ^ te.z;
  a[3] ^= to.w ^ te.w;
}

template <int G, int BLK, int WPS>
__global__ __launch_bounds__(BLK, WPS) void k_rec_gf_v2(u64* out, const u64* shares, size_t stride, const u128* lam, int m, size_t N) {
  extern __shared__ uint4 Tdyn[];  // [mpad][8][16]
  const int mpad = (m + G - 1) / G * G;
  for (int e = threadIdx.x; e < mpad * 128; e += BLK) {
    const int i = e >> 7, p = (e >> 4) & 7, j = e & 15;
    u128 l0 = i < m ? lam[i] : (u128)0;
    for (int k = 0; k < p; ++k) l0 = Gf128::mulx4(l0);
    const u128 l1 = Gf128::mulx(l0), l2 = Gf128::mulx(l1), l3 = Gf128::mulx(l2);
    const u128 v = (j & 1 ? l0 : (u128)0) ^ (j & 2 ? l1 : (u128)0) ^ (j & 4 ? l2 : (u128)0) ^ (j & 8 ? l3 : (u128)0);
    Tdyn[e] = make_uint4((u32)v, (u32)(v >> 32), (u32)(v >> 64), (u32)(v >> 96));
  }
  __syncthreads();
  const unsigned char* Tb = reinterpret_cast<const unsigned char*>(Tdyn);
  for (size_t s = (size_t)blockIdx.x * BLK + threadIdx.x; s < N; s += (size_t)gridDim.x * BLK) {
    u32 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][c] = 0;
    for (int i0 = 0; i0 < m; i0 += G) {
      u32 w[G][4];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        u64x2 v;
        v.x = v.y = 0;
        if (i0 + j < m) v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(shares + ((size_t)(i0 + j) * stride + s) * 2));
        w[j][0] = (u32)v.x;
        w[j][1] = (u32)(v.x >> 32);
        w[j][2] = (u32)v.y;
        w[j][3] = (u32)(v.y >> 32);
      }
      const u32 gbase = (u32)i0 * 2048u;
#pragma unroll
      for (int wd = 0; wd < 4; ++wd) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const u32 o = w[j][wd] & 0xF0F0F0F0u, e = (w[j][wd] << 4) & 0xF0F0F0F0u;
          gf_lookup2<0>(acc[wd], Tb, gbase, o, e, j);
          gf_lookup2<1>(acc[wd], Tb, gbase, o, e, j);
          gf_lookup2<2>(acc[wd], Tb, gbase, o, e, j);
          gf_lookup2<3>(acc[wd], Tb, gbase, o, e, j);
        }
      }
    }
    u32 r[4] = {acc[3][0], acc[3][1], acc[3][2], acc[3][3]};
#pragma unroll
    for (int wd = 2; wd >= 0; --wd) {
      gf_mulx32(r);
#pragma unroll
      for (int c = 0; c < 4; ++c) r[c] ^= acc[wd][c];
    }
    u64x2 o;
    o.x = (u64)r[0] | ((u64)r[1] << 32);
    o.y = (u64)r[2] | ((u64)r[3] << 32);
    __builtin_nontemporal_store(o, reinterpret_cast<u64x2*>(out + s * 2));
  }
}


// ---- reconstruct variant 3: variant 2 with the LDS reads issued by hand, one batch (4 lookups) ahead of the xors -------
// The compiler waits for a batch's reads right after issuing them; here batch b+1 is in flight while batch b is folded
// in (s_waitcnt lgkmcnt(4): LDS returns in order).  PF: the next group's shares are fetched while this group is worked on.
// batch IDX of a group of G parties: word wd = IDX / (2G), party j = (IDX / 2) % G, half = IDX & 1 (0: odd nibbles)
template <int G, int IDX>
__device__ __forceinline__ void gf_issue4(u32x4 (&buf)[4], u32 gbase, const u32 (&w)[G][4]) {
  constexpr int wd = IDX / (2 * G), j = (IDX / 2) % G, half = IDX & 1;
  const u32 m = half ? ((w[j][wd] << 4) & 0xF0F0F0F0u) : (w[j][wd] & 0xF0F0F0F0u);
  lds_read128<j * 2048 + (0 + (1 - half)) * 256>(buf[0], add_byte<0>(gbase, m));
  lds_read128<j * 2048 + (2 + (1 - half)) * 256>(buf[1], add_byte<1>(gbase, m));
  lds_read128<j * 2048 + (4 + (1 - half)) * 256>(buf[2], add_byte<2>(gbase, m));
  lds_read128<j * 2048 + (6 + (1 - half)) * 256>(buf[3], add_byte<3>(gbase, m));
}
template <int G, int IDX>
__device__ __forceinline__ void gf_pipe(u32x4 (&A)[4], u32x4 (&B)[4], u32x4 (&acc)[4], u32 gbase, const u32 (&w)[G][4]) {
  constexpr int NB = 8 * G;
  if constexpr (IDX < NB) {
    if constexpr (IDX + 1 < NB) gf_issue4<G, IDX + 1>((IDX & 1) ? A : B, gbase, w);
    u32x4(&cur)[4] = (IDX & 1) ? B : A;
    if constexpr (IDX + 1 < NB) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
    constexpr int wd = IDX / (2 * G);
    acc[wd] ^= (cur[0] ^ cur[1]) ^ (cur[2] ^ cur[3]);
    gf_pipe<G, IDX + 1>(A, B, acc, gbase, w);
  }
}

template <int G, int BLK, int WPS, bool PF>
__global__ __launch_bounds__(BLK, WPS) void k_rec_gf_v3(u64* out, const u64* shares, size_t stride, const u128* lam, int m, size_t N) {
  extern __shared__ uint4 Tdyn[];  // [mpad][8][16]
  const int mpad = (m + G - 1) / G * G;
  for (int e = threadIdx.x; e < mpad * 128; e += BLK) {
    const int i = e >> 7, p = (e >> 4) & 7, j = e & 15;
    u128 l0 = i < m ? lam[i] : (u128)0;
    for (int k = 0; k < p; ++k) l0 = Gf128::mulx4(l0);
    const u128 l1 = Gf128::mulx(l0), l2 = Gf128::mulx(l1), l3 = Gf128::mulx(l2);
    const u128 v = (j & 1 ? l0 : (u128)0) ^ (j & 2 ? l1 : (u128)0) ^ (j & 4 ? l2 : (u128)0) ^ (j & 8 ? l3 : (u128)0);
    Tdyn[e] = make_uint4((u32)v, (u32)(v >> 32), (u32)(v >> 64), (u32)(v >> 96));
  }
  __syncthreads();
  const u32 tbase = (u32)(uintptr_t)Tdyn;  // low half of the flat address = LDS byte address
  auto load_group = [&](u32 (&w)[G][4], int i0, size_t s) {
#pragma unroll
    for (int j = 0; j < G; ++j) {
      u64x2 v;
      v.x = v.y = 0;
      if (i0 + j < m) v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(shares + ((size_t)(i0 + j) * stride + s) * 2));
      w[j][0] = (u32)v.x;
      w[j][1] = (u32)(v.x >> 32);
      w[j][2] = (u32)v.y;
      w[j][3] = (u32)(v.y >> 32);
    }
  };
  for (size_t s = (size_t)blockIdx.x * BLK + threadIdx.x; s < N; s += (size_t)gridDim.x * BLK) {
    u32x4 acc[4] = {0, 0, 0, 0};
    u32 w[G][4], wn[G][4];
    load_group(w, 0, s);
    for (int i0 = 0; i0 < m; i0 += G) {
      if constexpr (PF) {
        if (i0 + G < m) load_group(wn, i0 + G, s);
      }
      const u32 gbase = tbase + (u32)i0 * 2048u;
      u32x4 A[4], B[4];
      gf_issue4<G, 0>(A, gbase, w);
      gf_pipe<G, 0>(A, B, acc, gbase, w);
      if constexpr (PF) {
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
          for (int c = 0; c < 4; ++c) w[j][c] = wn[j][c];
      } else {
        if (i0 + G < m) load_group(w, i0 + G, s);
      }
    }
    u32 r[4] = {acc[3].x, acc[3].y, acc[3].z, acc[3].w};
#pragma unroll
    for (int wd = 2; wd >= 0; --wd) {
      gf_mulx32(r);
      r[0] ^= acc[wd].x;
      r[1] ^= acc[wd].y;
      r[2] ^= acc[wd].z;
      r[3] ^= acc[wd].w;
    }
    u64x2 o;
    o.x = (u64)r[0] | ((u64)r[1] << 32);
    o.y = (u64)r[2] | ((u64)r[3] << 32);
    __builtin_nontemporal_store(o, reinterpret_cast<u64x2*>(out + s * 2));
  }
}


// ---- reconstruct variant 5: variant 3 squeezed under 64 registers (8 waves per SIMD) ---------------------------------
// tools/oprate.hip: v_xor_b32 issues every ~2.7 cycles per SIMD only when >= 4 waves are ready to issue; with half of a
// SIMD's 4 waves parked on LDS returns it drops to the ~4.6-cycle rate.  So: batches of 2 lookups (16 buffer registers),
// groups of G = 2 parties with the next group prefetched, two 1024-thread workgroups per CU.
template <int G, int IDX>
__device__ __forceinline__ void gf5_issue2(u32x4 (&buf)[2], u32 gbase, const u32 (&w)[G][4]) {
  // batch IDX: word wd = IDX / (4G), party j = (IDX / 4) % G, half = (IDX / 2) & 1 (0: odd nibbles), pair = IDX & 1 (bytes 0-1 / 2-3)
  constexpr int wd = IDX / (4 * G), j = (IDX / 4) % G, half = (IDX / 2) & 1, pair = IDX & 1;
  const u32 m = half ? ((w[j][wd] << 4) & 0xF0F0F0F0u) : (w[j][wd] & 0xF0F0F0F0u);
  lds_read128<j * 2048 + (4 * pair + 0 + (1 - half)) * 256>(buf[0], add_byte<2 * pair>(gbase, m));
  lds_read128<j * 2048 + (4 * pair + 2 + (1 - half)) * 256>(buf[1], add_byte<2 * pair + 1>(gbase, m));
}
template <int G, int IDX>
__device__ __forceinline__ void gf5_pipe(u32x4 (&A)[2], u32x4 (&B)[2], u32x4 (&acc)[4], u32 gbase, const u32 (&w)[G][4]) {
  constexpr int NB = 16 * G;
  if constexpr (IDX < NB) {
    if constexpr (IDX + 1 < NB) gf5_issue2<G, IDX + 1>((IDX & 1) ? A : B, gbase, w);
    u32x4(&cur)[2] = (IDX & 1) ? B : A;
    if constexpr (IDX + 1 < NB) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(cur[0]), "+v"(cur[1]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]));
    acc[IDX / (4 * G)] ^= cur[0] ^ cur[1];
    gf5_pipe<G, IDX + 1>(A, B, acc, gbase, w);
  }
}

template <int G, int BLK, int WPS>
__global__ __launch_bounds__(BLK, WPS) void k_rec_gf_v5(u64* out, const u64* shares, size_t stride, const u128* lam, int m, size_t N) {
  extern __shared__ uint4 Tdyn[];
  const int mpad = (m + G - 1) / G * G;
  for (int e = threadIdx.x; e < mpad * 128; e += BLK) {
    const int i = e >> 7, p = (e >> 4) & 7, j = e & 15;
    u128 l0 = i < m ? lam[i] : (u128)0;
    for (int k = 0; k < p; ++k) l0 = Gf128::mulx4(l0);
    const u128 l1 = Gf128::mulx(l0), l2 = Gf128::mulx(l1), l3 = Gf128::mulx(l2);
    const u128 v = (j & 1 ? l0 : (u128)0) ^ (j & 2 ? l1 : (u128)0) ^ (j & 4 ? l2 : (u128)0) ^ (j & 8 ? l3 : (u128)0);
    Tdyn[e] = make_uint4((u32)v, (u32)(v >> 32), (u32)(v >> 64), (u32)(v >> 96));
  }
  __syncthreads();
  const u32 tbase = (u32)(uintptr_t)Tdyn;
  for (size_t s = (size_t)blockIdx.x * BLK + threadIdx.x; s < N; s += (size_t)gridDim.x * BLK) {
    u32x4 acc[4] = {0, 0, 0, 0};
    u32 w[G][4], wn[G][4];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(shares + ((size_t)j * stride + s) * 2));
      w[j][0] = (u32)v.x, w[j][1] = (u32)(v.x >> 32), w[j][2] = (u32)v.y, w[j][3] = (u32)(v.y >> 32);
    }
    for (int i0 = 0; i0 < m; i0 += G) {
#pragma unroll
      for (int j = 0; j < G; ++j) {
        u64x2 v;
        v.x = v.y = 0;
        if (i0 + G + j < m) v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(shares + ((size_t)(i0 + G + j) * stride + s) * 2));
        wn[j][0] = (u32)v.x, wn[j][1] = (u32)(v.x >> 32), wn[j][2] = (u32)v.y, wn[j][3] = (u32)(v.y >> 32);
      }
      const u32 gbase = tbase + (u32)i0 * 2048u;
      u32x4 A[2], B[2];
      gf5_issue2<G, 0>(A, gbase, w);
      gf5_pipe<G, 0>(A, B, acc, gbase, w);
#pragma unroll
      for (int j = 0; j < G; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) w[j][c] = wn[j][c];
    }
    u32 r[4] = {acc[3].x, acc[3].y, acc[3].z, acc[3].w};
#pragma unroll
    for (int wd = 2; wd >= 0; --wd) {
      gf_mulx32(r);
      r[0] ^= acc[wd].x;
      r[1] ^= acc[wd].y;
      r[2] ^= acc[wd].z;
      r[3] ^= acc[wd].w;
    }
    u64x2 o;
    o.x = (u64)r[0] | ((u64)r[1] << 32);
    o.y = (u64)r[2] | ((u64)r[3] << 32);
    __builtin_nontemporal_store(o, reinterpret_cast<u64x2*>(out + s * 2));
  }
}

// ---- share variant 1: Horner with the node in a scalar register ----------------------------------------------------------
// y * a for a wave-uniform a < 64: a shifted copy of y per set bit of a (static shifts behind scalar branches), the at
// most five bits that leave the top folded back with x^128 = x^7 + x^2 + x + 1
template <int B>
__device__ __forceinline__ void gf_shl_xor(u32 (&r)[4], u32& ov, const u32 (&y)[4]) {
  if constexpr (B == 0) {
    r[0] ^= y[0];
    r[1] ^= y[1];
    r[2] ^= y[2];
    r[3] ^= y[3];
  } else {
    r[0] ^= y[0] << B;
    r[1] ^= __builtin_amdgcn_alignbit(y[1], y[0], 32 - B);
    r[2] ^= __builtin_amdgcn_alignbit(y[2], y[1], 32 - B);
    r[3] ^= __builtin_amdgcn_alignbit(y[3], y[2], 32 - B);
    ov ^= y[3] >> (32 - B);
  }
}
__device__ __forceinline__ void gf_mul_small_add(u32 (&y)[4], u32 a_uniform, const u32 (&c)[4]) {
  u32 r[4] = {c[0], c[1], c[2], c[3]}, ov = 0;
  if (a_uniform & 1) gf_shl_xor<0>(r, ov, y);
  if (a_uniform & 2) gf_shl_xor<1>(r, ov, y);
  if (a_uniform & 4) gf_shl_xor<2>(r, ov, y);
  if (a_uniform & 8) gf_shl_xor<3>(r, ov, y);
  if (a_uniform & 16) gf_shl_xor<4>(r, ov, y);
  if (a_uniform & 32) gf_shl_xor<5>(r, ov, y);
  if (a_uniform & 64) gf_shl_xor<6>(r, ov, y);
  if (a_uniform & 128) gf_shl_xor<7>(r, ov, y);
  r[0] ^= ov ^ (ov << 1) ^ (ov << 2) ^ (ov << 7);
  y[0] = r[0];
  y[1] = r[1];
  y[2] = r[2];
  y[3] = r[3];
}

struct Nodes8 {
  unsigned char a[256];
};

template <int TT, int BLK, int WPS>
__global__ __launch_bounds__(BLK, WPS) void k_share_gf_v1(u64* shares, size_t stride, const u64* secrets, const u64* coeffs,
                                                          size_t cstride, Nodes8 nodes, int n, size_t N) {
  for (size_t s = (size_t)blockIdx.x * BLK + threadIdx.x; s < N; s += (size_t)gridDim.x * BLK) {
    u32 c[TT + 1][4];
#pragma unroll
    for (int k = 0; k <= TT; ++k) {
      const u64* p = k == 0 ? secrets + s * 2 : coeffs + ((size_t)(k - 1) * cstride + s) * 2;
      const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(p));
      c[k][0] = (u32)v.x;
      c[k][1] = (u32)(v.x >> 32);
      c[k][2] = (u32)v.y;
      c[k][3] = (u32)(v.y >> 32);
    }
    for (int i = 0; i < n; ++i) {
      const u32 a = __builtin_amdgcn_readfirstlane((u32)nodes.a[i]);
      u32 y[4] = {c[TT][0], c[TT][1], c[TT][2], c[TT][3]};
#pragma unroll
      for (int k = TT - 1; k >= 0; --k) gf_mul_small_add(y, a, c[k]);
      u64x2 o;
      o.x = (u64)y[0] | ((u64)y[1] << 32);
      o.y = (u64)y[2] | ((u64)y[3] << 32);
      __builtin_nontemporal_store(o, reinterpret_cast<u64x2*>(shares + ((size_t)i * stride + s) * 2));
    }
  }
}

// variant 2: the set bits of the node walked by a scalar loop (one shifted copy per trip, shift amount in an SGPR)
__device__ __forceinline__ void gf_mul_small_add_loop(u32 (&y)[4], u32 a_uniform, const u32 (&c)[4]) {
  u32 r[4] = {c[0], c[1], c[2], c[3]}, ov = 0;
  if (a_uniform & 1) gf_shl_xor<0>(r, ov, y);
  u32 rest = a_uniform >> 1;
  u32 b = 1;
  while (rest) {  // scalar
    const u32 z = __builtin_ctz(rest);
    b += z;
    rest >>= z + 1;
    const u32 rs = 32 - b;
    r[0] ^= y[0] << b;
    r[1] ^= __builtin_amdgcn_alignbit(y[1], y[0], rs);
    r[2] ^= __builtin_amdgcn_alignbit(y[2], y[1], rs);
    r[3] ^= __builtin_amdgcn_alignbit(y[3], y[2], rs);
    ov ^= y[3] >> rs;
    b += 1;
  }
  r[0] ^= ov ^ (ov << 1) ^ (ov << 2) ^ (ov << 7);
  y[0] = r[0];
  y[1] = r[1];
  y[2] = r[2];
  y[3] = r[3];
}

template <int TT, int BLK, int WPS>
__global__ __launch_bounds__(BLK, WPS) void k_share_gf_v2(u64* shares, size_t stride, const u64* secrets, const u64* coeffs,
                                                          size_t cstride, Nodes8 nodes, int n, size_t N) {
  for (size_t s = (size_t)blockIdx.x * BLK + threadIdx.x; s < N; s += (size_t)gridDim.x * BLK) {
    u32 c[TT + 1][4];
#pragma unroll
    for (int k = 0; k <= TT; ++k) {
      const u64* p = k == 0 ? secrets + s * 2 : coeffs + ((size_t)(k - 1) * cstride + s) * 2;
      const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(p));
      c[k][0] = (u32)v.x;
      c[k][1] = (u32)(v.x >> 32);
      c[k][2] = (u32)v.y;
      c[k][3] = (u32)(v.y >> 32);
    }
    for (int i = 0; i < n; ++i) {
      const u32 a = __builtin_amdgcn_readfirstlane((u32)nodes.a[i]);
      u32 y[4] = {c[TT][0], c[TT][1], c[TT][2], c[TT][3]};
#pragma unroll
      for (int k = TT - 1; k >= 0; --k) gf_mul_small_add_loop(y, a, c[k]);
      u64x2 o;
      o.x = (u64)y[0] | ((u64)y[1] << 32);
      o.y = (u64)y[2] | ((u64)y[3] << 32);
      __builtin_nontemporal_store(o, reinterpret_cast<u64x2*>(shares + ((size_t)i * stride + s) * 2));
    }
  }
}

int main(int argc, char** argv) {
  const size_t N = argc > 1 ? (size_t)std::atof(argv[1]) : 5000000;
  constexpr int M = 40, TT = 13;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time_it = [&](auto launch, int reps) {
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
  };

  // ---- co-issue probe ----
  {
    u32* sink;
    CK(hipMalloc(&sink, 2048 * 256 * 4));
    const int iters = 2000;
    std::printf("-- co-issue probe: 2048 blocks x 256 threads, %d iterations of 16 units (NL ds_read_b128 + NV v_xor_b32 per unit)\n", iters);
#define CO(NL, NV)                                                                                                  \
  {                                                                                                                 \
    const float ms = time_it([&] { hipLaunchKernelGGL((k_coissue<NL, NV>), dim3(2048), dim3(256), 0, 0, sink, iters); }, 3); \
    const double units = 2048.0 * 4 * iters * 16;                                                                   \
    std::printf("NL=%d NV=%2d  %8.3f ms   %6.2f clk/unit/CU at 2.4 GHz\n", NL, NV, ms, ms * 1e-3 * 2.4e9 * 256 / units); \
  }
    CO(1, 0) CO(2, 0) CO(0, 4) CO(0, 5) CO(0, 8) CO(0, 10) CO(1, 4) CO(1, 5) CO(1, 8) CO(2, 8) CO(2, 10)
#undef CO
    CK(hipFree(sink));
  }

  // ---- data ----
  std::vector<u64> hl(2 * M);
  u64 x = 0x9E3779B97F4A7C15ull;
  auto rnd = [&] {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return x;
  };
  for (auto& v : hl) v = rnd();
  BigTable<Gf128> big;
  for (int i = 0; i < M; ++i) big.v[i] = ((u128)hl[2 * i + 1] << 64) | hl[2 * i];
  u128* lam_dev;
  CK(hipMalloc(&lam_dev, M * 16));
  CK(hipMemcpy(lam_dev, big.v, M * 16, hipMemcpyHostToDevice));
  u64 *sh, *o0, *o1, *cf;
  CK(hipMalloc(&sh, (size_t)M * N * 16));
  CK(hipMalloc(&o0, N * 16));
  CK(hipMalloc(&o1, N * 16));
  CK(hipMalloc(&cf, (size_t)(TT + 1) * N * 16));
  {
    AesKey key;
    for (int i = 0; i < 44; ++i) key.rk[i] = 0x9E3779B9u * (i + 1);
    for (int i = 0; i < 256; ++i) key.te0[i] = 0x85EBCA6Bu * (i + 7) ^ (i << 13);
    aes_key_round1(key);
    auto kern = &k_prg_blocks<>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, AES4_LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(AES4_GRID_CAP), dim3(ABLOCK), AES4_LDS_BYTES, 0, sh, key, 1ull, (size_t)M * N);
    hipLaunchKernelGGL(kern, dim3(AES4_GRID_CAP), dim3(ABLOCK), AES4_LDS_BYTES, 0, cf, key, 1ull << 40, (size_t)(TT + 1) * N);
    CK(hipDeviceSynchronize());
  }
  const double rec_bytes = (double)(M + 1) * 16 * N, share_bytes = (double)(TT + 1 + M) * 16 * N;
  auto same = [&](const u64* a, const u64* b, size_t words) {
    std::vector<u64> ha(words), hb(words);
    CK(hipMemcpy(ha.data(), a, words * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), b, words * 8, hipMemcpyDeviceToHost));
    size_t d = 0;
    for (size_t i = 0; i < words; ++i) d += ha[i] != hb[i];
    return d;
  };

  auto same_rows = [&](const u64* a, const u64* b) {  // the last 100000 secrets (or all) of every row
    const size_t cnt = N < 100000 ? N : 100000;
    size_t d = 0;
    for (int i = 0; i < M; ++i) d += same(a + ((size_t)i * N + (N - cnt)) * 2, b + ((size_t)i * N + (N - cnt)) * 2, 2 * cnt);
    return d;
  };
  std::printf("-- reconstruct, n = %d, N = %zu (%.0f B per secret)\n", M, N, rec_bytes / N);
  {
    const unsigned g = (unsigned)((N + 255) / 256);
    const float ms = time_it([&] { hipLaunchKernelGGL(k_recover_gf128<>, dim3(g), dim3(256), 0, 0, o0, sh, N, big, M, N); }, 5);
    std::printf("%-44s %8.3f ms  %6.2f TB/s  %6.2f G secrets/s\n", "library k_recover_gf128", ms, rec_bytes / ms / 1e9, N / ms / 1e6);
  }
#define REC(G, BLK, WPS)                                                                                             \
  {                                                                                                                 \
    const unsigned g = (unsigned)((N + BLK - 1) / BLK);                                                             \
    CK(hipMemset(o1, 0, N * 16));                                                                                   \
    const float ms = time_it([&] { hipLaunchKernelGGL((k_rec_gf_v1<M, G, BLK, WPS>), dim3(g), dim3(BLK), 0, 0, o1, sh, N, lam_dev, N); }, 5); \
    std::printf("v1 G=%2d block %4d waves/SIMD>=%d               %8.3f ms  %6.2f TB/s  %6.2f G secrets/s  diff %zu\n", G, BLK, WPS, ms, \
                rec_bytes / ms / 1e9, N / ms / 1e6, same(o0, o1, 2 * N));                                            \
  }
  REC(20, 256, 2)
#undef REC

#define REC2(G, BLK, WPS, GRIDPERCU)                                                                                 \
  {                                                                                                                 \
    auto kern = &k_rec_gf_v2<G, BLK, WPS>;                                                                          \
    const int mpad = (M + G - 1) / G * G;                                                                           \
    const size_t lds = (size_t)mpad * 2048;                                                                         \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    size_t gb = (N + BLK - 1) / BLK;                                                                                \
    if (gb > 256 * GRIDPERCU) gb = 256 * GRIDPERCU;                                                                 \
    CK(hipMemset(o1, 0, N * 16));                                                                                   \
    const float ms = time_it([&] { hipLaunchKernelGGL(kern, dim3((unsigned)gb), dim3(BLK), lds, 0, o1, sh, N, lam_dev, M, N); }, 5); \
    std::printf("v2 G=%2d block %4d wps %d grid %5zu lds %6zu   %8.3f ms  %6.2f TB/s  %6.2f G secrets/s  diff %zu\n", G, BLK, WPS, gb, lds, \
                ms, rec_bytes / ms / 1e9, N / ms / 1e6, same(o0, o1, 2 * N));                                        \
  }
  REC2(4, 1024, 1, 1)
#undef REC2
#define REC3(G, BLK, WPS, PF, GRIDPERCU)                                                                             \
  {                                                                                                                 \
    auto kern = &k_rec_gf_v3<G, BLK, WPS, PF>;                                                                      \
    const int mpad = (M + G - 1) / G * G;                                                                           \
    const size_t lds = (size_t)mpad * 2048;                                                                         \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    size_t gb = (N + BLK - 1) / BLK;                                                                                \
    if (gb > 256 * GRIDPERCU) gb = 256 * GRIDPERCU;                                                                 \
    CK(hipMemset(o1, 0, N * 16));                                                                                   \
    const float ms = time_it([&] { hipLaunchKernelGGL(kern, dim3((unsigned)gb), dim3(BLK), lds, 0, o1, sh, N, lam_dev, M, N); }, 5); \
    std::printf("v3 G=%2d block %4d wps %d pf %d grid %5zu lds %6zu %8.3f ms  %6.2f TB/s  %6.2f G secrets/s  diff %zu\n", G, BLK, WPS, (int)PF, \
                gb, lds, ms, rec_bytes / ms / 1e9, N / ms / 1e6, same(o0, o1, 2 * N));                               \
  }
  REC3(5, 512, 2, true, 2)
#undef REC3
#define REC5(G, BLK, WPS, GRIDPERCU)                                                                                 \
  {                                                                                                                 \
    auto kern = &k_rec_gf_v5<G, BLK, WPS>;                                                                          \
    const int mpad = (M + G - 1) / G * G;                                                                           \
    const size_t lds = (size_t)mpad * 2048;                                                                         \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    size_t gb = (N + BLK - 1) / BLK;                                                                                \
    if (gb > 256 * GRIDPERCU) gb = 256 * GRIDPERCU;                                                                 \
    CK(hipMemset(o1, 0, N * 16));                                                                                   \
    const float ms = time_it([&] { hipLaunchKernelGGL(kern, dim3((unsigned)gb), dim3(BLK), lds, 0, o1, sh, N, lam_dev, M, N); }, 5); \
    std::printf("v5 G=%2d block %4d wps %d grid %5zu lds %6zu      %8.3f ms  %6.2f TB/s  %6.2f G secrets/s  diff %zu\n", G, BLK, WPS, \
                gb, lds, ms, rec_bytes / ms / 1e9, N / ms / 1e6, same(o0, o1, 2 * N));                               \
  }
  REC5(2, 1024, 8, 2) REC5(4, 1024, 8, 2) REC5(2, 1024, 4, 2) REC5(5, 1024, 4, 2) REC5(5, 512, 4, 2)
#undef REC5
  std::printf("-- share, n = %d, t = %d, N = %zu (%.0f B per secret)\n", M, TT, N, share_bytes / N);
  u64* shares2;
  CK(hipMalloc(&shares2, (size_t)M * N * 16));
  {
    BigTable<Gf128> al;
    for (int i = 0; i < M; ++i) al.v[i] = (u128)(i + 1);
    const unsigned g = (unsigned)((N + 255) / 256);
    const float ms = time_it([&] {
      hipLaunchKernelGGL((k_share<Gf128, 1, 16, true>), dim3(g), dim3(256), 0, 0, Gf128::Ctx{}, sh, N, cf, cf + 2 * N, N, al, TT, M, N);
    }, 3);
    std::printf("%-44s %8.3f ms  %6.2f TB/s  %6.2f G secrets/s\n", "library k_share<Gf128,1,16,smallx>", ms, share_bytes / ms / 1e9, N / ms / 1e6);
  }
  Nodes8 nodes;
  for (int i = 0; i < 256; ++i) nodes.a[i] = (unsigned char)(i + 1);
#define SHR(KERN, BLK, WPS, name)                                                                                   \
  {                                                                                                                 \
    const unsigned g = (unsigned)((N + BLK - 1) / BLK);                                                             \
    const float ms = time_it([&] { hipLaunchKernelGGL((KERN<TT, BLK, WPS>), dim3(g), dim3(BLK), 0, 0, shares2, N, cf, cf + 2 * N, N, nodes, M, N); }, 3); \
    std::printf("%-30s block %4d wps %d %8.3f ms  %6.2f TB/s  %6.2f G secrets/s  diff %zu\n", name, BLK, WPS, ms, share_bytes / ms / 1e9, \
                N / ms / 1e6, same_rows(sh, shares2));                                                                \
  }
  SHR(k_share_gf_v1, 256, 1, "v1 static shifts") SHR(k_share_gf_v1, 256, 2, "v1 static shifts") SHR(k_share_gf_v1, 128, 1, "v1 static shifts")
  SHR(k_share_gf_v2, 256, 1, "v2 scalar bit loop") SHR(k_share_gf_v2, 256, 2, "v2 scalar bit loop")
#undef SHR
  return 0;
}
