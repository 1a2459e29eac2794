// tools/aes_bench.hip -- AES-128-CTR block rate of the single-table form (one T-table + rotations, 32 KiB of LDS per
// 256-thread workgroup) against the library's four-table PRG kernel (no rotations, one SDWA move per table address; 128 KiB
// of LDS, one 1024-thread workgroup per CU) and against the four-table forms it replaced (kept here only: the compiler's
// v_bfe_u32 + v_lshl_add_u32 addressing, and the same addresses from three fast-class opcodes).
// The table contents are arbitrary here: all kernels compute the same function of them, and are compared word for word.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/_build/aes_bench tools/aes_bench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../secure-computation-library_amd/csrc/kernels.hpp"
using namespace sclhip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1);} } while (0)

// the single-table form (aes_ctr_multi: one T-table + rotations) in the geometry it was shipped with
__global__ __launch_bounds__(256) void k_prg_blocks1(u64* dst, AesKey key, u64 counter0, size_t nblocks) {
  SCL_AES_PROLOGUE(key)
  const size_t G = (size_t)gridDim.x * 256;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nblocks; q += 4 * G) {
    u64 ctr[4], lo[4], hi[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) ctr[b] = counter0 + q + b * G;
    aes_ctr_multi<4>(te0, key, ctr, lo, hi);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (q + b * G < nblocks) {
        u64x2 w;
        w.x = lo[b];
        w.y = hi[b];
        *reinterpret_cast<u64x2*>(dst + 2 * (q + b * G)) = w;
      }
    }
  }
}

// ---- the four-table form the library shipped before: word (r * 256 + x) * 32 + c = entry x of table r, copy c ------------
#define OLD_AES4_PROLOGUE(key)                                                                     \
  extern __shared__ __align__(16) u32 aes4_lds[];                                                  \
  for (int e_ = threadIdx.x; e_ < 4 * 256 * 32; e_ += ABLOCK) {                                    \
    const int r_ = e_ >> 13;                                                                       \
    const u32 v_ = (key).te0[(e_ >> 5) & 255];                                                     \
    aes4_lds[e_] = r_ == 0 ? v_ : (v_ << (8 * r_)) | (v_ >> (32 - 8 * r_));                        \
  }                                                                                                \
  __syncthreads();
struct OldAes4 {  // the lane's copy of tables 0 / 1 (t0, t0 + 8192 words) and 2 / 3 (t2, t2 + 8192 words)
  const u32* t0;
  const u32* t2;
  __device__ __forceinline__ void block(const AesKey& key, u64 counter, u64& out_lo, u64& out_hi) const {
#define O_T0(x) t0[(x) << 5]
#define O_T1(x) t0[((x) << 5) + 8192]
#define O_T2(x) t2[(x) << 5]
#define O_T3(x) t2[((x) << 5) + 8192]
#define O_SB(x) ((O_T0(x) >> 8) & 255u)
    u32 s0 = (u32)counter ^ key.rk[0], s1 = (u32)(counter >> 32) ^ key.rk[1], s2 = 0x89ABCDEFu ^ key.rk[2], s3 = 0x01234567u ^ key.rk[3];
#pragma unroll
    for (int r = 1; r < 10; ++r) {
      const u32 u0 = O_T0(s0 & 255) ^ O_T1((s1 >> 8) & 255) ^ O_T2((s2 >> 16) & 255) ^ O_T3(s3 >> 24) ^ key.rk[4 * r + 0];
      const u32 u1 = O_T0(s1 & 255) ^ O_T1((s2 >> 8) & 255) ^ O_T2((s3 >> 16) & 255) ^ O_T3(s0 >> 24) ^ key.rk[4 * r + 1];
      const u32 u2 = O_T0(s2 & 255) ^ O_T1((s3 >> 8) & 255) ^ O_T2((s0 >> 16) & 255) ^ O_T3(s1 >> 24) ^ key.rk[4 * r + 2];
      const u32 u3 = O_T0(s3 & 255) ^ O_T1((s0 >> 8) & 255) ^ O_T2((s1 >> 16) & 255) ^ O_T3(s2 >> 24) ^ key.rk[4 * r + 3];
      s0 = u0; s1 = u1; s2 = u2; s3 = u3;
    }
    const u32 o0 = (O_SB(s0 & 255) | (O_SB((s1 >> 8) & 255) << 8) | (O_SB((s2 >> 16) & 255) << 16) | (O_SB(s3 >> 24) << 24)) ^ key.rk[40];
    const u32 o1 = (O_SB(s1 & 255) | (O_SB((s2 >> 8) & 255) << 8) | (O_SB((s3 >> 16) & 255) << 16) | (O_SB(s0 >> 24) << 24)) ^ key.rk[41];
    const u32 o2 = (O_SB(s2 & 255) | (O_SB((s3 >> 8) & 255) << 8) | (O_SB((s0 >> 16) & 255) << 16) | (O_SB(s1 >> 24) << 24)) ^ key.rk[42];
    const u32 o3 = (O_SB(s3 & 255) | (O_SB((s0 >> 8) & 255) << 8) | (O_SB((s1 >> 16) & 255) << 16) | (O_SB(s2 >> 24) << 24)) ^ key.rk[43];
    out_lo = (u64)o0 | ((u64)o1 << 32);
    out_hi = (u64)o2 | ((u64)o3 << 32);
#undef O_T0
#undef O_T1
#undef O_T2
#undef O_T3
#undef O_SB
  }
};

// NB independent blocks per lane through AES (OldAes4 / Aes4f over the old layout, or the library's Aes4)
template <int NB, class AES>
__device__ __forceinline__ void run_blocks(const AES& aes, u64* dst, const AesKey& key, u64 counter0, size_t nblocks) {
  const size_t G = (size_t)gridDim.x * ABLOCK;
  for (size_t q = (size_t)blockIdx.x * ABLOCK + threadIdx.x; q < nblocks; q += NB * G) {
    u64 lo[NB], hi[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) aes.block(key, counter0 + q + b * G, lo[b], hi[b]);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (q + b * G < nblocks) {
        u64x2 w;
        w.x = lo[b];
        w.y = hi[b];
        *reinterpret_cast<u64x2*>(dst + 2 * (q + b * G)) = w;
      }
    }
  }
}

template <int NB>
__global__ __launch_bounds__(ABLOCK) void k_prg_blocks4u(u64* dst, AesKey key, u64 counter0, size_t nblocks) {
  OLD_AES4_PROLOGUE(key)
  run_blocks<NB>(OldAes4{aes4_lds + (threadIdx.x & 31), aes4_lds + (threadIdx.x & 31) + 2 * 8192}, dst, key, counter0, nblocks);
}

// ---- variant: table addresses from fast-class opcodes ----------------------------------------------------------------
// tools/oprate.hip: v_bfe_u32 + v_lshl_add_u32 (4.4 + 4.5 cycles per wave instruction) is what the compiler makes of
// te[(s >> 8k) & 255] with 128-byte entries; (s >> (8k - 7)) & 0x7F80 is v_lshrrev_b32 + v_and_b32 (2.4 + 2.5) and the base
// goes in with a v_add_u32 (2.7) or folds into a v_and_or_b32 (4.6).
struct Aes4f {
  u32 b0, b2;  // LDS byte addresses of the lane's copy of table 0 and of table 2 (tables 1 / 3 sit 32 KiB above them)
  // (word >> SH) & 0x7F80, + base: three fast-class instructions, written out because the compiler turns the C form
  // back into v_bfe_u32 + v_lshl_add_u32.  SH < 0: shift left by -SH.
  template <int SH>
  static __device__ __forceinline__ u32 addr(u32 s, u32 base) {
    u32 t;
    if constexpr (SH >= 0) asm("v_lshrrev_b32 %0, %1, %2" : "=v"(t) : "n"(SH), "v"(s));
    else asm("v_lshlrev_b32 %0, %1, %2" : "=v"(t) : "n"(-SH), "v"(s));
    asm("v_and_b32 %0, 0x7f80, %0" : "+v"(t));
    asm("v_add_u32 %0, %1, %0" : "+v"(t) : "v"(base));
    return t;
  }
  template <int OFF>
  static __device__ __forceinline__ u32 ld(u32 a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const __attribute__((address_space(3))) u32*>((uintptr_t)(a + OFF));
#else
    return a + OFF;
#endif
  }
  __device__ __forceinline__ void block(const AesKey& key, u64 counter, u64& out_lo, u64& out_hi) const {
#define F_T0(s) ld<0>(addr<-7>(s, b0))
#define F_T1(s) ld<32768>(addr<1>(s, b0))
#define F_T2(s) ld<0>(addr<9>(s, b2))
#define F_T3(s) ld<32768>(addr<17>(s, b2))
#define F_SB0(s) ((ld<0>(addr<-7>(s, b0)) >> 8) & 255u)
#define F_SB1(s) ((ld<0>(addr<1>(s, b0)) >> 8) & 255u)
#define F_SB2(s) ((ld<0>(addr<9>(s, b0)) >> 8) & 255u)
#define F_SB3(s) ((ld<0>(addr<17>(s, b0)) >> 8) & 255u)
    u32 s0 = (u32)counter ^ key.rk[0], s1 = (u32)(counter >> 32) ^ key.rk[1], s2 = 0x89ABCDEFu ^ key.rk[2], s3 = 0x01234567u ^ key.rk[3];
#pragma unroll
    for (int r = 1; r < 10; ++r) {
      const u32 u0 = F_T0(s0) ^ F_T1(s1) ^ F_T2(s2) ^ F_T3(s3) ^ key.rk[4 * r + 0];
      const u32 u1 = F_T0(s1) ^ F_T1(s2) ^ F_T2(s3) ^ F_T3(s0) ^ key.rk[4 * r + 1];
      const u32 u2 = F_T0(s2) ^ F_T1(s3) ^ F_T2(s0) ^ F_T3(s1) ^ key.rk[4 * r + 2];
      const u32 u3 = F_T0(s3) ^ F_T1(s0) ^ F_T2(s1) ^ F_T3(s2) ^ key.rk[4 * r + 3];
      s0 = u0; s1 = u1; s2 = u2; s3 = u3;
    }
    const u32 o0 = (F_SB0(s0) | (F_SB1(s1) << 8) | (F_SB2(s2) << 16) | (F_SB3(s3) << 24)) ^ key.rk[40];
    const u32 o1 = (F_SB0(s1) | (F_SB1(s2) << 8) | (F_SB2(s3) << 16) | (F_SB3(s0) << 24)) ^ key.rk[41];
    const u32 o2 = (F_SB0(s2) | (F_SB1(s3) << 8) | (F_SB2(s0) << 16) | (F_SB3(s1) << 24)) ^ key.rk[42];
    const u32 o3 = (F_SB0(s3) | (F_SB1(s0) << 8) | (F_SB2(s1) << 16) | (F_SB3(s2) << 24)) ^ key.rk[43];
    out_lo = (u64)o0 | ((u64)o1 << 32);
    out_hi = (u64)o2 | ((u64)o3 << 32);
#undef F_T0
#undef F_T1
#undef F_T2
#undef F_T3
#undef F_SB0
#undef F_SB1
#undef F_SB2
#undef F_SB3
  }
};

template <int NB>
__global__ __launch_bounds__(ABLOCK) void k_prg_blocks4f(u64* dst, AesKey key, u64 counter0, size_t nblocks) {
  OLD_AES4_PROLOGUE(key)
  const u32 lane_base = (u32)(uintptr_t)aes4_lds + 4u * (threadIdx.x & 31);  // low half of the flat address = LDS byte address
  run_blocks<NB>(Aes4f{lane_base, lane_base + 65536u}, dst, key, counter0, nblocks);
}

// ---- the library's form (kernels.hpp, Aes4): one SDWA move per table address, at 1 / 2 / 4 blocks per lane ------------------
template <int NB>
__global__ __launch_bounds__(ABLOCK) void k_prg_blocks4p(u64* dst, AesKey key, u64 counter0, size_t nblocks) {
  SCL_AES4_PROLOGUE(key)
  run_blocks<NB>(aes, dst, key, counter0, nblocks);
}

// the same kernel with a clock reading: s_memtime ticks per s_memrealtime tick (100 MHz) around the block loop, one per workgroup
__global__ __launch_bounds__(ABLOCK) void k_prg_blocks4p_clock(u64* dst, AesKey key, u64 counter0, size_t nblocks, u64* stamps) {
  SCL_AES4_PROLOGUE(key)
  const u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  run_blocks<4>(aes, dst, key, counter0, nblocks);
  const u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

int main() {
  const size_t nblocks = (size_t)1 << 28;
  AesKey key;
  for (int i = 0; i < 44; ++i) key.rk[i] = 0x9E3779B9u * (i + 1);
  for (int i = 0; i < 256; ++i) key.te0[i] = 0x85EBCA6Bu * (i + 7) ^ (i << 13);
  aes_key_round1(key);
  aes_key_range(key, 0, nblocks + (1u << 20));  // every counter below stays under 2^32: round 1 from the lower word only
  u64 *a, *b;
  CK(hipMalloc(&a, nblocks * 16));
  CK(hipMalloc(&b, nblocks * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time_it = [&](auto launch, const char* name) {
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    std::printf("%-44s %8.3f ms  %6.1f G blocks/s\n", name, ms, nblocks / ms / 1e6);
  };
  time_it([&] { hipLaunchKernelGGL(k_prg_blocks1, dim3(1024), dim3(256), 0, 0, a, key, 12345ull, nblocks); }, "one table + rotations, 4 x 256 threads per CU");
  {
    auto kern = &k_prg_blocks<>;  // the library's kernel: four tables, one 1024-thread workgroup per CU
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, AES4_LDS_BYTES));
    time_it([&] { hipLaunchKernelGGL(kern, dim3(AES4_GRID_CAP), dim3(ABLOCK), AES4_LDS_BYTES, 0, b, key, 12345ull, nblocks); },
            "library: k_prg_blocks (four tables, SDWA addresses)");
  }
#define RUN4U(NB, name)                                                                                              \
  {                                                                                                                      \
    auto kern = &k_prg_blocks4u<NB>;                                                                                     \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, AES4_LDS_BYTES)); \
    time_it([&] { hipLaunchKernelGGL(kern, dim3(AES4_GRID_CAP), dim3(ABLOCK), AES4_LDS_BYTES, 0, b, key, 12345ull, nblocks); }, name); \
  }
  RUN4U(1, "four tables, compiler addresses, 1 block/lane")
  RUN4U(2, "four tables, compiler addresses, 2 blocks/lane")
  RUN4U(4, "four tables, compiler addresses, 4 blocks/lane")
#define RUN4F(NB, name)                                                                                              \
  {                                                                                                                      \
    auto kern = &k_prg_blocks4f<NB>;                                                                                     \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, AES4_LDS_BYTES)); \
    time_it([&] { hipLaunchKernelGGL(kern, dim3(AES4_GRID_CAP), dim3(ABLOCK), AES4_LDS_BYTES, 0, b, key, 12345ull, nblocks); }, name); \
  }
  RUN4F(2, "four tables, fast-opcode addresses, 2 blocks/lane")
  RUN4F(4, "four tables, fast-opcode addresses, 4 blocks/lane")
#define RUN4P(NB, name)                                                                                              \
  {                                                                                                                      \
    auto kern = &k_prg_blocks4p<NB>;                                                                                     \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, AES4_LDS_BYTES)); \
    time_it([&] { hipLaunchKernelGGL(kern, dim3(AES4_GRID_CAP), dim3(ABLOCK), AES4_LDS_BYTES, 0, b, key, 12345ull, nblocks); }, name); \
  }
  RUN4P(1, "four tables, one SDWA move per address, 1 block/lane")
  RUN4P(2, "four tables, one SDWA move per address, 2 blocks/lane")
  RUN4P(4, "four tables, one SDWA move per address, 4 blocks/lane")
  CK(hipGetLastError());
  std::vector<u64> ha(1 << 20), hb(1 << 20);
  CK(hipMemcpy(ha.data(), a + (nblocks - (1 << 19)) * 2, ha.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b + (nblocks - (1 << 19)) * 2, hb.size() * 8, hipMemcpyDeviceToHost));
  size_t diff = 0;
  for (size_t i = 0; i < ha.size(); ++i) diff += ha[i] != hb[i];
  std::printf("last 2^19 blocks: %zu differing words\n", diff);
  {  // in-kernel clock of the library's form under full load
    u64* st;
    CK(hipMalloc(&st, AES4_GRID_CAP * 16));
    auto kern = &k_prg_blocks4p_clock;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, AES4_LDS_BYTES));
    for (int r = 0; r < 6; ++r) hipLaunchKernelGGL(kern, dim3(AES4_GRID_CAP), dim3(ABLOCK), AES4_LDS_BYTES, 0, a, key, 1ull, nblocks, st);
    CK(hipDeviceSynchronize());
    std::vector<u64> h(AES4_GRID_CAP * 2);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    double tt = 0, rr = 0;
    for (int bq = 0; bq < AES4_GRID_CAP; ++bq) tt += (double)h[2 * bq], rr += (double)h[2 * bq + 1];
    std::printf("in-kernel clock of k_prg_blocks (4 blocks per lane), sixth launch in a row: %.2f GHz; %.3g shader cycles per workgroup loop\n",
                tt / rr * 0.1, tt / AES4_GRID_CAP);
  }
  return diff != 0;
}
