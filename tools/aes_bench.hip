// tools/aes_bench.hip -- AES-128-CTR block rate of the PRG kernel (one T-table + rotations, 32 KiB of LDS per 256-thread
// workgroup) against a four-table variant (no rotations; 128 KiB of LDS shared by one 1024-thread workgroup per CU).
// The table contents are arbitrary here: both kernels compute the same function of them, and are compared word for word.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/_build/aes_bench tools/aes_bench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../secure-computation-library_amd/csrc/kernels.hpp"
using namespace sclhip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1);} } while (0)

constexpr int TPB4 = 1024;
// table r, entry x, copy c at word (r * 256 + x) * 32 + c; lane reads copy lane & 31
template <int NB>
__device__ __forceinline__ void aes4_multi(const u32* t0, const u32* t2, const AesKey& key, const u64 (&ctr)[NB], u64 (&lo)[NB],
                                           u64 (&hi)[NB]) {
#define T0(x) t0[(x) << 5]
#define T1(x) t0[((x) << 5) + 8192]
#define T2(x) t2[(x) << 5]
#define T3(x) t2[((x) << 5) + 8192]
  u32 s[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    s[b][0] = (u32)ctr[b] ^ key.rk[0];
    s[b][1] = (u32)(ctr[b] >> 32) ^ key.rk[1];
    s[b][2] = 0x89ABCDEFu ^ key.rk[2];
    s[b][3] = 0x01234567u ^ key.rk[3];
  }
#pragma unroll 1
  for (int r = 1; r < 10; ++r) {
    const u32 k0 = key.rk[4 * r], k1 = key.rk[4 * r + 1], k2 = key.rk[4 * r + 2], k3 = key.rk[4 * r + 3];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const u32 s0 = s[b][0], s1 = s[b][1], s2 = s[b][2], s3 = s[b][3];
      s[b][0] = T0(s0 & 255) ^ T1((s1 >> 8) & 255) ^ T2((s2 >> 16) & 255) ^ T3(s3 >> 24) ^ k0;
      s[b][1] = T0(s1 & 255) ^ T1((s2 >> 8) & 255) ^ T2((s3 >> 16) & 255) ^ T3(s0 >> 24) ^ k1;
      s[b][2] = T0(s2 & 255) ^ T1((s3 >> 8) & 255) ^ T2((s0 >> 16) & 255) ^ T3(s1 >> 24) ^ k2;
      s[b][3] = T0(s3 & 255) ^ T1((s0 >> 8) & 255) ^ T2((s1 >> 16) & 255) ^ T3(s2 >> 24) ^ k3;
    }
  }
#define SB(x) ((T0(x) >> 8) & 255u)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const u32 s0 = s[b][0], s1 = s[b][1], s2 = s[b][2], s3 = s[b][3];
    const u32 o0 = (SB(s0 & 255) | (SB((s1 >> 8) & 255) << 8) | (SB((s2 >> 16) & 255) << 16) | (SB(s3 >> 24) << 24)) ^ key.rk[40];
    const u32 o1 = (SB(s1 & 255) | (SB((s2 >> 8) & 255) << 8) | (SB((s3 >> 16) & 255) << 16) | (SB(s0 >> 24) << 24)) ^ key.rk[41];
    const u32 o2 = (SB(s2 & 255) | (SB((s3 >> 8) & 255) << 8) | (SB((s0 >> 16) & 255) << 16) | (SB(s1 >> 24) << 24)) ^ key.rk[42];
    const u32 o3 = (SB(s3 & 255) | (SB((s0 >> 8) & 255) << 8) | (SB((s1 >> 16) & 255) << 16) | (SB(s2 >> 24) << 24)) ^ key.rk[43];
    lo[b] = (u64)o0 | ((u64)o1 << 32);
    hi[b] = (u64)o2 | ((u64)o3 << 32);
  }
#undef SB
#undef T0
#undef T1
#undef T2
#undef T3
}

template <int NB>
__global__ __launch_bounds__(TPB4) void k_prg_blocks4(u64* dst, AesKey key, u64 counter0, size_t nblocks) {
  extern __shared__ u32 tl[];  // 4 tables x 256 entries x 32 copies
  for (int e = threadIdx.x; e < 4 * 256 * 32; e += TPB4) {
    const int r = e >> 13, x = (e >> 5) & 255;
    const u32 v = key.te0[x];
    tl[e] = r == 0 ? v : (v << (8 * r)) | (v >> (32 - 8 * r));
  }
  __syncthreads();
  const u32* t0 = tl + (threadIdx.x & 31);
  const u32* t2 = t0 + 2 * 8192;
  const size_t G = (size_t)gridDim.x * TPB4;
  for (size_t q = (size_t)blockIdx.x * TPB4 + threadIdx.x; q < nblocks; q += NB * G) {
    u64 ctr[NB], lo[NB], hi[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) ctr[b] = counter0 + q + b * G;
    aes4_multi<NB>(t0, t2, key, ctr, lo, hi);
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

int main() {
  const size_t nblocks = (size_t)1 << 28;
  AesKey key;
  for (int i = 0; i < 44; ++i) key.rk[i] = 0x9E3779B9u * (i + 1);
  for (int i = 0; i < 256; ++i) key.te0[i] = 0x85EBCA6Bu * (i + 7) ^ (i << 13);
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
  time_it([&] { hipLaunchKernelGGL(k_prg_blocks, dim3(1024), dim3(256), 0, 0, a, key, 12345ull, nblocks); }, "one table + rotations, 4 x 256 threads per CU");
  {
    auto kern = &k_prg_blocks4<4>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    time_it([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(TPB4), 131072, 0, b, key, 12345ull, nblocks); }, "four tables, 1024 threads per CU, 4 blocks/lane");
  }
  {
    auto kern = &k_prg_blocks4<2>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    time_it([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(TPB4), 131072, 0, b, key, 12345ull, nblocks); }, "four tables, 1024 threads per CU, 2 blocks/lane");
  }
  CK(hipGetLastError());
  std::vector<u64> ha(1 << 20), hb(1 << 20);
  CK(hipMemcpy(ha.data(), a + (nblocks - (1 << 19)) * 2, ha.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b + (nblocks - (1 << 19)) * 2, hb.size() * 8, hipMemcpyDeviceToHost));
  size_t diff = 0;
  for (size_t i = 0; i < ha.size(); ++i) diff += ha[i] != hb[i];
  std::printf("last 2^19 blocks: %zu differing words\n", diff);
  return diff != 0;
}
