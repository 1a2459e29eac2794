// tools/aes_bench.hip -- AES-128-CTR block rate of the single-table form (one T-table + rotations, 32 KiB of LDS per
// 256-thread workgroup) against the library's four-table PRG kernel (no rotations; 128 KiB of LDS, one 1024-thread workgroup per CU).
// The table contents are arbitrary here: both kernels compute the same function of them, and are compared word for word.
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

// four tables, NB independent blocks per lane, each through the fully unrolled single-block form (Aes4::block)
template <int NB>
__global__ __launch_bounds__(ABLOCK) void k_prg_blocks4u(u64* dst, AesKey key, u64 counter0, size_t nblocks) {
  SCL_AES4_PROLOGUE(key)
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
  time_it([&] { hipLaunchKernelGGL(k_prg_blocks1, dim3(1024), dim3(256), 0, 0, a, key, 12345ull, nblocks); }, "one table + rotations, 4 x 256 threads per CU");
  {
    auto kern = &k_prg_blocks;  // the library's kernel: four tables, one 1024-thread workgroup per CU
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, AES4_LDS_BYTES));
    time_it([&] { hipLaunchKernelGGL(kern, dim3(AES4_GRID_CAP), dim3(ABLOCK), AES4_LDS_BYTES, 0, b, key, 12345ull, nblocks); },
            "four tables, 1024 threads per CU (k_prg_blocks)");
  }
#define RUN4U(NB, name)                                                                                              \
  {                                                                                                                      \
    auto kern = &k_prg_blocks4u<NB>;                                                                                     \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, AES4_LDS_BYTES)); \
    time_it([&] { hipLaunchKernelGGL(kern, dim3(AES4_GRID_CAP), dim3(ABLOCK), AES4_LDS_BYTES, 0, b, key, 12345ull, nblocks); }, name); \
  }
  RUN4U(1, "four tables, unrolled rounds, 1 block/lane")
  RUN4U(2, "four tables, unrolled rounds, 2 blocks/lane")
  RUN4U(4, "four tables, unrolled rounds, 4 blocks/lane")
  CK(hipGetLastError());
  std::vector<u64> ha(1 << 20), hb(1 << 20);
  CK(hipMemcpy(ha.data(), a + (nblocks - (1 << 19)) * 2, ha.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b + (nblocks - (1 << 19)) * 2, hb.size() * 8, hipMemcpyDeviceToHost));
  size_t diff = 0;
  for (size_t i = 0; i < ha.size(); ++i) diff += ha[i] != hb[i];
  std::printf("last 2^19 blocks: %zu differing words\n", diff);
  return diff != 0;
}
