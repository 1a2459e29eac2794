// tools/gfpos_bench.hip -- k_recover_gf128_pos (csrc/kernels.hpp) with compiler-visible LDS reads (the library's) and with the
// hand-issued reads of rounds 2-3 one, two or three batches ahead of their s_waitcnt (tools/gfpos_asm.hpp, NBUF = 2, 3, 4), at C4's shard size, against k_recover_gf128 word for word.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/_build/gfpos_bench tools/gfpos_bench.hip
// usage: gfpos_bench [N=12500000] [m=40]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gfpos_asm.hpp"
using namespace sclhip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1);} } while (0)

__global__ void k_fill(u64* p, size_t n, u64 seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    u64 x = seed + i * 0x9E3779B97F4A7C15ull;
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    p[i] = x;
  }
}

int main(int argc, char** argv) {
  const size_t N = argc > 1 ? strtoull(argv[1], 0, 10) : 12500000;
  const int m = argc > 2 ? atoi(argv[2]) : 40;
  BigTable<Gf128> big;
  u64 x = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < m; ++i) {
    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
    const u64 lo = x;
    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
    big.v[i] = ((u128)x << 64) | lo;
  }
  u64 *sh, *ref, *out;
  CK(hipMalloc(&sh, (size_t)m * N * 16));
  CK(hipMalloc(&ref, N * 16));
  CK(hipMalloc(&out, N * 16));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, sh, (size_t)m * N * 2, 777ull);
  hipLaunchKernelGGL(k_recover_gf128<>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, ref, sh, N, big, m, N, (const u64*)nullptr);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t lds = gfpos_lds_bytes((size_t)m);
  auto run = [&](auto kern, int blk, int grid, const char* name) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(out, 0, N * 16));
    auto launch = [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(blk), lds, 0, out, sh, N, big, m, N); };
    launch();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    std::vector<u64> a(N * 2), b(N * 2);
    CK(hipMemcpy(a.data(), ref, N * 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), out, N * 16, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 10;
    std::printf("%-46s %7.3f ms  %5.2f G secrets/s  %5.2f TB/s-equivalent (%.3f of 8)  differing words: %zu\n", name, ms, N / ms / 1e6,
                (double)(m + 1) * 16 * N / ms / 1e9, (double)(m + 1) * 16 * N / ms / 1e9 / 8, diff);
  };
  run(&k_recover_gf128_pos<512, 2>, 512, 512, "pos tables, compiler-visible reads (NBUF 0)");
  run(&k_recover_gf128_pos<512, 2, GfposPipeAsm<2>>, 512, 512, "pos tables, asm reads 1 batch ahead (round 3)");
  run(&k_recover_gf128_pos<512, 2>, 512, 512, "pos tables, compiler-visible reads (again)");
  run(&k_recover_gf128_pos<1024, 1>, 1024, 256, "compiler-visible, one 1024-thread workgroup");
  run(&k_recover_gf128_pos<512, 2, GfposPipeFill<4>>, 512, 512, "compiler-visible + 4 filler vector ops per stage");
  run(&k_recover_gf128_pos<512, 2, GfposPipeFill<8>>, 512, 512, "compiler-visible + 8 filler vector ops per stage");
  run(&k_recover_gf128_pos<512, 2, GfposPipeFill<12>>, 512, 512, "compiler-visible + 12 filler vector ops per stage");
  run(&k_recover_gf128_pos<512, 2, GfposPipeFill<16>>, 512, 512, "compiler-visible + 16 filler vector ops per stage");
  run(&k_recover_gf128_pos<512, 2, GfposPipeAsm<3>>, 512, 512, "pos tables, reads 2 batches ahead");
  run(&k_recover_gf128_pos<512, 2, GfposPipeAsm<4>>, 512, 512, "pos tables, reads 3 batches ahead");
  run(&k_recover_gf128_pos<1024, 1, GfposPipeAsm<3>>, 1024, 256, "pos tables, 2 ahead, one 1024-thread workgroup");
  run(&k_recover_gf128_pos<512, 2, GfposPipeAsm<2>>, 512, 512, "pos tables, asm reads 1 batch ahead (again)");
  return 0;
}
