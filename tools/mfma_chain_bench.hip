// tools/mfma_chain_bench.hip -- issue rate of v_mfma_i32_16x16x64_i8 / 32x32x32 as a function of how many independent
// accumulator chains rotate (1 = every instruction depends on the one before), one and two waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/_build/mfma_chain_bench tools/mfma_chain_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1);} } while (0)
constexpr int ITERS = 4096;

template <int CH, bool BIG>
__global__ __launch_bounds__(512) void k(int* out) {
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
  int r = 0;
  if constexpr (BIG) {
    v16i acc[CH];
    for (int j = 0; j < CH; ++j)
      for (int e = 0; e < 16; ++e) acc[j][e] = 0;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (int j = 0; j < CH; ++j)
      for (int e = 0; e < 16; ++e) r ^= acc[j][e];
  } else {
    v4i acc[CH];
    for (int j = 0; j < CH; ++j)
      for (int e = 0; e < 4; ++e) acc[j][e] = 0;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (int j = 0; j < CH; ++j)
      for (int e = 0; e < 4; ++e) r ^= acc[j][e];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int CH, bool BIG>
void run(int* out, int threads) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<CH, BIG>), dim3(256), dim3(threads), 0, 0, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<CH, BIG>), dim3(256), dim3(threads), 0, 0, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double ns = ms * 1e6 / 3.0 / ((double)ITERS * CH) / (threads / 256);
  std::printf("%s  %d chain(s)  %d wave(s)/SIMD: %6.2f ns of SIMD time per instruction\n", BIG ? "32x32x32" : "16x16x64", CH,
              threads / 256, ns);
}

int main() {
  int* out;
  CK(hipMalloc(&out, 256 * 512 * 4));
  run<1, false>(out, 256); run<2, false>(out, 256); run<4, false>(out, 256); run<8, false>(out, 256);
  run<1, false>(out, 512); run<2, false>(out, 512); run<4, false>(out, 512); run<8, false>(out, 512);
  run<1, true>(out, 256); run<2, true>(out, 256); run<4, true>(out, 256);
  run<1, true>(out, 512); run<2, true>(out, 512); run<4, true>(out, 512);
  return 0;
}
