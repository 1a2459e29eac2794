// tools/gemm_bench.hip -- k_gemm_mfma_m61 (csrc/gemm_mfma.hpp) on square shapes: the kernel as shipped, and (-DGEMM_PROBE_NO_LOADS)
// with every k-step reading the SAME fragments -- what the loop reaches when its operands always hit L1, i.e. how much of the
// shipped kernel's time is operand delivery rather than matrix instructions.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DGEMM_PROBE_NO_LOADS] -o tools/_build/gemm_bench tools/gemm_bench.hip
// usage: gemm_bench [n = 4096]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../secure-computation-library_amd/csrc/gemm_mfma.hpp"
using namespace sclhip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1);} } while (0)

__global__ void k_fill(u64* p, size_t n, u64 seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    u64 x = seed + i * 0x9E3779B97F4A7C15ull;
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    p[i] = x & M61::P;
  }
}

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 4096;
  const size_t kt = (n + 31) / 32, tiles = (n + 31) / 32;
  u64 *A, *B, *C;
  u64x2 *Ap, *Bp;
  CK(hipMalloc(&A, n * n * 8));
  CK(hipMalloc(&B, n * n * 8));
  CK(hipMalloc(&C, n * n * 8));
  CK(hipMalloc(&Ap, tiles * kt * MF_LIMBS * 64 * 16));
  CK(hipMalloc(&Bp, tiles * kt * MF_LIMBS * 64 * 16));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, n * n, 1ull);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, B, n * n, 2ull);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto planes = [&] {
    hipLaunchKernelGGL(k_gemm_planes_a<>, dim3(65536), dim3(256), 0, 0, Ap, A, n, n, n, kt);
    hipLaunchKernelGGL(k_gemm_planes_b<>, dim3(65536), dim3(256), 0, 0, Bp, B, n, n, n, kt);
  };
  const size_t wgs = ((tiles + 1) / 2) * ((tiles + 1) / 2);
  auto maink = [&] { hipLaunchKernelGGL(k_gemm_mfma_m61<>, dim3((unsigned)wgs), dim3(256), 0, 0, C, n, Ap, Bp, n, n, kt, kt, (size_t)0); };
  auto time = [&](auto&& fn, const char* what) {
    for (int i = 0; i < 3; ++i) fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 10; ++i) fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 10;
    std::printf("%-28s n=%zu  %8.3f ms  %7.2f T multiply-adds/s\n", what, n, ms, (double)n * n * n / ms / 1e9);
  };
  time(planes, "digit planes of A and B");
  time(maink,
#if defined(GEMM_PROBE_NO_LOADS)
       "main kernel, same fragments"
#else
       "main kernel"
#endif
  );
  CK(hipGetLastError());
  return 0;
}
