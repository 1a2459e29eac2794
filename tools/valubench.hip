// tools/valubench.hip -- integer VALU issue rates on gfx950 for the instructions the field arithmetic uses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1);} } while (0)
constexpr int ITERS = 4096, CH = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(u64* out, u32 seed) {
  u64 a[CH];
  u32 b = seed | 1, c = threadIdx.x * 2654435761u + seed;
  for (int j = 0; j < CH; ++j) a[j] = (u64)threadIdx.x * 0x9E3779B97F4A7C15ull + j;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      if constexpr (OP == 0) a[j] = (u64)(u32)a[j] * b + a[j];                     // v_mad_u64_u32
      if constexpr (OP == 1) a[j] = (u32)a[j] * b;                                  // v_mul_lo_u32
      if constexpr (OP == 2) a[j] = __umulhi((u32)a[j], b);                         // v_mul_hi_u32
      if constexpr (OP == 3) a[j] = __umul24((u32)a[j] & 0xffffff, b & 0xffffff) + c;        // v_mad_u32_u24
      if constexpr (OP == 4) a[j] = a[j] + (((u64)c << 32) | b);                    // 64-bit add
      if constexpr (OP == 5) a[j] = (u32)a[j] ^ (c + (u32)(a[j] >> 3));             // 32-bit alu
      if constexpr (OP == 6) a[j] = (a[j] >> 61) + (a[j] & 0x1FFFFFFFFFFFFFFFull) + b;  // mersenne fold
    }
  }
  u64 r = 0;
  for (int j = 0; j < CH; ++j) r ^= a[j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
  const int blocks = 256 * 8;
  u64* out;
  CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
  const char* names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "add u64", "alu32 x2", "mersenne fold"};
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
#define RUN(OP)                                                                                          \
  {                                                                                                      \
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u);                               \
    CK(hipDeviceSynchronize());                                                                          \
    CK(hipEventRecord(a));                                                                               \
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u);   \
    CK(hipEventRecord(b));                                                                               \
    CK(hipEventSynchronize(b));                                                                          \
    float ms;                                                                                            \
    CK(hipEventElapsedTime(&ms, a, b));                                                                  \
    const double ops = 5.0 * blocks * 256 * (double)ITERS * CH;                                          \
    const double per_s = ops / (ms * 1e-3);                                                              \
    std::printf("%-16s %8.2f T lane-ops/s  = %.2f lanes/clk/SIMD @2.4GHz -> %.1f cycles per wave64 instr\n", names[OP], \
                per_s / 1e12, per_s / (1024 * 2.4e9), 64.0 / (per_s / (1024 * 2.4e9)));                  \
  }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
  return 0;
}
