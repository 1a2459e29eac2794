// tools/coexec_bench.hip -- do i8 matrix instructions and integer VALU instructions overlap on one gfx950 SIMD?
// The Mersenne61 MFMA share kernel (csrc/share_mfma.hpp) needs ~1 matrix instruction (32 cycles) per ~7 VALU
// instructions of recombination; this measures what the SIMD can sustain for that mix
//   mode 0  matrix instructions only                     mode 1  VALU only
//   mode 2  one wave per SIMD issuing {1 MFMA, KV VALU} interleaved
//   mode 3  two waves per SIMD: one issues only MFMAs, the other only VALU
//   mode 4  two waves per SIMD, both interleaved as in mode 2
// build: hipcc -O3 --offload-arch=gfx950 -o coexec_bench tools/coexec_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1);} } while (0)
constexpr int ITERS = 2048;

template <int KV>
__device__ __forceinline__ void valu_group(long long (&t)[8], int x, int y) {
#pragma unroll
  for (int j = 0; j < KV; ++j) {
    u64 carry;
    asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(t[j & 7]), "=s"(carry) : "v"(x), "v"(y));
  }
}

// DO_M: this wave issues matrix instructions, DO_V: VALU groups
template <int KV>
__device__ __forceinline__ void body(bool do_m, bool do_v, int* out) {
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
#ifdef AB_IN_AGPR
  asm volatile("" : "=a"(a) : "0"(a));
  asm volatile("" : "=a"(b) : "0"(b));
#endif
  v16i acc[4];
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 16; ++e) acc[j][e] = 0;
  long long t[8];
  for (int j = 0; j < 8; ++j) t[j] = threadIdx.x + j;
  int x = threadIdx.x * 3 + 1, y = threadIdx.x ^ 0x55;
  if (do_m && do_v) {
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#ifdef READ_ACC
        // the VALU group's multiplicand comes from a tile finished two matrix instructions ago (as the share kernels'
        // recombination reads the previous word's accumulators)
        valu_group<KV>(t, acc[(j + 2) & 3][5], y);
#else
        valu_group<KV>(t, x, y);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (do_m) {
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (do_v) {
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        valu_group<KV>(t, x, y);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  int r = 0;
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 16; ++e) r ^= acc[j][e];
  for (int j = 0; j < 8; ++j) r ^= (int)t[j] ^ (int)(t[j] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// VALU only, written in C so that the compiler emits plain v_mad_i64_i32 without hazard nops: CH independent chains
template <int CH>
__global__ __launch_bounds__(512) void kv(int* out, int x, int y) {
  long long t[CH];
  for (int j = 0; j < CH; ++j) t[j] = threadIdx.x + j;
  const int xx = x + threadIdx.x, yy = y ^ threadIdx.x;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < CH; ++j) t[j] = (long long)(int)t[j] * (int)(yy + r) + t[j];
  }
  int r = 0;
  for (int j = 0; j < CH; ++j) r ^= (int)t[j] ^ (int)(t[j] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int CH>
void runv(int* out, int threads) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL((kv<CH>), dim3(256), dim3(threads), 0, 0, out, 3, 5);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((kv<CH>), dim3(256), dim3(threads), 0, 0, out, 3, 5);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double ns = ms * 1e6 / 3.0 / (ITERS * 4.0 * CH) / (threads / 256);
  std::printf("VALU only, %d wave(s) per SIMD, %d chains: %.2f ns of SIMD time per v_mad_i64_i32 (= %.1f cycles at 2.4 GHz)\n",
              threads / 256, CH, ns, ns * 2.4);
}

template <int MODE, int KV>
__global__ __launch_bounds__(512) void k(int* out) {
  const int w = threadIdx.x >> 6;  // waves 0..3 land on SIMDs 0..3, waves 4..7 again on 0..3
  if constexpr (MODE == 0) body<KV>(true, false, out);
  if constexpr (MODE == 1) body<KV>(false, true, out);
  if constexpr (MODE == 2 || MODE == 4) body<KV>(true, true, out);
  if constexpr (MODE == 3) {
    if (w < 4) body<KV>(true, false, out);
    else body<KV>(false, true, out);
  }
}

template <int MODE, int KV>
void run(int* out) {
  const int threads = (MODE >= 3) ? 512 : 256, blocks = 256;  // one workgroup per CU
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<MODE, KV>), dim3(blocks), dim3(threads), 0, 0, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<MODE, KV>), dim3(blocks), dim3(threads), 0, 0, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double ns_per_group = ms * 1e6 / 3.0 / (ITERS * 4.0);  // one group = 1 MFMA and/or KV VALU per issuing wave
  std::printf("mode %d  KV=%2d  %7.2f ns per {MFMA,%d VALU} group per wave  (= %.1f cycles at 2.4 GHz)\n", MODE, KV,
              ns_per_group, KV, ns_per_group * 2.4);
}

int main() {
  int* out;
  CK(hipMalloc(&out, 256 * 512 * 4));
  run<0, 7>(out);
  run<1, 4>(out);
  run<1, 7>(out);
  run<1, 8>(out);
  run<2, 4>(out);
  run<2, 6>(out);
  run<2, 7>(out);
  run<2, 8>(out);
  run<2, 12>(out);
  run<3, 4>(out);
  run<3, 7>(out);
  run<3, 8>(out);
  run<4, 4>(out);
  run<4, 7>(out);
  runv<8>(out, 256);
  runv<8>(out, 512);
  runv<2>(out, 256);
  runv<2>(out, 512);
  return 0;
}
