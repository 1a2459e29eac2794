// tools/gf2_mfma_probe.hip -- is a GF(2) matrix product (GF(2^128) multiplication by constants = a 128 x 128 bit matrix per
// party) worth doing on the matrix cores?  v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (e2m1) operands holds one bit per
// 4-bit element and is the densest form the hardware has: 32 x 32 x 64 multiply-adds per instruction.
//   1. operand maps: the guide documents the bf16 maps only ("other dtypes: check the map with exact integer data").  With
//      0/1 data the f32 sums are exact, so a candidate map either reproduces the host product or it does not.
//   2. issue rate: back-to-back instructions on independent accumulators, operands in registers; then with the A operand
//      re-read from LDS for every instruction (what a kernel with the matrix in LDS does).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/_build/gf2_mfma_probe tools/gf2_mfma_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1);} } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t u32;

// fp4 e2m1: 0x0 = 0, 0x1 = 0.5, 0x2 = 1, 0x4 = 2.  A entries 0 / 2.0, B entries 0 / 0.5: a product is 0 or 1.
constexpr int FMT_FP4 = 4;
constexpr int SCALE_ONE = 0x7F7F7F7F;  // E8M0 127 = 2^0 in every byte

__device__ __forceinline__ v16f mfma_fp4(v8i a, v8i b, v16f c) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, FMT_FP4, FMT_FP4, 0, SCALE_ONE, 0, SCALE_ONE);
}

// one instruction on operands the host packed; C out as the lanes hold it
__global__ void k_one(const u32* a_words, const u32* b_words, float* c_out) {
  const int l = threadIdx.x;
  v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < 4; ++j) {
    a[j] = (int)a_words[l * 4 + j];
    b[j] = (int)b_words[l * 4 + j];
  }
  v16f c = {};
  c = mfma_fp4(a, b, c);
  for (int r = 0; r < 16; ++r) c_out[l * 16 + r] = c[r];
}

// issue rate: NACC independent accumulators, REPS rounds, operands in registers
template <int NACC>
__global__ void k_rate(float* sink, int reps, long long* cycles) {
  v8i a = {(int)threadIdx.x, 1, 2, 3, 0, 0, 0, 0}, b = {4, 5, (int)threadIdx.x, 7, 0, 0, 0, 0};
  v16f c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = v16f{};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = mfma_fp4(a, b, c[i]);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += c[i][r];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// the same with the A operand read from LDS before every instruction (1 KiB per instruction, lane-linear), B from
// registers shared by NT column tiles
template <int NT>
__global__ void k_rate_lds(float* sink, int reps, long long* cycles, int frags) {
  extern __shared__ uint4 lds[];
  for (int i = threadIdx.x; i < frags * 64; i += blockDim.x) lds[i] = make_uint4(i, i * 3, i * 5, i * 7);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  v8i b[NT];
  for (int i = 0; i < NT; ++i) b[i] = v8i{4 + i, 5, (int)threadIdx.x, 7, 0, 0, 0, 0};
  v16f c[NT][4];
  for (int i = 0; i < NT; ++i)
    for (int m = 0; m < 4; ++m) c[i][m] = v16f{};
  const long long t0 = __builtin_amdgcn_s_memtime();
  int f = 0;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const uint4 w = lds[f * 64 + lane];
      f = f + 1 == frags ? 0 : f + 1;
      const v8i a = {(int)w.x, (int)w.y, (int)w.z, (int)w.w, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < NT; ++i) c[i][m] = mfma_fp4(a, b[i], c[i][m]);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NT; ++i)
    for (int m = 0; m < 4; ++m)
      for (int r = 0; r < 16; ++r) s += c[i][m][r];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  rs ^= rs << 13;
  rs ^= rs >> 7;
  rs ^= rs << 17;
  return rs;
}

int main() {
  // ---- 1. operand maps ---------------------------------------------------------------------------------------------
  // candidate: lane l = (h = l >> 5, r = l & 31) holds A[row r][k = 32 h + j] and B[k = 32 h + j][col r], j = 0..31, element j in
  // nibble j & 7 of register j >> 3 (low nibble first); C: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)
  std::vector<int> A(32 * 64), B(64 * 32);
  for (auto& v : A) v = (int)(rnd() & 1);
  for (auto& v : B) v = (int)(rnd() & 1);
  std::vector<u32> aw(64 * 4, 0), bw(64 * 4, 0);
  for (int l = 0; l < 64; ++l) {
    const int h = l >> 5, r = l & 31;
    for (int j = 0; j < 32; ++j) {
      const int k = 32 * h + j;
      if (A[r * 64 + k]) aw[l * 4 + (j >> 3)] |= 0x4u << (4 * (j & 7));  // 2.0
      if (B[k * 32 + r]) bw[l * 4 + (j >> 3)] |= 0x1u << (4 * (j & 7));  // 0.5
    }
  }
  u32 *da, *db;
  float* dc;
  CK(hipMalloc(&da, aw.size() * 4));
  CK(hipMalloc(&db, bw.size() * 4));
  CK(hipMalloc(&dc, 64 * 16 * 4));
  CK(hipMemcpy(da, aw.data(), aw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, bw.data(), bw.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, da, db, dc);
  CK(hipDeviceSynchronize());
  std::vector<float> C(64 * 16);
  CK(hipMemcpy(C.data(), dc, C.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int reg = 0; reg < 16; ++reg) {
      const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
      int want = 0;
      for (int k = 0; k < 64; ++k) want += A[row * 64 + k] * B[k * 32 + col];
      if (C[l * 16 + reg] != (float)want) {
        if (bad < 5) std::printf("  mismatch lane %d reg %d: got %g want %d\n", l, reg, C[l * 16 + reg], want);
        ++bad;
      }
    }
  std::printf("fp4 32x32x64 operand map (lane = 32 h + r holds k = 32 h + j, nibble j & 7 of register j >> 3): %s (%d of 1024 wrong)\n",
              bad ? "WRONG" : "confirmed with exact 0/1 data", bad);

  // ---- 2. issue rate -----------------------------------------------------------------------------------------------
  float* sink;
  long long* dcy;
  CK(hipMalloc(&sink, 1024 * 1024 * 4));
  CK(hipMalloc(&dcy, 8));
  auto report = [&](const char* what, double mfmas_per_wave, int waves_per_simd) {
    long long cy = 0;
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&cy, dcy, 8, hipMemcpyDeviceToHost));
    // s_memtime ticks at 100 MHz on this part; convert with the event time instead
    std::printf("  %-58s s_memtime delta %lld for %.0f MFMAs per wave (%d waves per SIMD)\n", what, cy, mfmas_per_wave, waves_per_simd);
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timed = [&](auto launch, const char* what, double mfmas_per_wave, int waves_per_cu) {
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // all 256 CUs busy: MFMAs per SIMD = mfmas_per_wave * waves_per_cu / 4
    const double per_simd = mfmas_per_wave * waves_per_cu / 4.0;
    const double macs = per_simd * 4 * 256 * 32.0 * 32 * 64;
    std::printf("  %-62s %8.3f ms  %.1f ns per MFMA per SIMD  = %.2f P bit-MAC/s chip-wide\n", what, ms, ms * 1e6 / per_simd,
                macs / (ms * 1e-3) / 1e15);
  };
  const int reps = 20000;
  timed([&] { hipLaunchKernelGGL(k_rate<4>, dim3(256), dim3(256), 0, 0, sink, reps, dcy); }, "registers, 4 accumulators, 1 wave per SIMD", 4.0 * reps, 4);
  timed([&] { hipLaunchKernelGGL(k_rate<4>, dim3(256), dim3(512), 0, 0, sink, reps, dcy); }, "registers, 4 accumulators, 2 waves per SIMD", 4.0 * reps, 8);
  timed([&] { hipLaunchKernelGGL(k_rate<8>, dim3(256), dim3(256), 0, 0, sink, reps, dcy); }, "registers, 8 accumulators, 1 wave per SIMD", 8.0 * reps, 4);
  {
    auto k1 = &k_rate_lds<1>;
    auto k2 = &k_rate_lds<2>;
    const int frags = 128;  // 128 KiB of A fragments
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, frags * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, frags * 1024));
    const int r2 = 5000;
    timed([&] { hipLaunchKernelGGL(k1, dim3(256), dim3(256), frags * 1024, 0, sink, r2, dcy, frags); }, "A from LDS per MFMA (1 column tile), 1 wave per SIMD", 4.0 * r2, 4);
    timed([&] { hipLaunchKernelGGL(k1, dim3(256), dim3(512), frags * 1024, 0, sink, r2, dcy, frags); }, "A from LDS per MFMA (1 column tile), 2 waves per SIMD", 4.0 * r2, 8);
    timed([&] { hipLaunchKernelGGL(k2, dim3(256), dim3(256), frags * 1024, 0, sink, r2, dcy, frags); }, "A from LDS per 2 MFMAs (2 column tiles), 1 wave per SIMD", 8.0 * r2, 4);
    timed([&] { hipLaunchKernelGGL(k2, dim3(256), dim3(512), frags * 1024, 0, sink, r2, dcy, frags); }, "A from LDS per 2 MFMAs (2 column tiles), 2 waves per SIMD", 8.0 * r2, 8);
  }
  (void)report;
  return 0;
}
