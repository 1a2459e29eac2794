// tools/gf2_mfma_recover_bench.hip -- k_recover_gf128_mfma (tools/recover_gf_mfma.hpp): the GF(2^128) reconstruct as a
// GF(2) matrix product on the matrix cores, C4's shape (40 parties, 1.25e7 secrets) in two launches of 20 parties, checked
// against the host's Gf128::mul on a sample of secrets, and timed.  Result (profiles/r3_gf2_mfma_recover.txt): correct, and
// 25 ns per matrix instruction per SIMD against 16-17 ns for the same instruction on idle operands (tools/gf2_mfma_probe.hip):
// 3.06 ms at C4's shard size where the nibble-table kernel of the library takes 2.7 -- not shipped.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/_build/gf2_mfma_recover_bench tools/gf2_mfma_recover_bench.hip
// usage: gf2_mfma_recover_bench [m=40] [N=12500000]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "recover_gf_mfma.hpp"
using namespace sclhip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1);} } while (0)

static u64 rs = 0x9E3779B97F4A7C15ull;
static u64 rnd() {
  rs ^= rs << 13;
  rs ^= rs >> 7;
  rs ^= rs << 17;
  return rs;
}

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
  const int m = argc > 1 ? atoi(argv[1]) : 40;
  const size_t N = argc > 2 ? strtoull(argv[2], 0, 10) : 12500000;
  std::vector<u128> lam(m);
  for (auto& v : lam) v = ((u128)rnd() << 64) | rnd();
  u64 *shares, *out;
  CK(hipMalloc(&shares, (size_t)m * N * 16));
  CK(hipMalloc(&out, N * 16));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, shares, (size_t)m * N * 2, 12345ull);
  const int passes = (m + GFM_PARTIES - 1) / GFM_PARTIES;
  uint4* tab;
  CK(hipMalloc(&tab, (size_t)passes * GFM_PARTIES * GFM_FRAG_WORDS * 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_recover_gf128_mfma), hipFuncAttributeMaxDynamicSharedMemorySize,
                         GFM_PARTIES * GFM_FRAG_WORDS * 4));
  auto run = [&]() {
    for (int p = 0; p < passes; ++p) {
      const int i0 = p * GFM_PARTIES, mp = m - i0 < GFM_PARTIES ? m - i0 : GFM_PARTIES;
      GfmLambdas L{};
      for (int i = 0; i < mp; ++i) L.v[i] = lam[i0 + i];
      uint4* t = tab + (size_t)p * GFM_PARTIES * (GFM_FRAG_WORDS / 4);
      const int m4 = (mp + 3) / 4 * 4;
      hipLaunchKernelGGL(k_gf_mfma_table, dim3(m4), dim3(256), 0, 0, t, L);
      const size_t tiles = (N + 63) / 64, blocks = (tiles + 7) / 8;
      hipLaunchKernelGGL(k_recover_gf128_mfma, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(GFM_BLOCK),
                         (size_t)m4 * GFM_FRAG_WORDS * 4, 0, out, shares + (size_t)i0 * N * 2, N, t, mp, m4, N, p ? out : nullptr);
    }
  };
  run();
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  // check a sample against the host field arithmetic
  const size_t sample[] = {0, 1, 31, 32, 63, 64, 12345, N / 2, N - 65, N - 2, N - 1};
  int bad = 0;
  for (size_t s : sample) {
    if (s >= N) continue;
    u128 want = 0;
    for (int i = 0; i < m; ++i) {
      u64 w[2];
      CK(hipMemcpy(w, shares + ((size_t)i * N + s) * 2, 16, hipMemcpyDeviceToHost));
      want ^= Gf128::mul(Gf128::Ctx{}, Gf128::ld(w), lam[i]);
    }
    u64 g[2];
    CK(hipMemcpy(g, out + s * 2, 16, hipMemcpyDeviceToHost));
    if (Gf128::ld(g) != want) {
      if (bad < 4) std::printf("  mismatch at secret %zu: got %016llx%016llx want %016llx%016llx\n", s, (unsigned long long)g[1], (unsigned long long)g[0],
                               (unsigned long long)(u64)(want >> 64), (unsigned long long)(u64)want);
      ++bad;
    }
  }
  std::printf("m = %d, N = %zu: %s\n", m, N, bad ? "WRONG" : "sample matches the host's Gf128::mul");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 10;
  CK(hipEventRecord(e0));
  for (int k = 0; k < reps; ++k) run();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double bytes = (double)(m + 1) * 16 * N;
  std::printf("reconstruct (%d launches of <= %d parties): %.3f ms = %.2f G secrets/s = %.2f TB/s-equivalent (%.3f of 8 TB/s); %.1f ns per matrix instruction per SIMD\n",
              passes, GFM_PARTIES, ms, N / ms / 1e6, bytes / ms / 1e9, bytes / ms / 1e9 / 8, ms * 1e6 / ((double)N / 64 * m * 16 / 1024));
  return bad ? 1 : 0;
}
