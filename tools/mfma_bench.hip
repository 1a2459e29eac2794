// tools/mfma_bench.hip -- the Mersenne61 matrix-core share kernels side by side, outside the library:
// k_share_mfma_m61 (two waves per SIMD, word bursts) against k_share_mfma_m61_pipe (one wave per SIMD, matrix
// and VALU instructions interleaved).  Checks that the two agree bit for bit and a sample against a host Horner.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/_build/mfma_bench tools/mfma_bench.hip
//   k_share_mfma_m61_p16's experiment switches (csrc/share_mfma.hpp): -DMF16_STAGGER=0 (all eight waves in step),
//   -DMF16_ABL=<bits> (1 no matrix instructions, 2 no recombination, 4 no stores, 8 no recode / fetch, 16 no fragment loads,
//   32 no barriers; results then wrong by construction) -- profiles/r3_p16_ablation.txt; counters: tools/p16_sq.sh
// usage: mfma_bench [n=128] [t=42] [N=10000000]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#ifdef MF16_STAMP
// diagnostic build (-DMF16_STAMP): waves 0 (early) and 4 (late) of every workgroup accumulate s_memtime deltas between the hook
// points of k_share_mfma_m61_p16's trip (csrc/share_mfma.hpp, MF16_HOOK) and write them out at the end; printed below as mean
// cycles per trip and phase, with the in-kernel clock (s_memtime ticks per s_memrealtime tick x 100 MHz).  The stamps cost
// cycles of their own (each waits for outstanding LDS / scalar-memory operations): read the split, not the total.
__device__ unsigned long long* mf16_stamp_buf;
#define MF16_HOOK_DECL                                                                                     \
  const bool stamp_on = (wu & 3) == 0;                                                                     \
  unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_trips = 0;                      \
  unsigned long long stamp_last = __builtin_amdgcn_s_memtime();                                            \
  const unsigned long long stamp_t0 = stamp_last, stamp_r0 = __builtin_amdgcn_s_memrealtime();
#define MF16_HOOK(p)                                                                                       \
  do {                                                                                                     \
    if (stamp_on) {                                                                                        \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                          \
      stamp_acc[p] += t_ - stamp_last;                                                                     \
      stamp_last = t_;                                                                                     \
      if ((p) == 9) ++stamp_trips;                                                                         \
    }                                                                                                      \
  } while (0)
#define MF16_HOOK_END                                                                                      \
  if (stamp_on && lane == 0) {                                                                             \
    unsigned long long* o_ = mf16_stamp_buf + ((size_t)blockIdx.x * 2 + (wu >> 2)) * 16;                   \
    for (int p_ = 0; p_ < 10; ++p_) o_[p_] = stamp_acc[p_];                                                \
    o_[10] = stamp_trips;                                                                                  \
    o_[11] = __builtin_amdgcn_s_memtime() - stamp_t0;                                                      \
    o_[12] = __builtin_amdgcn_s_memrealtime() - stamp_r0;                                                  \
  }
#endif
#include "../secure-computation-library_amd/csrc/share_mfma.hpp"
using namespace sclhip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1);} } while (0)

static u64 rng_state = 0x9E3779B97F4A7C15ull;
static u64 rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 128, t = argc > 2 ? atoi(argv[2]) : 42;
  const size_t N = argc > 3 ? strtoull(argv[3], 0, 10) : 10000000;
  constexpr int KS = 2, MT = 4;
  if (t + 1 > 64 || n > 128 || n <= 96) { std::printf("needs 96 < n <= 128, t <= 63\n"); return 1; }
  const M61::Ctx ctx{};
  // table (as mfma_table in capi.hip)
  const int ROWB = mf_rowb(KS);
  std::vector<unsigned char> host(mf_a_bytes(KS, MT), 0);
  for (int i = 0; i < n; ++i) {
    u64 v = 1;
    for (int k = 0; k <= t; ++k) {
      if (k) v = M61::mul(ctx, v, (u64)(i + 1));
      const u64 digits = mf_recode(v);
      for (int l = 0; l < MF_LIMBS; ++l) host[((size_t)(l * MT + i / 32) * 32 + (i % 32)) * ROWB + k] = (unsigned char)(digits >> (8 * l));
    }
  }
  unsigned char* tab;
  CK(hipMalloc(&tab, host.size()));
  CK(hipMemcpy(tab, host.data(), host.size(), hipMemcpyHostToDevice));
  std::vector<u64> hc((size_t)(t + 1) * N);
  for (auto& x : hc) x = rnd() % M61::P;
  for (int k = 0; k <= t && N > 2; ++k) { hc[(size_t)k * N] = M61::P - 1; hc[(size_t)k * N + 1] = 0; }
  u64 *c, *out0, *out1;
  CK(hipMalloc(&c, hc.size() * 8));
  CK(hipMemcpy(c, hc.data(), hc.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&out0, (size_t)n * N * 8));
  CK(hipMalloc(&out1, (size_t)n * N * 8));
  CK(hipMemset(out0, 0xAA, (size_t)n * N * 8));
  CK(hipMemset(out1, 0x55, (size_t)n * N * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time_it = [&](auto launch, const char* name) {
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    std::printf("%-28s %8.3f ms  %6.2f G secrets/s  %6.0f GB/s\n", name, ms, N / ms / 1e6, (double)(t + 1 + n) * 8 * N / ms / 1e6);
  };
  {
    const size_t shmem = mf_b_bytes(KS, MT);
    auto kern = &k_share_mfma_m61<KS, MT, true, 512>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const size_t nblocks = (N + 63) / 64;
    const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);
    time_it([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shmem, 0, out0, N, c, c + N, N, tab, t, n, N); }, "k_share_mfma_m61 (bursts)");
  }
  {
    const size_t shmem = mf_b_bytes(KS, MT, 1);
    auto kern = &k_share_mfma_m61_pipe<KS>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const size_t nblocks = (N + 31) / 32;
    const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);
    time_it([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, 0, out1, N, c, c + N, N, tab, t, n, N); }, "k_share_mfma_m61_pipe");
  }
  u64* out2;
  CK(hipMalloc(&out2, (size_t)n * N * 8));
  CK(hipMemset(out2, 0x33, (size_t)n * N * 8));
  {
    const size_t shmem = 2 * mf_b_bytes(KS, MT, 1);
    auto kern = &k_share_mfma_m61_p16<>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const size_t nblocks = (N + 31) / 32;
    const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);
#ifdef MF16_STAMP
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, (size_t)grid * 2 * 16 * 8));
    CK(hipMemset(stamps, 0, (size_t)grid * 2 * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(mf16_stamp_buf), &stamps, sizeof(stamps)));
#endif
    time_it([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shmem, 0, out2, N, c, c + N, N, tab, t, n, N); }, "k_share_mfma_m61_p16");
#ifdef MF16_STAMP
    {
      std::vector<unsigned long long> h((size_t)grid * 2 * 16);
      CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
      static const char* names[10] = {"loop edge -> trip start", "trip-start barrier", "fragment loads (16 ds_read_b128 + wait)",
                                      "word 3 + first quarter of word 2", "recode (with its wait for the coefficients)",
                                      "second quarter of word 2 + fetch issue", "rest of word 2", "mid-trip barrier",
                                      "words 1 and 0 + last recombination", "fold + stores"};
      for (int g = 0; g < 2; ++g) {
        double acc[13] = {0};
        for (unsigned b = 0; b < grid; ++b)
          for (int p = 0; p < 13; ++p) acc[p] += (double)h[((size_t)b * 2 + g) * 16 + p];
        const double trips = acc[10];
        double sum = 0;
        for (int p = 0; p < 10; ++p) sum += acc[p];
        std::printf("  %s waves (wave %d of each workgroup): %.0f trips each, in-kernel clock %.2f GHz, %.0f cycles per trip\n",
                    g ? "late" : "early", 4 * g, trips / grid, acc[11] / acc[12] * 0.1, sum / trips);
        for (int p = 0; p < 10; ++p) std::printf("    %-46s %7.0f cycles  %5.1f %%\n", names[p], acc[p] / trips, 100 * acc[p] / sum);
      }
    }
#endif
    std::vector<u64> a0((size_t)n * N), a2((size_t)n * N);
    CK(hipMemcpy(a0.data(), out0, a0.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(a2.data(), out2, a2.size() * 8, hipMemcpyDeviceToHost));
    size_t d2 = 0, firstbad = (size_t)-1;
    for (size_t i = 0; i < a0.size(); ++i)
      if (a0[i] != a2[i]) { if (firstbad == (size_t)-1) firstbad = i; ++d2; }
    std::printf("p16 vs bursts: %zu differing values", d2);
    if (d2) std::printf(" (first at party %zu secret %zu: %llx vs %llx)", firstbad / N, firstbad % N, (unsigned long long)a2[firstbad], (unsigned long long)a0[firstbad]);
    std::printf("\n");
  }
  if (argc > 5) {  // where the p16 kernel's time goes on the memory side: share rows that alias (results are wrong by construction)
    const size_t shmem = 2 * mf_b_bytes(KS, MT, 1);
    auto kern = &k_share_mfma_m61_p16<>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const size_t nblocks = (N + 31) / 32;
    const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);
    time_it([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shmem, 0, out2, N, c, c + N, N, tab, t, n, N); }, "p16: as shipped");
    time_it([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shmem, 0, out2, (size_t)0, c, c + N, N, tab, t, n, N); }, "p16: all share rows alias");
    time_it([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shmem, 0, out2, N, c, c, (size_t)0, tab, t, n, N); }, "p16: all coefficient rows alias");
    time_it([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shmem, 0, out2, (size_t)0, c, c, (size_t)0, tab, t, n, N); }, "p16: both alias");
    time_it([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shmem, 0, out2 + 4, N, c, c + N, N, tab, t, n, N - 4); }, "p16: share rows 32 B into a line");
    return 0;
  }
  if (argc > 4) {  // ablations of the pipelined kernel (results are wrong by construction)
    const size_t shmem = mf_b_bytes(KS, MT, 1);
    const size_t nblocks = (N + 31) / 32;
    const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);
#define ABL_RUN(ABL, name)                                                                                             \
  {                                                                                                                    \
    auto kern = &k_share_mfma_m61_pipe<KS, ABL>;                                                                       \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
    time_it([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, 0, out0, N, c, c + N, N, tab, t, n, N); }, name); \
  }
    ABL_RUN(1, "pipe: no matrix instrs")
    ABL_RUN(2, "pipe: no recombination")
    ABL_RUN(4, "pipe: no stores")
    ABL_RUN(8, "pipe: no recode/fetch")
    ABL_RUN(3, "pipe: no MFMA, no recomb")
    ABL_RUN(7, "pipe: only recode/fetch")
    ABL_RUN(12, "pipe: no stores, no recode")
    ABL_RUN(14, "pipe: MFMA + epilogue VALU")
    ABL_RUN(6, "pipe: MFMA + recode only")
    ABL_RUN(15, "pipe: nothing")
    return 0;
  }
  CK(hipGetLastError());
  std::vector<u64> h0((size_t)n * N), h1((size_t)n * N);
  CK(hipMemcpy(h0.data(), out0, h0.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1.data(), out1, h1.size() * 8, hipMemcpyDeviceToHost));
  size_t diff = 0;
  for (size_t i = 0; i < h0.size(); ++i) diff += h0[i] != h1[i];
  size_t bad = 0;
  for (size_t s = 0; s < N; s += (s < 64 ? 1 : N / 97 + 1))
    for (int i = 0; i < n; ++i) {
      u64 y = 0;
      for (int k = t; k >= 0; --k) y = M61::add(ctx, M61::mul(ctx, y, (u64)(i + 1)), hc[(size_t)k * N + s]);
      bad += y != h1[(size_t)i * N + s];
    }
  std::printf("pipe vs bursts: %zu differing values; pipe vs host Horner sample: %zu wrong\n", diff, bad);
  return diff || bad;
}
