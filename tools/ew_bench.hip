// tools/ew_bench.hip -- the element-wise kernels of csrc/kernels.hpp that are NOT stream-bound: inverse / divide by Montgomery's
// simultaneous inversion (k_ew_inv, chain length R * VEC per lane) against the one-Fermat-chain-per-element k_ew<F, 4 | 5>, and
// GF(2^128) multiply / inverse / divide on the LDS window table (k_ew_gf128_mul, k_ew_inv_rolled<GfLdsArith>) against Gf128::mul's register-only form -- word for
// word on the same inputs (zeros planted at lane, workgroup-tile and batch boundaries), then timed.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/_build/ew_bench tools/ew_bench.hip
// usage: ew_bench [N=10000000] [field: m61 m61r m127 m127r mont128 secp gf | all]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../secure-computation-library_amd/csrc/kernels.hpp"
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
// raw words -> canonical elements (any canonical value will do; zeros at chosen places)
template <class F>
__global__ void k_canon(typename F::Ctx ctx, u64* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    typename F::E v = F::ld(p + i * F::LIMBS);
    if constexpr (F::LIMBS == 1) v = M61::from_le_word(ctx, v);
    else if constexpr (F::TAG == 1) v = M127::from_le_word(ctx, v);
    else if constexpr (F::TAG == 2) { if (v >= ctx.p) v -= ctx.p; }
    else if constexpr (F::LIMBS == 4) { v.w[3] &= 0x7FFFFFFFFFFFFFFFull; }
    const bool plant = i < 3 || i % 1009 == 0 || (i & 255) == 255 || (i & 4095) == 0 || i + 2 >= n;
    if (plant && (i % 3 != 1)) v = F::zero();
    F::st(p + i * F::LIMBS, v);
  }
}

static hipEvent_t e0, e1;
template <class Fn>
static double time_ms(Fn&& launch, int reps = 5) {
  launch();
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

static size_t diff_words(const u64* x, const u64* y, size_t words) {
  std::vector<u64> a(words), b(words);
  CK(hipMemcpy(a.data(), x, words * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), y, words * 8, hipMemcpyDeviceToHost));
  size_t d = 0;
  for (size_t i = 0; i < words; ++i) d += a[i] != b[i];
  return d;
}

template <class F, bool DIV, int VEC, int R>
static void run_batch(const char* name, typename F::Ctx ctx, u64* out, const u64* ref, const u64* a, const u64* b, size_t n, unsigned* flag,
                      double base_ms) {
  const size_t npacks = n / VEC;
  const size_t tiles = (npacks + (size_t)BLOCK * R - 1) / ((size_t)BLOCK * R);
  CK(hipMemset(out, 0xAB, n * F::LIMBS * 8));
  CK(hipMemset(flag, 0, 4));
  auto launch = [&] { hipLaunchKernelGGL((k_ew_inv<F, DIV, VEC, R>), dim3((unsigned)tiles), dim3(BLOCK), 0, 0, ctx, out, a, b, npacks, flag); };
  const double ms = time_ms(launch);
  unsigned h = 0;
  CK(hipMemcpy(&h, flag, 4, hipMemcpyDeviceToHost));
  const size_t d = diff_words(out, ref, n * F::LIMBS);
  const double E = F::LIMBS * 8.0;
  std::printf("%-8s %s batch L=%2d  %8.3f ms  %7.2f G/s  %6.0f GB/s  x%.1f vs per-element  flag=%u  diff=%zu\n", name, DIV ? "div" : "inv", R * VEC, ms,
              n / ms / 1e6, (DIV ? 3 : 2) * E * n / ms / 1e6, base_ms / ms, h, d);
  std::fflush(stdout);
}

template <class F, bool DIV, int VEC, int R, int W = 2, int G = 4, bool KEEP = false>
static void run_reread(const char* name, typename F::Ctx ctx, u64* out, const u64* ref, const u64* a, const u64* b, size_t n, unsigned* flag,
                       double base_ms) {
  const size_t npacks = n / VEC;
  const size_t tiles = (npacks + (size_t)BLOCK * R - 1) / ((size_t)BLOCK * R);
  CK(hipMemset(out, 0xAB, n * F::LIMBS * 8));
  CK(hipMemset(flag, 0, 4));
  auto launch = [&] { hipLaunchKernelGGL((k_ew_inv<F, DIV, VEC, R, G, W, KEEP>), dim3((unsigned)tiles), dim3(BLOCK), 0, 0, ctx, out, a, b, npacks, flag); };
  const double ms = time_ms(launch);
  unsigned h = 0;
  CK(hipMemcpy(&h, flag, 4, hipMemcpyDeviceToHost));
  const size_t d = diff_words(out, ref, n * F::LIMBS);
  const double E = F::LIMBS * 8.0;
  std::printf("%-8s %s %s L=%2d W=%d G=%d  %8.3f ms  %7.2f G/s  %6.0f GB/s  x%.1f vs per-element  flag=%u  diff=%zu\n", name, DIV ? "div" : "inv", KEEP ? "grouped" : "reread", R * VEC, W, G, ms,
              n / ms / 1e6, (DIV ? 3 : 2) * E * n / ms / 1e6, base_ms / ms, h, d);
  std::fflush(stdout);
}

template <class F, int VEC>
static void run_field(const char* name, typename F::Ctx ctx, size_t n) {
  const size_t words = n * F::LIMBS;
  u64 *a, *b, *ref, *out;
  unsigned* flag;
  CK(hipMalloc(&a, words * 8));
  CK(hipMalloc(&b, words * 8));
  CK(hipMalloc(&ref, words * 8));
  CK(hipMalloc(&out, words * 8));
  CK(hipMalloc(&flag, 64));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, a, words, 11ull);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, b, words, 22ull);
  hipLaunchKernelGGL((k_canon<F>), dim3(4096), dim3(256), 0, 0, ctx, a, n);
  hipLaunchKernelGGL((k_canon<F>), dim3(4096), dim3(256), 0, 0, ctx, b, n);
  CK(hipDeviceSynchronize());
  const size_t npacks = n / VEC;
  const unsigned g = (unsigned)((npacks + BLOCK - 1) / BLOCK);
  const double E = F::LIMBS * 8.0;
  for (int div = 0; div < 2; ++div) {
    CK(hipMemset(flag, 0, 4));
    double base;
    if (div) base = time_ms([&] { hipLaunchKernelGGL((k_ew<F, 5, VEC, true>), dim3(g), dim3(BLOCK), 0, 0, ctx, ref, a, b, npacks, flag); }, 2);
    else base = time_ms([&] { hipLaunchKernelGGL((k_ew<F, 4, VEC, true>), dim3(g), dim3(BLOCK), 0, 0, ctx, ref, a, b, npacks, flag); }, 2);
    std::printf("%-8s %s per-element  %8.3f ms  %7.2f G/s  %6.0f GB/s\n", name, div ? "div" : "inv", base, n / base / 1e6, (div ? 3 : 2) * E * n / base / 1e6);
    if (div) {
      run_batch<F, true, VEC, 4>(name, ctx, out, ref, a, b, n, flag, base);
      run_batch<F, true, VEC, 8>(name, ctx, out, ref, a, b, n, flag, base);
      if constexpr (F::LIMBS <= 2) run_batch<F, true, VEC, 16>(name, ctx, out, ref, a, b, n, flag, base);
      if constexpr (std::is_same<F, M61>::value || std::is_same<F, M127>::value) {
        run_reread<F, true, VEC, 16, 3>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, true, VEC, 32, 2>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, true, VEC, 16, 3, 4, true>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, true, VEC, 8, 4, 4, true>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, true, VEC, 8, 3, 4, true>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, true, VEC, 12, 3, 4, true>(name, ctx, out, ref, a, b, n, flag, base);
      }
    } else {
      run_batch<F, false, VEC, 2>(name, ctx, out, ref, a, b, n, flag, base);
      run_batch<F, false, VEC, 4>(name, ctx, out, ref, a, b, n, flag, base);
      run_batch<F, false, VEC, 8>(name, ctx, out, ref, a, b, n, flag, base);
      if constexpr (F::LIMBS <= 2) run_batch<F, false, VEC, 16>(name, ctx, out, ref, a, b, n, flag, base);
      if constexpr (F::LIMBS == 2 || VEC == 1) run_batch<F, false, VEC, 32>(name, ctx, out, ref, a, b, n, flag, base);
      if constexpr (std::is_same<F, M61>::value) {
        run_reread<F, false, VEC, 16, 4>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 16, 3>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 16, 4, 8>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 24, 3>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 32, 3>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 32, 3, 8>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 32, 2>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 16, 3, 4, true>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 16, 2, 4, true>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 8, 4, 4, true>(name, ctx, out, ref, a, b, n, flag, base);
      }
      if constexpr (std::is_same<F, M127>::value) {
        run_reread<F, false, VEC, 16, 3, 4, true>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 16, 2, 4, true>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 16, 4>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 16, 3>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 24, 3>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 32, 2>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 32, 3>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 32, 3, 2>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 28, 3, 4>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 28, 3, 2>(name, ctx, out, ref, a, b, n, flag, base);
        run_reread<F, false, VEC, 40, 2, 4>(name, ctx, out, ref, a, b, n, flag, base);
      }
    }
  }
  CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(ref)); CK(hipFree(out)); CK(hipFree(flag));
}

// the rolled chain (prefix products in scratch memory): any field, GF(2^128) with its products on the LDS table
template <class F, class ARITH, bool DIV, int L, int BLK>
static void run_rolled(const char* name, typename F::Ctx ctx, u64* out, const u64* ref, const u64* a, const u64* b, size_t n, unsigned* flag,
                       double base_ms) {
  auto kern = &k_ew_inv_rolled<F, ARITH, DIV, L, BLK>;
  const int lds = BLK * (int)ARITH::LDS_PER_LANE;
  if (lds) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const size_t tiles = (n + (size_t)BLK * L - 1) / ((size_t)BLK * L);
  CK(hipMemset(out, 0xAB, n * F::LIMBS * 8));
  CK(hipMemset(flag, 0, 4));
  const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(BLK), lds, 0, ctx, out, a, b, n, flag); });
  unsigned h = 0;
  CK(hipMemcpy(&h, flag, 4, hipMemcpyDeviceToHost));
  const size_t d = diff_words(out, ref, n * F::LIMBS);
  const double E = F::LIMBS * 8.0;
  std::printf("%-8s %s rolled L=%3d BLK=%3d  %8.3f ms  %7.2f G/s  %6.0f GB/s  x%.1f vs per-element  flag=%u  diff=%zu\n", name, DIV ? "div" : "inv", L, BLK, ms,
              n / ms / 1e6, (DIV ? 3 : 2) * E * n / ms / 1e6, base_ms / ms, h, d);
  std::fflush(stdout);
}

template <class F>
static void run_field_rolled(const char* name, typename F::Ctx ctx, size_t n) {
  const size_t words = n * F::LIMBS;
  u64 *a, *b, *ref, *out;
  unsigned* flag;
  CK(hipMalloc(&a, words * 8));
  CK(hipMalloc(&b, words * 8));
  CK(hipMalloc(&ref, words * 8));
  CK(hipMalloc(&out, words * 8));
  CK(hipMalloc(&flag, 64));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, a, words, 11ull);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, b, words, 22ull);
  hipLaunchKernelGGL((k_canon<F>), dim3(4096), dim3(256), 0, 0, ctx, a, n);
  hipLaunchKernelGGL((k_canon<F>), dim3(4096), dim3(256), 0, 0, ctx, b, n);
  CK(hipDeviceSynchronize());
  const unsigned g = (unsigned)((n + BLOCK - 1) / BLOCK);
  typedef FieldArith<F> A;
  double base = time_ms([&] { hipLaunchKernelGGL((k_ew<F, 4, 1, true>), dim3(g), dim3(BLOCK), 0, 0, ctx, ref, a, b, n, flag); }, 1);
  run_rolled<F, A, false, 16, 256>(name, ctx, out, ref, a, b, n, flag, base);
  run_rolled<F, A, false, 32, 256>(name, ctx, out, ref, a, b, n, flag, base);
  run_rolled<F, A, false, 64, 256>(name, ctx, out, ref, a, b, n, flag, base);
  run_rolled<F, A, false, 128, 256>(name, ctx, out, ref, a, b, n, flag, base);
  run_rolled<F, A, false, 64, 64>(name, ctx, out, ref, a, b, n, flag, base);
  base = time_ms([&] { hipLaunchKernelGGL((k_ew<F, 5, 1, true>), dim3(g), dim3(BLOCK), 0, 0, ctx, ref, a, b, n, flag); }, 1);
  run_rolled<F, A, true, 32, 256>(name, ctx, out, ref, a, b, n, flag, base);
  run_rolled<F, A, true, 64, 256>(name, ctx, out, ref, a, b, n, flag, base);
  CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(ref)); CK(hipFree(out)); CK(hipFree(flag));
}

// the ladder Gf128::inv was until round 5 (127 squarings by general product + 127 products), for the record
__global__ void k_gf_inv_ladder(u64* dst, const u64* a, size_t n) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const Gf128::Ctx c{};
  const u128 x = Gf128::ld(a + 2 * q);
  u128 r = 1, sq = x;
  for (int i = 1; i < 128; ++i) {
    sq = Gf128::mul(c, sq, sq);
    r = Gf128::mul(c, r, sq);
  }
  Gf128::st(dst + 2 * q, x == 0 ? (u128)0 : r);
}

template <int BLK, int W = 4>
static void run_gf_mul(u64* out, const u64* ref, const u64* a, const u64* b, size_t n, double base_ms) {
  auto kern = &k_ew_gf128_mul<BLK, W>;
  const int lds = BLK * (16 << W);
  std::printf("(window bits %d)\n", W);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const size_t blocks = (n + BLK - 1) / BLK;
  CK(hipMemset(out, 0xAB, n * 16));
  for (unsigned cap : {0u, 2560u, 10240u}) {  // one pack per lane, or a resident grid that strides
    const unsigned grid = cap && blocks > cap ? cap : (unsigned)blocks;
    const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(BLK), lds, 0, out, a, b, n); });
    const size_t d = diff_words(out, ref, n * 2);
    std::printf("gf2_128  mul LDS comb BLK=%3d grid=%8u  %8.3f ms  %7.2f G/s  %6.0f GB/s  x%.1f  diff=%zu\n", BLK, grid, ms, n / ms / 1e6,
                48.0 * n / ms / 1e6, base_ms / ms, d);
    std::fflush(stdout);
  }
}

static void run_gf_all(size_t n) {
  u64 *a, *b, *ref, *out;
  unsigned* flag;
  CK(hipMalloc(&a, n * 16));
  CK(hipMalloc(&b, n * 16));
  CK(hipMalloc(&ref, n * 16));
  CK(hipMalloc(&out, n * 16));
  CK(hipMalloc(&flag, 64));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, a, n * 2, 11ull);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, b, n * 2, 22ull);
  hipLaunchKernelGGL((k_canon<Gf128>), dim3(4096), dim3(256), 0, 0, Gf128::Ctx{}, a, n);
  hipLaunchKernelGGL((k_canon<Gf128>), dim3(4096), dim3(256), 0, 0, Gf128::Ctx{}, b, n);
  CK(hipDeviceSynchronize());
  const unsigned g = (unsigned)((n + BLOCK - 1) / BLOCK);
  const Gf128::Ctx ctx{};
  double base = time_ms([&] { hipLaunchKernelGGL((k_ew<Gf128, 2, 1, true>), dim3(g), dim3(BLOCK), 0, 0, ctx, ref, a, b, n, flag); }, 2);
  std::printf("gf2_128  mul registers  %8.3f ms  %7.2f G/s  %6.0f GB/s\n", base, n / base / 1e6, 48.0 * n / base / 1e6);
  run_gf_mul<64>(out, ref, a, b, n, base);
  run_gf_mul<128>(out, ref, a, b, n, base);
  run_gf_mul<256>(out, ref, a, b, n, base);
  run_gf_mul<64, 3>(out, ref, a, b, n, base);
  run_gf_mul<256, 3>(out, ref, a, b, n, base);
  run_gf_mul<64, 2>(out, ref, a, b, n, base);
  run_gf_mul<256, 2>(out, ref, a, b, n, base);
  const size_t nl = n / 16;  // the old ladder is slow: a sixteenth of the batch
  double ladder = time_ms([&] { hipLaunchKernelGGL(k_gf_inv_ladder, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, 0, out, a, nl); }, 1);
  std::printf("gf2_128  inv ladder of round 4 (254 register products), %zu elements  %8.3f ms  %7.4f G/s\n", nl, ladder, nl / ladder / 1e6);
  ladder *= 16.0;
  base = time_ms([&] { hipLaunchKernelGGL((k_ew<Gf128, 4, 1, true>), dim3(g), dim3(BLOCK), 0, 0, ctx, ref, a, b, n, flag); }, 1);
  std::printf("gf2_128  inv Itoh-Tsujii per element, register products  %8.3f ms  %7.3f G/s  x%.1f vs the ladder\n", base, n / base / 1e6, ladder / base);
  std::printf("(below: x vs the ladder)\n");
  run_rolled<Gf128, GfLdsArith<64>, false, 1, 64>("gf2_128", ctx, out, ref, a, b, n, flag, ladder);
  run_rolled<Gf128, GfLdsArith<64>, false, 8, 64>("gf2_128", ctx, out, ref, a, b, n, flag, ladder);
  run_rolled<Gf128, GfLdsArith<64>, false, 16, 64>("gf2_128", ctx, out, ref, a, b, n, flag, ladder);
  run_rolled<Gf128, GfLdsArith<64>, false, 32, 64>("gf2_128", ctx, out, ref, a, b, n, flag, ladder);
  run_rolled<Gf128, GfLdsArith<64>, false, 64, 64>("gf2_128", ctx, out, ref, a, b, n, flag, ladder);
  run_rolled<Gf128, GfLdsArith<256>, false, 32, 256>("gf2_128", ctx, out, ref, a, b, n, flag, ladder);
  std::printf("(window bits 3)\n");
  run_rolled<Gf128, GfLdsArith<64, 3>, false, 32, 64>("gf2_128", ctx, out, ref, a, b, n, flag, ladder);
  run_rolled<Gf128, GfLdsArith<256, 3>, false, 32, 256>("gf2_128", ctx, out, ref, a, b, n, flag, ladder);
  base = time_ms([&] { hipLaunchKernelGGL((k_ew<Gf128, 5, 1, true>), dim3(g), dim3(BLOCK), 0, 0, ctx, ref, a, b, n, flag); }, 1);
  std::printf("gf2_128  div per element (Itoh-Tsujii, register products)  %8.3f ms  %7.3f G/s\n", base, n / base / 1e6);
  run_rolled<Gf128, GfLdsArith<64>, true, 32, 64>("gf2_128", ctx, out, ref, a, b, n, flag, base);
  run_rolled<Gf128, GfLdsArith<64>, true, 64, 64>("gf2_128", ctx, out, ref, a, b, n, flag, base);
}

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 10000000;
  const char* which = argc > 2 ? argv[2] : "all";
  auto want = [&](const char* f) { return !std::strcmp(which, "all") || !std::strcmp(which, f); };
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  if (want("m61")) run_field<M61, 2>("m61", M61::Ctx{}, n);
  if (want("m127")) run_field<M127, 1>("m127", M127::Ctx{}, n);
  const Mont128::Ctx mc = Mont128::make_ctx((((u128)0xFFFFFFFFFFFFFFFFull) << 64) | (u128)0xFFFFFFFFFFFFFF61ull);
  if (want("m61r")) run_field_rolled<M61>("m61", M61::Ctx{}, n);
  if (want("m127r")) run_field_rolled<M127>("m127", M127::Ctx{}, n);
  if (want("mont128")) run_field<Mont128, 1>("mont128", mc, n);
  if (want("mont128")) run_field_rolled<Mont128>("mont128", mc, n);
  if (want("secp")) run_field<Secp256k1Scalar, 1>("secp", Secp256k1Scalar::Ctx{}, n / 4);
  if (want("secp")) run_field_rolled<Secp256k1Scalar>("secp", Secp256k1Scalar::Ctx{}, n / 4);
  if (want("gf")) run_gf_all(n);
  return 0;
}
