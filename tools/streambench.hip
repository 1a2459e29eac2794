// tools/streambench.hip -- the headline kernels (Shamir (10,3) over Mersenne61: k_share_small, k_recover_fixed) on PLAIN
// hipMalloc allocations against variants that change only how the DRAM streams are walked: workgroup size, the order in
// which a workgroup touches the party rows, which workgroups run side by side.  Every variant is checked word for word
// against the library kernel's output.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/_build/streambench tools/streambench.hip
// run:   tools/_build/streambench [N secrets, default 1e8] [rounds, default 3]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../secure-computation-library_amd/csrc/kernels.hpp"
using namespace sclhip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1);} } while (0)

constexpr int NP = 10, TT = 3;

// which pack a thread owns.  MAP 0: block b owns packs [b*BLK, (b+1)*BLK).  MAP 1: XCD-contiguous -- blocks are dealt
// round-robin over the 8 XCDs, so block b is the (b/8)-th block of XCD b%8; give each XCD one contiguous eighth.
template <int MAP, int BLK>
__device__ __forceinline__ size_t pack_of(size_t npacks) {
  if constexpr (MAP == 0) {
    return (size_t)blockIdx.x * BLK + threadIdx.x;
  } else {
    const size_t per = (gridDim.x + 7) / 8;  // blocks per XCD
    const size_t lb = (size_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    return lb * BLK + threadIdx.x;
  }
}

struct Lam {
  u64 v[16];
};

// ROT: rows are visited starting at row (blockIdx % NP)
template <int BLK, int MAP, bool ROT>
__global__ __launch_bounds__(BLK) void k_rec(u64* out, const u64* shares, size_t stride, Lam lam, size_t npacks) {
  extern __shared__ u32 occupancy_pad1[];
  const size_t q = pack_of<MAP, BLK>(npacks);
  if (q >= npacks) return;
  const size_t off = q * 2;
  u64x2 x[NP];
  const int r0 = ROT ? (int)(blockIdx.x % NP) : 0;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    int row = i + r0;
    if (row >= NP) row -= NP;
    x[i] = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(shares + (size_t)row * stride + off));
  }
  u128 a0 = 0, a1 = 0;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    int row = i + r0;
    if (row >= NP) row -= NP;
    const u64 l = lam.v[row];
    a0 += (u128)l * x[i].x;
    a1 += (u128)l * x[i].y;
  }
  u64x2 r;
  r.x = M61::fold128(a0);
  r.y = M61::fold128(a1);
  __builtin_nontemporal_store(r, reinterpret_cast<u64x2*>(out + off));
}

template <int BLK, int MAP, bool ROT>
__global__ __launch_bounds__(BLK) void k_shr(u64* shares, size_t stride, const u64* secrets, const u64* coeffs, size_t cstride,
                                             SmallVdm tab, size_t npacks) {
  __shared__ u32 V[NP * (TT + 1)];
  for (int i = threadIdx.x; i < NP * (TT + 1); i += BLK) V[i] = tab.v[i];
  __syncthreads();
  const size_t q = pack_of<MAP, BLK>(npacks);
  if (q >= npacks) return;
  const size_t off = q * 2;
  Pack<M61, 2> c[TT + 1];
  c[0] = load_pack<M61, 2, true>(secrets + off);
#pragma unroll
  for (int k = 1; k <= TT; ++k) c[k] = load_pack<M61, 2, true>(coeffs + (size_t)(k - 1) * cstride + off);
  const int r0 = ROT ? (int)(blockIdx.x % NP) : 0;
  for (int ii = 0; ii < NP; ++ii) {
    int i = ii + r0;
    if (i >= NP) i -= NP;
    const u32* row = V + i * (TT + 1);
    SmallAcc<M61> acc[2];
    acc[0].init();
    acc[1].init();
#pragma unroll
    for (int k = 1; k <= TT; ++k) {
      const u32 w = row[k];
      acc[0].mac(c[k].v[0], w);
      acc[1].mac(c[k].v[1], w);
    }
    Pack<M61, 2> y;
    y.v[0] = acc[0].fold(M61::Ctx{}, c[0].v[0]);
    y.v[1] = acc[1].fold(M61::Ctx{}, c[0].v[1]);
    store_pack<M61, 2, true>(shares + (size_t)i * stride + off, y);
  }
}

// two packs per thread, BLK apart: a workgroup covers 2 * BLK * 16 contiguous bytes of every row
template <int BLK>
__global__ __launch_bounds__(BLK) void k_rec2(u64* out, const u64* shares, size_t stride, Lam lam, size_t npacks) {
  extern __shared__ u32 occupancy_pad[];
  const size_t q0 = (size_t)blockIdx.x * (2 * BLK) + threadIdx.x;
  if (q0 >= npacks) return;
  const bool two = q0 + BLK < npacks;
  u64x2 x[NP], y[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) x[i] = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(shares + (size_t)i * stride + q0 * 2));
  if (two) {
#pragma unroll
    for (int i = 0; i < NP; ++i) y[i] = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(shares + (size_t)i * stride + (q0 + BLK) * 2));
  }
  u128 a0 = 0, a1 = 0;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    a0 += (u128)lam.v[i] * x[i].x;
    a1 += (u128)lam.v[i] * x[i].y;
  }
  u64x2 r;
  r.x = M61::fold128(a0);
  r.y = M61::fold128(a1);
  __builtin_nontemporal_store(r, reinterpret_cast<u64x2*>(out + q0 * 2));
  if (two) {
    a0 = a1 = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      a0 += (u128)lam.v[i] * y[i].x;
      a1 += (u128)lam.v[i] * y[i].y;
    }
    r.x = M61::fold128(a0);
    r.y = M61::fold128(a1);
    __builtin_nontemporal_store(r, reinterpret_cast<u64x2*>(out + (q0 + BLK) * 2));
  }
}

// share with a choice of store flavour (NTS) and PK packs per thread, BLK apart
template <int BLK, bool NTS, int PK>
__global__ __launch_bounds__(BLK) void k_shr2(u64* shares, size_t stride, const u64* secrets, const u64* coeffs, size_t cstride,
                                              SmallVdm tab, size_t npacks) {
  extern __shared__ u32 dynpad[];
  __shared__ u32 V[NP * (TT + 1)];
  for (int i = threadIdx.x; i < NP * (TT + 1); i += BLK) V[i] = tab.v[i];
  __syncthreads();
  const size_t q0 = (size_t)blockIdx.x * (PK * BLK) + threadIdx.x;
  Pack<M61, 2> c[PK][TT + 1];
#pragma unroll
  for (int p = 0; p < PK; ++p) {
    const size_t q = q0 + (size_t)p * BLK;
    if (q < npacks) {
      c[p][0] = load_pack<M61, 2, true>(secrets + q * 2);
#pragma unroll
      for (int k = 1; k <= TT; ++k) c[p][k] = load_pack<M61, 2, true>(coeffs + (size_t)(k - 1) * cstride + q * 2);
    }
  }
  for (int i = 0; i < NP; ++i) {
    const u32* row = V + i * (TT + 1);
#pragma unroll
    for (int p = 0; p < PK; ++p) {
      const size_t q = q0 + (size_t)p * BLK;
      if (q < npacks) {
        SmallAcc<M61> acc[2];
        acc[0].init();
        acc[1].init();
#pragma unroll
        for (int k = 1; k <= TT; ++k) {
          const u32 w = row[k];
          acc[0].mac(c[p][k].v[0], w);
          acc[1].mac(c[p][k].v[1], w);
        }
        Pack<M61, 2> y;
        y.v[0] = acc[0].fold(M61::Ctx{}, c[p][0].v[0]);
        y.v[1] = acc[1].fold(M61::Ctx{}, c[p][0].v[1]);
        store_pack<M61, 2, NTS>(shares + (size_t)i * stride + q * 2, y);
      }
    }
  }
}

int main(int argc, char** argv) {
  const size_t N = argc > 1 ? (size_t)std::atof(argv[1]) : 100000000;
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 2;
  const size_t npacks = N / 2;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // two sets of operands in separate plain allocations: every variant is timed on both (the same kernel runs up to
  // 12 % apart from one allocation to the next, so a variant only counts if it wins on both)
  u64 *secrets[2], *coeffs[2], *shares[2], *out[2], *ref_sh, *ref_out;
  for (int a = 0; a < 2; ++a) {
    CK(hipMalloc(&secrets[a], N * 8));
    CK(hipMalloc(&coeffs[a], (size_t)TT * N * 8));
    CK(hipMalloc(&shares[a], (size_t)NP * N * 8));
    CK(hipMalloc(&out[a], N * 8));
  }
  const size_t pitched_elems = (size_t)NP * (N + ((size_t)48 << 20) / 8);
  u64* pitched[2];
  for (int a = 0; a < 2; ++a) CK(hipMalloc(&pitched[a], pitched_elems * 8));
  CK(hipMalloc(&ref_sh, (size_t)NP * N * 8));
  CK(hipMalloc(&ref_out, N * 8));
  {
    AesKey key;
    for (int i = 0; i < 44; ++i) key.rk[i] = 0x9E3779B9u * (i + 1);
    for (int i = 0; i < 256; ++i) key.te0[i] = 0x85EBCA6Bu * (i + 7) ^ (i << 13);
    aes_key_round1(key);
    auto kern = &k_vector_random<M61>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, AES4_LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(AES4_GRID_CAP), dim3(ABLOCK), AES4_LDS_BYTES, 0, M61::Ctx{}, secrets[0], key, 1ull, N);
    hipLaunchKernelGGL(kern, dim3(AES4_GRID_CAP), dim3(ABLOCK), AES4_LDS_BYTES, 0, M61::Ctx{}, coeffs[0], key, 1ull << 40, (size_t)TT * N);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(secrets[1], secrets[0], N * 8, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(coeffs[1], coeffs[0], (size_t)TT * N * 8, hipMemcpyDeviceToDevice));
  }
  SmallVdm sv;
  for (int i = 0; i < NP; ++i) {
    u32 pw = 1;
    for (int k = 0; k <= TT; ++k) {
      sv.v[i * (TT + 1) + k] = pw;
      pw *= (u32)(i + 1);
    }
  }
  Lam lam;
  Table<M61> lamt;
  {
    const M61::Ctx ctx{};
    for (int i = 0; i < NP; ++i) {
      u64 num = 1, den = 1;
      for (int j = 0; j < NP; ++j) {
        if (j == i) continue;
        num = M61::mul(ctx, num, M61::sub(ctx, 0, (u64)(j + 1)));
        den = M61::mul(ctx, den, M61::sub(ctx, (u64)(i + 1), (u64)(j + 1)));
      }
      lam.v[i] = lamt.v[i] = M61::mul(ctx, num, M61::inv(ctx, den));
    }
  }
  auto time_it = [&](auto launch, int reps) {
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
  };
  auto diff = [&](const u64* a, const u64* b, size_t words) {
    static unsigned long long* cnt = nullptr;
    if (!cnt) CK(hipMalloc(&cnt, 8));
    CK(hipMemset(cnt, 0, 8));
    hipLaunchKernelGGL(k_count_diff, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, 0, cnt, a, b, words);
    unsigned long long h = 0;
    CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
    return (size_t)h;
  };
  const double sb = 112.0 * N, rb = 88.0 * N;
  {
    const unsigned g = (unsigned)((npacks + 255) / 256);
    hipLaunchKernelGGL((k_share_small<M61, 2>), dim3(g), dim3(256), 0, 0, M61::Ctx{}, ref_sh, N, secrets[0], coeffs[0], N, sv, TT, NP, npacks);
    hipLaunchKernelGGL((k_recover_fixed<M61, 2, NP, true>), dim3(g), dim3(256), 0, 0, M61::Ctx{}, ref_out, ref_sh, N, lamt, npacks);
    CK(hipDeviceSynchronize());
    std::printf("reference round trip: diff %zu\n", diff(ref_out, secrets[0], N));
  }
  for (int round = 0; round < rounds; ++round) {
    std::printf("== round %d: ms on operand set A / B (plain hipMalloc allocations, N = %zu)\n", round, N);
    auto report = [&](const char* kind, const char* name, float a, float b, double bytes, size_t d) {
      std::printf("%-8s %-44s %7.3f / %7.3f ms  %5.0f / %5.0f GB/s  diff %zu\n", kind, name, a, b, bytes / a / 1e6, bytes / b / 1e6, d);
    };
    {
      const unsigned g = (unsigned)((npacks + 255) / 256);
      float ms[2];
      for (int a = 0; a < 2; ++a)
        ms[a] = time_it([&] { hipLaunchKernelGGL((k_share_small<M61, 2>), dim3(g), dim3(256), 0, 0, M61::Ctx{}, shares[a], N, secrets[a], coeffs[a], N, sv, TT, NP, npacks); }, 10);
      report("share", "library k_share_small b256", ms[0], ms[1], sb, diff(ref_sh, shares[0], (size_t)NP * N) + diff(ref_sh, shares[1], (size_t)NP * N));
    }
#define SHR2(BLK, NTS, PK, LDSB)                                                                                     \
  {                                                                                                                 \
    const unsigned g = (unsigned)((npacks + PK * BLK - 1) / (PK * BLK));                                            \
    auto kern = &k_shr2<BLK, NTS, PK>;                                                                              \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)); \
    float ms[2];                                                                                                    \
    for (int a = 0; a < 2; ++a)                                                                                     \
      ms[a] = time_it([&] { hipLaunchKernelGGL(kern, dim3(g), dim3(BLK), LDSB, 0, shares[a], N, secrets[a], coeffs[a], N, sv, npacks); }, 10); \
    char nm[96];                                                                                                    \
    std::snprintf(nm, sizeof nm, "b%d nt-store %d packs %d lds %d B", BLK, (int)NTS, PK, LDSB);                      \
    report("share", nm, ms[0], ms[1], sb, round == 0 ? diff(ref_sh, shares[0], (size_t)NP * N) + diff(ref_sh, shares[1], (size_t)NP * N) : 0); \
  }
    SHR2(256, true, 1, 0) SHR2(64, true, 1, 19 * 1024)
#undef SHR2
    {
      const unsigned g = (unsigned)((npacks + 255) / 256);
      float ms[2];
      for (int a = 0; a < 2; ++a)
        ms[a] = time_it([&] { hipLaunchKernelGGL((k_recover_fixed<M61, 2, NP, true>), dim3(g), dim3(256), 0, 0, M61::Ctx{}, out[a], shares[a], N, lamt, npacks); }, 10);
      report("recover", "library k_recover_fixed b256", ms[0], ms[1], rb, diff(ref_out, out[0], N) + diff(ref_out, out[1], N));
    }
#define REC1(BLK, LDSB)                                                                                             \
  {                                                                                                                 \
    const unsigned g = (unsigned)((npacks + BLK - 1) / BLK);                                                        \
    auto kern = &k_rec<BLK, 0, false>;                                                                              \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)); \
    float ms[2];                                                                                                    \
    for (int a = 0; a < 2; ++a)                                                                                     \
      ms[a] = time_it([&] { hipLaunchKernelGGL(kern, dim3(g), dim3(BLK), LDSB, 0, out[a], shares[a], N, lam, npacks); }, 10); \
    char nm[96];                                                                                                    \
    std::snprintf(nm, sizeof nm, "b%d lds %d B", BLK, LDSB);                                                        \
    report("recover", nm, ms[0], ms[1], rb, round == 0 ? diff(ref_out, out[0], N) + diff(ref_out, out[1], N) : 0);   \
  }
    REC1(256, 0) REC1(64, 19 * 1024)
#undef REC1
#define REC2(BLK, LDSB)                                                                                             \
  {                                                                                                                 \
    const unsigned g = (unsigned)((npacks + 2 * BLK - 1) / (2 * BLK));                                              \
    auto kern = &k_rec2<BLK>;                                                                                       \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)); \
    float ms[2];                                                                                                    \
    for (int a = 0; a < 2; ++a)                                                                                     \
      ms[a] = time_it([&] { hipLaunchKernelGGL(kern, dim3(g), dim3(BLK), LDSB, 0, out[a], shares[a], N, lam, npacks); }, 10); \
    char nm[96];                                                                                                    \
    std::snprintf(nm, sizeof nm, "2 packs b%d lds %d B", BLK, LDSB);                                                \
    report("recover", nm, ms[0], ms[1], rb, round == 0 ? diff(ref_out, out[0], N) + diff(ref_out, out[1], N) : 0);   \
  }
    REC2(64, 19 * 1024)
#undef REC2
    // row pitch sweep: the library kernels with the share matrix at a pitch of N + pad elements (needs pitched buffers)
    for (size_t pad_bytes : {(size_t)0, (size_t)256, (size_t)1024, (size_t)4096 + 256, (size_t)65536 + 4096 + 256, ((size_t)1 << 20) + 65536 + 4096 + 256,
                             ((size_t)2 << 20), ((size_t)2 << 20) + 4096 + 256, ((size_t)16 << 20) + 256, ((size_t)47 << 20) + 4096}) {
      const size_t pitch = N + pad_bytes / 8;
      if (pitch * NP > pitched_elems) continue;
      const unsigned g = (unsigned)((npacks + 255) / 256), g64 = (unsigned)((npacks + 63) / 64);
      float ms_s[2], ms_r[2];
      for (int a = 0; a < 2; ++a) {
        ms_s[a] = time_it([&] { hipLaunchKernelGGL((k_share_small<M61, 2>), dim3(g), dim3(256), 0, 0, M61::Ctx{}, pitched[a], pitch, secrets[a], coeffs[a], N, sv, TT, NP, npacks); }, 10);
        ms_r[a] = time_it([&] { hipLaunchKernelGGL((k_recover_fixed<M61, 2, NP, true, 64>), dim3(g64), dim3(64), 19456, 0, M61::Ctx{}, out[a], pitched[a], pitch, lamt, npacks); }, 10);
      }
      std::printf("pitch N + %9zu B: share %7.3f / %7.3f ms  recover(b64, 8 waves/CU) %7.3f / %7.3f ms  diff %zu\n", pad_bytes, ms_s[0], ms_s[1],
                  ms_r[0], ms_r[1], diff(ref_out, out[0], N) + diff(ref_out, out[1], N));
    }
  }
  return 0;
}
