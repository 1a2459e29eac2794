// tools/membench.hip -- HBM ceilings on this box for the access shapes the engine uses, and A/B variants
// of the reconstruct kernel.  Build: hipcc -O3 --offload-arch=gfx950 -o tools/_build/membench tools/membench.hip
// Run on the GPU box: tools/_build/membench [N secrets, default 1e8]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/scl_hip/detail/field.hpp"

using namespace sclhip;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      std::printf("%s: %s\n", #x, hipGetErrorString(e));                       \
      std::exit(1);                                                            \
    }                                                                          \
  } while (0)

template <bool NT>
__device__ __forceinline__ u64x2 ld(const u64x2* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <bool NT>
__device__ __forceinline__ void st(u64x2* p, u64x2 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// ---- ceilings
template <bool NT>
__global__ __launch_bounds__(256) void k_read(const u64x2* a, size_t n16, u64* sink) {
  u64 acc = 0;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n16; q += (size_t)gridDim.x * 256) {
    const u64x2 v = ld<NT>(a + q);
    acc ^= v.x ^ v.y;
  }
  if (acc == 0x1234567ull) sink[0] = acc;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_write(u64x2* a, size_t n16) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n16; q += (size_t)gridDim.x * 256) {
    u64x2 v;
    v.x = q;
    v.y = ~q;
    st<NT>(a + q, v);
  }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_copy(u64x2* d, const u64x2* a, size_t n16) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n16; q += (size_t)gridDim.x * 256) st<NT>(d + q, ld<NT>(a + q));
}
// 10 input streams, 1 output stream, no arithmetic beyond xor: the access shape of reconstruct
template <bool NT, int M>
__global__ __launch_bounds__(256) void k_shape(u64x2* out, const u64x2* sh, size_t stride16, size_t n16) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n16; q += (size_t)gridDim.x * 256) {
    u64x2 x[M];
#pragma unroll
    for (int i = 0; i < M; ++i) x[i] = ld<NT>(sh + i * stride16 + q);
    u64x2 r = x[0];
#pragma unroll
    for (int i = 1; i < M; ++i) r ^= x[i];
    st<NT>(out + q, r);
  }
}

struct Lam {
  u64 v[16];
};

// ---- reconstruct variants (M61, lambda in SGPRs)
template <bool NT, int M, int BLK>
__global__ __launch_bounds__(BLK) void k_rec(u64x2* out, const u64x2* sh, size_t stride16, Lam lam, size_t n16) {
  for (size_t q = (size_t)blockIdx.x * BLK + threadIdx.x; q < n16; q += (size_t)gridDim.x * BLK) {
    u64x2 x[M];
#pragma unroll
    for (int i = 0; i < M; ++i) x[i] = ld<NT>(sh + i * stride16 + q);
    u128 a0 = 0, a1 = 0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      a0 += (u128)lam.v[i] * x[i].x;
      a1 += (u128)lam.v[i] * x[i].y;
    }
    u64x2 r;
    r.x = M61::fold128(a0);
    r.y = M61::fold128(a1);
    st<NT>(out + q, r);
  }
}

// two packs per thread, BLK apart (each instruction still a contiguous 1 KiB per wave)
template <bool NT, int M>
__global__ __launch_bounds__(256) void k_rec2(u64x2* out, const u64x2* sh, size_t stride16, Lam lam, size_t n16) {
  const size_t q0 = (size_t)blockIdx.x * 512 + threadIdx.x;
  u64x2 x[M], y[M];
  const bool in0 = q0 < n16, in1 = q0 + 256 < n16;
  if (!in0) return;
#pragma unroll
  for (int i = 0; i < M; ++i) x[i] = ld<NT>(sh + i * stride16 + q0);
  if (in1) {
#pragma unroll
    for (int i = 0; i < M; ++i) y[i] = ld<NT>(sh + i * stride16 + q0 + 256);
  }
  u128 a0 = 0, a1 = 0, b0 = 0, b1 = 0;
#pragma unroll
  for (int i = 0; i < M; ++i) {
    a0 += (u128)lam.v[i] * x[i].x;
    a1 += (u128)lam.v[i] * x[i].y;
  }
  u64x2 r;
  r.x = M61::fold128(a0);
  r.y = M61::fold128(a1);
  st<NT>(out + q0, r);
  if (in1) {
#pragma unroll
    for (int i = 0; i < M; ++i) {
      b0 += (u128)lam.v[i] * y[i].x;
      b1 += (u128)lam.v[i] * y[i].y;
    }
    r.x = M61::fold128(b0);
    r.y = M61::fold128(b1);
    st<NT>(out + q0 + 256, r);
  }
}

// tiled layout: element (row i, pack q) at ((q / TP) * M + i) * TP + q % TP  -- rows interleaved per tile of TP packs
template <bool NT, int M>
__global__ __launch_bounds__(256) void k_rec_tiled(u64x2* out, const u64x2* sh, size_t TP, Lam lam, size_t n16) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n16; q += (size_t)gridDim.x * 256) {
    const size_t base = (q / TP) * M * TP + (q % TP);
    u64x2 x[M];
#pragma unroll
    for (int i = 0; i < M; ++i) x[i] = ld<NT>(sh + base + i * TP);
    u128 a0 = 0, a1 = 0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      a0 += (u128)lam.v[i] * x[i].x;
      a1 += (u128)lam.v[i] * x[i].y;
    }
    u64x2 r;
    r.x = M61::fold128(a0);
    r.y = M61::fold128(a1);
    st<NT>(out + q, r);
  }
}

template <class F>
float timeit(F&& launch, int reps = 10) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipGetLastError());
  return ms / reps;
}

int main(int argc, char** argv) {
  const size_t N = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 100000000ull;
  const int M = 10;
  const size_t n16 = N / 2, stride16 = n16;
  u64x2 *sh, *out;
  u64* sink;
  CK(hipMalloc(&sh, (size_t)M * n16 * 16));
  CK(hipMalloc(&out, n16 * 16));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(sh, 0x11, (size_t)M * n16 * 16));
  Lam lam;
  for (int i = 0; i < 16; ++i) lam.v[i] = 0x0123456789abcdefull * (i + 3) & M61::P;
  const size_t all16 = (size_t)M * n16;
  auto grid = [](size_t items, int blk, long cap) {
    size_t b = (items + blk - 1) / blk;
    if (cap > 0 && b > (size_t)cap) b = cap;
    return dim3((unsigned)b);
  };
  std::printf("N=%zu secrets, %d rows of %.2f GB\n", N, M, n16 * 16 / 1e9);
  for (long cap : {0L, 4096L, 16384L}) {
    std::printf("-- grid cap %ld (0 = one pack per thread)\n", cap);
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_read<true>, grid(all16, 256, cap), dim3(256), 0, 0, sh, all16, sink); });
    std::printf("read-only  nt   %7.0f GB/s\n", all16 * 16 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_read<false>, grid(all16, 256, cap), dim3(256), 0, 0, sh, all16, sink); });
    std::printf("read-only       %7.0f GB/s\n", all16 * 16 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_write<true>, grid(all16, 256, cap), dim3(256), 0, 0, sh, all16); });
    std::printf("write-only nt   %7.0f GB/s\n", all16 * 16 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_write<false>, grid(all16, 256, cap), dim3(256), 0, 0, sh, all16); });
    std::printf("write-only      %7.0f GB/s\n", all16 * 16 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_copy<true>, grid(all16 / 2, 256, cap), dim3(256), 0, 0, sh + all16 / 2, sh, all16 / 2); });
    std::printf("copy       nt   %7.0f GB/s\n", all16 * 16 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_shape<true, M>), grid(n16, 256, cap), dim3(256), 0, 0, out, sh, stride16, n16); });
    std::printf("10-in/1-out nt  %7.0f GB/s\n", (M + 1) * n16 * 16 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_shape<false, M>), grid(n16, 256, cap), dim3(256), 0, 0, out, sh, stride16, n16); });
    std::printf("10-in/1-out     %7.0f GB/s\n", (M + 1) * n16 * 16 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_rec<true, M, 256>), grid(n16, 256, cap), dim3(256), 0, 0, out, sh, stride16, lam, n16); });
    std::printf("recover b256 nt %7.0f GB/s\n", (M + 1) * n16 * 16 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_rec<true, M, 512>), grid(n16, 512, cap), dim3(512), 0, 0, out, sh, stride16, lam, n16); });
    std::printf("recover b512 nt %7.0f GB/s\n", (M + 1) * n16 * 16 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_rec<true, M, 1024>), grid(n16, 1024, cap), dim3(1024), 0, 0, out, sh, stride16, lam, n16); });
    std::printf("recover b1024 nt%7.0f GB/s\n", (M + 1) * n16 * 16 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_rec<true, M, 64>), grid(n16, 64, cap), dim3(64), 0, 0, out, sh, stride16, lam, n16); });
    std::printf("recover b64 nt  %7.0f GB/s\n", (M + 1) * n16 * 16 / ms / 1e6);
    if (cap == 0) {
      ms = timeit([&] { hipLaunchKernelGGL((k_rec2<true, M>), dim3((unsigned)((n16 + 511) / 512)), dim3(256), 0, 0, out, sh, stride16, lam, n16); });
      std::printf("recover 2packs  %7.0f GB/s\n", (M + 1) * n16 * 16 / ms / 1e6);
    }
  }
  // ---- stream-count and row-stride sensitivity of the many-streams access shape (one pack per thread)
  std::printf("-- k-in/1-out, contiguous rows\n");
  {
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL((k_shape<true, 1>), grid(n16, 256, 0), dim3(256), 0, 0, out, sh, stride16, n16); });
    std::printf("1-in/1-out   %7.0f GB/s\n", 2 * n16 * 16 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_shape<true, 2>), grid(n16, 256, 0), dim3(256), 0, 0, out, sh, stride16, n16); });
    std::printf("2-in/1-out   %7.0f GB/s\n", 3 * n16 * 16 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_shape<true, 4>), grid(n16, 256, 0), dim3(256), 0, 0, out, sh, stride16, n16); });
    std::printf("4-in/1-out   %7.0f GB/s\n", 5 * n16 * 16 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_shape<true, 8>), grid(n16, 256, 0), dim3(256), 0, 0, out, sh, stride16, n16); });
    std::printf("8-in/1-out   %7.0f GB/s\n", 9 * n16 * 16 / ms / 1e6);
  }
  std::printf("-- 10-in/1-out vs row stride padding (bytes)\n");
  {
    const size_t n16s = (N * 9 / 10) / 2;  // leave room for padding inside the allocation
    for (size_t pad16 : {(size_t)0, (size_t)16, (size_t)64, (size_t)128, (size_t)256, (size_t)272, (size_t)1024, (size_t)4096 + 16,
                         (size_t)65536 + 16, (size_t)(1 << 20) + 272, (size_t)3000000}) {
      const size_t st16 = n16s + pad16;
      float ms = timeit([&] { hipLaunchKernelGGL((k_rec<true, M, 256>), grid(n16s, 256, 0), dim3(256), 0, 0, out, sh, st16, lam, n16s); });
      std::printf("pad %10zu B  stride %% 4096 = %4zu : %7.0f GB/s\n", pad16 * 16, (st16 * 16) % 4096, (M + 1) * n16s * 16 / ms / 1e6);
    }
  }
  std::printf("-- recover on a tiled layout [tile][party][TP packs]\n");
  for (size_t TP : {(size_t)256, (size_t)512, (size_t)1024, (size_t)2048, (size_t)4096, (size_t)8192, (size_t)16384, (size_t)32768,
                    (size_t)65536, (size_t)(1 << 18), (size_t)(1 << 20), (size_t)(4096 + 256), (size_t)(4096 + 16), (size_t)(8192 + 64)}) {
    const size_t n16t = n16 / TP * TP;
    float ms = timeit([&] { hipLaunchKernelGGL((k_rec_tiled<true, M>), grid(n16t, 256, 0), dim3(256), 0, 0, out, sh, TP, lam, n16t); });
    std::printf("tile %8zu packs (%8zu B per row-tile): %7.0f GB/s\n", TP, TP * 16, (M + 1) * n16t * 16 / ms / 1e6);
  }
  std::printf("-- SoA rows, stride = 640 MiB + delta\n");
  {
    const size_t n16s = ((size_t)640 << 20) / 16;  // 640 MiB rows
    for (size_t delta : {(size_t)0, (size_t)4096, (size_t)16384, (size_t)65536, (size_t)(65536 + 4096), (size_t)(1 << 18), (size_t)(1 << 20),
                         (size_t)((1 << 20) + 65536), (size_t)(3 << 20), (size_t)((5 << 20) + 65536 + 256)}) {
      const size_t st16 = n16s + delta / 16;
      float ms = timeit([&] { hipLaunchKernelGGL((k_rec<true, M, 256>), grid(n16s, 256, 0), dim3(256), 0, 0, out, sh, st16, lam, n16s); });
      std::printf("delta %9zu B : %7.0f GB/s\n", delta, (M + 1) * n16s * 16 / ms / 1e6);
    }
  }
  return 0;
}
