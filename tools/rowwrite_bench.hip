// tools/rowwrite_bench.hip -- HBM write rate of the matrix-core share kernel's store pattern, without the arithmetic:
// a workgroup of 8 waves writes RUN bytes of each of 128 rows per trip (wave w rows 16w .. 16w+15; a lane 32 bytes as two
// 16-byte stores), block b of a trip = blockIdx.x + trip * gridDim.x as in k_share_mfma_m61_p16 (RUN = 256 there).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/_build/rowwrite_bench tools/rowwrite_bench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1);} } while (0)

// RUN bytes per row per trip (256, 512, 1024, 2048); NT: non-temporal stores
template <int RUN, bool NT>
__global__ __launch_bounds__(512, 1) void k_rows(u64* out, size_t stride, size_t N) {
  constexpr int COLS = RUN / 8;  // secrets per trip
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, kb = lane >> 4;
  const int row = 16 * w + r16;
  const size_t nblocks = N / COLS;
  for (size_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    u64* rowp = out + (size_t)row * stride + blk * COLS + 4 * kb;
#pragma unroll
    for (int ct = 0; ct < COLS / 16; ++ct) {
      u64x2 o0 = {blk + ct, (u64)row}, o1 = {blk, (u64)lane};
      u64x2* dst = reinterpret_cast<u64x2*>(rowp + 16 * ct);
      if (NT) {
        __builtin_nontemporal_store(o0, dst);
        __builtin_nontemporal_store(o1, dst + 1);
      } else {
        dst[0] = o0;
        dst[1] = o1;
      }
    }
  }
}

// the same bytes with the pieces permuted between the four lanes of a row (lanes 16 apart): store instruction j of a tile writes
// the j-th 64 contiguous bytes of the row's 128 (lane piece 16 bytes at 16 * kb), instead of four 16-byte pieces 32 bytes apart
template <int RUN, bool NT>
__global__ __launch_bounds__(512, 1) void k_rows_perm(u64* out, size_t stride, size_t N) {
  constexpr int COLS = RUN / 8;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, kb = lane >> 4;
  const int row = 16 * w + r16;
  const size_t nblocks = N / COLS;
  for (size_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    u64* rowp = out + (size_t)row * stride + blk * COLS + 2 * kb;
#pragma unroll
    for (int ct = 0; ct < COLS / 16; ++ct) {
      u64x2 o0 = {blk + ct, (u64)row}, o1 = {blk, (u64)lane};
      u64x2* dst = reinterpret_cast<u64x2*>(rowp + 16 * ct);
      if (NT) {
        __builtin_nontemporal_store(o0, dst);
        __builtin_nontemporal_store(o1, dst + 4);
      } else {
        dst[0] = o0;
        dst[4] = o1;
      }
    }
  }
}

int main(int argc, char** argv) {
  const size_t N = argc > 1 ? strtoull(argv[1], 0, 10) : 20000000;
  const int rows = 128;
  u64* out;
  CK(hipMalloc(&out, (size_t)rows * N * 8));
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
    std::printf("%-56s %8.3f ms  %6.0f GB/s written\n", name, ms, (double)rows * 8 * N / ms / 1e6);
  };
#define RUN_IT(RUN, NT, GRID, name) time_it([&] { hipLaunchKernelGGL((k_rows<RUN, NT>), dim3(GRID), dim3(512), 0, 0, out, N, N); }, name)
  RUN_IT(256, true, 256, "256 B per row per trip, streaming stores, 256 groups");
  RUN_IT(256, false, 256, "256 B per row per trip, cached stores,    256 groups");
  RUN_IT(512, true, 256, "512 B per row per trip, streaming stores, 256 groups");
  RUN_IT(1024, true, 256, "1 KiB per row per trip, streaming stores, 256 groups");
  RUN_IT(2048, true, 256, "2 KiB per row per trip, streaming stores, 256 groups");
  RUN_IT(256, true, 512, "256 B per row per trip, streaming stores, 512 groups");
  RUN_IT(256, true, 1024, "256 B per row per trip, streaming stores, 1024 groups");
  RUN_IT(1024, true, 1024, "1 KiB per row per trip, streaming stores, 1024 groups");
#define RUN_P(RUN, NT, GRID, name) time_it([&] { hipLaunchKernelGGL((k_rows_perm<RUN, NT>), dim3(GRID), dim3(512), 0, 0, out, N, N); }, name)
  RUN_P(256, true, 256, "256 B per row per trip, 64-byte pieces, streaming");
  RUN_P(256, false, 256, "256 B per row per trip, 64-byte pieces, cached");
  RUN_P(1024, true, 256, "1 KiB per row per trip, 64-byte pieces, streaming");
  CK(hipGetLastError());
  return 0;
}
