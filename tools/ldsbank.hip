// tools/ldsbank.hip -- how many banks does a 64-lane ds_read_b32 spread over?  16 waves per CU read LDS in a loop with four
// address patterns; LDS-array cycles per instruction per CU from the wall time.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/_build/ldsbank tools/ldsbank.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef unsigned u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1);} } while (0)

// MODE 0: lane l reads word l (64 distinct words, 64 consecutive banks if there are 64)
// MODE 1: lanes l and l + 32 read words l and l + 2048 (bank l mod 32 or 64 both times, different addresses)
// MODE 2: lanes l and l + 32 read words l and 32 + l + 2048 (distinct banks if there are 64, the same if there are 32)
// MODE 3: all lanes read the same word (broadcast)
template <int MODE>
__global__ __launch_bounds__(1024) void k_lds(u32* out, int iters) {
  __shared__ u32 t[16384];
  for (int i = threadIdx.x; i < 16384; i += 1024) t[i] = i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  u32 idx = MODE == 0 ? lane : MODE == 1 ? (lane & 31) + (lane >> 5) * 2048 : MODE == 2 ? (lane & 31) + (lane >> 5) * (2048 + 32) : 5;
  u32 acc = 0;
  u32 a = (u32)(uintptr_t)t + 4 * idx;  // LDS byte address; the 16 reads of an iteration differ in the offset field only
  for (int i = 0; i < iters; ++i) {
    u32 v[16];
    asm volatile(
        "ds_read_b32 %0, %16 offset:0\n\tds_read_b32 %1, %16 offset:256\n\tds_read_b32 %2, %16 offset:512\n\t"
        "ds_read_b32 %3, %16 offset:768\n\tds_read_b32 %4, %16 offset:1024\n\tds_read_b32 %5, %16 offset:1280\n\t"
        "ds_read_b32 %6, %16 offset:1536\n\tds_read_b32 %7, %16 offset:1792\n\tds_read_b32 %8, %16 offset:2048\n\t"
        "ds_read_b32 %9, %16 offset:2304\n\tds_read_b32 %10, %16 offset:2560\n\tds_read_b32 %11, %16 offset:2816\n\t"
        "ds_read_b32 %12, %16 offset:3072\n\tds_read_b32 %13, %16 offset:3328\n\tds_read_b32 %14, %16 offset:3584\n\t"
        "ds_read_b32 %15, %16 offset:3840\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]),
          "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
        : "v"(a)
        : "memory");
#pragma unroll
    for (int u = 0; u < 16; ++u) acc ^= v[u];
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

// ---- 16-byte accesses (round 4): what a 64-lane ds_read_b128 / ds_write_b128 costs by address pattern ----------------------
// MODE 0: lane l at byte 16 l (1 KiB contiguous per instruction: k_share_gf_tiles' per-lane coefficient slots)
// MODE 1: lane l at byte 16 (hash(l) & 15) of a 256-byte table (k_recover_gf128_pos: a random entry of one bank row)
// MODE 2: lane l at byte 16 l with the 16-byte slots of lanes 4 apart swapped in bank order ((l ^ (l >> 2 & 3)) ...): a swizzle
// MODE 3: as 0, but ds_write_b128
typedef u32 u32x4v __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(1024) void k_lds128(u32* out, int iters) {
  __shared__ __attribute__((aligned(16))) u32 t[16384];
  for (int i = threadIdx.x; i < 16384; i += 1024) t[i] = i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  u32 slot = lane;
  if (MODE == 1) slot = (lane * 2654435761u >> 13) & 15;
  if (MODE == 2) slot = lane ^ ((lane >> 3) & 7);
  const u32 a = (u32)(uintptr_t)t + 16 * slot;
  u32x4v acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 3) {
      u32x4v v = acc;
      asm volatile(
          "ds_write_b128 %0, %1 offset:0\n\tds_write_b128 %0, %1 offset:1024\n\tds_write_b128 %0, %1 offset:2048\n\t"
          "ds_write_b128 %0, %1 offset:3072\n\tds_write_b128 %0, %1 offset:4096\n\tds_write_b128 %0, %1 offset:5120\n\t"
          "ds_write_b128 %0, %1 offset:6144\n\tds_write_b128 %0, %1 offset:7168\n\ts_waitcnt lgkmcnt(0)"
          :
          : "v"(a), "v"(v)
          : "memory");
      acc.x += 1;
    } else {
      u32x4v v[8];
      asm volatile(
          "ds_read_b128 %0, %8 offset:0\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\t"
          "ds_read_b128 %3, %8 offset:3072\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\t"
          "ds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
          : "v"(a)
          : "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// ---- round 6: a RANDOM entry of a table of 2^LOGE 16-byte entries per lane (what a window of LOGE bits of a share looks up) ----
// LOGE 4 is the 256-byte table of k_recover_gf128_pos (one bank row: conflict-free by construction); 8 is the byte window's
// 4 KiB table (16 bank rows: lanes that pick the same bank quad in different rows collide).  Sixteen random slots per lane are
// drawn BEFORE the loop (sixteen address registers; the loop itself is the reads and the xors of the probes above -- a first
// form that drew new slots every iteration spent as long on its address arithmetic as on the reads), 16 waves per CU x 256 CUs
// of different lanes: the figure is a mean over 65 536 random wave patterns.  SAME: every lane of a wave the same entry.
template <int LOGE, bool SAME>
__global__ __launch_bounds__(1024) void k_lds128_random(u32* out, int iters) {
  __shared__ __attribute__((aligned(16))) u32 t[16384];   // 64 KiB: eight regions of 8 KiB
  for (int i = threadIdx.x; i < 16384; i += 1024) t[i] = i * 2654435761u;
  __syncthreads();
  const u32 base = (u32)(uintptr_t)t;
  u32 x = ((SAME ? (threadIdx.x >> 6) : threadIdx.x) + blockIdx.x * 1024u) * 2654435761u + 12345u;
  constexpr u32 MASK = (1u << LOGE) - 1;
  u32 a[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    x = x * 1664525u + 1013904223u;
    a[u] = base + (u32)(u & 7) * 8192u + (((x >> 13) & MASK) << 4);
  }
  u32x4v acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; i += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u32x4v v[8];
      asm volatile(
          "ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
          "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
          : "v"(a[8 * h + 0]), "v"(a[8 * h + 1]), "v"(a[8 * h + 2]), "v"(a[8 * h + 3]), "v"(a[8 * h + 4]), "v"(a[8 * h + 5]),
            "v"(a[8 * h + 6]), "v"(a[8 * h + 7])
          : "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}
// How many lanes does the LDS serve per cycle of a ds_read_b128, and does it merge equal addresses?  Lane l reads entry
// (l / GROUP) of a table whose entries sit STRIDE16 x 16 bytes apart: GROUP lanes share an address (a broadcast if the
// hardware merges them), distinct addresses fall into the same bank quad when STRIDE16 is a multiple of 16.
template <int GROUP, int STRIDE16>
__global__ __launch_bounds__(1024) void k_lds128_pattern(u32* out, int iters) {
  __shared__ __attribute__((aligned(16))) u32 t[16384];
  for (int i = threadIdx.x; i < 16384; i += 1024) t[i] = i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const u32 a = (u32)(uintptr_t)t + 16u * (u32)STRIDE16 * (u32)(lane / GROUP);
  u32x4v acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    u32x4v v[8];
    asm volatile(
        "ds_read_b128 %0, %8 offset:0\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\t"
        "ds_read_b128 %3, %8 offset:48\n\tds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\t"
        "ds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(a)
        : "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u];
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

int main() {
  u32* out;
  CK(hipMalloc(&out, 256 * 1024 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 4000;
  auto run = [&](auto kern, const char* name) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, out, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = 16.0 * iters * 16;  // waves x iterations x reads
    std::printf("%-64s %7.3f ms  %5.2f cycles per ds_read_b32 per CU at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
  };
  run(k_lds<0>, "64 lanes, 64 consecutive words");
  run(k_lds<1>, "lanes l and l+32: same word index mod 64, different address");
  run(k_lds<2>, "lanes l and l+32: word indices 32 apart mod 64");
  run(k_lds<3>, "all lanes one word");
  auto run128 = [&](auto kern, const char* name) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, out, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = 16.0 * iters * 8;
    std::printf("%-64s %7.3f ms  %5.2f cycles per 16-byte instruction per CU at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
  };
  run128(k_lds128<0>, "ds_read_b128, lane l at byte 16 l (1 KiB contiguous)");
  run128(k_lds128<1>, "ds_read_b128, every lane a random entry of one 256-byte table");
  run128(k_lds128<2>, "ds_read_b128, 16 l with slots swizzled within groups of 8");
  run128(k_lds128<3>, "ds_write_b128, lane l at byte 16 l (1 KiB contiguous)");
  std::printf("# round 6: ds_read_b128, every lane a RANDOM entry of a table of 2^k 16-byte entries (sixteen slots per lane drawn before the loop)\n");
  run128(k_lds128_random<4, false>, "random entry of a 256-byte table (16 entries: 4-bit window)");
  run128(k_lds128_random<5, false>, "random entry of a 512-byte table (32 entries: 5-bit window)");
  run128(k_lds128_random<6, false>, "random entry of a 1 KiB table (64 entries: 6-bit window)");
  run128(k_lds128_random<7, false>, "random entry of a 2 KiB table (128 entries: 7-bit window)");
  run128(k_lds128_random<8, false>, "random entry of a 4 KiB table (256 entries: 8-bit window)");
  run128(k_lds128_random<9, false>, "random entry of an 8 KiB table (512 entries)");
  run128(k_lds128_random<8, true>, "4 KiB table, all lanes of a wave the same random entry");
  std::printf("# which lanes collide: lane l reads entry l / GROUP, entries STRIDE x 16 bytes apart\n");
  run128(k_lds128_pattern<1, 16>, "64 distinct addresses, all in ONE bank quad (stride 256 B)");
  run128(k_lds128_pattern<4, 16>, "16 distinct addresses in one bank quad, 4 lanes each");
  run128(k_lds128_pattern<16, 16>, "4 distinct addresses in one bank quad, 16 lanes each");
  run128(k_lds128_pattern<32, 16>, "2 distinct addresses in one bank quad, 32 lanes each");
  run128(k_lds128_pattern<16, 17>, "4 distinct addresses in 4 bank quads, 16 lanes each");
  run128(k_lds128_pattern<4, 17>, "16 distinct addresses in 16 bank quads, 4 lanes each");
  run128(k_lds128_pattern<2, 17>, "32 distinct addresses, 2 per bank quad two rows apart, 2 lanes each");
  CK(hipGetLastError());
  return 0;
}
