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
  CK(hipGetLastError());
  return 0;
}
