// tools/oprate.hip -- issue cost of single vector-ALU opcodes on gfx950: cycles per wave64 instruction per SIMD for streams of
// ONE opcode on CH independent registers, at 8, 4, 2 and 1 waves per SIMD.  Which integer / logic opcodes run at the
// 2-cycle rate and which at 4 decides what a "lookup" or a "fold" costs in the VALU-bound kernels.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/_build/oprate tools/oprate.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef unsigned int u32;
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1);} } while (0)
constexpr int ITERS = 2048, CH = 8;

#define OPS(X)                                                                                        \
  X(0, "v_xor_b32 (vop2)", "v_xor_b32 %0, %1, %0", 1)                                                 \
  X(1, "v_and_b32 (vop2)", "v_and_b32 %0, %1, %0", 1)                                                 \
  X(2, "v_add_u32 (vop2)", "v_add_u32 %0, %1, %0", 1)                                                 \
  X(3, "v_lshlrev_b32 imm", "v_lshlrev_b32 %0, 3, %0", 1)                                             \
  X(4, "v_lshrrev_b32 vgpr", "v_lshrrev_b32 %0, %1, %0", 1)                                           \
  X(5, "v_bfe_u32", "v_bfe_u32 %0, %0, 8, 8", 1)                                                      \
  X(6, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 7, %1", 1)                                           \
  X(7, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 24", 1)                                          \
  X(8, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2", 1)                                              \
  X(9, "v_add_u32_sdwa byte", "v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", 1) \
  X(10, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2", 1)                                                 \
  X(11, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2", 1)                                           \
  X(12, "v_mov_b32", "v_mov_b32 %0, %1", 1)                                                           \
  X(13, "v_cndmask_b32 vcc", "v_cndmask_b32 %0, %0, %1, vcc", 1)                                      \
  X(14, "v_xor_b32 e64 (sgpr op)", "v_xor_b32_e64 %0, %0, s4", 1)                                     \
  X(15, "v_xnor_b32", "v_xnor_b32 %0, %1, %0", 1)                                                     \
  X(16, "v_or3_b32", "v_or3_b32 %0, %0, %1, %2", 1)                                                   \
  X(17, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2", 1)                                                 \
  X(18, "v_bfi_b32", "v_bfi_b32 %0, %1, %0, %2", 1)                                                   \
  X(19, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1", 1)                                                 \
  X(20, "v_xor_b32 inline const", "v_xor_b32 %0, 7, %0", 1)                                           \
  X(21, "v_and_b32 literal", "v_and_b32 %0, 0xf0f0f0f0, %0", 1)                                       \
  X(22, "v_lshlrev_b32 vgpr", "v_lshlrev_b32 %0, %1, %0", 1)                                          \
  X(23, "v_lshrrev_b32 imm", "v_lshrrev_b32 %0, 8, %0", 1)                                            \
  X(24, "v_or_b32 (vop2)", "v_or_b32 %0, %1, %0", 1)                                                  \
  X(25, "v_sub_u32 (vop2)", "v_sub_u32 %0, %1, %0", 1)                                                \
  X(26, "v_and_b32 sgpr", "v_and_b32 %0, s4, %0", 1)                                                  \
  X(27, "v_mad_u64_u32", "v_mad_u64_u32 %0, s[6:7], %1, %2, %0", 2)                                   \
  X(28, "v_mov_b32 dpp row_shr", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", 1)                   \
  X(29, "v_bitop3_b32 (a^b^c)", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96", 1)                          \
  X(30, "v_lshl_add_u64", "v_lshl_add_u64 %0, %0, 3, %0", 2)                                         \
  X(31, "v_lshlrev_b64", "v_lshlrev_b64 %0, 3, %0", 2)                                               \
  X(32, "v_lshrrev_b64", "v_lshrrev_b64 %0, 3, %0", 2)

template <int OP>
__global__ __launch_bounds__(256) void k(u32* out, u32 seed) {
  typename std::conditional<OP == 27 || OP >= 30, u64, u32>::type a[CH];
  u32 b = seed | 1, c = threadIdx.x * 2654435761u + seed;
  for (int j = 0; j < CH; ++j) a[j] = threadIdx.x * 0x9E3779B9u + j;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
#define X(ID, NAME, ASM, N) \
  if constexpr (OP == ID) asm volatile(ASM : "+v"(a[j]) : "v"(b), "v"(c) : "vcc", "s4", "s6", "s7");
      OPS(X)
#undef X
    }
  }
  u32 r = 0;
  for (int j = 0; j < CH; ++j) r ^= (u32)a[j];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
}
// fully dependent chain of v_xor_b32 (one register)
__global__ __launch_bounds__(256) void k_dep(u32* out, u32 seed) {
  u32 a = threadIdx.x, b = seed | 1;
  for (int i = 0; i < ITERS * CH; ++i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "v"(b));
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = a;
}

int main() {
  u32* out;
  CK(hipMalloc(&out, (size_t)256 * 8 * 256 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // in-kernel clock estimate: a dependent v_xor chain issues one instruction per >= 4 cycles per wave; report both clocks
  const int occupancies[] = {8, 4, 2, 1};  // 256-thread blocks per CU (= waves per SIMD)
  std::printf("%-26s", "opcode \\ waves per SIMD");
  for (int o : occupancies) std::printf("  %8d", o);
  std::printf("   (cycles per wave64 instruction per SIMD at 2.4 GHz)\n");
  auto run = [&](auto kern, const char* name) {
    std::printf("%-26s", name);
    for (int o : occupancies) {
      const int blocks = 256 * o;
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 12345u);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 12345u);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double instr_per_simd = 5.0 * o * (double)ITERS * CH;  // o waves per SIMD, each ITERS * CH instructions
      std::printf("  %8.2f", ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
    std::printf("\n");
  };
#define X(ID, NAME, ASM, N) run(k<ID>, NAME);
  OPS(X)
#undef X
  run(k_dep, "v_xor_b32 dependent chain");
  return 0;
}
