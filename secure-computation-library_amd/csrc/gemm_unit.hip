// csrc/gemm_unit.hip -- the general matrix-core product's kernels (gemm_mfma.hpp) as a translation unit of their own.
//
// The units of capi.hip are compiled with -amdgpu-mfma-vgpr-form=1 (matrix-instruction accumulators in architectural registers:
// the sharing kernels recombine them in place on the vector ALU).  k_gemm_mfma_m61 wants the opposite: its fifteen accumulator
// tiles (240 registers) belong in the accumulation registers, which the vector ALU does not touch inside the k loop, leaving the
// 256 architectural ones to the two register images of the operand fragments.  Compiled with the flag the kernel shuttles
// accumulator tiles between the two files inside its loop (v_accvgpr_read / _write by the hundred).  So: this unit, without it.
// The three launchers below are all the library sees (internal linkage to the outside: csrc/exports.map).
#include <hip/hip_runtime.h>

#include "gemm_mfma.hpp"

namespace sclhip {

hipError_t gemm_launch_planes_a(u64x2* planes, const u64* A, size_t lda, size_t M, size_t K, size_t ktiles, hipStream_t st) {
  const size_t g = (((M + 31) / 32) * ktiles * 64 + 255) / 256;
  hipLaunchKernelGGL(k_gemm_planes_a<>, dim3((unsigned)(g < (1u << 20) ? g : (1u << 20))), dim3(256), 0, st, planes, A, lda, M, K, ktiles);
  return hipGetLastError();
}

hipError_t gemm_launch_planes_b(u64x2* planes, const u64* B, size_t ldb, size_t K, size_t N, size_t ktiles, hipStream_t st) {
  const size_t g = (((N + 31) / 32) * ktiles * 64 + 255) / 256;
  hipLaunchKernelGGL(k_gemm_planes_b<>, dim3((unsigned)(g < (1u << 20) ? g : (1u << 20))), dim3(256), 0, st, planes, B, ldb, K, N, ktiles);
  return hipGetLastError();
}

// splits = 1: the product into C (row pitch ldc); splits > 1: slice y of the k-steps writes its partial product to C + y * cslice
hipError_t gemm_launch_main(u64* C, size_t ldc, const u64x2* Ap, const u64x2* Bp, size_t M, size_t N, size_t ktiles, size_t kslice,
                            size_t cslice, size_t splits, hipStream_t st) {
  const size_t wgs = (((M + 31) / 32 + 1) / 2) * (((N + 31) / 32 + 1) / 2);
  hipLaunchKernelGGL(k_gemm_mfma_m61<>, dim3((unsigned)(wgs < (1u << 24) ? wgs : (1u << 24)), (unsigned)splits), dim3(256), 0, st, C, ldc,
                     Ap, Bp, M, N, ktiles, kslice, cslice);
  return hipGetLastError();
}

}  // namespace sclhip
