// tools/bitop3_probe.hip -- the truth-table convention of gfx950's v_bitop3_b32: with the operands 0xF0, 0xCC, 0xAA the low byte
// of the result IS the table, i.e. table bit (4 a + 2 b + c) is the result for operand bits (a, b, c) of (src0, src1, src2).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/_build/bitop3_probe tools/bitop3_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int T>
__device__ unsigned one(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_bitop3_b32(a, b, c, T); }
__global__ void k(unsigned* out, unsigned a, unsigned b, unsigned c) {
  out[0] = one<0x96>(a, b, c);
  out[1] = one<0xE4>(a, b, c);
  out[2] = one<0xCA>(a, b, c);
  out[3] = one<0x78>(a, b, c);
  out[4] = one<0x1B>(a, b, c);
}
int main() {
  unsigned* d;
  hipMalloc(&d, 64);
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, 0xF0u, 0xCCu, 0xAAu);
  unsigned h[5];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const unsigned want[5] = {0x96, 0xE4, 0xCA, 0x78, 0x1B};
  int bad = 0;
  for (int i = 0; i < 5; ++i) {
    std::printf("table 0x%02X on (0xF0, 0xCC, 0xAA) -> 0x%02X%s\n", want[i], h[i] & 0xFF, (h[i] & 0xFF) == want[i] ? "" : "   <-- differs");
    bad += (h[i] & 0xFF) != want[i];
  }
  std::printf(bad ? "convention differs\n" : "convention confirmed: table bit 4a+2b+c; a^b^c = 0x96, (c ? a : b) = 0xE4\n");
  return bad;
}
