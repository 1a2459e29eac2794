// tools/aes_bitslice_bench.hip -- bit-sliced AES-128-CTR on gfx950's three-input bitwise instruction (v_bitop3_b32), sized and
// measured against the library's four-table kernel (k_prg_blocks): VERDICT r3 "next" #2.
//
// Form: a lane holds 32 counter blocks as 128 bit planes (plane = one u32, bit j = block j of the lane), so every bitwise
// instruction works on 64 x 32 = 2048 blocks per wave.  Counter blocks are generated directly in plane form -- block
// j * 64 + lane of a wave's 2048-block group: the counter's bits 0..5 are the lane number (a plane is all-zeros or all-ones
// per lane), bits 6..10 are the five constant patterns 0xAAAAAAAA .. 0xFFFF0000, everything above is wave-uniform -- there is
// no input transpose.  SubBytes is the generated three-input-LUT circuit (tools/gen_aes_bitslice.py: 82 operations per byte,
// from Boyar & Peralta's 128-gate depth-16 circuit), ShiftRows is a renaming, MixColumns + AddRoundKey are three-way xors
// (round-key bits are wave-uniform 0 / ~0 masks from scalar registers), the ciphertext planes go through four 32 x 32 bit
// transposes and leave as 16-byte stores, 1 KiB contiguous per store instruction.
// Real AES: the key schedule and S-box are computed on the host from their definitions, and the output is compared word for
// word with the library's table kernel on the same key and counters.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/_build/aes_bitslice_bench tools/aes_bitslice_bench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../secure-computation-library_amd/csrc/kernels.hpp"
using namespace sclhip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1);} } while (0)

#define LUT3(tt, a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), (tt))
__device__ __forceinline__ u32 x3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }  // a ^ b ^ c
#include "aes_bitslice_gen.inc"

struct BsKey {
  u32 rk[44];  // round keys, little-endian column words as in AesKey
};

// round-key plane of state byte k, bit i, round r: all-ones where the key bit is set (wave-uniform: scalar registers)
template <int R, int K, int I>
__device__ __forceinline__ u32 kplane(const BsKey& key) {
  return (u32)(-(int)((key.rk[4 * R + K / 4] >> (8 * (K % 4) + I)) & 1u));
}

// MixColumns + AddRoundKey of one row of a column (the key plane of row ROW needs ROW as a template argument)
template <int R, int C, int ROW, bool LAST>
__device__ __forceinline__ void mix_row(u32 (&q)[16][8], const u32 (&a)[4][8], const u32 (&s)[8], const u32 (&t7)[4], const BsKey& key) {
  constexpr int K = 4 * C + ROW, N = (ROW + 1) & 3;
  if constexpr (LAST) {  // the last round has no MixColumns
    q[K][0] = a[ROW][0] ^ kplane<R, K, 0>(key);
    q[K][1] = a[ROW][1] ^ kplane<R, K, 1>(key);
    q[K][2] = a[ROW][2] ^ kplane<R, K, 2>(key);
    q[K][3] = a[ROW][3] ^ kplane<R, K, 3>(key);
    q[K][4] = a[ROW][4] ^ kplane<R, K, 4>(key);
    q[K][5] = a[ROW][5] ^ kplane<R, K, 5>(key);
    q[K][6] = a[ROW][6] ^ kplane<R, K, 6>(key);
    q[K][7] = a[ROW][7] ^ kplane<R, K, 7>(key);
  } else {
    // out[i] = a_r[i] ^ s[i] ^ xtime(a_r ^ a_n)[i] ^ key, s = a_0 ^ a_1 ^ a_2 ^ a_3, xtime(v)[i] = v[i-1] ^ (i in {0,1,3,4}) v[7]
    q[K][0] = x3(a[ROW][0], s[0], t7[ROW]) ^ kplane<R, K, 0>(key);
    q[K][1] = x3(a[ROW][1], s[1], t7[ROW]) ^ x3(a[ROW][0], a[N][0], kplane<R, K, 1>(key));
    q[K][2] = x3(x3(a[ROW][2], s[2], a[ROW][1]), a[N][1], kplane<R, K, 2>(key));
    q[K][3] = x3(a[ROW][3], s[3], t7[ROW]) ^ x3(a[ROW][2], a[N][2], kplane<R, K, 3>(key));
    q[K][4] = x3(a[ROW][4], s[4], t7[ROW]) ^ x3(a[ROW][3], a[N][3], kplane<R, K, 4>(key));
    q[K][5] = x3(x3(a[ROW][5], s[5], a[ROW][4]), a[N][4], kplane<R, K, 5>(key));
    q[K][6] = x3(x3(a[ROW][6], s[6], a[ROW][5]), a[N][5], kplane<R, K, 6>(key));
    q[K][7] = x3(x3(a[ROW][7], s[7], a[ROW][6]), a[N][6], kplane<R, K, 7>(key));
  }
}

// SubBytes on the four bytes that ShiftRows sends into column C, then MixColumns + AddRoundKey of that column into q
template <int R, int C, bool LAST>
__device__ __forceinline__ void column(u32 (&q)[16][8], const u32 (&p)[16][8], const BsKey& key) {
  u32 a[4][8];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int src = 4 * ((C + r) & 3) + r;
    u32 u[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) u[j] = p[src][7 - j];
    aes_sbox_planes(sb, u);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[r][7 - j] = sb[j];
  }
  u32 s[8], t7[4];
  if constexpr (!LAST) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = x3(a[0][i], a[1][i], a[2][i]) ^ a[3][i];
#pragma unroll
    for (int r = 0; r < 4; ++r) t7[r] = a[r][7] ^ a[(r + 1) & 3][7];
  }
  mix_row<R, C, 0, LAST>(q, a, s, t7, key);
  mix_row<R, C, 1, LAST>(q, a, s, t7, key);
  mix_row<R, C, 2, LAST>(q, a, s, t7, key);
  mix_row<R, C, 3, LAST>(q, a, s, t7, key);
}

template <int R>
__device__ __forceinline__ void rounds(u32 (&p)[16][8], const BsKey& key) {
  if constexpr (R <= 10) {
    u32 q[16][8];
    column<R, 0, R == 10>(q, p, key);
    column<R, 1, R == 10>(q, p, key);
    column<R, 2, R == 10>(q, p, key);
    column<R, 3, R == 10>(q, p, key);
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) p[k][i] = q[k][i];
    rounds<R + 1>(p, key);
  }
}

// 32 x 32 bit transpose: afterwards bit q of x[j] is what bit j of x[q] was
__device__ __forceinline__ void transpose32(u32 (&x)[32]) {
#define BS_STAGE(S, M)                                                 \
  _Pragma("unroll") for (int k = 0; k < 32; ++k) if ((k & S) == 0) {   \
    const u32 a = x[k], b = x[k + S];                                  \
    x[k] = LUT3(0xE4, b << S, a, (u32)M);      /* bits of b << S where M is set, of a elsewhere */ \
    x[k + S] = LUT3(0xE4, b, a >> S, (u32)M);  /* bits of b where M is set, of a >> S elsewhere */ \
  }
  BS_STAGE(16, 0xFFFF0000u)
  BS_STAGE(8, 0xFF00FF00u)
  BS_STAGE(4, 0xF0F0F0F0u)
  BS_STAGE(2, 0xCCCCCCCCu)
  BS_STAGE(1, 0xAAAAAAAAu)
#undef BS_STAGE
}

// counter0 must be a multiple of 2048 and no group may cross a multiple of 2^32 (the bench's launches: checked on the host)
template <int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_aes_bitslice(uint4* dst, BsKey key, u64 counter0,
                                                                                                    size_t ngroups) {
  const u32 lane = threadIdx.x;
  const u64 nonce = 0x0123456789ABCDEFull;  // prg.h:34-43
  for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const u64 base = counter0 + (u64)g * 2048u;  // wave-uniform; its low 11 bits are zero
    u32 p[16][8];
    // input block = LE64(counter) || LE64(nonce); byte k bit i of the counter is counter bit 8 k + i
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int bit = 8 * k + i;
        u32 v;
        if (bit < 6) v = (u32)(-(int)((lane >> bit) & 1u));
        else if (bit == 6) v = 0xAAAAAAAAu;
        else if (bit == 7) v = 0xCCCCCCCCu;
        else if (bit == 8) v = 0xF0F0F0F0u;
        else if (bit == 9) v = 0xFF00FF00u;
        else if (bit == 10) v = 0xFFFF0000u;
        else v = (u32)(-(int)((base >> bit) & 1u));
        p[k][i] = v ^ (u32)(-(int)((key.rk[k / 4] >> (8 * (k % 4) + i)) & 1u));
      }
#pragma unroll
    for (int k = 8; k < 16; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        p[k][i] = (u32)(-(int)(((nonce >> (8 * (k - 8) + i)) & 1u) ^ ((key.rk[k / 4] >> (8 * (k % 4) + i)) & 1u)));
    rounds<1>(p, key);
    // word w of block j: planes of bytes 4w .. 4w+3
    u32 x[4][32];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
#pragma unroll
      for (int b = 0; b < 32; ++b) x[w][b] = p[4 * w + b / 8][b % 8];
      transpose32(x[w]);
    }
    u32x4* out = reinterpret_cast<u32x4*>(dst) + (size_t)g * 2048u + lane;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      u32x4 v;
      v.x = x[0][j], v.y = x[1][j], v.z = x[2][j], v.w = x[3][j];
      __builtin_nontemporal_store(v, out + (size_t)j * 64);
    }
  }
}

// ---- host: real AES tables and key schedule from their definitions ------------------------------------------------------
static unsigned char g_sbox[256];
static void make_sbox() {
  auto mul = [](unsigned a, unsigned b) {
    unsigned r = 0;
    while (b) {
      if (b & 1) r ^= a;
      a = ((a << 1) ^ ((a & 0x80) ? 0x11b : 0)) & 0xff;
      b >>= 1;
    }
    return r;
  };
  for (int x = 0; x < 256; ++x) {
    unsigned inv = 0;
    for (int y = 1; y < 256 && x; ++y)
      if (mul(x, y) == 1) inv = y;
    unsigned s = inv, r = inv;
    for (int i = 0; i < 4; ++i) {
      s = ((s << 1) | (s >> 7)) & 0xff;
      r ^= s;
    }
    g_sbox[x] = (unsigned char)(r ^ 0x63);
  }
}
static void make_key(const unsigned char seed[16], AesKey& k) {
  unsigned char rk[176];
  std::memcpy(rk, seed, 16);
  unsigned rcon = 1;
  for (int i = 16; i < 176; i += 4) {
    unsigned char t[4] = {rk[i - 4], rk[i - 3], rk[i - 2], rk[i - 1]};
    if (i % 16 == 0) {
      const unsigned char t0 = t[0];
      t[0] = (unsigned char)(g_sbox[t[1]] ^ rcon);
      t[1] = g_sbox[t[2]];
      t[2] = g_sbox[t[3]];
      t[3] = g_sbox[t0];
      rcon = ((rcon << 1) ^ ((rcon & 0x80) ? 0x11b : 0)) & 0xff;
    }
    for (int j = 0; j < 4; ++j) rk[i + j] = (unsigned char)(rk[i - 16 + j] ^ t[j]);
  }
  for (int w = 0; w < 44; ++w) k.rk[w] = (u32)rk[4 * w] | ((u32)rk[4 * w + 1] << 8) | ((u32)rk[4 * w + 2] << 16) | ((u32)rk[4 * w + 3] << 24);
  for (int x = 0; x < 256; ++x) {
    const unsigned s = g_sbox[x], s2 = ((s << 1) ^ ((s & 0x80) ? 0x11b : 0)) & 0xff, s3 = s2 ^ s;
    k.te0[x] = s2 | (s << 8) | (s << 16) | (s3 << 24);
  }
  aes_key_round1(k);
}

int main(int argc, char** argv) {
  const size_t nblocks = argc > 1 ? strtoull(argv[1], 0, 10) : ((size_t)1 << 28);
  const u64 counter0 = 3u * 2048u;
  make_sbox();
  AesKey key;
  const unsigned char seed[16] = {'b', 'i', 't', 's', 'l', 'i', 'c', 'e', 'd', ' ', 'a', 'e', 's', 0, 1, 2};
  make_key(seed, key);
  aes_key_range(key, counter0, nblocks);
  BsKey bk;
  std::memcpy(bk.rk, key.rk, sizeof bk.rk);
  if (nblocks % 2048 || counter0 + nblocks > (1ull << 32)) {
    std::printf("nblocks must be a multiple of 2048 and stay below 2^32\n");
    return 1;
  }
  u64 *a, *b;
  CK(hipMalloc(&a, nblocks * 16));
  CK(hipMalloc(&b, nblocks * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time_it = [&](auto launch, const char* name) {
    launch();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    std::printf("%-64s %8.3f ms  %6.1f G blocks/s\n", name, ms, nblocks / ms / 1e6);
  };
  {
    auto kern = &k_prg_blocks<>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, AES4_LDS_BYTES));
    time_it([&] { hipLaunchKernelGGL(kern, dim3(AES4_GRID_CAP), dim3(ABLOCK), AES4_LDS_BYTES, 0, a, key, counter0, nblocks); },
            "library: k_prg_blocks (four LDS tables)");
  }
  auto check = [&](const char* name) {
    std::vector<u64> ha(1 << 18), hb(1 << 18);
    size_t diff = 0;
    for (size_t off : {(size_t)0, (nblocks / 2) & ~(size_t)2047, nblocks - (1 << 17)}) {
      CK(hipMemcpy(ha.data(), a + off * 2, ha.size() * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hb.data(), b + off * 2, hb.size() * 8, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < ha.size(); ++i) diff += ha[i] != hb[i];
    }
    std::printf("    %s against the table kernel, 3 x 2^17 blocks: %zu differing words\n", name, diff);
  };
  const size_t ngroups = nblocks / 2048;
#define RUN_BS(WPE, GRID, name)                                                                                       \
  {                                                                                                                    \
    CK(hipMemset(b, 0, nblocks * 16));                                                                                 \
    time_it([&] { hipLaunchKernelGGL((k_aes_bitslice<WPE>), dim3(GRID), dim3(64), 0, 0, reinterpret_cast<uint4*>(b), bk, counter0, ngroups); }, name); \
    check(name);                                                                                                       \
  }
  RUN_BS(2, 256 * 4 * 2, "bit-sliced, 2 waves per SIMD (<= 256 registers)")
  RUN_BS(3, 256 * 4 * 3, "bit-sliced, 3 waves per SIMD (<= 168 registers)")
  RUN_BS(4, 256 * 4 * 4, "bit-sliced, 4 waves per SIMD (<= 128 registers)")
  return 0;
}
