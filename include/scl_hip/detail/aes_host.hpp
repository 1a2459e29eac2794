// include/scl_hip/detail/aes_host.hpp -- AES-128 on the host, for the PRG draws that are too small to be worth a kernel
// launch (util::PRG::next below scl::hip::prgHostBytes()).
//
// The reference's PRG (src/scl/util/prg.cc:28-146) is AES-128 with the key = the seed and block i =
// AES(LE64(counter_i) || LE64(0x0123456789ABCDEF)); the batch paths draw those blocks on the GPU (csrc/kernels.hpp,
// k_prg_blocks and the fused sharing kernels).  A per-secret call of the reference's API draws two to four blocks: here
// they come from this file -- the AES-NI instructions where the CPU has them (what prg.cc uses), a byte-wise FIPS-197
// implementation otherwise -- so that such a call costs what the reference's does instead of a device round trip.
#ifndef SCL_HIP_DETAIL_AES_HOST_HPP
#define SCL_HIP_DETAIL_AES_HOST_HPP

#include <cstddef>
#include <cstdint>
#include <cstring>

#if defined(__x86_64__) && !defined(SCL_HIP_NO_AESNI)
#include <immintrin.h>
#endif

namespace scl::detail {

class Aes128Host {
 public:
  explicit Aes128Host(const unsigned char key[16]) {
    // FIPS-197 key expansion: w[i] = w[i-4] ^ (i % 4 == 0 ? SubWord(RotWord(w[i-1])) ^ rcon : w[i-1])
    std::memcpy(m_rk[0], key, 16);
    unsigned char rcon = 1;
    for (int r = 1; r <= 10; ++r) {
      const unsigned char* p = m_rk[r - 1];
      unsigned char* q = m_rk[r];
      q[0] = p[0] ^ sbox(p[13]) ^ rcon;
      q[1] = p[1] ^ sbox(p[14]);
      q[2] = p[2] ^ sbox(p[15]);
      q[3] = p[3] ^ sbox(p[12]);
      for (int i = 4; i < 16; ++i) q[i] = p[i] ^ q[i - 4];
      rcon = xtime(rcon);
    }
#if defined(__x86_64__) && !defined(SCL_HIP_NO_AESNI)
    m_ni = __builtin_cpu_supports("aes") && __builtin_cpu_supports("sse2");
#endif
  }

  void encrypt(const unsigned char in[16], unsigned char out[16]) const {
#if defined(__x86_64__) && !defined(SCL_HIP_NO_AESNI)
    if (m_ni) return encryptNi(in, out);
#endif
    unsigned char s[16];
    for (int i = 0; i < 16; ++i) s[i] = in[i] ^ m_rk[0][i];
    for (int r = 1; r <= 10; ++r) {
      unsigned char t[16];
      // SubBytes + ShiftRows: state is column-major, row i of column c at s[4c + i]; row i rotates left by i columns
      for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 4; ++i) t[4 * c + i] = sbox(s[4 * ((c + i) & 3) + i]);
      if (r < 10) {
        for (int c = 0; c < 4; ++c) {  // MixColumns: (2 3 1 1 / 1 2 3 1 / 1 1 2 3 / 3 1 1 2)
          const unsigned char a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
          const unsigned char all = a0 ^ a1 ^ a2 ^ a3;
          s[4 * c] = a0 ^ all ^ xtime(a0 ^ a1);
          s[4 * c + 1] = a1 ^ all ^ xtime(a1 ^ a2);
          s[4 * c + 2] = a2 ^ all ^ xtime(a2 ^ a3);
          s[4 * c + 3] = a3 ^ all ^ xtime(a3 ^ a0);
        }
      } else {
        std::memcpy(s, t, 16);
      }
      for (int i = 0; i < 16; ++i) s[i] ^= m_rk[r][i];
    }
    std::memcpy(out, s, 16);
  }

  /// blocks [counter, counter + nblocks) of the reference's PRG stream (prg.cc:82-84, 124-146)
  void prgBlocks(unsigned char* dst, std::size_t nblocks, std::uint64_t counter) const {
    for (std::size_t b = 0; b < nblocks; ++b) {
      unsigned char in[16];
      const std::uint64_t c = counter + b, nonce = 0x0123456789ABCDEFull;
      for (int i = 0; i < 8; ++i) {
        in[i] = (unsigned char)(c >> (8 * i));
        in[8 + i] = (unsigned char)(nonce >> (8 * i));
      }
      encrypt(in, dst + 16 * b);
    }
  }

 private:
  static unsigned char xtime(unsigned char a) { return (unsigned char)((a << 1) ^ ((a & 0x80) ? 0x1B : 0)); }
  static unsigned char gmul(unsigned char a, unsigned char b) {
    unsigned char r = 0;
    for (int i = 0; i < 8; ++i) {
      if (b & 1) r ^= a;
      a = xtime(a);
      b >>= 1;
    }
    return r;
  }
  /// the S-box from its definition: multiplicative inverse in GF(2^8) mod x^8+x^4+x^3+x+1, then the affine map.
  /// NOT constant-time: the 256-byte table is indexed by key- and state-dependent bytes, so on a CPU without AES-NI (the only
  /// place this form runs; the AES-NI path above has no data-dependent memory access) cache timing can leak the PRG seed to a
  /// co-resident observer.  The reference itself requires AES-NI (src/scl/util/prg.cc uses _mm_aesenc_si128 unconditionally);
  /// this fallback exists so that the mirror's known answers can be checked on any build machine.
  static unsigned char sbox(unsigned char x) {
    static const Table t;
    return t.s[x];
  }
  struct Table {
    unsigned char s[256];
    Table() {
      for (int x = 0; x < 256; ++x) {
        unsigned char inv = 0;
        if (x) {  // x^254
          unsigned char p = (unsigned char)x, acc = 1;
          for (int e = 254; e; e >>= 1) {
            if (e & 1) acc = gmul(acc, p);
            p = gmul(p, p);
          }
          inv = acc;
        }
        unsigned char y = inv;
        for (int k = 1; k <= 4; ++k) y ^= (unsigned char)((inv << k) | (inv >> (8 - k)));
        s[x] = y ^ 0x63;
      }
    }
  };

#if defined(__x86_64__) && !defined(SCL_HIP_NO_AESNI)
  __attribute__((target("aes,sse2"))) void encryptNi(const unsigned char in[16], unsigned char out[16]) const {
    __m128i s = _mm_xor_si128(_mm_loadu_si128(reinterpret_cast<const __m128i*>(in)),
                              _mm_loadu_si128(reinterpret_cast<const __m128i*>(m_rk[0])));
    for (int r = 1; r < 10; ++r) s = _mm_aesenc_si128(s, _mm_loadu_si128(reinterpret_cast<const __m128i*>(m_rk[r])));
    s = _mm_aesenclast_si128(s, _mm_loadu_si128(reinterpret_cast<const __m128i*>(m_rk[10])));
    _mm_storeu_si128(reinterpret_cast<__m128i*>(out), s);
  }
  bool m_ni = false;
#endif
  unsigned char m_rk[11][16];
};

}  // namespace scl::detail

#endif
