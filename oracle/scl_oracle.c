/* oracle/scl_oracle.c -- TEST INFRASTRUCTURE, not product code.  See scl_oracle.h.
 *
 * Plain-C CPU restatement of the reference hot path.  Citations are relative
 * to /root/reference.  Nothing here is reachable from the product library
 * (secure-computation-library_amd/); it is loaded by tests/, smoke() and
 * bench.py's cpu_baseline leg only.
 */
#define _POSIX_C_SOURCE 200809L
#include "scl_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

typedef unsigned __int128 u128;
typedef __int128 i128;

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------ AES-128 */
/* FIPS-197 AES-128 encryption.  The reference uses the AES-NI instructions
 * (src/scl/util/prg.cc:34-80); the portable path below computes the same
 * function from the specification, the NI path uses the same instructions. */

static unsigned char g_sbox[256];
static int g_sbox_ready = 0;
static int g_force_ni = -1;

static unsigned char gf256_mul(unsigned char a, unsigned char b) {
  unsigned char r = 0;
  while (b) {
    if (b & 1) r ^= a;
    a = (unsigned char)((a << 1) ^ ((a & 0x80) ? 0x1b : 0));
    b >>= 1;
  }
  return r;
}

static void sbox_init(void) {
  if (g_sbox_ready) return;
  for (int x = 0; x < 256; ++x) {
    /* multiplicative inverse in GF(2^8) (0 -> 0) by exhaustive search, then the affine map */
    unsigned char inv = 0;
    if (x) {
      for (int y = 1; y < 256; ++y)
        if (gf256_mul((unsigned char)x, (unsigned char)y) == 1) { inv = (unsigned char)y; break; }
    }
    unsigned char s = inv, r = inv;
    for (int i = 0; i < 4; ++i) {
      s = (unsigned char)((s << 1) | (s >> 7));
      r ^= s;
    }
    g_sbox[x] = (unsigned char)(r ^ 0x63);
  }
  g_sbox_ready = 1;
}

/* round keys as 11 x 16 bytes, column-major state order = byte order in memory */
static void aes128_expand(const unsigned char key[16], unsigned char rk[176]) {
  sbox_init();
  memcpy(rk, key, 16);
  unsigned char rcon = 1;
  for (int i = 16; i < 176; i += 4) {
    unsigned char t[4] = {rk[i - 4], rk[i - 3], rk[i - 2], rk[i - 1]};
    if (i % 16 == 0) {
      unsigned char t0 = t[0];
      t[0] = (unsigned char)(g_sbox[t[1]] ^ rcon);
      t[1] = g_sbox[t[2]];
      t[2] = g_sbox[t[3]];
      t[3] = g_sbox[t0];
      rcon = (unsigned char)((rcon << 1) ^ ((rcon & 0x80) ? 0x1b : 0));
    }
    for (int j = 0; j < 4; ++j) rk[i + j] = (unsigned char)(rk[i - 16 + j] ^ t[j]);
  }
}

static void aes128_encrypt_c(const unsigned char rk[176], const unsigned char in[16],
                             unsigned char out[16]) {
  unsigned char s[16], t[16];
  for (int i = 0; i < 16; ++i) s[i] = (unsigned char)(in[i] ^ rk[i]);
  for (int round = 1; round <= 10; ++round) {
    /* SubBytes + ShiftRows: byte (row r, col c) lives at index 4c+r */
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 4; ++r) t[4 * c + r] = g_sbox[s[4 * ((c + r) & 3) + r]];
    if (round < 10) {
      for (int c = 0; c < 4; ++c) {
        unsigned char a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
        s[4 * c + 0] = (unsigned char)(gf256_mul(a0, 2) ^ gf256_mul(a1, 3) ^ a2 ^ a3);
        s[4 * c + 1] = (unsigned char)(a0 ^ gf256_mul(a1, 2) ^ gf256_mul(a2, 3) ^ a3);
        s[4 * c + 2] = (unsigned char)(a0 ^ a1 ^ gf256_mul(a2, 2) ^ gf256_mul(a3, 3));
        s[4 * c + 3] = (unsigned char)(gf256_mul(a0, 3) ^ a1 ^ a2 ^ gf256_mul(a3, 2));
      }
    } else {
      memcpy(s, t, 16);
    }
    for (int i = 0; i < 16; ++i) s[i] ^= rk[16 * round + i];
  }
  memcpy(out, s, 16);
}

#if defined(__x86_64__)
__attribute__((target("aes,sse2"))) static void aes128_encrypt_ni(const unsigned char rk[176],
                                                                  const unsigned char in[16],
                                                                  unsigned char out[16]) {
  const __m128i* k = (const __m128i*)rk;
  __m128i m = _mm_loadu_si128((const __m128i*)in);
  m = _mm_xor_si128(m, _mm_loadu_si128(k));
  for (int r = 1; r < 10; ++r) m = _mm_aesenc_si128(m, _mm_loadu_si128(k + r));
  m = _mm_aesenclast_si128(m, _mm_loadu_si128(k + 10));
  _mm_storeu_si128((__m128i*)out, m);
}
static int have_ni(void) {
  if (g_force_ni == 0) return 0;
  return __builtin_cpu_supports("aes");
}
#else
static int have_ni(void) { return 0; }
#endif

void sclo_aes_force(int use_aesni) { g_force_ni = use_aesni; }

/* ---------------------------------------------------------------------- PRG */
/* util::PRG (include/scl/util/prg.h:64-173, src/scl/util/prg.cc:88-146):
 *   key   = seed zero-padded / truncated to 16 bytes        (prg.cc:88-101)
 *   block = AES_key( LE64(counter) || LE64(PRG_NONCE) ), counter from 0
 *           (prg.h:34-43, prg.cc:82-84: _mm_set_epi64x(NONCE, counter))
 *   next(buf,n) emits ceil(n/16) whole blocks, keeps the first n bytes and
 *   discards the rest; nothing is buffered across calls     (prg.cc:124-146) */
#define SCLO_PRG_NONCE 0x0123456789ABCDEFULL

typedef struct {
  unsigned char rk[176];
  uint64_t counter;
  int ni;
} sclo_prg_t;

static void prg_init(sclo_prg_t* g, const unsigned char* seed, size_t seed_len) {
  unsigned char key[16] = {0};
  if (seed) memcpy(key, seed, seed_len > 16 ? 16 : seed_len);
  aes128_expand(key, g->rk);
  g->counter = 0;
  g->ni = have_ni();
}

static void prg_block(const sclo_prg_t* g, uint64_t counter, unsigned char out[16]) {
  unsigned char in[16];
  const uint64_t nonce = SCLO_PRG_NONCE;
  memcpy(in, &counter, 8); /* little-endian host, as the reference assumes */
  memcpy(in + 8, &nonce, 8);
#if defined(__x86_64__)
  if (g->ni) { aes128_encrypt_ni(g->rk, in, out); return; }
#endif
  aes128_encrypt_c(g->rk, in, out);
}

static void prg_next(sclo_prg_t* g, unsigned char* buf, size_t n) {
  if (n == 0) return;
  size_t nblocks = (n + 15) / 16;
  unsigned char blk[16];
  for (size_t i = 0; i < nblocks; ++i) {
    prg_block(g, g->counter++, blk);
    size_t off = 16 * i;
    memcpy(buf + off, blk, n - off < 16 ? n - off : 16);
  }
}

int sclo_prg(const unsigned char* seed, size_t seed_len, const size_t* sizes, size_t ncalls,
             unsigned char* out) {
  sclo_prg_t g;
  prg_init(&g, seed, seed_len);
  for (size_t i = 0; i < ncalls; ++i) {
    prg_next(&g, out, sizes[i]);
    out += sizes[i];
  }
  return SCLO_OK;
}

int sclo_prg_blocks(const unsigned char* seed, size_t seed_len, uint64_t counter0, size_t nblocks,
                    unsigned char* out) {
  sclo_prg_t g;
  prg_init(&g, seed, seed_len);
  for (size_t i = 0; i < nblocks; ++i) prg_block(&g, counter0 + i, out + 16 * i);
  return SCLO_OK;
}

/* ------------------------------------------------------------- hex helpers */
/* util::fromHexString<T> (include/scl/util/str.h:49-76): big-endian hex, no 0x,
 * odd length / bad character are errors, bits shifted past the word are lost. */
static int hex_parse(const char* s, u128* out, int bits) {
  size_t n = strlen(s);
  if (n % 2) return SCLO_BAD_HEX_LEN;
  u128 t = 0;
  for (size_t i = 0; i < n; ++i) {
    char c = s[i];
    unsigned v;
    if (c >= '0' && c <= '9') v = (unsigned)(c - '0');
    else if (c >= 'a' && c <= 'f') v = (unsigned)(c - 'a' + 10);
    else if (c >= 'A' && c <= 'F') v = (unsigned)(c - 'A' + 10);
    else return SCLO_BAD_HEX_CHAR;
    t = (t << 4) + v;
    if (bits == 64) t &= (u128)0xFFFFFFFFFFFFFFFFULL;
  }
  *out = t;
  return SCLO_OK;
}

static int hex_u64(uint64_t v, char* out, size_t outlen, int pad16) {
  char tmp[17];
  int len = 0;
  if (pad16) {
    for (int i = 15; i >= 0; --i) tmp[len++] = "0123456789abcdef"[(v >> (4 * i)) & 15];
  } else {
    int started = 0;
    for (int i = 15; i >= 0; --i) {
      unsigned d = (unsigned)((v >> (4 * i)) & 15);
      if (d || started || i == 0) { tmp[len++] = "0123456789abcdef"[d]; started = 1; }
    }
  }
  if ((size_t)len + 1 > outlen) return -1;
  memcpy(out, tmp, (size_t)len);
  out[len] = 0;
  return len;
}

/* =============================================================== Mersenne61 */
/* include/scl/math/fields/mersenne61.h:29-49, src/scl/math/fields/mersenne61.cc:33-100,
 * helpers src/scl/math/fields/small_ff.h:28-92 */
#define P61 0x1FFFFFFFFFFFFFFFULL

static inline uint64_t m61_add(uint64_t a, uint64_t b) { /* small_ff.h:29-34 */
  uint64_t t = a + b;
  return t >= P61 ? t - P61 : t;
}
static inline uint64_t m61_sub(uint64_t a, uint64_t b) { /* small_ff.h:40-46 */
  return b > a ? a + P61 - b : a - b;
}
static inline uint64_t m61_neg(uint64_t a) { return a ? P61 - a : 0; } /* small_ff.h:52-56 */
static inline uint64_t m61_mul(uint64_t x, uint64_t y) { /* mersenne61.cc:58-69 */
  u128 z = (u128)x * y;
  uint64_t a = (uint64_t)(z >> 61);
  uint64_t b = (uint64_t)z & P61;
  return m61_add(a, b);
}
static inline int m61_inv(uint64_t* out, uint64_t v) { /* small_ff.h:61-92, S = int64 */
  if (v == 0) return SCLO_ZERO_INVERSE;
  int64_t k = 0, nk = 1, r = (int64_t)P61, nr = (int64_t)v;
  while (nr != 0) {
    int64_t q = r / nr, tmp;
    tmp = nk; nk = k - q * tmp; k = tmp;
    tmp = nr; nr = r - q * tmp; r = tmp;
  }
  if (k < 0) k += (int64_t)P61;
  *out = (uint64_t)k;
  return SCLO_OK;
}
static inline uint64_t m61_from_int(int v) { /* mersenne61.cc:37-40 */
  return v < 0 ? (uint64_t)((int64_t)v + (int64_t)P61) : (uint64_t)v;
}
static inline uint64_t m61_from_bytes(const unsigned char* src) { /* mersenne61.cc:86-90 */
  uint64_t w;
  memcpy(&w, src, 8);
  return w % P61;
}
static inline uint64_t m61_ld(const uint64_t* p) { return p[0]; }
static inline void m61_st(uint64_t* p, uint64_t v) { p[0] = v; }
static inline int m61_is_zero(uint64_t v) { return v == 0; }

#define FE uint64_t
#define FN(name) m61_##name
#define LIMBS 1
#include "scl_oracle_generic.inc"
#undef FE
#undef FN
#undef LIMBS

/* ============================================================== Mersenne127 */
/* include/scl/math/fields/mersenne127.h:29-49, src/scl/math/fields/mersenne127.cc:33-128 */
#define P127 ((((u128)0x7FFFFFFFFFFFFFFFULL) << 64) | (u128)0xFFFFFFFFFFFFFFFFULL)

static inline u128 m127_add(u128 a, u128 b) {
  u128 t = a + b;
  return t >= P127 ? t - P127 : t;
}
static inline u128 m127_sub(u128 a, u128 b) { return b > a ? a + P127 - b : a - b; }
static inline u128 m127_neg(u128 a) { return a ? P127 - a : 0; }

/* 128x128 -> 256 by four 64x64 products (mersenne127.cc:66-83) */
static inline void mul_wide(u128 x, u128 y, u128* hi, u128* lo) {
  uint64_t a = (uint64_t)(x >> 64), b = (uint64_t)x, c = (uint64_t)(y >> 64), d = (uint64_t)y;
  u128 ac = (u128)a * c, ad = (u128)a * d, bc = (u128)b * c, bd = (u128)b * d;
  u128 carry = (u128)(uint64_t)ad + (u128)(uint64_t)bc + (bd >> 64);
  *hi = ac + (ad >> 64) + (bc >> 64) + (carry >> 64);
  *lo = (ad << 64) + (bc << 64) + bd;
}
static inline u128 m127_mul(u128 x, u128 y) { /* mersenne127.cc:87-97 */
  u128 hi, lo;
  mul_wide(x, y, &hi, &lo);
  u128 a = (hi << 1) | (lo >> 127);
  u128 b = lo & P127;
  return m127_add(a, b);
}
static inline int m127_inv(u128* out, u128 v) { /* small_ff.h:61-92, S = __int128 */
  if (v == 0) return SCLO_ZERO_INVERSE;
  i128 k = 0, nk = 1, r = (i128)P127, nr = (i128)v;
  while (nr != 0) {
    i128 q = r / nr, tmp;
    tmp = nk; nk = k - q * tmp; k = tmp;
    tmp = nr; nr = r - q * tmp; r = tmp;
  }
  if (k < 0) k += (i128)P127;
  *out = (u128)k;
  return SCLO_OK;
}
static inline u128 m127_from_int(int v) { /* mersenne127.cc:37-40 */
  return v < 0 ? (u128)((i128)v + (i128)P127) : (u128)v;
}
static inline u128 m127_from_bytes(const unsigned char* src) { /* mersenne127.cc:114-118 */
  u128 w;
  memcpy(&w, src, 16);
  return w % P127;
}
static inline u128 m127_ld(const uint64_t* p) { return ((u128)p[1] << 64) | p[0]; }
static inline void m127_st(uint64_t* p, u128 v) { p[0] = (uint64_t)v; p[1] = (uint64_t)(v >> 64); }
static inline int m127_is_zero(u128 v) { return v == 0; }

#define FE u128
#define FN(name) m127_##name
#define LIMBS 2
#include "scl_oracle_generic.inc"
#undef FE
#undef FN
#undef LIMBS

/* ================================================================= MONT128 */
/* PARITY UNPINNED plug-in field: a generic odd prime p < 2^128 with elements
 * held in Montgomery form x*R mod p, R = 2^128 -- the N=2-limb analogue of the
 * reference's only Montgomery code, include/scl/math/fields/ff_ops_gmp.h:44-260
 * (instantiated there for the 256-bit secp256k1 fields only). */
static u128 g_mp = 0, g_mmc = 0, g_mr2 = 0, g_mone = 0;

static inline u128 mont_reduce_once(u128 t, int carry) { /* ff_ops_gmp.h:96-99 */
  return (carry || t >= g_mp) ? t - g_mp : t;
}
static inline u128 mont128_add(u128 a, u128 b) { /* montyModAdd, ff_ops_gmp.h:128-134 */
  u128 t = a + b;
  return mont_reduce_once(t, t < a);
}
static inline u128 mont128_sub(u128 a, u128 b) { /* montyModSub, ff_ops_gmp.h:142-148 */
  return a >= b ? a - b : a - b + g_mp;
}
static inline u128 mont128_neg(u128 a) { return a ? g_mp - a : 0; } /* ff_ops_gmp.h:156-162 */
/* montyModMul (ff_ops_gmp.h:174-191) computes a*b/R mod p; any correct REDC yields
 * the same unique residue.  t = (a*b + ((a*b mod R)*mc mod R)*p) / R */
static inline u128 mont128_mul(u128 a, u128 b) {
  u128 thi, tlo, mhi, mlo;
  mul_wide(a, b, &thi, &tlo);
  u128 m = tlo * g_mmc;
  mul_wide(m, g_mp, &mhi, &mlo);
  u128 lo = tlo + mlo; /* == 0 mod 2^128 by construction */
  u128 c0 = lo < tlo;
  u128 hi = thi + mhi;
  int carry = hi < thi;
  u128 hi2 = hi + c0;
  carry |= hi2 < hi;
  return mont_reduce_once(hi2, carry);
}
static inline u128 mont128_pow(u128 base, u128 e) { /* montyModExp, ff_ops_gmp.h:225-237 */
  u128 r = g_mone;
  int started = 0;
  for (int i = 127; i >= 0; --i) {
    if (started) r = mont128_mul(r, r);
    if ((e >> i) & 1) { r = mont128_mul(r, base); started = 1; }
  }
  return r;
}
static inline int mont128_inv(u128* out, u128 v) { /* montyModInv: Fermat, ff_ops_gmp.h:250-260 */
  if (v == 0) return SCLO_ZERO_INVERSE;
  *out = mont128_pow(v, g_mp - 2);
  return SCLO_OK;
}
static inline u128 mont128_to_mont(u128 x) { return mont128_mul(x, g_mr2); } /* montyIn */
static inline u128 mont128_from_mont(u128 x) { return mont128_mul(x, 1); }   /* montyRedc */
static inline u128 mont128_from_int(int v) { /* montyInFromInt, ff_ops_gmp.h:108-114 */
  u128 x = (u128)(v < 0 ? -(int64_t)v : (int64_t)v);
  if (v < 0) x = g_mp - x;
  return mont128_to_mont(x);
}
static inline u128 mont128_from_bytes(const unsigned char* src) { /* montyFromBytes: BIG-endian, ff_ops_gmp.h:279-290 */
  u128 x = 0;
  for (int i = 0; i < 16; ++i) x = (x << 8) | src[i];
  return mont128_to_mont(x);
}
static inline u128 mont128_ld(const uint64_t* p) { return ((u128)p[1] << 64) | p[0]; }
static inline void mont128_st(uint64_t* p, u128 v) { p[0] = (uint64_t)v; p[1] = (uint64_t)(v >> 64); }
static inline int mont128_is_zero(u128 v) { return v == 0; }

int sclo_mont128_set_prime(const uint64_t p[2]) {
  u128 pp = ((u128)p[1] << 64) | p[0];
  if (!(pp & 1) || pp < 3) return SCLO_BAD_ARG;
  g_mp = pp;
  /* mc = -p^{-1} mod 2^128 by Newton iteration (5 doublings from 3 correct bits... 7 to be safe) */
  u128 inv = pp; /* correct to 3 bits for odd p */
  for (int i = 0; i < 7; ++i) inv *= 2 - pp * inv;
  g_mmc = (u128)0 - inv;
  /* R mod p and R^2 mod p by repeated doubling */
  u128 r = 1 % pp;
  for (int i = 0; i < 256; ++i) {
    u128 t = r + r;
    int c = t < r;
    r = (c || t >= pp) ? t - pp : t;
    if (i == 127) g_mone = r;
  }
  g_mr2 = r;
  return SCLO_OK;
}
void sclo_mont128_get_prime(uint64_t p[2]) { p[0] = (uint64_t)g_mp; p[1] = (uint64_t)(g_mp >> 64); }

static void mont128_ensure(void) {
  if (g_mp == 0) {
    /* default p = 2^128 - 159 (the largest prime below 2^128) */
    const uint64_t p[2] = {0xFFFFFFFFFFFFFF61ULL, 0xFFFFFFFFFFFFFFFFULL};
    sclo_mont128_set_prime(p);
  }
}

#define FE u128
#define FN(name) mont128_##name
#define LIMBS 2
#include "scl_oracle_generic.inc"
#undef FE
#undef FN
#undef LIMBS

/* ================================================================ GF(2^128) */
/* PARITY UNPINNED plug-in field (absent from the reference):
 * GF(2)[x] / (x^128 + x^7 + x^2 + x + 1), bit i of the little-endian 128-bit
 * word is the coefficient of x^i.  Bitwise shift-xor multiplication. */
static inline u128 gf128_add(u128 a, u128 b) { return a ^ b; }
static inline u128 gf128_sub(u128 a, u128 b) { return a ^ b; }
static inline u128 gf128_neg(u128 a) { return a; }
static inline u128 gf128_mul(u128 a, u128 b) {
  u128 r = 0;
  for (int i = 0; i < 128; ++i) {
    if ((b >> i) & 1) r ^= a;
    int top = (int)(a >> 127);
    a <<= 1;
    if (top) a ^= 0x87;
  }
  return r;
}
static inline int gf128_inv(u128* out, u128 v) { /* a^(2^128-2) */
  if (v == 0) return SCLO_ZERO_INVERSE;
  u128 r = 1, sq = v;
  for (int i = 1; i < 128; ++i) { /* exponent bits 1..127 set */
    sq = gf128_mul(sq, sq);
    r = gf128_mul(r, sq);
  }
  *out = r;
  return SCLO_OK;
}
static inline u128 gf128_from_int(int v) { return (u128)(v < 0 ? -(int64_t)v : (int64_t)v); }
static inline u128 gf128_from_bytes(const unsigned char* src) {
  u128 w;
  memcpy(&w, src, 16);
  return w;
}
static inline u128 gf128_ld(const uint64_t* p) { return ((u128)p[1] << 64) | p[0]; }
static inline void gf128_st(uint64_t* p, u128 v) { p[0] = (uint64_t)v; p[1] = (uint64_t)(v >> 64); }
static inline int gf128_is_zero(u128 v) { return v == 0; }

#define FE u128
#define FN(name) gf128_##name
#define LIMBS 2
#include "scl_oracle_generic.inc"
#undef FE
#undef FN
#undef LIMBS

/* ======================================================== the two 256-bit Montgomery primes of secp256k1 */
/* FF<Secp256k1Scalar> (include/scl/math/fields/secp256k1_scalar.h, src/scl/math/fields/secp256k1_scalar.cc:47-135) and
 * FF<Secp256k1Field> (secp256k1_field.h, src/scl/math/fields/secp256k1_field.cc:43-135):
 * 4 x 64-bit limbs, values held in Montgomery form x*2^256 mod p, arithmetic from the mpn Montgomery family
 * include/scl/math/fields/ff_ops_gmp.h:44-314.  Prime p = the order of the secp256k1 group, or the prime the curve is
 * defined over (2^256 - 2^32 - 977).  One set of routines; secpq_select() points it at either prime and re-derives the
 * Montgomery constants from it (this oracle is single-threaded test infrastructure). */
typedef struct { uint64_t w[4]; } fe256;
static const uint64_t SQ_PRIMES[2][4] = {
    {0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL},   /* group order */
    {0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL}};  /* field prime */
static const uint64_t* SQ_P = SQ_PRIMES[0];
static uint64_t g_sq_mc0 = 0; /* -p^{-1} mod 2^64 (low limb of RedParams::mc, secp256k1_scalar.cc:62-67, secp256k1_field.cc:53-58) */
static fe256 g_sq_one, g_sq_r2;

static int sq_geq_p(const uint64_t a[4]) {
  for (int i = 3; i >= 0; --i) {
    if (a[i] > SQ_P[i]) return 1;
    if (a[i] < SQ_P[i]) return 0;
  }
  return 1;
}
static uint64_t sq_add_n(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  u128 c = 0;
  for (int i = 0; i < 4; ++i) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static uint64_t sq_sub_n(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}
static inline fe256 secpq_add(fe256 a, fe256 b) { /* montyModAdd, ff_ops_gmp.h:128-134 */
  fe256 r;
  uint64_t carry = sq_add_n(r.w, a.w, b.w);
  if (carry || sq_geq_p(r.w)) sq_sub_n(r.w, r.w, SQ_P);
  return r;
}
static inline fe256 secpq_sub(fe256 a, fe256 b) { /* montyModSub, ff_ops_gmp.h:142-148 */
  fe256 r;
  if (sq_sub_n(r.w, a.w, b.w)) sq_add_n(r.w, r.w, SQ_P);
  return r;
}
static inline fe256 secpq_neg(fe256 a) { /* montyModNeg: 0 - a, ff_ops_gmp.h:156-162 */
  fe256 z = {{0, 0, 0, 0}};
  return secpq_sub(z, a);
}
/* montyModMul, ff_ops_gmp.h:174-191: interleaved (CIOS) multiplication, a*b/2^256 mod p */
static inline fe256 secpq_mul(fe256 a, fe256 b) {
  uint64_t u[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) { c += (u128)a.w[j] * b.w[i] + u[j]; u[j] = (uint64_t)c; c >>= 64; }
    c += u[4]; u[4] = (uint64_t)c; u[5] = (uint64_t)(c >> 64);
    const uint64_t q = g_sq_mc0 * u[0];
    c = 0;
    for (int j = 0; j < 4; ++j) { c += (u128)q * SQ_P[j] + u[j]; u[j] = (uint64_t)c; c >>= 64; }
    c += u[4]; u[4] = (uint64_t)c; u[5] += (uint64_t)(c >> 64);
    for (int j = 0; j < 5; ++j) u[j] = u[j + 1];
    u[5] = 0;
  }
  fe256 r = {{u[0], u[1], u[2], u[3]}};
  if (u[4] || sq_geq_p(r.w)) sq_sub_n(r.w, r.w, SQ_P);
  return r;
}
static inline int secpq_is_zero(fe256 v) { return (v.w[0] | v.w[1] | v.w[2] | v.w[3]) == 0; }
static inline int secpq_inv(fe256* out, fe256 v) { /* montyModInv: a^(p-2), ff_ops_gmp.h:225-260 */
  if (secpq_is_zero(v)) return SCLO_ZERO_INVERSE;
  uint64_t e[4] = {SQ_P[0] - 2, SQ_P[1], SQ_P[2], SQ_P[3]};
  fe256 r = g_sq_one;
  for (int i = 255; i >= 0; --i) {
    r = secpq_mul(r, r);
    if ((e[i / 64] >> (i % 64)) & 1) r = secpq_mul(r, v);
  }
  *out = r;
  return SCLO_OK;
}
static inline fe256 secpq_to_mont(fe256 x) { return secpq_mul(x, g_sq_r2); } /* montyIn: x*2^256 mod p */
static inline fe256 secpq_from_mont(fe256 x) { fe256 one = {{1, 0, 0, 0}}; return secpq_mul(x, one); } /* montyRedc */
static inline fe256 secpq_from_int(int v) { /* montyInFromInt, ff_ops_gmp.h:108-114 */
  fe256 x = {{(uint64_t)(v < 0 ? -(int64_t)v : (int64_t)v), 0, 0, 0}};
  if (v < 0) { fe256 pp = {{SQ_P[0], SQ_P[1], SQ_P[2], SQ_P[3]}}; sq_sub_n(x.w, pp.w, x.w); }
  return secpq_to_mont(x);
}
static inline fe256 secpq_from_bytes(const unsigned char* src) { /* montyFromBytes: big-endian, ff_ops_gmp.h:279-290 */
  fe256 x;
  for (int i = 3; i >= 0; --i) {
    uint64_t w = 0;
    for (int j = 0; j < 8; ++j) w = (w << 8) | *src++;
    x.w[i] = w;
  }
  return secpq_to_mont(x);
}
static inline fe256 secpq_ld(const uint64_t* p) { fe256 r = {{p[0], p[1], p[2], p[3]}}; return r; }
static inline void secpq_st(uint64_t* p, fe256 v) { memcpy(p, v.w, 32); }

static void secpq_ensure(void) {
  if (g_sq_mc0) return;
  uint64_t inv = SQ_P[0]; /* Newton: -p^{-1} mod 2^64 */
  for (int i = 0; i < 6; ++i) inv *= 2 - SQ_P[0] * inv;
  g_sq_mc0 = (uint64_t)0 - inv;
  /* R mod p and R^2 mod p by 512 modular doublings of 1 */
  fe256 r = {{1, 0, 0, 0}};
  for (int i = 0; i < 512; ++i) {
    fe256 t;
    uint64_t c = sq_add_n(t.w, r.w, r.w);
    if (c || sq_geq_p(t.w)) sq_sub_n(t.w, t.w, SQ_P);
    r = t;
    if (i == 255) g_sq_one = r;
  }
  g_sq_r2 = r;
}

static void secpq_select(int field) { /* SCLO_SECP256K1_SCALAR or SCLO_SECP256K1_FIELD */
  const uint64_t* want = SQ_PRIMES[field == SCLO_SECP256K1_FIELD ? 1 : 0];
  if (want != SQ_P) {
    SQ_P = want;
    g_sq_mc0 = 0;
  }
  secpq_ensure();
}
#define IS_SECP256(field) ((field) == SCLO_SECP256K1_SCALAR || (field) == SCLO_SECP256K1_FIELD)

#define FE fe256
#define FN(name) secpq_##name
#define LIMBS 4
#include "scl_oracle_generic.inc"
#undef FE
#undef FN
#undef LIMBS

/* ================================================================ dispatch */
/* ------------------------------------------------------------------------------------------------
 * Rings Z2k<K>, tag SCLO_Z2K(K) = 0x100 + K, 1 <= K <= 128 (include/scl/math/z2k.h:39-320,
 * include/scl/math/z2k/z2k_ops.h:32-150).  One limb for K <= 64, two above, like Z2k::ValueType
 * (z2k.h:44-45).  The reference computes on the whole word and masks only when a value is compared,
 * written or printed (z2k_ops.h:97-141); every result here is stored masked, which is the same residue
 * mod 2^K.  byteSize = (K-1)/8 + 1 (z2k.h:50-52) is the stride of read / Vector::random.
 * The reference has no Serializer for Z2k, so the wire functions refuse ring tags. */
#define SCLO_IS_RING(field) ((field) > 0x100 && (field) <= 0x100 + 128)
#define RING_K(field) ((field)-0x100)

static u128 ring_mask(int K) { return K >= 128 ? ~(u128)0 : (((u128)1 << K) - 1); }
static size_t ring_limbs(int K) { return K <= 64 ? 1 : 2; }
static size_t ring_bytes(int K) { return (size_t)(K - 1) / 8 + 1; }
static u128 ring_ld(const uint64_t* p, int K) { return K <= 64 ? (u128)p[0] : (((u128)p[1] << 64) | p[0]); }
static void ring_st(uint64_t* p, int K, u128 v) {
  v &= ring_mask(K);
  p[0] = (uint64_t)v;
  if (K > 64) p[1] = (uint64_t)(v >> 64);
}
/* fromBytes: load the word, mask (z2k_ops.h:107-112); only the first byteSize bytes can matter */
static u128 ring_from_bytes(const unsigned char* src, int K) {
  u128 v = 0;
  for (size_t b = 0; b < ring_bytes(K); ++b) v |= (u128)src[b] << (8 * b);
  return v & ring_mask(K);
}
/* invert (z2k_ops.h:80-93): odd values only; Newton steps doubling the correct bits from 5 */
static int ring_inv(u128* out, u128 v, int K) {
  if (!(v & 1)) return SCLO_NOT_INVERTIBLE_2K;
  const int wide = K > 64;
  u128 z = (v * 3) ^ 2;
  for (size_t bits = 5; bits <= (size_t)K; bits *= 2) {
    z *= 2 - v * z;
    if (!wide) z = (uint64_t)z; /* the reference's word is 64 bits here */
  }
  *out = z;
  return SCLO_OK;
}

static int ring_ew(int K, int op, uint64_t* dst, const uint64_t* a, const uint64_t* b, size_t n) {
  const size_t L = ring_limbs(K);
  for (size_t i = 0; i < n; ++i) {
    u128 x = ring_ld(a + i * L, K), y = b ? ring_ld(b + i * L, K) : 0, t;
    int st = SCLO_OK;
    switch (op) {
      case SCLO_ADD: x += y; break;
      case SCLO_SUB: x -= y; break;
      case SCLO_MUL: x *= y; break;
      case SCLO_NEG: x = (u128)0 - x; break;
      case SCLO_INV: st = ring_inv(&x, x, K); break;
      case SCLO_DIV: st = ring_inv(&t, y, K); x *= t; break; /* operator/= (z2k.h:183-186) */
      default: return SCLO_BAD_ARG;
    }
    if (st) return st;
    ring_st(dst + i * L, K, x);
  }
  return SCLO_OK;
}

static int ring_from_bytes_n(int K, const unsigned char* src, size_t n, uint64_t* dst) {
  for (size_t i = 0; i < n; ++i) ring_st(dst + i * ring_limbs(K), K, ring_from_bytes(src + i * ring_bytes(K), K));
  return SCLO_OK;
}

/* Vector::random (vector.h:507-519): ONE prg.next(n * byteSize), then read at stride byteSize */
static int ring_vector_random(int K, const unsigned char* seed, size_t seed_len, size_t n, uint64_t* out) {
  sclo_prg_t g;
  prg_init(&g, seed, seed_len);
  unsigned char* buf = (unsigned char*)malloc(n * ring_bytes(K) + 16);
  if (!buf) return SCLO_BAD_ARG;
  prg_next(&g, buf, n * ring_bytes(K));
  ring_from_bytes_n(K, buf, n, out);
  free(buf);
  return SCLO_OK;
}

/* additiveShare (additive.h:41-53) per secret on one PRG: n-1 draws of T::random (one prg.next(byteSize)
 * each, i.e. one AES block), last = secret - sum */
static int ring_additive_share(int K, const unsigned char* seed, size_t seed_len, const uint64_t* secrets,
                               size_t N, size_t n, uint64_t* shares) {
  const size_t L = ring_limbs(K);
  if (n == 0) return SCLO_BAD_ARG;
  sclo_prg_t g;
  prg_init(&g, seed, seed_len);
  for (size_t s = 0; s < N; ++s) {
    u128 sum = 0;
    for (size_t i = 0; i + 1 < n; ++i) {
      unsigned char buf[16];
      prg_next(&g, buf, ring_bytes(K));
      const u128 r = ring_from_bytes(buf, K);
      ring_st(shares + (s * n + i) * L, K, r);
      sum += r;
    }
    ring_st(shares + (s * n + n - 1) * L, K, ring_ld(secrets + s * L, K) - sum);
  }
  return SCLO_OK;
}

static int ring_additive_recover(int K, const uint64_t* shares, size_t n, size_t N, uint64_t* out) {
  const size_t L = ring_limbs(K);
  for (size_t s = 0; s < N; ++s) {
    u128 sum = 0;
    for (size_t i = 0; i < n; ++i) sum += ring_ld(shares + (s * n + i) * L, K);
    ring_st(out + s * L, K, sum);
  }
  return SCLO_OK;
}

static int ring_dot(int K, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  const size_t L = ring_limbs(K);
  u128 acc = 0;
  for (size_t i = 0; i < n; ++i) acc += ring_ld(a + i * L, K) * (b ? ring_ld(b + i * L, K) : 1);
  ring_st(out, K, acc);
  return SCLO_OK;
}

static int ring_matmul(int K, const uint64_t* A, const uint64_t* B, size_t n, size_t k, size_t m, uint64_t* C) {
  const size_t L = ring_limbs(K);
  for (size_t i = 0; i < n; ++i)
    for (size_t j = 0; j < m; ++j) {
      u128 acc = 0;
      for (size_t l = 0; l < k; ++l) acc += ring_ld(A + (i * k + l) * L, K) * ring_ld(B + (l * m + j) * L, K);
      ring_st(C + (i * m + j) * L, K, acc);
    }
  return SCLO_OK;
}

int sclo_limbs(int field) {
  if (SCLO_IS_RING(field)) return (int)ring_limbs(RING_K(field));
  return field == SCLO_M61 ? 1 : (field >= 1 && field <= 3) ? 2 : IS_SECP256(field) ? 4 : -1;
}

const char* sclo_field_name(int field) {
  switch (field) {
    case SCLO_M61: return "Mersenne61";   /* mersenne61.h:38 */
    case SCLO_M127: return "Mersenne127"; /* mersenne127.h:38 */
    case SCLO_MONT128: return "Mont128";
    case SCLO_GF2_128: return "GF(2^128)";
    case SCLO_SECP256K1_SCALAR: return "secp256k1_order"; /* secp256k1_scalar.h NAME */
    case SCLO_SECP256K1_FIELD: return "secp256k1_field";  /* secp256k1_field.h NAME */
    default: return "";
  }
}

const char* sclo_status_message(int status) {
  switch (status) {
    case SCLO_OK: return "";
    case SCLO_ZERO_INVERSE: return "0 not invertible modulo prime"; /* small_ff.h:70 */
    case SCLO_BAD_HEX_LEN: return "odd-length hex string";          /* str.h:52 */
    case SCLO_BAD_HEX_CHAR: return "encountered invalid hex character"; /* str.h:38 */
    case SCLO_ERROR_DETECTED: return "error detected during recovery";  /* shamir.h:135 */
    case SCLO_NOT_INVERTIBLE_2K: return "value not invertible modulo 2^K"; /* z2k_ops.h:82 */
    default: return "bad argument";
  }
}

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
/* FIELD_SWITCH(field, BODY): BODY(P) is expanded once per field with P = its prefix */
#define FIELD_SWITCH(field, BODY)                          \
  switch (field) {                                         \
    case SCLO_M61: { BODY(m61_) } break;                   \
    case SCLO_M127: { BODY(m127_) } break;                 \
    case SCLO_MONT128: { mont128_ensure(); BODY(mont128_) } break; \
    case SCLO_GF2_128: { BODY(gf128_) } break;             \
    case SCLO_SECP256K1_SCALAR: { secpq_select(field); BODY(secpq_) } break; \
    case SCLO_SECP256K1_FIELD: { secpq_select(field); BODY(secpq_) } break; \
    default: break;                                        \
  }                                                        \
  return SCLO_BAD_ARG;

int sclo_ew(int field, int op, uint64_t* dst, const uint64_t* a, const uint64_t* b, size_t n) {
  if (SCLO_IS_RING(field)) return ring_ew(RING_K(field), op, dst, a, b, n);
#define BODY(P) return CAT(P, ew)(op, dst, a, b, n);
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_from_int(int field, int v, uint64_t* dst) {
  if (SCLO_IS_RING(field)) { /* Z2k(ValueType) from a sign-extended int */
    ring_st(dst, RING_K(field), (u128)(__int128)v);
    return SCLO_OK;
  }
#define BODY(P) CAT(P, st)(dst, CAT(P, from_int)(v)); return SCLO_OK;
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_from_bytes(int field, const unsigned char* src, size_t n, uint64_t* dst) {
  if (SCLO_IS_RING(field)) return ring_from_bytes_n(RING_K(field), src, n, dst);
  const size_t L = (size_t)sclo_limbs(field);
#define BODY(P) for (size_t i = 0; i < n; ++i) CAT(P, st)(dst + i * L, CAT(P, from_bytes)(src + i * L * 8)); return SCLO_OK;
  FIELD_SWITCH(field, BODY)
#undef BODY
}

/* convertTo(string): parse big-endian hex into the value type, then "% p"
 * (mersenne61.cc:42-46, mersenne127.cc:42-46) */
/* montyFromString (include/scl/math/fields/ff_ops_gmp.h:370-398): an odd-length string gets a leading
 * "0" (no error), more than 64 digits is an error, and the digits are cut into 16-character limbs FROM
 * THE LEFT -- the first chunk is the top limb (index (n-1)/16), a short last chunk becomes limb 0 as it
 * stands.  For lengths that are not a multiple of 16 this is not the usual big-endian reading; restated
 * as is.  The empty string leaves the element 0. */
static int hex_parse_limbs(const char* s, fe256* out, size_t nlimbs) {
  size_t n = strlen(s);
  fe256 t = {{0, 0, 0, 0}};
  *out = t;
  if (n == 0) return SCLO_OK;
  /* "hex string too large to parse" beyond 64 digits (ff_ops_gmp.h:377-379).  The template's bound is 64 whatever N is: at
   * N = 2 a string of 33 .. 64 digits makes it write limbs 2 and 3 of a two-limb value -- undefined behaviour in the
   * reference; refused here */
  if (n > 16 * nlimbs) return SCLO_BAD_ARG;
  char buf[66];
  if (n % 2) {
    buf[0] = '0';
    memcpy(buf + 1, s, n + 1);
    ++n;
  } else {
    memcpy(buf, s, n + 1);
  }
  int cidx = (int)((n - 1) / 16);
  for (size_t i = 0; i < n && cidx >= 0; i += 16) {
    size_t end = i + 16 < n ? i + 16 : n;
    uint64_t w = 0;
    for (size_t j = i; j < end; ++j) {
      char ch = buf[j];
      unsigned v;
      if (ch >= '0' && ch <= '9') v = (unsigned)(ch - '0');
      else if (ch >= 'a' && ch <= 'f') v = (unsigned)(ch - 'a' + 10);
      else if (ch >= 'A' && ch <= 'F') v = (unsigned)(ch - 'A' + 10);
      else return SCLO_BAD_HEX_CHAR;
      w = (w << 4) | v;
    }
    t.w[cidx--] = w;
  }
  *out = t;
  return SCLO_OK;
}
static int hex_parse256(const char* s, fe256* out) { return hex_parse_limbs(s, out, 4); }

int sclo_from_hex(int field, const char* hex, uint64_t* dst) {
  if (IS_SECP256(field)) {
    fe256 t;
    int st256 = hex_parse256(hex, &t);
    if (st256) return st256;
    secpq_select(field);
    secpq_st(dst, hex[0] ? secpq_to_mont(t) : t);
    return SCLO_OK;
  }
  if (field == SCLO_MONT128) { /* the Montgomery family's montyFromString at two limbs (pinned by oracle/_ref, field tag 2) */
    fe256 t;
    int st2 = hex_parse_limbs(hex, &t, 2);
    if (st2) return st2;
    mont128_ensure();
    const u128 raw = ((u128)t.w[1] << 64) | t.w[0];
    mont128_st(dst, hex[0] ? mont128_to_mont(raw) : raw);
    return SCLO_OK;
  }
  u128 v;
  int st = hex_parse(hex, &v, field == SCLO_M61 ? 64 : 128);
  if (st) return st;
  switch (field) {
    case SCLO_M61: dst[0] = (uint64_t)v % P61; return SCLO_OK;
    case SCLO_M127: m127_st(dst, v % P127); return SCLO_OK;
    case SCLO_GF2_128: gf128_st(dst, v); return SCLO_OK;
    default: return SCLO_BAD_ARG;
  }
}

/* toString: std::hex of the u64 (mersenne61.cc:97-100); for u128 the reference prints
 * the top word (if non-zero) then the low word WITHOUT zero padding
 * (src/scl/util/str.cc:23-39) -- reproduced as is. */
int sclo_to_hex(int field, const uint64_t* a, char* out, size_t outlen) {
  if (IS_SECP256(field)) { /* montyToString: value out of Montgomery form, hex without leading zeros */
    secpq_select(field);
    const fe256 v = secpq_from_mont(secpq_ld(a));
    char tmp[65];
    int len = 0, started = 0;
    for (int i = 3; i >= 0; --i)
      for (int nib = 15; nib >= 0; --nib) {
        unsigned d = (unsigned)((v.w[i] >> (4 * nib)) & 15);
        if (d || started || (i == 0 && nib == 0)) { tmp[len++] = "0123456789abcdef"[d]; started = 1; }
      }
    if ((size_t)len + 1 > outlen) return SCLO_BAD_ARG;
    memcpy(out, tmp, (size_t)len);
    out[len] = 0;
    return SCLO_OK;
  }
  if (field == SCLO_M61) return hex_u64(a[0], out, outlen, 0) < 0 ? SCLO_BAD_ARG : SCLO_OK;
  if (field == SCLO_M127) {
    if (a[0] == 0 && a[1] == 0) return hex_u64(0, out, outlen, 0) < 0 ? SCLO_BAD_ARG : SCLO_OK;
    int len = 0;
    if (a[1]) {
      len = hex_u64(a[1], out, outlen, 0);
      if (len < 0) return SCLO_BAD_ARG;
    }
    return hex_u64(a[0], out + len, outlen - (size_t)len, 0) < 0 ? SCLO_BAD_ARG : SCLO_OK;
  }
  if (field == SCLO_MONT128 || field == SCLO_GF2_128) {
    u128 v = ((u128)a[1] << 64) | a[0];
    if (field == SCLO_MONT128) { mont128_ensure(); v = mont128_from_mont(v); }
    uint64_t hi = (uint64_t)(v >> 64), lo = (uint64_t)v;
    int len = 0;
    if (hi) {
      len = hex_u64(hi, out, outlen, 0);
      if (len < 0) return SCLO_BAD_ARG;
    }
    return hex_u64(lo, out + len, outlen - (size_t)len, hi != 0) < 0 ? SCLO_BAD_ARG : SCLO_OK;
  }
  return SCLO_BAD_ARG;
}

int sclo_exp(int field, const uint64_t* base, size_t e, uint64_t* dst) {
#define BODY(P) CAT(P, st)(dst, CAT(P, exp)(CAT(P, ld)(base), e)); return SCLO_OK;
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_vector_random(int field, const unsigned char* seed, size_t seed_len, size_t n,
                       uint64_t* out) {
  if (SCLO_IS_RING(field)) return ring_vector_random(RING_K(field), seed, seed_len, n, out);
  const size_t L = (size_t)sclo_limbs(field);
  sclo_prg_t g;
  prg_init(&g, seed, seed_len);
  unsigned char* buf = (unsigned char*)malloc(n * L * 8 + 16);
  if (!buf) return SCLO_BAD_ARG;
  prg_next(&g, buf, n * L * 8);
  int st = sclo_from_bytes(field, buf, n, out);
  free(buf);
  return st;
}

int sclo_shamir_share(int field, const unsigned char* seed, size_t seed_len,
                      const uint64_t* secrets, size_t N, size_t t, size_t n, uint64_t* shares) {
#define BODY(P) return CAT(P, shamir_share)(seed, seed_len, secrets, N, t, n, shares);
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_shamir_share_packed(int field, const unsigned char* seed, size_t seed_len, const uint64_t* secrets, size_t N,
                             size_t t, size_t n, size_t W, uint64_t* shares) {
  if (W == 0) return SCLO_BAD_ARG;
#define BODY(P) return CAT(P, shamir_share_packed)(seed, seed_len, secrets, N, t, n, W, shares);
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_shamir_share_coeffs(int field, const uint64_t* secrets, const uint64_t* coeffs, size_t N,
                             size_t t, size_t n, uint64_t* shares) {
#define BODY(P) return CAT(P, shamir_share_coeffs)(secrets, coeffs, N, t, n, shares);
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_shamir_recover(int field, const uint64_t* shares, size_t n, size_t N, uint64_t* out) {
#define BODY(P) return CAT(P, shamir_recover)(shares, n, N, out);
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_shamir_recover_lambda(int field, const uint64_t* shares, const uint64_t* lambda,
                               size_t n, size_t N, uint64_t* out) {
#define BODY(P) return CAT(P, shamir_recover_lambda)(shares, lambda, n, N, out);
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_shamir_recover_at(int field, const uint64_t* shares, const uint64_t* alphas,
                           const uint64_t* x, size_t m, size_t N, uint64_t* out) {
#define BODY(P) return CAT(P, shamir_recover_at)(shares, alphas, x, m, N, out);
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_shamir_recover_d(int field, const uint64_t* shares, size_t n, size_t t, size_t N,
                          uint64_t* out, unsigned char* status) {
#define BODY(P) return CAT(P, shamir_recover_d)(shares, n, t, N, out, status);
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_shamir_recover_c(int field, const uint64_t* shares, const uint64_t* alphas, size_t count, size_t N,
                          uint64_t* f_out, uint64_t* e_out, unsigned char* status, unsigned* nerr) {
#define BODY(P) return CAT(P, shamir_recover_c)(shares, alphas, count, N, f_out, e_out, status, nerr);
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_lagrange_basis(int field, const uint64_t* nodes, size_t m, const uint64_t* x,
                        uint64_t* out) {
#define BODY(P) return CAT(P, lagrange_api)(nodes, m, x, out);
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_additive_share(int field, const unsigned char* seed, size_t seed_len,
                        const uint64_t* secrets, size_t N, size_t n, uint64_t* shares) {
  if (SCLO_IS_RING(field)) return ring_additive_share(RING_K(field), seed, seed_len, secrets, N, n, shares);
#define BODY(P) return CAT(P, additive_share)(seed, seed_len, secrets, N, n, shares);
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_additive_recover(int field, const uint64_t* shares, size_t n, size_t N, uint64_t* out) {
  if (SCLO_IS_RING(field)) return ring_additive_recover(RING_K(field), shares, n, N, out);
#define BODY(P) return CAT(P, additive_recover)(shares, n, N, out);
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_dot(int field, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  if (SCLO_IS_RING(field)) return ring_dot(RING_K(field), a, b, n, out);
#define BODY(P) CAT(P, st)(out, CAT(P, vdot)(a, b, n)); return SCLO_OK;
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_sum(int field, const uint64_t* a, size_t n, uint64_t* out) {
  if (SCLO_IS_RING(field)) return ring_dot(RING_K(field), a, NULL, n, out);
#define BODY(P) CAT(P, st)(out, CAT(P, vsum)(a, n)); return SCLO_OK;
  FIELD_SWITCH(field, BODY)
#undef BODY
}

/* Vector::scalarMultiply: r_i = scalar * v_i (vector.h:274-285) */
int sclo_scalar_mul(int field, const uint64_t* a, const uint64_t* scalar, size_t n, uint64_t* out) {
  if (SCLO_IS_RING(field)) {
    const size_t L_ = ring_limbs(RING_K(field));
    for (size_t i = 0; i < n; ++i)
      ring_st(out + i * L_, RING_K(field), ring_ld(scalar, RING_K(field)) * ring_ld(a + i * L_, RING_K(field)));
    return SCLO_OK;
  }
  const size_t L = (size_t)sclo_limbs(field);
#define BODY(P) for (size_t i = 0; i < n; ++i) CAT(P, st)(out + i * L, CAT(P, mul)(CAT(P, ld)(scalar), CAT(P, ld)(a + i * L))); return SCLO_OK;
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_poly_eval(int field, const uint64_t* coeffs, size_t ncoeff, const uint64_t* xs,
                   size_t nx, uint64_t* out) {
#define BODY(P) return CAT(P, poly_eval_api)(coeffs, ncoeff, xs, nx, out);
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_vandermonde(int field, size_t n, size_t m, const uint64_t* xs, uint64_t* out) {
#define BODY(P) CAT(P, vandermonde)(n, m, xs, out); return SCLO_OK;
  FIELD_SWITCH(field, BODY)
#undef BODY
}

int sclo_matmul(int field, const uint64_t* A, const uint64_t* B, size_t n, size_t k, size_t m,
                uint64_t* C) {
  if (SCLO_IS_RING(field)) return ring_matmul(RING_K(field), A, B, n, k, m, C);
#define BODY(P) CAT(P, matmul)(A, B, n, k, m, C); return SCLO_OK;
  FIELD_SWITCH(field, BODY)
#undef BODY
}

size_t sclo_wire_vector(int field, const uint64_t* elems, size_t n, unsigned char* out) {
  const size_t L = (size_t)sclo_limbs(field), bs = 8 * L;
  if (out) {
    const uint32_t cnt = (uint32_t)n;
    memcpy(out, &cnt, 4);
    for (size_t i = 0; i < n; ++i) {
      if (IS_SECP256(field)) { /* montyToBytes: out of Montgomery form, big-endian (ff_ops_gmp.h:298-314) */
        secpq_select(field);
        const fe256 v = secpq_from_mont(secpq_ld(elems + i * L));
        for (int b = 0; b < 32; ++b) out[4 + i * bs + b] = (unsigned char)(v.w[3 - b / 8] >> (8 * (7 - b % 8)));
      } else if (field == SCLO_MONT128) { /* gmp family: out of Montgomery form, big-endian (ff_ops_gmp.h:298-314) */
        mont128_ensure();
        u128 v = mont128_from_mont(mont128_ld(elems + i * L));
        for (int b = 0; b < 16; ++b) out[4 + i * bs + b] = (unsigned char)(v >> (8 * (15 - b)));
      } else {
        memcpy(out + 4 + i * bs, elems + i * L, bs); /* ff::toBytes = memcpy of the value (mersenne61.cc:92-95) */
      }
    }
  }
  return 4 + n * bs;
}

int sclo_unwire_vector(int field, const unsigned char* in, size_t nbytes, uint64_t* elems, size_t capacity,
                       size_t* n) {
  const size_t L = (size_t)sclo_limbs(field), bs = 8 * L;
  uint32_t cnt;
  if (nbytes < 4) return SCLO_BAD_ARG;
  memcpy(&cnt, in, 4);
  if (4 + (size_t)cnt * bs > nbytes || cnt > capacity) return SCLO_BAD_ARG;
  *n = cnt;
  return sclo_from_bytes(field, in + 4, cnt, elems);
}

/* Serializer<Matrix> (matrix.h:910-963): u32 rows, u32 cols, then the vector image of the row-major values */
size_t sclo_wire_matrix(int field, const uint64_t* elems, size_t rows, size_t cols, unsigned char* out) {
  if (out) {
    const uint32_t r = (uint32_t)rows, c = (uint32_t)cols;
    memcpy(out, &r, 4);
    memcpy(out + 4, &c, 4);
  }
  return 8 + sclo_wire_vector(field, elems, rows * cols, out ? out + 8 : NULL);
}

int sclo_unwire_matrix(int field, const unsigned char* in, size_t nbytes, uint64_t* elems, size_t capacity,
                       size_t* rows, size_t* cols) {
  uint32_t r, c;
  size_t n = 0;
  if (nbytes < 12) return SCLO_BAD_ARG;
  memcpy(&r, in, 4);
  memcpy(&c, in + 4, 4);
  const int st = sclo_unwire_vector(field, in + 8, nbytes - 8, elems, capacity, &n);
  if (st != SCLO_OK) return st;
  if (n != (size_t)r * c) return SCLO_BAD_ARG; /* the reference trusts the sender here (matrix.h:420) */
  *rows = r;
  *cols = c;
  return SCLO_OK;
}

int sclo_time_shamir(int field, size_t N, size_t t, size_t n, const unsigned char* seed,
                     size_t seed_len, double* share_s, double* recover_s, uint64_t* mismatches,
                     uint64_t* checksum) {
#define BODY(P) return CAT(P, time_shamir)(N, t, n, seed, seed_len, share_s, recover_s, mismatches, checksum);
  FIELD_SWITCH(field, BODY)
#undef BODY
}
