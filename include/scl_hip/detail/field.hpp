// include/scl_hip/detail/field.hpp -- per-lane finite-field arithmetic for the MI355X engine.
//
// One struct per field with the same static interface, usable from device
// kernels and from the host-side table builders (Lagrange basis, Vandermonde,
// reduction epilogues).  All values are canonical (in [0,p)) between calls;
// because every residue has a unique canonical representative, any evaluation
// order / lazy reduction here is bit-identical to the reference's
// reduce-after-every-step code (SURVEY.md section 8a note C).
//
// What each field follows in the reference (paths relative to /root/reference):
//   M61   include/scl/math/fields/mersenne61.h:29-49, src/scl/math/fields/mersenne61.cc:33-100
//   M127  include/scl/math/fields/mersenne127.h:29-49, src/scl/math/fields/mersenne127.cc:33-128
//   add/sub/neg  src/scl/math/fields/small_ff.h:28-56
//   MONT128  new 2-limb Montgomery field, modelled on include/scl/math/fields/ff_ops_gmp.h:44-260
//   GF2_128  new binary field (not in the reference)
//
// Interface (F = field struct):
//   F::E                  element type            F::LIMBS   uint64 limbs per element
//   F::Ctx                per-launch field parameters (empty for the Mersenne fields)
//   F::add/sub/mul/neg(ctx,..)   canonical in, canonical out
//   F::inv(ctx,a)         Fermat inverse; inv(0) = 0 (callers flag the zero)
//   F::one(ctx), F::from_u64(ctx,v) (v small, < p), F::from_le_word(ctx, raw) (FF::read: "% p")
//   F::Acc                lazy dot-product accumulator: acc_zero, mac(acc,a,b), acc_add(acc,a),
//                         acc_fold(ctx,acc) -> canonical; good for >= F::ACC_TERMS terms
//   F::KC / F::KAcc       multiply-accumulate against a PREPARED constant (a table entry that is the same for
//                         every lane: Lagrange rows, Vandermonde entries): kc_make(ctx,c) once per entry,
//                         kmac(ctx,acc,k,x) per term, kacc_fold(ctx,acc) -> canonical; good for F::K_TERMS terms.
//                         The Mersenne fields split the constant so that the term is a handful of independent
//                         32x32 multiply-adds into 64-bit columns with no carries; the others alias Acc / mac.
#pragma once

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SCL_HD __host__ __device__ __forceinline__
#else
#define SCL_HD inline __attribute__((always_inline))
#endif

namespace sclhip {

typedef uint32_t u32;
typedef uint64_t u64;
typedef unsigned __int128 u128;

struct W256 {
  u128 hi, lo;
};

// 128 x 128 -> 256 bits from four 64 x 64 -> 128 products
SCL_HD W256 mulwide(u128 x, u128 y) {
  const u64 x0 = (u64)x, x1 = (u64)(x >> 64), y0 = (u64)y, y1 = (u64)(y >> 64);
  const u128 p00 = (u128)x0 * y0, p01 = (u128)x0 * y1, p10 = (u128)x1 * y0, p11 = (u128)x1 * y1;
  const u128 mid = (p00 >> 64) + (u64)p01 + (u64)p10;  // < 3 * 2^64
  W256 r;
  r.lo = (u128)(u64)p00 | (mid << 64);
  r.hi = p11 + (p01 >> 64) + (p10 >> 64) + (mid >> 64);
  return r;
}

SCL_HD u128 bswap128(u128 v) {
  return ((u128)__builtin_bswap64((u64)v) << 64) | __builtin_bswap64((u64)(v >> 64));
}

// ---- lazy product sums for the Montgomery fields -------------------------------------------------------
// A sum of K products of L-limb operands (32-bit limbs) kept as 2L-1 column sums of 32x32 partial products:
// column k = sum over i+j=k of a_i*b_j, held as a 64-bit low word plus a count of its carries.  That is one
// v_mad_u64_u32 and one v_addc per partial product and no reduction until the end, half the multiplies of the
// interleaved Montgomery product per term.  Good for K <= 2^24 products.
template <int L>
struct LazyCols {
  u64 lo[2 * L - 1];
  u32 hi[2 * L - 1];
};

SCL_HD void lazy_col_mad(u64& lo, u32& hi, u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
#else
  const u64 s = lo + (u64)a * b;
  hi += s < lo ? 1u : 0u;
  lo = s;
#endif
}

// acc += a * b, one 32 x 32 -> 64 multiply-add.  Written out so that the product is never widened or split
// from its accumulation (the compiler turns "acc += (u64)a * b" on loop-carried limbs into a 64 x 32 product
// or into a multiply by zero followed by a 64-bit add).
SCL_HD void mad32(u64& acc, u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  u64 carry_unused;  // an SGPR pair of the compiler's choosing (naming vcc costs a hazard nop per instruction)
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry_unused) : "v"(a), "v"(b));
#else
  acc += (u64)a * b;
#endif
}

template <int L>
SCL_HD void lazy_zero(LazyCols<L>& c) {
#pragma unroll
  for (int k = 0; k < 2 * L - 1; ++k) {
    c.lo[k] = 0;
    c.hi[k] = 0;
  }
}

template <int L>
SCL_HD void lazy_mac(LazyCols<L>& c, const u32* a, const u32* b) {
#pragma unroll
  for (int i = 0; i < L; ++i) {
#pragma unroll
    for (int j = 0; j < L; ++j) lazy_col_mad(c.lo[i + j], c.hi[i + j], a[i], b[j]);
  }
}

// column sums -> the NT 32-bit limbs of the integer they stand for (which must be < 2^(32 NT))
template <int L, int NT>
SCL_HD void lazy_limbs(const LazyCols<L>& c, u32* t) {
  u64 cy = 0;
#pragma unroll
  for (int k = 0; k < 2 * L - 1; ++k) {
    const u64 s = c.lo[k] + cy;
    const u32 wrap = s < cy ? 1u : 0u;
    t[k] = (u32)s;
    cy = (s >> 32) + ((u64)(c.hi[k] + wrap) << 32);
  }
#pragma unroll
  for (int k = 2 * L - 1; k < NT; ++k) {
    t[k] = (u32)cy;
    cy >>= 32;
  }
}

// NR rounds of word-32 Montgomery reduction of t[0..NT): afterwards t[NR..NT) = (T + Q p) / 2^(32 NR) with
// Q < 2^(32 NR); p has L limbs, mc = -p^-1 mod 2^32
template <int L, int NR, int NT, typename PF>
SCL_HD void lazy_redc(u32* t, PF p, u32 mc) {
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const u32 q = t[i] * mc;
    u64 cy = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) {
      const u64 s = (u64)q * p(j) + t[i + j] + cy;
      t[i + j] = (u32)s;
      cy = s >> 32;
    }
#pragma unroll
    for (int k = i + L; k < NT; ++k) {
      const u64 s = (u64)t[k] + cy;
      t[k] = (u32)s;
      cy = s >> 32;
    }
  }
}

// KC interface of the fields that have no cheaper prepared-constant form: it is their Acc / mac
#define SCL_KC_IS_ACC()                                                                              \
  typedef E KC;                                                                                      \
  typedef Acc KAcc;                                                                                  \
  enum { K_TERMS = ACC_TERMS };                                                                      \
  static SCL_HD KC kc_make(const Ctx&, const E& c) { return c; }                                     \
  static SCL_HD KAcc kacc_zero() { return acc_zero(); }                                              \
  static SCL_HD void kmac(const Ctx& ctx, KAcc& a, const KC& k, const E& x) { mac(ctx, a, k, x); }   \
  static SCL_HD E kacc_fold(const Ctx& ctx, const KAcc& a) { return acc_fold(ctx, a); }

// ------------------------------------------------------------------ Mersenne61
struct M61 {
  typedef u64 E;
  struct Ctx {};
  enum { LIMBS = 1, ACC_TERMS = 64, TAG = 0 };
  static constexpr u64 P = 0x1FFFFFFFFFFFFFFFull;

  static SCL_HD E zero() { return 0; }
  static SCL_HD E one(const Ctx&) { return 1; }
  static SCL_HD E from_u64(const Ctx&, u64 v) { return v; }
  static SCL_HD bool is_zero(E a) { return a == 0; }
  static SCL_HD bool eq(E a, E b) { return a == b; }
  static SCL_HD E ld(const u64* p) { return p[0]; }
  static SCL_HD void st(u64* p, E v) { p[0] = v; }
  static SCL_HD u32 low32(E v) { return (u32)v; }

  static SCL_HD E add(const Ctx&, E a, E b) {
    const u64 t = a + b;
    return t >= P ? t - P : t;
  }
  static SCL_HD E sub(const Ctx&, E a, E b) { return a >= b ? a - b : a + P - b; }
  static SCL_HD E neg(const Ctx&, E a) { return a ? P - a : 0; }
  // z < 2^125 -> canonical
  static SCL_HD E reduce_product(u128 z) {
    const u64 r = (u64)(z >> 61) + ((u64)z & P);
    return r >= P ? r - P : r;
  }
  static SCL_HD E mul(const Ctx&, E a, E b) { return reduce_product((u128)a * b); }
  static SCL_HD E sqr(const Ctx& c, E a) { return mul(c, a, a); }
  // FF::read = native little-endian u64 "% p" (mersenne61.cc:86-90)
  static SCL_HD E from_le_word(const Ctx&, u64 w) {
    const u64 r = (w & P) + (w >> 61);
    return r >= P ? r - P : r;
  }

  // y*x + c for a small constant x < 2^32, LAZY: inputs y, c < 2^62 (not necessarily canonical),
  // result < 2^61 + 4 and congruent to y*x + c.  Two 32x32->64 multiplies, one fold:
  //   y*x = h*2^32 + lo32(l),  l = yl*x,  h = yh*x + (l >> 32);   h*2^32 = (h >> 29) + ((h & (2^29-1)) << 32) mod p
  enum { SMALL_BITS = 32 };
  static SCL_HD u64 muladd_small_lazy(u64 y, u32 x, u64 c) {
    const u64 l = (u64)(u32)y * x;
    const u64 h = (u64)(u32)(y >> 32) * x + (l >> 32);              // < 2^62 + 2^32
    const u64 v = (u64)(u32)l | ((h & 0x1FFFFFFFull) << 32);         // < 2^61
    const u64 s = v + (h >> 29) + c;                                 // < 2^63
    return (s & P) + (s >> 61);
  }
  static SCL_HD E canon(u64 r) { return r >= P ? r - P : r; }       // r < 2^61 + P
  static SCL_HD u64 muladd_small_lazy(const Ctx&, u64 y, u32 x, u64 c) { return muladd_small_lazy(y, x, c); }
  static SCL_HD E muladd_small(const Ctx&, E y, u32 x, E c) { return canon(muladd_small_lazy(y, x, c)); }

  struct Acc {
    u128 v;
  };
  static SCL_HD Acc acc_zero() { return Acc{0}; }
  static SCL_HD void mac(const Ctx&, Acc& acc, E a, E b) { acc.v += (u128)a * b; }  // product < 2^122
  static SCL_HD void acc_add(const Ctx&, Acc& acc, E a) { acc.v += a; }
  // any 128-bit value -> canonical: sum the three 61-bit digits (2^61 = 1 mod p), fold once more
  static SCL_HD E fold128(u128 z) {
    const u64 lo = (u64)z, hi = (u64)(z >> 64);
    const u64 s = (lo & P) + (((lo >> 61) | (hi << 3)) & P) + (hi >> 58);
    const u64 r = (s & P) + (s >> 61);
    return r >= P ? r - P : r;
  }
  static SCL_HD E acc_fold(const Ctx&, const Acc& acc) { return fold128(acc.v); }

  // Prepared constant c: 21-bit limbs of c (w[0..2]) and of c*2^32 mod p (w[3..5]).  With x = x0 + x1*2^32,
  //   c*x = sum_j 2^(21j) * (x0*w[j] + x1*w[3+j])   (mod p):
  // six independent 32x32 multiply-adds into three 64-bit columns, each product < 2^53, so 1024 terms fit
  // without a carry; x may be any 64-bit value.
  struct KC {
    u32 w[6];
  };
  struct KAcc {
    u64 c[3];
  };
  enum { K_TERMS = 1024 };
  static SCL_HD KC kc_make(const Ctx& ctx, E c) {
    const E h = mul(ctx, c, (E)1 << 32);
    KC k;
    k.w[0] = (u32)(c & 0x1FFFFF);
    k.w[1] = (u32)((c >> 21) & 0x1FFFFF);
    k.w[2] = (u32)(c >> 42);
    k.w[3] = (u32)(h & 0x1FFFFF);
    k.w[4] = (u32)((h >> 21) & 0x1FFFFF);
    k.w[5] = (u32)(h >> 42);
    return k;
  }
  static SCL_HD KAcc kacc_zero() { return KAcc{{0, 0, 0}}; }
  static SCL_HD void kmac(const Ctx&, KAcc& a, const KC& k, E x) {
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      mad32(a.c[j], x0, k.w[j]);
      mad32(a.c[j], x1, k.w[3 + j]);
    }
  }
  static SCL_HD E kacc_fold(const Ctx&, const KAcc& a) {
    return fold128((u128)a.c[0] + ((u128)a.c[1] << 21) + ((u128)a.c[2] << 42));
  }

  // x^(2^k) * y
  static SCL_HD E sqn_mul(const Ctx& c, E x, int k, E y) {
    for (int i = 0; i < k; ++i) x = sqr(c, x);
    return mul(c, x, y);
  }
  // Fermat: a^(p-2), p-2 = 2^61-3 = (2^59-1)*4 + 1
  static SCL_HD E inv(const Ctx& c, E a) {
    const E x2 = sqn_mul(c, a, 1, a);      // 2^2-1
    const E x4 = sqn_mul(c, x2, 2, x2);    // 2^4-1
    const E x8 = sqn_mul(c, x4, 4, x4);
    const E x16 = sqn_mul(c, x8, 8, x8);
    const E x32 = sqn_mul(c, x16, 16, x16);
    const E x48 = sqn_mul(c, x32, 16, x16);
    const E x56 = sqn_mul(c, x48, 8, x8);
    const E x58 = sqn_mul(c, x56, 2, x2);
    const E x59 = sqn_mul(c, x58, 1, a);
    return sqn_mul(c, x59, 2, a);
  }
};

// ----------------------------------------------------------------- Mersenne127
struct M127 {
  typedef u128 E;
  struct Ctx {};
  enum { LIMBS = 2, ACC_TERMS = 1 << 30, TAG = 1 };
  static SCL_HD u128 P() { return (((u128)0x7FFFFFFFFFFFFFFFull) << 64) | (u128)0xFFFFFFFFFFFFFFFFull; }

  static SCL_HD E zero() { return 0; }
  static SCL_HD E one(const Ctx&) { return 1; }
  static SCL_HD E from_u64(const Ctx&, u64 v) { return v; }
  static SCL_HD bool is_zero(E a) { return a == 0; }
  static SCL_HD bool eq(E a, E b) { return a == b; }
  static SCL_HD u32 low32(E v) { return (u32)v; }
  static SCL_HD E ld(const u64* p) { return ((u128)p[1] << 64) | p[0]; }
  static SCL_HD void st(u64* p, E v) {
    p[0] = (u64)v;
    p[1] = (u64)(v >> 64);
  }

  static SCL_HD E add(const Ctx&, E a, E b) {
    const u128 t = a + b;  // < 2^128
    return t >= P() ? t - P() : t;
  }
  static SCL_HD E sub(const Ctx&, E a, E b) { return a >= b ? a - b : a + P() - b; }
  static SCL_HD E neg(const Ctx&, E a) { return a ? P() - a : 0; }
  // a*b as an un-normalised value < 2^128 congruent to the product (mersenne127.cc:87-93)
  static SCL_HD u128 mul_lazy(E a, E b) {
    const W256 z = mulwide(a, b);
    return ((z.hi << 1) | (z.lo >> 127)) + (z.lo & P());
  }
  static SCL_HD E mul(const Ctx&, E a, E b) {
    const u128 t = mul_lazy(a, b);
    return t >= P() ? t - P() : t;
  }
  // a^2 from three 64 x 64 products instead of four (a = a1 2^64 + a0, a1 < 2^63: a0^2 + 2 a0 a1 2^64 + a1^2 2^128) -- what the
  // Fermat chain of inv spends 126 of its 138 products on
  static SCL_HD E sqr(const Ctx&, E a) {
    const u64 a0 = (u64)a, a1 = (u64)(a >> 64);
    const u128 p00 = (u128)a0 * a0, mid2 = ((u128)a0 * a1) << 1, p11 = (u128)a1 * a1;  // a0 a1 < 2^127: the doubling fits
    W256 z;
    z.lo = p00 + (mid2 << 64);
    z.hi = p11 + (mid2 >> 64) + (z.lo < p00 ? 1 : 0);
    const u128 t = ((z.hi << 1) | (z.lo >> 127)) + (z.lo & P());
    return t >= P() ? t - P() : t;
  }
  // FF::read = 16-byte little-endian load "% p" (mersenne127.cc:114-118)
  static SCL_HD E from_le_word(const Ctx&, u128 w) {
    const u128 r = (w & P()) + (w >> 127);
    return r >= P() ? r - P() : r;
  }

  // y*x + c for a small constant x < 2^32, LAZY: y any u128 (congruent value), c < 2^127; result
  // < 2^127 + 2^34, congruent to y*x + c.  A 4-limb multiply chain (4 mads) gives 160 bits, one fold.
  enum { SMALL_BITS = 32 };
  static SCL_HD u128 muladd_small_lazy(u128 y, u32 x, u128 c) {
    const u64 t0 = (u64)(u32)y * x;
    const u64 t1 = (u64)(u32)(y >> 32) * x + (t0 >> 32);
    const u64 t2 = (u64)(u32)(y >> 64) * x + (t1 >> 32);
    const u64 t3 = (u64)(u32)(y >> 96) * x + (t2 >> 32);  // bits 96..159 of the product
    const u128 low = (u128)(u32)t0 | ((u128)(u32)t1 << 32) | ((u128)(u32)t2 << 64) | ((u128)(t3 & 0x7FFFFFFFull) << 96);
    const u128 s = low + c;  // < 2^128
    return (s & P()) + (s >> 127) + (t3 >> 31);
  }
  static SCL_HD E canon(u128 r) {  // r < 2^127 + 2^34
    const u128 f = (r & P()) + (r >> 127);
    return f >= P() ? f - P() : f;
  }
  static SCL_HD u128 muladd_small_lazy(const Ctx&, u128 y, u32 x, u128 c) { return muladd_small_lazy(y, x, c); }
  static SCL_HD E muladd_small(const Ctx&, E y, u32 x, E c) { return canon(muladd_small_lazy(y, x, c)); }

  struct Acc {
    u128 lo;
    u64 hi;
  };
  static SCL_HD Acc acc_zero() { return Acc{0, 0}; }
  static SCL_HD void acc_add_raw(Acc& acc, u128 a) {
    const u128 t = acc.lo + a;
    acc.hi += (t < a);
    acc.lo = t;
  }
  static SCL_HD void acc_add(const Ctx&, Acc& acc, E a) { acc_add_raw(acc, a); }
  static SCL_HD void mac(const Ctx&, Acc& acc, E a, E b) { acc_add_raw(acc, mul_lazy(a, b)); }
  // hi*2^128 + lo = 2*hi + lo (mod p)
  static SCL_HD E acc_fold(const Ctx&, const Acc& acc) {
    const u128 v = (acc.lo & P()) + (acc.lo >> 127) + ((u128)acc.hi << 1);
    const u128 r = (v & P()) + (v >> 127);
    return r >= P() ? r - P() : r;
  }

  // Prepared constant c: for each 32-bit limb position i of x, the 22-bit limbs w[6i..6i+5] of c*2^(32i) mod p:
  //   c*x = sum_j 2^(22j) * sum_i x_i*w[6i+j]   (mod p):
  // 24 independent 32x32 multiply-adds into six 64-bit columns, each product < 2^54, so 256 terms fit without a
  // carry (the general product costs 16 multiplies and as many carry chains again); x may be any 128-bit value.
  struct KC {
    u32 w[24];
  };
  struct KAcc {
    u64 c[6];
  };
  enum { K_TERMS = 256 };
  static SCL_HD KC kc_make(const Ctx& ctx, E c) {
    KC k;
    E v = c;
    for (int i = 0; i < 4; ++i) {
      for (int j = 0; j < 6; ++j) k.w[6 * i + j] = (u32)(v >> (22 * j)) & 0x3FFFFFu;
      v = mul(ctx, v, (E)1 << 32);
    }
    return k;
  }
  static SCL_HD KAcc kacc_zero() { return KAcc{{0, 0, 0, 0, 0, 0}}; }
  static SCL_HD void kmac(const Ctx&, KAcc& a, const KC& k, E x) {
    const u64 xl = (u64)x, xh = (u64)(x >> 64);
    const u32 xi[4] = {(u32)xl, (u32)(xl >> 32), (u32)xh, (u32)(xh >> 32)};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) mad32(a.c[j], xi[i], k.w[6 * i + j]);
  }
  static SCL_HD E kacc_fold(const Ctx& ctx, const KAcc& a) {
    Acc t = acc_zero();
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int sh = 22 * j;  // column weight 2^sh; the part at or above 2^128 goes to t.hi
      acc_add_raw(t, (u128)a.c[j] << sh);
      if (sh > 64) t.hi += a.c[j] >> (128 - sh);
    }
    return acc_fold(ctx, t);
  }

  static SCL_HD E sqn_mul(const Ctx& c, E x, int k, E y) {
    for (int i = 0; i < k; ++i) x = sqr(c, x);
    return mul(c, x, y);
  }
  // Fermat: p-2 = 2^127-3 = (2^125-1)*4 + 1
  static SCL_HD E inv(const Ctx& c, E a) {
    const E x2 = sqn_mul(c, a, 1, a);
    const E x4 = sqn_mul(c, x2, 2, x2);
    const E x8 = sqn_mul(c, x4, 4, x4);
    const E x16 = sqn_mul(c, x8, 8, x8);
    const E x32 = sqn_mul(c, x16, 16, x16);
    const E x64 = sqn_mul(c, x32, 32, x32);
    const E x96 = sqn_mul(c, x64, 32, x32);
    const E x112 = sqn_mul(c, x96, 16, x16);
    const E x120 = sqn_mul(c, x112, 8, x8);
    const E x124 = sqn_mul(c, x120, 4, x4);
    const E x125 = sqn_mul(c, x124, 1, a);
    return sqn_mul(c, x125, 2, a);
  }
};

// --------------------------------------------------------------------- MONT128
// Elements are Montgomery residues x*R mod p, R = 2^128 (ff_ops_gmp.h keeps its
// values the same way).  PARITY UNPINNED: no such field in the reference.
struct Mont128 {
  typedef u128 E;
  struct Ctx {
    u128 p, mc, one, r2;  // modulus, -p^-1 mod R, R mod p, R^2 mod p
    u128 k32;             // 2^32 * R mod p (undoes the extra word of the lazy accumulator's reduction)
    u64 bmu;              // floor(2^162 / p) for a full-width modulus (p >= 2^127), else 0: the small-node quotient estimate
  };
  enum { LIMBS = 2, ACC_TERMS = 1 << 24, TAG = 2 };

  // parameters for an odd modulus p >= 3 (host side)
  static inline Ctx make_ctx(u128 p) {
    Ctx c;
    c.p = p;
    u128 inv = p;  // correct to 3 bits for odd p; each Newton step doubles that
    for (int i = 0; i < 7; ++i) inv *= 2 - p * inv;
    c.mc = (u128)0 - inv;
    u128 r = 1 % p;
    c.one = 0;
    for (int i = 0; i < 256; ++i) {
      const u128 t = r + r;
      r = (t < r || t >= p) ? t - p : t;
      if (i == 127) c.one = r;
    }
    c.r2 = r;
    c.k32 = mul(c, (u128)1 << 32, c.r2);
    c.bmu = 0;
    if (p >> 127) {  // long division of 2^162 by p: the quotient has 35 or 36 bits
      u128 rem = 1;
      u64 q = 0;
      for (int i = 0; i < 162; ++i) {
        const bool top = (rem >> 127) != 0;
        rem <<= 1;
        q <<= 1;
        if (top || rem >= p) {
          rem -= p;
          q |= 1;
        }
      }
      c.bmu = q;
    }
    return c;
  }

  static SCL_HD E zero() { return 0; }
  static SCL_HD E one(const Ctx& c) { return c.one; }
  static SCL_HD bool is_zero(E a) { return a == 0; }
  static SCL_HD bool eq(E a, E b) { return a == b; }
  static SCL_HD u32 low32(E v) { return (u32)v; }
  static SCL_HD E ld(const u64* p) { return ((u128)p[1] << 64) | p[0]; }
  static SCL_HD void st(u64* p, E v) {
    p[0] = (u64)v;
    p[1] = (u64)(v >> 64);
  }
  static SCL_HD E add(const Ctx& c, E a, E b) {
    const u128 t = a + b;
    return (t < a || t >= c.p) ? t - c.p : t;
  }
  static SCL_HD E sub(const Ctx& c, E a, E b) { return a >= b ? a - b : a - b + c.p; }
  static SCL_HD E neg(const Ctx& c, E a) { return a ? c.p - a : 0; }
  // REDC of a 256-bit T < p*R
  static SCL_HD E redc(const Ctx& c, const W256& t) {
    const u128 m = t.lo * c.mc;
    const W256 mp = mulwide(m, c.p);
    const u128 lo = t.lo + mp.lo;  // == 0 mod R
    const u128 hi = t.hi + mp.hi;
    const bool c1 = hi < t.hi;
    const u128 hi2 = hi + (lo < t.lo ? 1 : 0);
    const bool c2 = hi2 < hi;
    return (c1 || c2 || hi2 >= c.p) ? hi2 - c.p : hi2;
  }
  // The Montgomery product a b R^-1 mod p, interleaved over 32-bit words -- the form of the reference's montyModMul over its
  // 64-bit limbs (ff_ops_gmp.h:174-191) -- column by column (product scanning): column k collects a_i b_j and m_i p_j for
  // i + j = k in ONE 96-bit accumulator (a v_mad_u64_u32 onto the running 64 bits + a carry into the third word per partial
  // product: no operand is ever widened or moved), m_k = low word * (-p^-1 mod 2^32) makes the low word vanish, the accumulator
  // moves down a word.  32 multiply-adds, 4 word products: about half the vector instructions of "128 x 128 -> 256, low product
  // by -p^-1, 128 x 128 -> 256, add" (redc(mulwide(a, b)), which stays for the lazy accumulators' final reduction).  The result
  // is the unique residue in [0, p) either way (the sum is < 2p before the last subtraction for a, b < p).
  static SCL_HD E mul(const Ctx& c, E a, E b) {
    const u32 aw[4] = {(u32)a, (u32)(a >> 32), (u32)(a >> 64), (u32)(a >> 96)};
    const u32 bw[4] = {(u32)b, (u32)(b >> 32), (u32)(b >> 64), (u32)(b >> 96)};
    const u32 pw[4] = {(u32)c.p, (u32)(c.p >> 32), (u32)(c.p >> 64), (u32)(c.p >> 96)};
    const u32 n0 = (u32)c.mc;
    u32 m[4], t[4];
    u64 lo = 0;
    u32 hi = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; ++k) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int j = 0; j < k; ++j) {
        lazy_col_mad(lo, hi, aw[j], bw[k - j]);
        lazy_col_mad(lo, hi, m[j], pw[k - j]);
      }
      lazy_col_mad(lo, hi, aw[k], bw[0]);
      m[k] = (u32)lo * n0;
      lazy_col_mad(lo, hi, m[k], pw[0]);   // the low word is 0 now
      lo = (lo >> 32) | ((u64)hi << 32);
      hi = 0;
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 4; k < 8; ++k) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int j = k - 3; j < 4; ++j) {
        lazy_col_mad(lo, hi, aw[j], bw[k - j]);
        lazy_col_mad(lo, hi, m[j], pw[k - j]);
      }
      t[k - 4] = (u32)lo;
      lo = (lo >> 32) | ((u64)hi << 32);
      hi = 0;
    }
    const u128 r = (u128)((u64)t[0] | ((u64)t[1] << 32)) | ((u128)((u64)t[2] | ((u64)t[3] << 32)) << 64);
    return (lo || r >= c.p) ? r - c.p : r;   // lo: the ninth word, 0 or 1
  }
  static SCL_HD E sqr(const Ctx& c, E a) { return mul(c, a, a); }
  // The Horner kernels' small-constant form (SMALLX): the node as a plain integer x < 2^32 (NOT its residue), y*x + c through
  // muladd_small below -- one lazy limb product and one Barrett step instead of a Montgomery product; full-width moduli only
  // (small_nodes_ok), the host hands those kernels the nodes' plain values
  enum { SMALL_BITS = 32 };

  // Small-node sharing.  A node power v that is a small plain integer multiplies a residue without leaving
  // Montgomery form, (x R) v = (x v) R, so share_i = sum_k c_k v_ik needs no Montgomery product: four 32 x 32
  // multiply-adds per term into per-limb 64-bit sums (seven terms of v < 2^29 cannot overflow them), then ONE reduction
  // of the < 2^160 sum S.  For a full-width modulus (2^127 <= p < 2^128, Ctx::bmu != 0) that is a Barrett step:
  // q' = floor(floor(S / 2^96) * bmu / 2^66), bmu = floor(2^162 / p), satisfies q - 1 <= q' <= q for q = floor(S / p) < 2^32
  // (both factors are rounded down, so q' <= q; the truncation of S costs < 2^96 / p <= 2^-31 and that of bmu
  // < floor(S / 2^96) / 2^66 < 1/4), so S - q' p < 2p takes ONE conditional subtraction (round 4; two with bmu = floor(2^160 / p)).
  // (ff_ops_gmp.h:174-191 reaches the same residue one montyModMul and one montyModAdd at a time.)
  struct SAcc {
    u64 a[4];
  };
  static SCL_HD void sacc_zero(SAcc& s) { s.a[0] = s.a[1] = s.a[2] = s.a[3] = 0; }
  static SCL_HD void sacc_mac(SAcc& s, E c, u32 v) {
    mad32(s.a[0], (u32)c, v);
    mad32(s.a[1], (u32)(c >> 32), v);
    mad32(s.a[2], (u32)(c >> 64), v);
    mad32(s.a[3], (u32)(c >> 96), v);
  }
  // c0 + sum_j a_j 2^(32 j) mod p, canonical; the sum is < p * 2^32
  static SCL_HD E sacc_fold(const Ctx& c, const SAcc& s, E c0) {
    u64 l0 = (u64)(u32)s.a[0] + (u32)c0;
    u64 l1 = (s.a[0] >> 32) + (u64)(u32)s.a[1] + (u32)(c0 >> 32);
    u64 l2 = (s.a[1] >> 32) + (u64)(u32)s.a[2] + (u32)(c0 >> 64);
    u64 l3 = (s.a[2] >> 32) + (u64)(u32)s.a[3] + (u32)(c0 >> 96);
    u64 l4 = (s.a[3] >> 32);
    l1 += l0 >> 32;
    l2 += l1 >> 32;
    l3 += l2 >> 32;
    l4 += l3 >> 32;  // < 2^32: S < p * 2^32 < 2^160
    const u128 slo = (u128)((u64)(u32)l0 | (l1 << 32)) | ((u128)((u64)(u32)l2 | (l3 << 32)) << 64);
    const u64 sh = (u64)(u32)l3 | (l4 << 32);  // floor(S / 2^96)
    const u32 q = (u32)(u64)(((u128)sh * c.bmu) >> 66);
    const u32 p0 = (u32)c.p, p1 = (u32)(c.p >> 32), p2 = (u32)(c.p >> 64), p3 = (u32)(c.p >> 96);
    u64 t0 = 0, t1, t2, t3;
    mad32(t0, q, p0);
    t1 = t0 >> 32;
    mad32(t1, q, p1);
    t2 = t1 >> 32;
    mad32(t2, q, p2);
    t3 = t2 >> 32;
    mad32(t3, q, p3);
    const u128 qlo = (u128)((u64)(u32)t0 | (t1 << 32)) | ((u128)((u64)(u32)t2 | (t3 << 32)) << 64);
    u128 r = slo - qlo;
    const u32 rh = (u32)l4 - (u32)(t3 >> 32) - (slo < qlo ? 1u : 0u);  // S - q p < 2p: rh <= 1
    if (rh || r >= c.p) r -= c.p;
    return r;
  }
  enum { SMALL_NODE_VALUE_BITS = 29 };
  static SCL_HD bool small_nodes_ok(const Ctx& c) { return c.bmu != 0; }
  // y * x + a for a plain x < 2^32, canonical (the Horner step between the groups of k_share_blocked)
  static SCL_HD E muladd_small(const Ctx& c, E y, u32 x, E a) {
    SAcc s;
    sacc_zero(s);
    sacc_mac(s, y, x);
    return sacc_fold(c, s, a);
  }
  static SCL_HD E muladd_small_lazy(const Ctx& c, E y, u32 x, E a) { return muladd_small(c, y, x, a); }
  static SCL_HD E muladd_small_lazy(E y, u32, E) { return y; }  // (never run: every SMALLX caller passes the context)
  static SCL_HD E canon(E r) { return r; }
  static SCL_HD E to_mont(const Ctx& c, u128 x) { return mul(c, x, c.r2); }
  static SCL_HD u128 from_mont(const Ctx& c, E a) { return mul(c, a, 1); }
  static SCL_HD E from_u64(const Ctx& c, u64 v) { return to_mont(c, v); }
  // fromBytes of the gmp family is BIG-endian (ff_ops_gmp.h:279-290); raw is the LE load
  static SCL_HD E from_le_word(const Ctx& c, u128 raw) { return to_mont(c, bswap128(raw)); }
  static SCL_HD E raw_from_le_word(u128 raw) { return bswap128(raw); }  // the integer, not a residue

  // Lazy accumulator: products summed unreduced (LazyCols), plain elements summed modularly beside them.
  // The fold reduces by FIVE 32-bit words instead of four, (T + Q p) / 2^160 < p (K p / 2^160 + 1) < 2p for any
  // K <= 2^32 terms and ANY odd p, so one conditional subtraction lands in [0,p); the spare factor 2^-32 is
  // taken back by one Montgomery product with k32 = 2^32 R.
  struct Acc {
    LazyCols<4> c;
    E e;
  };
  static SCL_HD Acc acc_zero() {
    Acc a;
    lazy_zero(a.c);
    a.e = 0;
    return a;
  }
  static SCL_HD void mac(const Ctx&, Acc& acc, E a, E b) {
    const u32 al[4] = {(u32)a, (u32)(a >> 32), (u32)(a >> 64), (u32)(a >> 96)};
    const u32 bl[4] = {(u32)b, (u32)(b >> 32), (u32)(b >> 64), (u32)(b >> 96)};
    lazy_mac<4>(acc.c, al, bl);
  }
  static SCL_HD void acc_add(const Ctx& c, Acc& acc, E a) { acc.e = add(c, acc.e, a); }
  static SCL_HD E fold_wide(const Ctx& c, const Acc& acc) {
    u32 t[10];
    lazy_limbs<4, 10>(acc.c, t);
    const u128 p = c.p;
    lazy_redc<4, 5, 10>(t, [p](int j) { return (u32)(p >> (32 * j)); }, (u32)c.mc);
    u128 r = (u128)t[5] | ((u128)t[6] << 32) | ((u128)t[7] << 64) | ((u128)t[8] << 96);
    if (t[9] || r >= p) r -= p;
    return r;  // = sum / 2^160
  }
  static SCL_HD E acc_fold(const Ctx& c, const Acc& acc) { return add(c, mul(c, fold_wide(c, acc), c.k32), acc.e); }
  // Tables of constants can carry the factor 2^32 themselves: with every mac(acc, a, table_scale(b)) the
  // fold needs no product at all.
  static SCL_HD E table_scale(const Ctx& c, E b) { return mul(c, b, c.k32); }
  static SCL_HD E acc_fold_scaled(const Ctx& c, const Acc& acc) { return add(c, fold_wide(c, acc), acc.e); }
  SCL_KC_IS_ACC()

  // a^(p-2) with fixed 4-bit windows: a^0 .. a^15 once (14 products), then four squarings and at most one product per nibble of
  // the exponent -- 128 + 32 + 14 = 174 products where bit-by-bit square-and-multiply takes 128 + popcount(p - 2) (255 for
  // 2^128 - 159).  The exponent is the same in every lane: the table index is wave-uniform.
  static SCL_HD E inv(const Ctx& c, E a) {
    const u128 e = c.p - 2;
    E tbl[16];
    tbl[0] = c.one;
    tbl[1] = a;
    for (int i = 2; i < 16; ++i) tbl[i] = mul(c, tbl[i - 1], a);
    E r = tbl[(unsigned)(e >> 124) & 15u];
    for (int i = 30; i >= 0; --i) {
      r = sqr(c, sqr(c, sqr(c, sqr(c, r))));
      const unsigned nib = (unsigned)(e >> (4 * i)) & 15u;
      if (nib) r = mul(c, r, tbl[nib]);
    }
    return r;
  }
};

// ------------------------------------------------------------------- GF(2^128)
// GF(2)[x]/(x^128 + x^7 + x^2 + x + 1); bit i of the little-endian word = coeff of x^i.
// PARITY UNPINNED: no such field in the reference.
struct Gf128 {
  typedef u128 E;
  struct Ctx {};
  enum { LIMBS = 2, ACC_TERMS = 1 << 30, TAG = 3 };

  static SCL_HD E zero() { return 0; }
  static SCL_HD E one(const Ctx&) { return 1; }
  static SCL_HD E from_u64(const Ctx&, u64 v) { return v; }
  static SCL_HD bool is_zero(E a) { return a == 0; }
  static SCL_HD bool eq(E a, E b) { return a == b; }
  static SCL_HD u32 low32(E v) { return (u32)v; }
  static SCL_HD E ld(const u64* p) { return ((u128)p[1] << 64) | p[0]; }
  static SCL_HD void st(u64* p, E v) {
    p[0] = (u64)v;
    p[1] = (u64)(v >> 64);
  }
  static SCL_HD E add(const Ctx&, E a, E b) { return a ^ b; }
  static SCL_HD E sub(const Ctx&, E a, E b) { return a ^ b; }
  static SCL_HD E neg(const Ctx&, E a) { return a; }
  // multiply by x^4 with reduction
  static SCL_HD E mulx4(E a) {
    const u64 t = (u64)(a >> 124);  // the 4 bits leaving the top
    return (a << 4) ^ (u128)((t << 7) ^ (t << 2) ^ (t << 1) ^ t);
  }
  static SCL_HD E mulx(E a) {
    const u64 t = (u64)(a >> 127);
    return (a << 1) ^ (u128)(t * 0x87);
  }
  // 256 -> 128 bits with x^128 = x^7 + x^2 + x + 1: lo ^ hi*r; the seven bits hi*r pushes past x^127 come down once more
  static SCL_HD E reduce256(E lo, E hi) {
    const u64 top = (u64)(hi >> 121) ^ (u64)(hi >> 126) ^ (u64)(hi >> 127);  // < 2^7
    const E hr = (hi << 7) ^ (hi << 2) ^ (hi << 1) ^ hi;
    return lo ^ hr ^ (E)((top << 7) ^ (top << 2) ^ (top << 1) ^ top);
  }
  // the multiples u(x)*a mod f of a for the sixteen polynomials u of degree < 4 (the window table of the comb product)
  static SCL_HD void window_table(E a, E* t) {
    t[0] = 0;
    t[1] = a;
    t[2] = mulx(a);
    t[4] = mulx(t[2]);
    t[8] = mulx(t[4]);
    t[3] = t[2] ^ a;
    t[5] = t[4] ^ a;
    t[6] = t[4] ^ t[2];
    t[7] = t[6] ^ a;
#pragma unroll
    for (int u = 9; u < 16; ++u) t[u] = t[8] ^ t[u - 8];
  }
  static SCL_HD E mul(const Ctx&, E a, E b) {
#if defined(__HIP_DEVICE_COMPILE__)
    // 4-bit windows over b, branch-free: tab = {a, a*x, a*x^2, a*x^3}.  The register-only form for device code that has no
    // LDS to spare; the element-wise kernels use the comb product below on a window table in LDS (k_ew_gf128, kernels.hpp)
    const E a1 = mulx(a), a2 = mulx(a1), a3 = mulx(a2);
    E r = 0;
    for (int k = 31; k >= 0; --k) {
      r = mulx4(r);
      const u32 nib = (u32)(b >> (4 * k)) & 15u;
      r ^= (nib & 1 ? a : (E)0) ^ (nib & 2 ? a1 : (E)0) ^ (nib & 4 ? a2 : (E)0) ^ (nib & 8 ? a3 : (E)0);
    }
    return r;
#else
    // Lopez-Dahab comb with 4-bit windows: nibble k of every 32-bit word of b selects a multiple of a that lands at that
    // word's offset; the 256-bit sum moves up four bits between nibble positions (seven shifts instead of thirty-one)
    E t[16];
    window_table(a, t);
    E lo = 0, hi = 0;
    for (int k = 7; k >= 0; --k) {
      for (int j = 0; j < 4; ++j) {
        const E m = t[(u32)(b >> (32 * j + 4 * k)) & 15u];
        lo ^= m << (32 * j);
        if (j) hi ^= m >> (128 - 32 * j);
      }
      if (k) {
        hi = (hi << 4) | (lo >> 124);
        lo <<= 4;
      }
    }
    return reduce256(lo, hi);
#endif
  }
  // Squaring is linear in characteristic 2: the bits of a move to the even positions of a 256-bit value (a zero between
  // every two), which is then reduced -- about a twentieth of a general product.
  static SCL_HD u32 spread16(u32 x) {  // bit i of the low half -> bit 2i
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
  }
  static SCL_HD u64 spread32(u32 x) { return (u64)spread16(x & 0xFFFFu) | ((u64)spread16(x >> 16) << 32); }
  static SCL_HD E sqr(const Ctx&, E a) {
    const E lo = (E)spread32((u32)a) | ((E)spread32((u32)(a >> 32)) << 64);
    const E hi = (E)spread32((u32)(a >> 64)) | ((E)spread32((u32)(a >> 96)) << 64);
    return reduce256(lo, hi);
  }
  // y*x + c for a small polynomial x (< 2^16, wave-uniform on the GPU): shift-xor per set bit of x,
  // then the <= 16 overflow bits are reduced with x^128 = x^7 + x^2 + x + 1.
  enum { SMALL_BITS = 16 };
  static SCL_HD E canon(E r) { return r; }
  static SCL_HD E muladd_small_lazy(E y, u32 x, E c) { return muladd_small(Ctx{}, y, x, c); }
  static SCL_HD E muladd_small_lazy(const Ctx&, E y, u32 x, E c) { return muladd_small(Ctx{}, y, x, c); }
  static SCL_HD E muladd_small(const Ctx&, E y, u32 x, E c) {
#if defined(__HIP_DEVICE_COMPILE__)
    // On the device x is WAVE-UNIFORM in every caller (a node read from LDS with a uniform index: the Horner kernels):
    // it goes into a scalar register and each of its bits is a scalar branch around a STATICALLY shifted copy of y
    // (v_alignbit with immediate amounts) -- the generic loop below shifts 128 bits by a per-lane amount per set bit,
    // about three times the instructions ((40,13) share: 1.19 -> 1.85 TB/s, tools/gf128_bench.hip).
    const u32 a = __builtin_amdgcn_readfirstlane(x);
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32), y2 = (u32)(y >> 64), y3 = (u32)(y >> 96);
    u32 r0 = (u32)c, r1 = (u32)(c >> 32), r2 = (u32)(c >> 64), r3 = (u32)(c >> 96), ov = 0;
    if (a & 1u) {
      r0 ^= y0;
      r1 ^= y1;
      r2 ^= y2;
      r3 ^= y3;
    }
#define SCL_GF_BIT(B)                                     \
  if (a & (1u << B)) {                                    \
    r0 ^= y0 << B;                                        \
    r1 ^= __builtin_amdgcn_alignbit(y1, y0, 32 - B);      \
    r2 ^= __builtin_amdgcn_alignbit(y2, y1, 32 - B);      \
    r3 ^= __builtin_amdgcn_alignbit(y3, y2, 32 - B);      \
    ov ^= y3 >> (32 - B);                                 \
  }
    SCL_GF_BIT(1) SCL_GF_BIT(2) SCL_GF_BIT(3) SCL_GF_BIT(4) SCL_GF_BIT(5) SCL_GF_BIT(6) SCL_GF_BIT(7)
    if (a >> 8) {  // nodes above 255 are rare: keep their bits out of the common path's way
      SCL_GF_BIT(8) SCL_GF_BIT(9) SCL_GF_BIT(10) SCL_GF_BIT(11) SCL_GF_BIT(12) SCL_GF_BIT(13) SCL_GF_BIT(14) SCL_GF_BIT(15)
    }
#undef SCL_GF_BIT
    // ov < 2^15 holds the bits shifted past x^127: ov * (x^7 + x^2 + x + 1) < 2^22
    r0 ^= ov ^ (ov << 1) ^ (ov << 2) ^ (ov << 7);
    return (u128)r0 | ((u128)r1 << 32) | ((u128)r2 << 64) | ((u128)r3 << 96);
#else
    u128 lo = 0;
    u64 ovf = 0;
    const u64 ytop = (u64)(y >> 64);
    while (x) {
      const int b = __builtin_ctz(x);
      x &= x - 1;
      lo ^= y << b;
      ovf ^= b ? (ytop >> (64 - b)) : 0;  // bits of y shifted past bit 127 (b <= 15)
    }
    return lo ^ c ^ (u128)((ovf << 7) ^ (ovf << 2) ^ (ovf << 1) ^ ovf);
#endif
  }
  static SCL_HD E from_le_word(const Ctx&, u128 w) { return w; }

  struct Acc {
    u128 v;
  };
  static SCL_HD Acc acc_zero() { return Acc{0}; }
  static SCL_HD void mac(const Ctx& c, Acc& acc, E a, E b) { acc.v ^= mul(c, a, b); }
  static SCL_HD void acc_add(const Ctx&, Acc& acc, E a) { acc.v ^= a; }
  static SCL_HD E acc_fold(const Ctx&, const Acc& acc) { return acc.v; }
  SCL_KC_IS_ACC()

  // Inverse a^(2^128 - 2) = (a^(2^127 - 1))^2 by Itoh-Tsujii: with b_k = a^(2^k - 1), b_(j+k) = b_j^(2^k) * b_k.  The chain
  // 1, 2, 3, 6, 12, 24, 48, 96, 120, 126, 127 takes 10 products and 127 squarings (254 products as a plain ladder); MUL is
  // the product to use (the LDS comb in the element-wise kernels, mul elsewhere).  inv(0) = 0 (callers flag the zero).
  template <class MUL>
  static SCL_HD E inv_chain(const Ctx& c, E a, MUL&& product) {
    // step s: cur <- cur^(2^k) * y with k = 1, 1, 3, 6, 12, 24, 48, 24, 6, 1 and y = a, a, then the step's own input five times,
    // then b24, b6, a.  One rolled loop (one copy of the product and of the squaring in the code, not ten and 127).
    E cur = a, b6 = 0, b24 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int s = 0; s < 10; ++s) {
      const int k = s < 2 ? 1 : s == 2 ? 3 : s == 3 ? 6 : s == 4 ? 12 : s == 5 ? 24 : s == 6 ? 48 : s == 7 ? 24 : s == 8 ? 6 : 1;
      const E y = (s < 2 || s == 9) ? a : s == 7 ? b24 : s == 8 ? b6 : cur;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
      for (int i = 0; i < k; ++i) cur = sqr(c, cur);
      cur = product(cur, y);
      if (s == 2) b6 = cur;
      if (s == 4) b24 = cur;
    }
    return sqr(c, cur);
  }
  static SCL_HD E inv(const Ctx& c, E a) {
    return inv_chain(c, a, [&](E x, E y) { return mul(c, x, y); });
  }
};

// --------------------------------------------------------------- 256-bit Montgomery primes (N = 4)
// The reference's mpn Montgomery family (include/scl/math/fields/ff_ops_gmp.h:44-314) instantiated for four 64-bit limbs:
//   FF<Secp256k1Scalar>  the order of the secp256k1 group (src/scl/math/fields/secp256k1_scalar.cc:47-135): the field
//                        Feldman / Pedersen VSS share over
//   FF<Secp256k1Field>   the prime the curve is defined over, 2^256 - 2^32 - 977 (src/scl/math/fields/secp256k1_field.cc:43-135)
// Four limbs holding the Montgomery residue x*2^256 mod p exactly like the reference's m_value.  PRM supplies the prime,
// -p^-1 mod 2^64 (low limb of RedParams::mc), 2^256 mod p, 2^512 mod p and the 32-bit limbs of 2^256 - p.
struct U256 {
  u64 w[4];
};

struct SecpOrderParams {  // p = FFFFFFFF FFFFFFFF FFFFFFFF FFFFFFFE BAAEDCE6 AF48A03B BFD25E8C D0364141
  enum { TAG = 4 };
  static SCL_HD u64 P(int i) {
    return i == 0 ? 0xBFD25E8CD0364141ull : i == 1 ? 0xBAAEDCE6AF48A03Bull : i == 2 ? 0xFFFFFFFFFFFFFFFEull : 0xFFFFFFFFFFFFFFFFull;
  }
  static constexpr u64 MC0 = 0x4B0DFF665588B13Full;  // secp256k1_scalar.cc:62-67
  static SCL_HD u64 ONE(int i) { return i == 0 ? 0x402DA1732FC9BEBFull : i == 1 ? 0x4551231950B75FC4ull : i == 2 ? 1ull : 0ull; }
  static SCL_HD u64 R2(int i) {
    return i == 0 ? 0x896CF21467D7D140ull : i == 1 ? 0x741496C20E7CF878ull : i == 2 ? 0xE697F5E45BCD07C6ull : 0x9D671CD581C69BC5ull;
  }
  static SCL_HD u32 C32(int j) {  // 2^256 - p = 1 45512319 50B75FC4 402DA173 2FC9BEBF
    return j == 0 ? 0x2FC9BEBFu : j == 1 ? 0x402DA173u : j == 2 ? 0x50B75FC4u : j == 3 ? 0x45512319u : j == 4 ? 1u : 0u;
  }
};

struct SecpFieldParams {  // p = 2^256 - 2^32 - 977 = FFFFFFFF ... FFFFFFFE FFFFFC2F
  enum { TAG = 7 };       // (5 and 6 are the rings' internal tags)
  static SCL_HD u64 P(int i) { return i == 0 ? 0xFFFFFFFEFFFFFC2Full : 0xFFFFFFFFFFFFFFFFull; }
  static constexpr u64 MC0 = 0xD838091DD2253531ull;  // secp256k1_field.cc:53-58 (low limb)
  static SCL_HD u64 ONE(int i) { return i == 0 ? 0x1000003D1ull : 0ull; }                                   // 2^256 mod p (the ONE of :93-94)
  static SCL_HD u64 R2(int i) { return i == 0 ? 0x000007A2000E90A1ull : i == 1 ? 1ull : 0ull; }              // (2^32 + 977)^2
  static SCL_HD u32 C32(int j) { return j == 0 ? 0x000003D1u : j == 1 ? 1u : 0u; }                          // 2^256 - p = 2^32 + 977
};

template <class PRM>
struct Mont256 {
  typedef U256 E;
  struct Ctx {};
  enum { LIMBS = 4, ACC_TERMS = 1 << 24, TAG = PRM::TAG, SMALL_BITS = 32 };  // SMALLX: nodes as plain integers < 2^32 (see Mont128)

  static SCL_HD u64 P(int i) { return PRM::P(i); }
  static constexpr u64 MC0 = PRM::MC0;
  static SCL_HD E make(u64 a, u64 b, u64 c, u64 d) {
    E r;
    r.w[0] = a; r.w[1] = b; r.w[2] = c; r.w[3] = d;
    return r;
  }
  static SCL_HD E zero() { return make(0, 0, 0, 0); }
  static SCL_HD E one(const Ctx&) { return make(PRM::ONE(0), PRM::ONE(1), PRM::ONE(2), PRM::ONE(3)); }  // 2^256 mod p
  static SCL_HD E r2() { return make(PRM::R2(0), PRM::R2(1), PRM::R2(2), PRM::R2(3)); }               // 2^512 mod p
  static SCL_HD bool is_zero(const E& a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3]) == 0; }
  static SCL_HD bool eq(const E& a, const E& b) {
    return ((a.w[0] ^ b.w[0]) | (a.w[1] ^ b.w[1]) | (a.w[2] ^ b.w[2]) | (a.w[3] ^ b.w[3])) == 0;
  }
  static SCL_HD u32 low32(const E& v) { return (u32)v.w[0]; }
  static SCL_HD E ld(const u64* p) { return make(p[0], p[1], p[2], p[3]); }
  static SCL_HD void st(u64* p, const E& v) {
    p[0] = v.w[0]; p[1] = v.w[1]; p[2] = v.w[2]; p[3] = v.w[3];
  }
  static SCL_HD bool geq_p(const E& a) {
    for (int i = 3; i >= 0; --i) {
      if (a.w[i] > P(i)) return true;
      if (a.w[i] < P(i)) return false;
    }
    return true;
  }
  static SCL_HD u64 add_n(E& r, const E& a, const E& b) {  // returns the carry out
    u128 c = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      c += (u128)a.w[i] + b.w[i];
      r.w[i] = (u64)c;
      c >>= 64;
    }
    return (u64)c;
  }
  static SCL_HD u64 sub_n(E& r, const E& a, const E& b) {  // returns the borrow out
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u128 d = (u128)a.w[i] - b.w[i] - borrow;
      r.w[i] = (u64)d;
      borrow = (u64)(d >> 64) & 1;
    }
    return borrow;
  }
  static SCL_HD E prime() { return make(P(0), P(1), P(2), P(3)); }
  static SCL_HD E add(const Ctx&, const E& a, const E& b) {  // montyModAdd, ff_ops_gmp.h:128-134
    E r;
    const u64 carry = add_n(r, a, b);
    if (carry || geq_p(r)) sub_n(r, r, prime());
    return r;
  }
  static SCL_HD E sub(const Ctx&, const E& a, const E& b) {  // montyModSub, ff_ops_gmp.h:142-148
    E r;
    if (sub_n(r, a, b)) add_n(r, r, prime());
    return r;
  }
  static SCL_HD E neg(const Ctx& c, const E& a) { return sub(c, zero(), a); }  // montyModNeg
  // montyModMul (ff_ops_gmp.h:174-191): interleaved Montgomery product a*b/2^256 mod p, valid for a < 2^256, b < p.
  // Column by column over 32-bit words (product scanning, as Mont128::mul): column k collects a_i b_j and m_i p_j for i + j = k in
  // one 96-bit accumulator -- a v_mad_u64_u32 and a carry per partial product, nothing widened or moved --, m_k = low word *
  // (-p^-1 mod 2^32) clears the low word, the accumulator moves down a word: 128 multiply-adds and 8 word products, about half
  // the vector instructions of the limb-by-limb form below (615 -> ~330 per product on gfx950).
  static SCL_HD E mul(const Ctx&, const E& a, const E& b) {
    u32 aw[8], bw[8], pw[8], m[8], t[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 4; ++i) {
      aw[2 * i] = (u32)a.w[i];
      aw[2 * i + 1] = (u32)(a.w[i] >> 32);
      bw[2 * i] = (u32)b.w[i];
      bw[2 * i + 1] = (u32)(b.w[i] >> 32);
      pw[2 * i] = (u32)P(i);
      pw[2 * i + 1] = (u32)(P(i) >> 32);
    }
    const u32 n0 = (u32)MC0;
    u64 lo = 0;
    u32 hi = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 8; ++k) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int j = 0; j < k; ++j) {
        lazy_col_mad(lo, hi, aw[j], bw[k - j]);
        lazy_col_mad(lo, hi, m[j], pw[k - j]);
      }
      lazy_col_mad(lo, hi, aw[k], bw[0]);
      m[k] = (u32)lo * n0;
      lazy_col_mad(lo, hi, m[k], pw[0]);   // the low word is 0 now
      lo = (lo >> 32) | ((u64)hi << 32);
      hi = 0;
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 8; k < 16; ++k) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int j = k - 7; j < 8; ++j) {
        lazy_col_mad(lo, hi, aw[j], bw[k - j]);
        lazy_col_mad(lo, hi, m[j], pw[k - j]);
      }
      t[k - 8] = (u32)lo;
      lo = (lo >> 32) | ((u64)hi << 32);
      hi = 0;
    }
    E r = make((u64)t[0] | ((u64)t[1] << 32), (u64)t[2] | ((u64)t[3] << 32), (u64)t[4] | ((u64)t[5] << 32), (u64)t[6] | ((u64)t[7] << 32));
    if (lo || geq_p(r)) sub_n(r, r, prime());   // lo: the seventeenth word, 0 or 1
    return r;
  }
  // the same product limb by limb over 64-bit words (the shape of the reference's loop; what mul was until round 6): kept for
  // tests/cxx/mont_mul_check.cc
  static SCL_HD E mul_by_limbs(const Ctx&, const E& a, const E& b) {
    u64 u[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u128 c = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c += (u128)a.w[j] * b.w[i] + u[j];
        u[j] = (u64)c;
        c >>= 64;
      }
      c += u[4];
      u[4] = (u64)c;
      u[5] = (u64)(c >> 64);
      const u64 q = MC0 * u[0];
      c = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c += (u128)q * P(j) + u[j];
        u[j] = (u64)c;
        c >>= 64;
      }
      c += u[4];
      u[4] = (u64)c;
      u[5] += (u64)(c >> 64);
      u[0] = u[1]; u[1] = u[2]; u[2] = u[3]; u[3] = u[4]; u[4] = u[5]; u[5] = 0;
    }
    E r = make(u[0], u[1], u[2], u[3]);
    if (u[4] || geq_p(r)) sub_n(r, r, prime());
    return r;
  }
  static SCL_HD E sqr(const Ctx& c, const E& a) { return mul(c, a, a); }
  static SCL_HD E to_mont(const Ctx& c, const E& x) { return mul(c, x, r2()); }            // montyIn
  static SCL_HD E from_mont(const Ctx& c, const E& a) { return mul(c, a, make(1, 0, 0, 0)); }  // montyRedc
  static SCL_HD E from_u64(const Ctx& c, u64 v) { return to_mont(c, make(v, 0, 0, 0)); }
  // fromBytes: 32 BIG-endian bytes -> value -> Montgomery form (montyFromBytes, ff_ops_gmp.h:279-290);
  // raw = the four little-endian 64-bit words of the byte string in memory order
  static SCL_HD E from_le_word(const Ctx& c, const E& raw) {
    return to_mont(c, make(__builtin_bswap64(raw.w[3]), __builtin_bswap64(raw.w[2]), __builtin_bswap64(raw.w[1]),
                           __builtin_bswap64(raw.w[0])));
  }
  static SCL_HD E raw_from_le_word(const E& raw) {  // the integer the bytes spell, not a residue
    return make(__builtin_bswap64(raw.w[3]), __builtin_bswap64(raw.w[2]), __builtin_bswap64(raw.w[1]),
                __builtin_bswap64(raw.w[0]));
  }
  // toBytes image (montyToBytes, ff_ops_gmp.h:298-314) as four little-endian words in memory order
  static SCL_HD E to_be_image(const Ctx& c, const E& a) {
    const E v = from_mont(c, a);
    return make(__builtin_bswap64(v.w[3]), __builtin_bswap64(v.w[2]), __builtin_bswap64(v.w[1]), __builtin_bswap64(v.w[0]));
  }
  // ---- products with SMALL plain integers --------------------------------------------------------------------
  // (x R) * v = (x v) R as integers: a residue in Montgomery form times a plain integer v stays in Montgomery form and
  // needs no Montgomery reduction at all, only "mod p" of a 288-bit integer -- which for these primes (2^256 minus a
  // short constant) is fold_top.  With v < 2^29 (the powers of the default nodes 1..n at small thresholds) a term is
  // eight 32 x 32 multiply-adds into per-limb 64-bit sums (seven terms cannot overflow them) against the 64 multiply-adds
  // and a share of an eight-round reduction of a full product: what k_share_small / k_share_blocked run on.
  struct SAcc {
    u64 a[8];
  };
  static SCL_HD void sacc_zero(SAcc& s) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s.a[j] = 0;
  }
  static SCL_HD void sacc_mac(SAcc& s, const E& c, u32 v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mad32(s.a[2 * i], (u32)c.w[i], v);
      mad32(s.a[2 * i + 1], (u32)(c.w[i] >> 32), v);
    }
  }
  // c0 + sum_j a_j 2^(32 j) mod p, canonical; the sum is < 2^288 (at most 7 terms of < p * 2^29, plus c0 < p)
  static SCL_HD E sacc_fold(const SAcc& s, const E& c0) {
    u32 r[9];
    u64 cy = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u64 t = (u64)(u32)s.a[j] + (j ? (s.a[j - 1] >> 32) : 0) + (u32)(c0.w[j >> 1] >> (32 * (j & 1))) + cy;  // < 2^35
      r[j] = (u32)t;
      cy = t >> 32;
    }
    r[8] = (u32)((s.a[7] >> 32) + cy);  // < 2^32: the whole sum is < 2^288
    const u32 again = fold_top(r, r[8]);
    fold_top(r, again);
    E v = make((u64)r[0] | ((u64)r[1] << 32), (u64)r[2] | ((u64)r[3] << 32), (u64)r[4] | ((u64)r[5] << 32),
               (u64)r[6] | ((u64)r[7] << 32));
    if (geq_p(v)) sub_n(v, v, prime());
    return v;
  }
  enum { SMALL_NODE_VALUE_BITS = 29 };
  static SCL_HD bool small_nodes_ok(const Ctx&) { return true; }
  // y * x + c for a plain x < 2^32, canonical (the Horner step between the groups of k_share_blocked)
  static SCL_HD E muladd_small(const Ctx&, const E& y, u32 x, const E& c) {
    SAcc s;
    sacc_zero(s);
    sacc_mac(s, y, x);
    return sacc_fold(s, c);
  }
  static SCL_HD E muladd_small_lazy(const E& y, u32 x, const E& c) { return muladd_small(Ctx{}, y, x, c); }
  static SCL_HD E muladd_small_lazy(const Ctx&, const E& y, u32 x, const E& c) { return muladd_small(Ctx{}, y, x, c); }
  static SCL_HD E canon(const E& r) { return r; }

  // Lazy accumulator: products summed unreduced (LazyCols), plain elements summed modularly beside them.
  // Fold: eight word-32 Montgomery rounds leave T' = (T + Q p) / 2^256 < (K + 1) p in nine limbs; the top limb
  // h comes back down as h * (2^256 - p) (p is 2^256 minus a 129-bit constant), twice at most, and one
  // conditional subtraction lands in [0,p) -- the same residue the reference reaches one montyModMul and one
  // montyModAdd at a time.
  struct Acc {
    LazyCols<8> c;
    E e;
  };
  static SCL_HD Acc acc_zero() {
    Acc a;
    lazy_zero(a.c);
    a.e = zero();
    return a;
  }
  static SCL_HD void mac(const Ctx&, Acc& acc, const E& a, const E& b) {
    u32 al[8], bl[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      al[2 * i] = (u32)a.w[i];
      al[2 * i + 1] = (u32)(a.w[i] >> 32);
      bl[2 * i] = (u32)b.w[i];
      bl[2 * i + 1] = (u32)(b.w[i] >> 32);
    }
    lazy_mac<8>(acc.c, al, bl);
  }
  static SCL_HD void acc_add(const Ctx& c, Acc& acc, const E& a) { acc.e = add(c, acc.e, a); }
  static SCL_HD u32 P32(int j) { return (u32)(P(j >> 1) >> (32 * (j & 1))); }
  static SCL_HD u32 C32(int j) { return PRM::C32(j); }  // 32-bit limbs of 2^256 - p
  static SCL_HD u32 fold_top(u32* r, u32 h) {  // r[0..8) += h * (2^256 - p); returns the carry out
    u64 cy = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u64 s = (u64)h * C32(j) + r[j] + cy;
      r[j] = (u32)s;
      cy = s >> 32;
    }
    return (u32)cy;
  }
  static SCL_HD E acc_fold(const Ctx& c, const Acc& acc) {
    u32 t[17];
    lazy_limbs<8, 17>(acc.c, t);
    lazy_redc<8, 8, 17>(t, [](int j) { return P32(j); }, (u32)MC0);
    u32* r = t + 8;
    const u32 again = fold_top(r, r[8]);
    fold_top(r, again);
    E v = make((u64)r[0] | ((u64)r[1] << 32), (u64)r[2] | ((u64)r[3] << 32), (u64)r[4] | ((u64)r[5] << 32),
               (u64)r[6] | ((u64)r[7] << 32));
    if (geq_p(v)) sub_n(v, v, prime());
    return add(c, v, acc.e);
  }
  static SCL_HD E table_scale(const Ctx&, const E& b) { return b; }
  static SCL_HD E acc_fold_scaled(const Ctx& c, const Acc& acc) { return acc_fold(c, acc); }
  SCL_KC_IS_ACC()

  // montyModInv (ff_ops_gmp.h:225-260): a^(p-2) by square-and-multiply from the top bit; inv(0) = 0
  // (fixed 4-bit windows as in Mont128::inv: 256 + 64 + 14 = 334 products against 256 + popcount(p - 2), ~450 for both primes)
  static SCL_HD E inv(const Ctx& c, const E& a) {
    const u64 e[4] = {P(0) - 2, P(1), P(2), P(3)};
    E tbl[16];
    tbl[0] = one(c);
    tbl[1] = a;
    for (int i = 2; i < 16; ++i) tbl[i] = mul(c, tbl[i - 1], a);
    E r = tbl[(unsigned)(e[3] >> 60) & 15u];
    for (int i = 62; i >= 0; --i) {
      r = sqr(c, sqr(c, sqr(c, sqr(c, r))));
      const unsigned nib = (unsigned)(e[i >> 4] >> (4 * (i & 15))) & 15u;
      if (nib) r = mul(c, r, tbl[nib]);
    }
    return is_zero(a) ? zero() : r;
  }
};

typedef Mont256<SecpOrderParams> Secp256k1Scalar;
typedef Mont256<SecpFieldParams> Secp256k1Field;

// ------------------------------------------------------------------------------- rings Z2k<K>
// scl::math::Z2k<K> of the reference (include/scl/math/z2k.h:39-320, include/scl/math/z2k/z2k_ops.h:32-150):
// arithmetic modulo 2^K on one 64-bit word (K <= 64) or one 128-bit word.  The reference wraps at the word and
// masks only when a value is compared, written or printed; results here are masked as they are produced, the
// same residue mod 2^K.  Not a field: only odd values have inverses (Newton iteration, z2k_ops.h:80-93).
template <typename W, int LIMBS_, int TAG_>
struct Z2kRing {
  typedef W E;
  struct Ctx {
    W mask;  // 2^K - 1
    int K;
  };
  enum { LIMBS = LIMBS_, ACC_TERMS = 1 << 30, TAG = TAG_, SMALL_BITS = 0 };

  static inline Ctx make_ctx(int K) {
    Ctx c;
    c.K = K;
    c.mask = K >= (int)(8 * sizeof(W)) ? ~(W)0 : (W)(((W)1 << K) - 1);
    return c;
  }
  static SCL_HD E zero() { return 0; }
  static SCL_HD E one(const Ctx& c) { return (E)1 & c.mask; }
  static SCL_HD E from_u64(const Ctx& c, u64 v) { return (E)v & c.mask; }
  static SCL_HD bool is_zero(E a) { return a == 0; }
  static SCL_HD bool eq(E a, E b) { return a == b; }
  static SCL_HD u32 low32(E v) { return (u32)v; }
  static SCL_HD E ld(const u64* p) {
    if constexpr (LIMBS_ == 1) return p[0];
    else return ((E)p[1] << 64) | p[0];
  }
  static SCL_HD void st(u64* p, E v) {
    p[0] = (u64)v;
    if constexpr (LIMBS_ == 2) p[1] = (u64)(v >> 64);
  }
  static SCL_HD E add(const Ctx& c, E a, E b) { return (a + b) & c.mask; }
  static SCL_HD E sub(const Ctx& c, E a, E b) { return (a - b) & c.mask; }
  static SCL_HD E mul(const Ctx& c, E a, E b) { return (a * b) & c.mask; }
  static SCL_HD E neg(const Ctx& c, E a) { return ((E)0 - a) & c.mask; }
  static SCL_HD E canon(E r) { return r; }
  static SCL_HD E from_le_word(const Ctx& c, E raw) { return raw & c.mask; }  // fromBytes, z2k_ops.h:107-112
  // invert (z2k_ops.h:80-93): z = 3v xor 2 is right to 5 bits, each step doubles that; even v has no inverse
  // (the caller raises "value not invertible modulo 2^K")
  static SCL_HD E inv(const Ctx& c, E v) {
    E z = (v * 3) ^ 2;
    for (int bits = 5; bits <= c.K; bits *= 2) z *= (E)2 - v * z;
    return z & c.mask;
  }
  struct Acc {
    E v;
  };
  static SCL_HD Acc acc_zero() { return Acc{0}; }
  static SCL_HD void mac(const Ctx&, Acc& acc, E a, E b) { acc.v += a * b; }
  static SCL_HD void acc_add(const Ctx&, Acc& acc, E a) { acc.v += a; }
  static SCL_HD E acc_fold(const Ctx& c, const Acc& acc) { return acc.v & c.mask; }
  SCL_KC_IS_ACC()
};
typedef Z2kRing<u64, 1, 5> Z2k64;
typedef Z2kRing<u128, 2, 6> Z2k128;

}  // namespace sclhip
