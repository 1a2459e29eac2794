// tests/cxx/mont128_fold_check.cc -- Mont128::sacc_fold (detail/field.hpp: the small-node sharing's ONE Barrett step with ONE
// conditional subtraction, quotient estimate floor(floor(S / 2^96) * floor(2^162 / p) / 2^66)) against plain long arithmetic: the
// 192-bit sum c0 + sum c_k v_k built by hand and reduced bit by bit.  Extreme operands (p - 1 everywhere, v = 2^29 - 1, seven
// terms) and five full-width moduli including 2^128 - 1 and 2^127 + 29 (odd; only the arithmetic matters here).  Host only;
// built and run by tests/test_cxx_api.py.
#include <cstdio>
#include <cstdint>
#include <random>
#include "scl_hip/detail/field.hpp"
using namespace sclhip;
typedef unsigned __int128 u128_t;
// S mod p for S = c0 + sum c_k v_k computed with 192-bit (hi:u64, lo:u128) long arithmetic, bit-serial reduction
static u128_t ref_fold(u128_t p, const u128_t* c, const uint32_t* v, int t, u128_t c0) {
  // accumulate into (hi, lo)
  u128_t lo = c0; uint64_t hi = 0;
  for (int k = 0; k < t; ++k) {
    // c[k] * v[k]: split c into two 64-bit halves
    u128_t a0 = (u128_t)(uint64_t)c[k] * v[k], a1 = (u128_t)(uint64_t)(c[k] >> 64) * v[k];
    u128_t add_lo = a0 + (a1 << 64); uint64_t add_hi = (uint64_t)(a1 >> 64) + (add_lo < a0 ? 1 : 0);
    u128_t nl = lo + add_lo; hi += add_hi + (nl < lo ? 1 : 0); lo = nl;
  }
  // reduce (hi, lo) mod p bit-serially: r = 0; for bits from top: r = 2r + bit mod p
  u128_t r = 0;
  for (int b = 191; b >= 0; --b) {
    int bit = b >= 128 ? (int)((hi >> (b - 128)) & 1) : (int)((lo >> b) & 1);
    bool top = (r >> 127) != 0;
    r = (r << 1) | (u128_t)bit;
    if (top || r >= p) r -= p;
  }
  return r;
}
int main() {
  std::mt19937_64 g(12345);
  const u128_t primes[] = {((u128_t)0xFFFFFFFFFFFFFFFFull << 64) | 0xFFFFFFFFFFFFFF61ull, ((u128_t)1 << 127) + 29 /* odd, not nec. prime: arithmetic only */,
                           ((u128_t)0x8000000000000000ull << 64) | 1, ((u128_t)0xFFFFFFFFFFFFFFFFull << 64) | 0xFFFFFFFFFFFFFFFFull,
                           ((u128_t)0xC000000000000001ull << 64) | 0x7ull};
  long bad = 0, n = 0;
  for (u128_t p : primes) {
    Mont128::Ctx ctx = Mont128::make_ctx(p);
    for (int it = 0; it < 400000; ++it) {
      int t = 1 + (int)(g() % 7);
      u128_t c[7]; uint32_t v[7];
      for (int k = 0; k < 7; ++k) {
        u128_t x = ((u128_t)g() << 64) | g();
        int mode = (int)(g() % 4);
        c[k] = mode == 0 ? p - 1 : mode == 1 ? p - 1 - (g() % 3) : x % p;
        v[k] = mode == 0 ? ((1u << 29) - 1) : (uint32_t)(g() % (1u << 29));
      }
      u128_t c0 = (g() % 3 == 0) ? p - 1 : ((((u128_t)g() << 64) | g()) % p);
      Mont128::SAcc s; Mont128::sacc_zero(s);
      for (int k = 0; k < t; ++k) Mont128::sacc_mac(s, c[k], v[k]);
      u128_t got = Mont128::sacc_fold(ctx, s, c0), want = ref_fold(p, c, v, t, c0);
      ++n; if (got != want) { if (bad < 5) std::printf("MISMATCH t=%d\n", t); ++bad; }
    }
  }
  std::printf("%ld folds, %ld mismatches\n", n, bad);
  return bad != 0;
}
