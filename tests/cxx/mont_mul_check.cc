// tests/cxx/mont_mul_check.cc -- Mont128::mul and Mont256<...>::mul (detail/field.hpp: the Montgomery product interleaved over 32-bit words, the
// form of the reference's montyModMul, ff_ops_gmp.h:174-191) against (1) the two-step form it replaced, redc(mulwide(a, b)),
// and (2) plain long arithmetic: a b R^-1 mod p with the 256-bit product reduced bit by bit and R^-1 applied as 128 halvings
// modulo p.  Operands at the edges (0, 1, p - 1, p - 2, all-ones words, R mod p) and random ones; moduli of every shape: the
// default 2^128 - 159, 2^127 - 1, the smallest (3, 7), one below a word boundary, random odd ones.  Host only; built and run
// by tests/test_cxx_api.py -- the kernels compile the same function.  Second half: Mont256::mul (both secp256k1 primes), the same
// column-wise form over eight words, against the limb-by-limb form it replaced (mul_by_limbs), operands at the edges -- 0, 1,
// p - 1, R mod p, all-ones (a < 2^256 need not be reduced: montyIn hands it raw values) -- and random.
#include <cstdint>
#include <cstdio>
#include <random>

#include "scl_hip/detail/field.hpp"
using namespace sclhip;
typedef unsigned __int128 u128_t;

static u128_t addmod(u128_t a, u128_t b, u128_t p) {
  const u128_t t = a + b;
  return (t < a || t >= p) ? t - p : t;
}
// a * b mod p by double-and-add (a, b < p)
static u128_t mulmod(u128_t a, u128_t b, u128_t p) {
  u128_t r = 0;
  for (int i = 127; i >= 0; --i) {
    r = addmod(r, r, p);
    if ((b >> i) & 1) r = addmod(r, a, p);
  }
  return r;
}
// x / 2 mod p for odd p
static u128_t halve(u128_t x, u128_t p) {
  if (!(x & 1)) return x >> 1;
  const u128_t s = x + p;            // may carry out of 128 bits
  const bool carry = s < x;
  return (s >> 1) | ((u128_t)(carry ? 1 : 0) << 127);
}

int main() {
  std::mt19937_64 g(20261005);
  auto rnd = [&]() { return ((u128_t)g() << 64) | g(); };
  u128_t moduli[14] = {((u128_t)0xFFFFFFFFFFFFFFFFull << 64) | 0xFFFFFFFFFFFFFF61ull, (((u128_t)1) << 127) - 1, 3, 7, 0xFFFFFFFBull,
                       ((u128_t)1 << 64) - 59, ((u128_t)1 << 96) - 17, ((u128_t)0x8000000000000000ull << 64) | 1,
                       ((u128_t)0xFFFFFFFFFFFFFFFFull << 64) | 0xFFFFFFFFFFFFFFFFull, 0, 0, 0, 0, 0};
  for (int i = 9; i < 14; ++i) moduli[i] = (rnd() >> (i == 9 ? 0 : (g() % 100))) | 1 | (i == 9 ? (u128_t)1 << 127 : 0);
  long n = 0, bad = 0;
  for (u128_t p : moduli) {
    if (p < 3) p = 5;
    const Mont128::Ctx c = Mont128::make_ctx(p);
    const u128_t edge[8] = {0, 1 % p, p - 1, (p - 2) % p, c.one, c.r2, (p >> 1), ((u128_t)0xFFFFFFFF00000000ull << 64 | 0xFFFFFFFFull) % p};
    for (int it = 0; it < 60000; ++it) {
      const u128_t a = it < 64 ? edge[it & 7] : rnd() % p, b = it < 64 ? edge[it >> 3] : (it % 3 ? rnd() % p : a);
      const u128_t got = Mont128::mul(c, a, b);
      const u128_t two_step = Mont128::redc(c, mulwide(a, b));
      bool ok = got == two_step && got < p;
      if (it < 2000) {             // (the long arithmetic is slow: a sample)
        u128_t want = mulmod(a, b, p);
        for (int k = 0; k < 128; ++k) want = halve(want, p);
        ok = ok && got == want;
      }
      ++n;
      if (!ok) {
        if (bad < 5) std::printf("MISMATCH modulus %016llx%016llx it %d\n", (unsigned long long)(p >> 64), (unsigned long long)p, it);
        ++bad;
      }
    }
  }
  auto check256 = [&](auto f) {
    using F = decltype(f);
    typename F::Ctx c{};
    auto rnd256 = [&]() { return F::make(g(), g(), g(), g()); };
    const typename F::E p = F::prime();
    typename F::E pm1 = p;
    pm1.w[0] -= 1;
    const typename F::E edge[6] = {F::zero(), F::make(1, 0, 0, 0), pm1, F::one(c), F::make(~0ull, ~0ull, ~0ull, ~0ull), F::make(0, 0, 0, 1ull << 63)};
    for (int it = 0; it < 300000; ++it) {
      typename F::E a = it < 36 ? edge[it % 6] : rnd256(), b = it < 36 ? edge[it / 6] : rnd256();
      if (F::geq_p(b)) F::sub_n(b, b, p);         // b < p; a stays raw
      const typename F::E x = F::mul(c, a, b), y = F::mul_by_limbs(c, a, b);
      ++n;
      if (!F::eq(x, y) || F::geq_p(x)) {
        if (bad < 5) std::printf("MISMATCH 256-bit product it %d\n", it);
        ++bad;
      }
    }
  };
  check256(Secp256k1Scalar{});
  check256(Secp256k1Field{});
  std::printf("%ld products, %ld mismatches\n", n, bad);
  return bad != 0;
}
