// tests/cxx/gf128_host_check.cc -- Gf128::mul (4-bit-window comb), Gf128::sqr (bit spread + fold) and Gf128::inv (Itoh-Tsujii
// chain) of detail/field.hpp on the host against a bit-serial shift-xor multiplier written here (x^128 + x^7 + x^2 + x + 1):
// random operands, the sparse and the all-ones corners, a * inv(a) = 1, inv(0) = 0, and the 254-product ladder the chain
// replaces; and Mersenne127's dedicated squaring against its general product.  Host only; built and run by tests/test_cxx_api.py.
#include <cstdint>
#include <cstdio>
#include <random>
#include "scl_hip/detail/field.hpp"
using namespace sclhip;
static u128 ref_mul(u128 a, u128 b) {
  u128 r = 0;
  for (int i = 127; i >= 0; --i) {
    const bool top = (r >> 127) != 0;
    r <<= 1;
    if (top) r ^= 0x87;
    if ((b >> i) & 1) r ^= a;
  }
  return r;
}
static u128 ref_inv(u128 a) {  // a^(2^128 - 2) = prod_{i=1..127} a^(2^i)
  u128 r = 1, sq = a;
  for (int i = 1; i < 128; ++i) {
    sq = ref_mul(sq, sq);
    r = ref_mul(r, sq);
  }
  return a ? r : 0;
}
int main() {
  std::mt19937_64 g(20261004);
  auto rnd = [&] { return ((u128)g() << 64) | g(); };
  const Gf128::Ctx c{};
  long bad = 0, n = 0;
  const u128 ones = ~(u128)0;
  const u128 corners[] = {0, 1, 2, 0x87, (u128)1 << 127, ones, ones >> 1, ((u128)1 << 127) | 1, (u128)0xFFFFFFFFull << 96, (u128)0xF << 124};
  for (u128 a : corners)
    for (u128 b : corners) {
      bad += Gf128::mul(c, a, b) != ref_mul(a, b);
      ++n;
    }
  for (u128 a : corners) {
    bad += Gf128::sqr(c, a) != ref_mul(a, a);
    bad += Gf128::inv(c, a) != ref_inv(a);
    n += 2;
  }
  for (int it = 0; it < 200000; ++it) {
    const u128 a = rnd(), b = rnd();
    bad += Gf128::mul(c, a, b) != ref_mul(a, b);
    bad += Gf128::sqr(c, a) != ref_mul(a, a);
    n += 2;
  }
  for (int it = 0; it < 2000; ++it) {
    const u128 a = rnd();
    const u128 i = Gf128::inv(c, a);
    bad += i != ref_inv(a);
    bad += Gf128::mul(c, a, i) != 1;
    n += 2;
  }
  // Mersenne127's three-product squaring (what its Fermat inverse runs on) against the general product
  {
    const M127::Ctx mc{};
    const u128 P = M127::P();
    const u128 mcorners[] = {0, 1, 2, P - 1, P - 2, (u128)1 << 64, ((u128)1 << 64) - 1, (u128)1 << 126, (u128)0x7FFFFFFFFFFFFFFFull << 64};
    for (u128 a : mcorners) {
      bad += M127::sqr(mc, a) != M127::mul(mc, a, a);
      ++n;
    }
    for (int it = 0; it < 500000; ++it) {
      const u128 a = rnd() % P;
      bad += M127::sqr(mc, a) != M127::mul(mc, a, a);
      ++n;
    }
  }
  std::printf("gf128 / m127 host check: %ld comparisons, %ld mismatches\n", n, bad);
  return bad != 0;
}
