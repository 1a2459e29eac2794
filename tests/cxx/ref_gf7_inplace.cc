// tests/cxx/ref_gf7_inplace.cc -- the reference's OWN worked example of its field plug-in boundary,
// /root/reference/test/scl/gf7.h + gf7.cc (a traits struct and the ff:: specialisations for the integers modulo 7), compiled
// IN PLACE -- the two files are handed to the compiler where they lie, nothing of them is copied -- against this repo's
// mirror headers: an include directory in which `scl` is a link to include/scl_hip makes the reference's
// `#include "scl/math/fields/ff_ops.h"` resolve to include/scl_hip/math/fields/ff_ops.h (tests/test_cxx_api.py builds
// it; CPU only, skipped where the reference is absent).  What it shows: a field defined the reference's way compiles
// unchanged and runs through FF / Vector / Polynomial / computeLagrangeBasis / scl::ss of the mirror.
// Cases: test/scl/ss/test_shamir.cc:144-160 (Berlekamp-Welch, the Wikipedia example) and the identities of
// test/scl/math/test_ff.cc:64-227 that need no hex parsing (gf7.cc has no convertTo(string)).
#include <cstdio>

#include "gf7.h"      // -I /root/reference/test/scl
#include "scl/scl.h"  // -I <alias directory>: include/scl_hip/scl.h

using namespace scl;
using FF = math::FF<test::GaloisField7>;

static int g_fail = 0, g_checks = 0;
#define REQUIRE(cond)                                                 \
  do {                                                                \
    ++g_checks;                                                       \
    if (!(cond)) {                                                    \
      ++g_fail;                                                       \
      std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
    }                                                                 \
  } while (0)

int main() {
  static_assert(!math::OnDevice<FF>, "a field without TAG / Impl has no kernels: every member takes the generic host path");
  // test_shamir.cc:144-160
  math::Vector<FF> bs = {FF(1), FF(5), FF(3), FF(6), FF(3), FF(2), FF(2)};
  math::Vector<FF> corrected = {FF(1), FF(6), FF(3), FF(6), FF(1), FF(2), FF(2)};
  auto s = ss::shamirRecoverC(bs);
  REQUIRE(s.err.evaluate(FF(2)) == FF::zero());
  REQUIRE(s.err.evaluate(FF(5)) == FF::zero());
  for (std::size_t i = 0; i < bs.size(); ++i) REQUIRE(s.f.evaluate(FF((int)i + 1)) == corrected[i]);
  // field identities over every pair of residues
  for (int a = 0; a < 7; ++a)
    for (int b = 0; b < 7; ++b) {
      const FF x(a), y(b);
      REQUIRE(x + y == FF((a + b) % 7));
      REQUIRE(x - y == FF(a - b));
      REQUIRE(x * y == FF(a * b));
      REQUIRE(x + y - y == x);
      if (b) REQUIRE(x / y * y == x);
    }
  REQUIRE(FF(-1) == FF(6) && FF(3).negated() == FF(4) && math::exp(FF(3), 6) == FF::one());
  bool threw = false;
  try {
    (void)FF(0).inverse();
  } catch (const std::logic_error& e) {
    threw = std::string(e.what()) == "0 not invertible modulo prime";
  }
  REQUIRE(threw);
  // sharing on a PRG and the three reconstructions
  auto prg = util::PRG::create("gf7 in place");
  for (int v = 0; v < 7; ++v) {
    // seven shares of a degree-2 polynomial: n = 3t + 1, what shamirRecoverC(shares) corrects t errors of (node 7 is 0 in GF(7):
    // the seventh share is the secret itself, as in the Wikipedia example's seven points)
    const auto shares = ss::shamirSecretShare(FF(v), 2, 7, prg);
    REQUIRE(shares.size() == 7);
    REQUIRE(ss::shamirRecoverP(shares) == FF(v));
    REQUIRE(ss::shamirRecoverD(shares.subVector(0, 5), 2) == FF(v));
    auto bad = shares;
    bad[3] += FF(1);
    REQUIRE(ss::shamirRecoverC(bad).f.constantTerm() == FF(v));
    const auto add = ss::additiveShare(FF(v), 4, prg);
    REQUIRE(add.sum() == FF(v));
  }
  // Vector members and the Lagrange basis
  const auto u = math::Vector<FF>::random(50, prg), w = math::Vector<FF>::random(50, prg);
  REQUIRE(u.add(w).subtract(w).equals(u));
  FF dot = FF::zero();
  for (std::size_t i = 0; i < 50; ++i) dot += u[i] * w[i];
  REQUIRE(u.dot(w) == dot);
  const auto basis = math::computeLagrangeBasis(math::Vector<FF>::range(1, 5), 0);
  FF one = FF::zero();
  for (const auto& l : basis) one += l;
  REQUIRE(one == FF::one());
  std::printf("ref_gf7_inplace: %d checks, %d failures\n", g_checks, g_fail);
  return g_fail ? 1 : 0;
}
