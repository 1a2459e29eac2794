// tests/cxx/test_scl_api.cc -- the reference's own tests for the hot path, restated against the C++
// mirror (include/scl_hip/), so that they read like test/scl/math/test_ff.cc, test_vector.cc,
// test_poly.cc, test_matrix.cc, test/scl/ss/test_shamir.cc, test_additive.cc and
// test/scl/util/test_prg.cc.  Known-answer values come from the real reference
// (tests/golden/golden_v1.json).
//
//   test_scl_api --host-only   scalar FF / Polynomial / hex / Lagrange-table cases (no GPU needed)
//   test_scl_api               everything (needs a GPU: Vector, Matrix, PRG and ss:: run HIP kernels)
#include <cstdio>
#include <cstring>
#include <functional>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include <scl_hip/scl.h>

#include "gf7_field.h"

using namespace scl;

static int g_fail = 0, g_checks = 0;
#define REQUIRE(...)                                                         \
  do {                                                                       \
    ++g_checks;                                                              \
    if (!(__VA_ARGS__)) {                                                           \
      ++g_fail;                                                              \
      std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #__VA_ARGS__);       \
    }                                                                        \
  } while (0)
#define REQUIRE_THROWS_MSG(expr, Type, msg)                                  \
  do {                                                                       \
    ++g_checks;                                                              \
    bool ok_ = false;                                                        \
    try {                                                                    \
      (void)(expr);                                                          \
    } catch (const Type& e_) {                                               \
      ok_ = std::string(e_.what()) == (msg);                                 \
      if (!ok_) std::printf("  wrong message '%s'\n", e_.what());            \
    } catch (...) {                                                          \
    }                                                                        \
    if (!ok_) {                                                              \
      ++g_fail;                                                              \
      std::printf("  FAILED %s:%d: %s should throw %s(\"%s\")\n", __FILE__, __LINE__, #expr, #Type, msg); \
    }                                                                        \
  } while (0)

// where a case runs: HOST = scalars and host containers only (no device call at any threshold); GPU = uses the batch
// API (hip::DeviceVector / hip::ShareMatrix): kernels, always; BOTH = host containers of a field with kernels: runs on the
// host below hip::hostThreshold() -- so also without a GPU -- and is run a second time with the threshold at 0, when
// every Vector / Matrix / PRG member goes through its kernel
enum Where { HOST = 0, GPU = 1, BOTH = 2 };
struct Case {
  const char* name;
  int where;
  std::function<void()> fn;
};
static std::vector<Case>& cases() {
  static std::vector<Case> c;
  return c;
}
struct Reg {
  Reg(const char* n, int g, std::function<void()> f) { cases().push_back({n, g, std::move(f)}); }
};
#define TEST_CASE(id, name, gpu) \
  static void id();              \
  static Reg reg_##id(name, gpu, id); \
  static void id()

using F61 = math::Fp<61>;
using F127 = math::Fp<127>;

// ---------------------------------------------------------------------------- scalars (host)
template <typename FF>
static void field_identities(const char* seed_tag) {
  // test/scl/math/test_ff.cc:64-227 -- identities on pseudo-random operands.  Operands come from
  // FF::fromString over a simple counter hash so that this case needs no PRG (no GPU).
  std::uint64_t st = 0x9E3779B97F4A7C15ull ^ (std::uint64_t)seed_tag[0];
  auto next = [&]() {
    char buf[33];
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    std::uint64_t a = st;
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    std::snprintf(buf, sizeof buf, "%016llx%016llx", (unsigned long long)a, (unsigned long long)st);
    return FF::fromString(buf);
  };
  const FF zero = FF::zero(), one = FF::one();
  for (int rep = 0; rep < 50; ++rep) {
    FF a = next(), b = next(), c = next();
    if (a == zero) a = one;
    if (b == zero) b = one;
    REQUIRE(a + b == b + a);
    REQUIRE(a * b == b * a);
    REQUIRE(c * (a + b) == c * a + c * b);
    REQUIRE((a + b) + c == a + (b + c));
    REQUIRE(a * a.inverse() == one);
    REQUIRE(a - a == zero);
    REQUIRE(-(a - b) == b - a);
    REQUIRE(a / b == (b / a).inverse());
    REQUIRE(a + zero == a);
    REQUIRE(a * one == a);
    REQUIRE(a * zero == zero);
    REQUIRE(-a + a == zero);
    unsigned char buf[FF::byteSize()];
    a.write(buf);
    REQUIRE(FF::read(buf) == a);
    REQUIRE(math::exp(a, 0) == one);
    REQUIRE(math::exp(a, 1) == a);
    REQUIRE(math::exp(a, 5) == a * a * a * a * a);
    FF d = a;
    REQUIRE(d++ == a);
    REQUIRE(d == a + one);
    REQUIRE(--d == a);
  }
  REQUIRE_THROWS_MSG(zero.inverse(), std::logic_error, "0 not invertible modulo prime");  // test_ff.cc:168-171
  REQUIRE(FF(-1) + one == zero);
  REQUIRE(FF(-5) == zero - FF(5));
}

TEST_CASE(ff_mersenne61, "FF<Mersenne61> identities + metadata", HOST) {
  field_identities<F61>("a");
  REQUIRE(std::string(F61::name()) == "Mersenne61");  // test_mersenne61.cc:28-33
  REQUIRE(F61::bitSize() == 61);
  REQUIRE(F61::byteSize() == 8);
  REQUIRE(F61::fromString("7b") == F61(0x7b));        // :41-47
  REQUIRE(F61(0x41621e).toString() == "41621e");
  REQUIRE(F61::fromString("1fffffffffffffff") == F61(0));  // p == 0
  REQUIRE_THROWS_MSG(F61::fromString("abc"), std::invalid_argument, "odd-length hex string");
  REQUIRE_THROWS_MSG(F61::fromString("zz"), std::invalid_argument, "encountered invalid hex character");
  std::stringstream ss;
  ss << F61(255);
  REQUIRE(ss.str() == "ff");
}

TEST_CASE(ff_mersenne127, "FF<Mersenne127> identities + metadata", HOST) {
  field_identities<F127>("b");
  REQUIRE(std::string(F127::name()) == "Mersenne127");  // test_mersenne127.cc:28-33
  REQUIRE(F127::bitSize() == 127);
  REQUIRE(F127::byteSize() == 16);
  REQUIRE(F127::fromString("80000000000000000000000000000000") == F127(1));  // :41-42: 2^127 = 1
  const auto v = F127::fromString("58797a14d0653d22a05c11c60e1aacf4");     // :44-45
  REQUIRE(v.toString() == "58797a14d0653d22a05c11c60e1aacf4");
  REQUIRE((std::is_same_v<math::Fp<62>, F127>));  // fp.h:34-43 selection
  REQUIRE((std::is_same_v<math::Fp<1>, F61>));
}

TEST_CASE(ff_secp256k1_scalar, "FF<Secp256k1Scalar> identities + metadata", HOST) {
  using FS = math::FF<math::ff::Secp256k1Scalar>;
  field_identities<FS>("s");
  REQUIRE(std::string(FS::name()) == "secp256k1_order");
  REQUIRE(FS::byteSize() == 32);
  REQUIRE(FS(123).toString() == "7b");
  REQUIRE(FS(-1).toString() == "fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364140");  // p - 1
  REQUIRE(FS::fromString("fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141") == FS(0));  // p = 0
  REQUIRE(FS::fromString("7b") == FS(123));
  REQUIRE(FS::fromString("abc") == FS(0xabc));  // odd length is padded, not rejected (ff_ops_gmp.h:383-386)
  REQUIRE_THROWS_MSG(FS::fromString(std::string(66, 'f')), std::invalid_argument, "hex string too large to parse");
}

TEST_CASE(ff_secp256k1_field, "FF<Secp256k1Field> identities + metadata", HOST) {
  // the prime the curve is defined over, 2^256 - 2^32 - 977 (src/scl/math/fields/secp256k1_field.cc:43-135)
  using FP = math::FF<math::ff::Secp256k1Field>;
  field_identities<FP>("p");
  REQUIRE(std::string(FP::name()) == "secp256k1_field");
  REQUIRE(FP::byteSize() == 32 && FP::bitSize() == 256);
  REQUIRE(FP(123).toString() == "7b");
  REQUIRE(FP(-1).toString() == "fffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc2e");  // p - 1
  REQUIRE(FP::fromString("fffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc2f") == FP(0));  // p = 0
  REQUIRE(FP::fromString("fffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc30") == FP(1));
  REQUIRE(FP(2).inverse() * FP(2) == FP::one());
  REQUIRE(FP::one().value().w[0] == 0x1000003D1ull && FP::one().value().w[1] == 0);  // the ONE of secp256k1_field.cc:93-94
  unsigned char buf[32];
  FP(-2).write(buf);
  REQUIRE(buf[0] == 0xff && buf[27] == 0xfe && buf[31] == 0x2d && FP::read(buf) == FP(-2));  // value, big-endian
}

TEST_CASE(ff_mont128_reference_kats, "FF<Mont128>: strings, ints, bytes as the reference's Montgomery templates give them at N = 2", HOST) {
  // known answers from tests/golden/golden_mont128.json: include/scl/math/fields/ff_ops_gmp.h compiled at two limbs
  // (oracle/ref_harness.cc, field tag 2), modulus 2^128 - 159.  value().w = the Montgomery image, like FF::m_value.
  using FM = math::FF<math::ff::Mont128>;
  auto image = [](const FM& x) {
    std::uint64_t w[2];
    x.toLimbs(w);
    char buf[40];
    if (w[1]) std::snprintf(buf, sizeof buf, "%llx%016llx", (unsigned long long)w[1], (unsigned long long)w[0]);
    else std::snprintf(buf, sizeof buf, "%llx", (unsigned long long)w[0]);
    return std::string(buf);
  };
  REQUIRE(image(FM(1)) == "9f" && image(FM(-1)) == "fffffffffffffffffffffffffffffec2" && image(FM(123)) == "4c65");
  REQUIRE(image(FM(-2147483647)) == "ffffffffffffffffffffffb080000000" && image(FM(65536)) == "9f0000");
  REQUIRE(image(FM::fromString("7b")) == "4c65" && FM::fromString("7b").toString() == "7b");
  REQUIRE(FM::fromString("") == FM(0) && FM::fromString("00") == FM(0));
  REQUIRE(FM::fromString("ffffffffffffffffffffffffffffff61") == FM(0));            // p = 0
  REQUIRE(FM::fromString("ffffffffffffffffffffffffffffffff").toString() == "9e");  // 2^128 - 1 = 158 mod p
  REQUIRE(FM::fromString("abc") == FM(0xabc));                                     // odd length is padded (ff_ops_gmp.h:383-386)
  // limbs are cut from the LEFT in 16-digit pieces; a short last piece becomes limb 0 as it stands (ff_ops_gmp.h:388-395)
  REQUIRE(FM::fromString("0123456789ABCDEFabcdef").toString() == "123456789abcdef0000000000abcdef");
  REQUIRE(image(FM::fromString("0123456789ABCDEFabcdef")) == "b4e81b4e81b4e771000000006ab4e771");
  REQUIRE(FM::fromString("123456789abcdef0f").toString() == "123456789abcdef000000000000000f");
  REQUIRE_THROWS_MSG(FM::fromString("zz"), std::invalid_argument, "encountered invalid hex character");
  REQUIRE_THROWS_MSG(FM::fromString(std::string(33, '1')), std::invalid_argument, "hex string too large to parse");
  unsigned char be[16];
  for (int i = 0; i < 16; ++i) be[i] = (unsigned char)i;                           // montyFromBytes: big-endian value
  REQUIRE(image(FM::read(be)) == "a03fdf7f1ebe5dfd9d3cdc7c1bbb51");
  unsigned char back[16];
  FM::read(be).write(back);
  REQUIRE(std::memcmp(back, be, 16) == 0);
  REQUIRE_THROWS_MSG(FM(0).inverse(), std::logic_error, "0 not invertible modulo prime");
}

TEST_CASE(ff_plugins, "plug-in fields: Mont128, GF(2^128) identities", HOST) {
  field_identities<math::FF<math::ff::Mont128>>("c");
  using G = math::FF<math::ff::GF2_128>;
  G a = G::fromString("0123456789abcdeffedcba9876543210"), b = G::fromString("ffeeddccbbaa99887766554433221100");
  REQUIRE(a + a == G::zero());
  REQUIRE(a * b == b * a);
  REQUIRE(a * a.inverse() == G::one());
  REQUIRE((a + b) * a == a * a + b * a);
  // x^127 * x = x^128 = x^7 + x^2 + x + 1
  REQUIRE(G::fromString("80000000000000000000000000000000") * G(2) == G(0x87));
  // a published product: GHASH step X_1 = C * H of test case 2 of the GCM specification (McGrew & Viega, appendix B),
  // H = 66e94bd4.., C = 0388dace.., X_1 = 5e2ec746.., each block's 128 bits reversed (GCM keeps x^0 in its top bit);
  // tests/test_plugin_field_pins.py holds the vectors and the dictionary
  const G gh = G::fromString("74d42c539a5f3211dc3451f72bd29766"), gc = G::fromString("1e7f4d8e9d4314cf49c56d06735b11c0");
  REQUIRE(gc * gh == G::fromString("ed7bcaca160da13411460e8962e3747a"));
  REQUIRE(G::fromString("ed7bcaca160da13411460e8962e3747a") / gh == gc);
}

TEST_CASE(small_forms, "detail/field.hpp: small-constant and lazy forms == generic mul/add", HOST) {
  // the device kernels' shortcuts (muladd_small, lazy accumulators) against plain mul/add on the host
  std::uint64_t st = 88172645463325252ull;
  auto rnd = [&]() {
    st ^= st << 13;
    st ^= st >> 7;
    st ^= st << 17;
    return st;
  };
  using namespace sclhip;
  for (int rep = 0; rep < 20000; ++rep) {
    const u32 x = rep < 64 ? (u32)(0xFFFFFFFFu >> (rep % 32)) : (u32)rnd();
    {
      const M61::Ctx c{};
      const u64 y = rep == 0 ? M61::P - 1 : rnd() % M61::P, a = rep == 1 ? M61::P - 1 : rnd() % M61::P;
      REQUIRE(M61::muladd_small(c, y, x, a) == M61::add(c, M61::mul(c, y, (u64)x), a));
      // lazy chain: three steps without intermediate canonicalisation, worst-case inputs included
      {
        u64 lz = y, want = y;
        for (int i = 0; i < 3; ++i) {
          lz = M61::muladd_small_lazy(lz, x, a);
          want = M61::add(c, M61::mul(c, want, (u64)x), a);
        }
        REQUIRE(M61::canon(lz) == want);
      }
      M61::Acc acc = M61::acc_zero();
      u64 want = 0;
      for (int i = 0; i < 64; ++i) {
        const u64 p = i & 1 ? M61::P - 1 : y, q = i & 2 ? M61::P - 1 : a;
        M61::mac(c, acc, p, q);
        want = M61::add(c, want, M61::mul(c, p, q));
      }
      REQUIRE(M61::acc_fold(c, acc) == want);
      REQUIRE(M61::from_le_word(c, rnd() | (rep < 3 ? ~0ull : 0)) < M61::P);
    }
    {
      const M127::Ctx c{};
      const u128 P = M127::P();
      const u128 y = rep == 0 ? P - 1 : (((u128)rnd() << 64) | rnd()) % P, a = rep == 1 ? P - 1 : (((u128)rnd() << 64) | rnd()) % P;
      REQUIRE(M127::muladd_small(c, y, x, a) == M127::add(c, M127::mul(c, y, (u128)x), a));
      {
        u128 lz = y, want = y;
        for (int i = 0; i < 3; ++i) {
          lz = M127::muladd_small_lazy(lz, x, a);
          want = M127::add(c, M127::mul(c, want, (u128)x), a);
        }
        REQUIRE(M127::canon(lz) == want);
      }
      M127::Acc acc = M127::acc_zero();
      u128 want = 0;
      for (int i = 0; i < 40; ++i) {
        const u128 p = i & 1 ? P - 1 : y, q = i & 2 ? P - 1 : a;
        M127::mac(c, acc, p, q);
        want = M127::add(c, want, M127::mul(c, p, q));
      }
      REQUIRE(M127::acc_fold(c, acc) == want);
    }
    {
      const Gf128::Ctx c{};
      const u128 y = ((u128)rnd() << 64) | rnd(), a = ((u128)rnd() << 64) | rnd();
      const u32 xs = x & 0xFFFF;
      REQUIRE(Gf128::muladd_small(c, y, xs, a) == Gf128::add(c, Gf128::mul(c, y, (u128)xs), a));
    }
  }
}

TEST_CASE(prepared_constants, "detail/field.hpp: kc_make / kmac / kacc_fold == mul/add, K_TERMS worst case", HOST) {
  // the table-driven kernels' multiply-accumulate against a prepared constant, on the host
  std::uint64_t st = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    st ^= st << 13;
    st ^= st >> 7;
    st ^= st << 17;
    return st;
  };
  using namespace sclhip;
  {
    const M61::Ctx c{};
    for (int rep = 0; rep < 300; ++rep) {
      M61::KAcc acc = M61::kacc_zero();
      u64 want = 0;
      const int terms = rep < 4 ? (int)M61::K_TERMS : 1 + (int)(rnd() % 64);
      for (int i = 0; i < terms; ++i) {
        // rep 0: every term the largest canonical product; rep 1: largest limbs (x need not be canonical)
        const u64 k = rep == 0 ? M61::P - 1 : rep == 1 ? 0x1FFFFFFFFFFFFFull : rnd() % M61::P;
        const u64 x = rep == 0 ? M61::P - 1 : rep == 1 ? ~0ull : rep == 2 ? rnd() : rnd() % M61::P;
        M61::kmac(c, acc, M61::kc_make(c, k), x);
        want = M61::add(c, want, M61::mul(c, k, M61::from_le_word(c, x)));
      }
      REQUIRE(M61::kacc_fold(c, acc) == want);
    }
  }
  {
    const M127::Ctx c{};
    const u128 P = M127::P();
    for (int rep = 0; rep < 300; ++rep) {
      M127::KAcc acc = M127::kacc_zero();
      u128 want = 0;
      const int terms = rep < 4 ? (int)M127::K_TERMS : 1 + (int)(rnd() % 48);
      for (int i = 0; i < terms; ++i) {
        const u128 r1 = ((u128)rnd() << 64) | rnd(), r2 = ((u128)rnd() << 64) | rnd();
        const u128 k = rep == 0 ? P - 1 : r1 % P;
        const u128 x = rep == 0 ? P - 1 : rep == 1 ? ~(u128)0 : rep == 2 ? r2 : r2 % P;
        M127::kmac(c, acc, M127::kc_make(c, k), x);
        want = M127::add(c, want, M127::mul(c, k, M127::from_le_word(c, x)));
      }
      REQUIRE(M127::kacc_fold(c, acc) == want);
    }
    // all limbs of the prepared constant at their maximum, all limbs of x at theirs, K_TERMS times
    M127::KC full;
    for (int i = 0; i < 24; ++i) full.w[i] = 0x3FFFFFu;
    M127::KAcc acc = M127::kacc_zero();
    for (int i = 0; i < (int)M127::K_TERMS; ++i) M127::kmac(c, acc, full, ~(u128)0);
    for (int j = 0; j < 6; ++j) REQUIRE(acc.c[j] == (u64)M127::K_TERMS * 4 * 0x3FFFFFull * 0xFFFFFFFFull);
    M61::KC full61;
    for (int i = 0; i < 6; ++i) full61.w[i] = 0x1FFFFFu;
    M61::KAcc a61 = M61::kacc_zero();
    for (int i = 0; i < (int)M61::K_TERMS; ++i) M61::kmac(M61::Ctx{}, a61, full61, ~0ull);
    for (int j = 0; j < 3; ++j) REQUIRE(a61.c[j] == (u64)M61::K_TERMS * 2 * 0x1FFFFFull * 0xFFFFFFFFull);
  }
}

TEST_CASE(poly_host, "Polynomial: create / evaluate / arithmetic", HOST) {
  using P = math::Polynomial<F61>;
  // test/scl/math/test_poly.cc:64-71: 4 + 5x + x^2 at 5 = 54
  const auto p = P::create({F61(4), F61(5), F61(1)});
  REQUIRE(p.evaluate(F61(5)) == F61(54));
  REQUIRE(p.degree() == 2);
  // trailing zeros are trimmed (poly.h:178-198)
  REQUIRE(P::create({F61(1), F61(2), F61(0), F61(0)}).degree() == 1);
  REQUIRE(P::create({F61(0), F61(0)}).isZero());
  const auto q = P::create({F61(1), F61(1)});
  const auto pq = p.multiply(q);  // (4+5x+x^2)(1+x) = 4 + 9x + 6x^2 + x^3
  REQUIRE(pq[0] == F61(4));
  REQUIRE(pq[1] == F61(9));
  REQUIRE(pq[2] == F61(6));
  REQUIRE(pq[3] == F61(1));
  const auto qr = pq.divide(q);
  REQUIRE(qr[1].isZero());
  REQUIRE(qr[0].evaluate(F61(7)) == p.evaluate(F61(7)));
  REQUIRE(p.add(q).evaluate(F61(3)) == p.evaluate(F61(3)) + q.evaluate(F61(3)));
  REQUIRE(p.subtract(p).isZero());
}

TEST_CASE(lagrange_host, "computeLagrangeBasis (host table code)", HOST) {
  // closed form for nodes 1..n at 0: (-1)^(i-1) C(n,i)   
  const auto lb = math::computeLagrangeBasis(math::Vector<F61>::range(1, 11), 0);
  const int want[10] = {10, -45, 120, -210, 252, -210, 120, -45, 10, -1};
  for (int i = 0; i < 10; ++i) REQUIRE(lb[i] == F61(want[i]));
  REQUIRE_THROWS_MSG(math::computeLagrangeBasis(math::Vector<F61>{F61(1), F61(2), F61(2)}, 0), std::logic_error,
                     "0 not invertible modulo prime");
  REQUIRE_THROWS_MSG(math::Vector<F61>::range(3, 1), std::invalid_argument, "invalid range");
  REQUIRE(math::Vector<F61>::range(2, 2).empty());
}

TEST_CASE(matrix_host, "Matrix: identity / transpose / invert on scalars", HOST) {
  using M = math::Matrix<F61>;
  auto m = M::fromVector(2, 2, {F61(1), F61(2), F61(3), F61(4)});
  REQUIRE(m.transpose()(0, 1) == F61(3));
  REQUIRE(M::identity(3).isIdentity());
  const auto inv = m.invert();
  // m * m^-1 = I checked entry by entry with scalars (test_matrix.cc:326-331 uses multiply)
  REQUIRE(m(0, 0) * inv(0, 0) + m(0, 1) * inv(1, 0) == F61(1));
  REQUIRE(m(0, 0) * inv(0, 1) + m(0, 1) * inv(1, 1) == F61(0));
  REQUIRE(m(1, 0) * inv(0, 0) + m(1, 1) * inv(1, 0) == F61(0));
  REQUIRE(m(1, 0) * inv(0, 1) + m(1, 1) * inv(1, 1) == F61(1));
  REQUIRE_THROWS_MSG(M(0, 2), std::invalid_argument, "n or m cannot be 0");
  REQUIRE_THROWS_MSG(M::fromVector(2, 2, {F61(1)}), std::invalid_argument, "invalid dimensions");
  REQUIRE_THROWS_MSG(M(2, 3).invert(), std::invalid_argument, "cannot invert non-square matrix");
}

// ---------------------------------------------------------------------------- GPU-backed
// ---------------------------------------------------------------------------- a user-defined field (host)
// The reference's field plug-in boundary (include/scl/math/fields/ff_ops.h:35-118): gf7_field.h defines GF(7) with the
// traits struct and the eleven specialisations only; everything below runs through the generic host paths.
using G7 = math::FF<usr::Gf7>;
static_assert(!math::OnDevice<G7> && math::OnDevice<F61>, "GF(7) has no kernels; the built-in fields do");

TEST_CASE(gf7_scalars, "user-defined field GF(7): FF<Gf7> over the eleven ff:: specialisations", HOST) {
  REQUIRE(std::string(G7::name()) == "GF(7)" && G7::byteSize() == 1 && G7::bitSize() == 8);
  for (int a = 0; a < 7; ++a)
    for (int b = 0; b < 7; ++b) {
      REQUIRE(G7(a) + G7(b) == G7((a + b) % 7));
      REQUIRE(G7(a) - G7(b) == G7((a - b + 7) % 7));
      REQUIRE(G7(a) * G7(b) == G7((a * b) % 7));
      if (b) REQUIRE((G7(a) / G7(b)) * G7(b) == G7(a));
    }
  REQUIRE(G7(-1) == G7(6) && G7(-15) == G7(6) && G7(7) == G7::zero() && -G7(3) == G7(4) && -G7(0) == G7(0));
  REQUIRE(G7(3).inverse() == G7(5) && G7(6).inverse() == G7(6));
  REQUIRE_THROWS_MSG(G7(0).inverse(), std::logic_error, "0 not invertible modulo prime");  // test_ff.cc:168-171
  REQUIRE(G7::fromString("0a") == G7(3) && G7(5).toString() == "5");
  unsigned char byte = 0;
  G7(4).write(&byte);
  REQUIRE(byte == 4 && G7::read(&byte) == G7(4));
  byte = 200;
  REQUIRE(G7::read(&byte) == G7(200 % 7));
  REQUIRE(math::exp(G7(3), 6) == G7(1) && math::exp(G7(3), 0) == G7(1) && math::exp(G7(3), 2) == G7(2));
  G7 x = G7::one();
  for (int i = 0; i < 6; ++i) x++;
  REQUIRE(x == G7::zero());  // the x++ walk of shamirSecretShare wraps in a 7-element field
}

TEST_CASE(gf7_containers, "user-defined field GF(7): Vector / Matrix / Polynomial / Lagrange on the host", HOST) {
  using V = math::Vector<G7>;
  const V a = {G7(1), G7(2), G7(3)}, b = {G7(6), G7(5), G7(4)};
  REQUIRE(a.add(b) == V({G7(0), G7(0), G7(0)}) && a.subtract(b) == V({G7(2), G7(4), G7(6)}));
  REQUIRE(a.multiplyEntryWise(b) == V({G7(6), G7(3), G7(5)}) && a.dot(b) == G7((6 + 10 + 12) % 7));
  REQUIRE(a.sum() == G7(6) && a.scalarMultiply(G7(3)) == V({G7(3), G7(6), G7(2)}) && a != b);
  REQUIRE_THROWS_MSG(a.add(V({G7(1)})), std::invalid_argument, "Vec sizes mismatch");
  REQUIRE(V::range(5, 9) == V({G7(5), G7(6), G7(0), G7(1)}));
  // Vandermonde rows x_i^j and their product with a coefficient column = Horner at the same nodes
  const auto vm = math::Matrix<G7>::vandermonde(4, 3);
  REQUIRE(vm(0, 0) == G7(1) && vm(2, 2) == G7(2) && vm(3, 2) == G7(2) && vm(3, 1) == G7(4));
  const V coeff = {G7(5), G7(1), G7(3)};  // 5 + x + 3x^2
  const auto p = math::Polynomial<G7>::create(coeff);
  const auto evals = vm.multiply(coeff);
  for (int i = 0; i < 4; ++i) REQUIRE(evals[i] == p.evaluate(G7(i + 1)));
  const auto prod = vm.multiply(coeff.toColumnMatrix());
  REQUIRE(prod.rows() == 4 && prod.cols() == 1 && prod(1, 0) == p.evaluate(G7(2)));
  // interpolation at 0 from three points of a degree-2 polynomial
  const auto lb = math::computeLagrangeBasis(V::range(1, 4), 0);
  REQUIRE(math::innerProd<G7>(evals.begin(), evals.begin() + 3, lb.begin()) == G7(5));
  REQUIRE(ss::shamirRecoverP(evals.subVector(3)) == G7(5));
  REQUIRE_THROWS_MSG(math::computeLagrangeBasis(V({G7(1), G7(8)}), 0), std::logic_error, "0 not invertible modulo prime");
  // error detection (shamir.h:116-139), short overload: t = 2, five shares at nodes 1..5; it re-derives share t + 1 .. 2t - 1
  // (index 3 only) from the first three -- the last share is never looked at (note E)
  V five;
  for (int i = 1; i <= 5; ++i) five.toStlVector().push_back(p.evaluate(G7(i)));
  REQUIRE(ss::shamirRecoverD(five, 2) == G7(5));
  V bad = five;
  bad[3] += G7(1);
  REQUIRE_THROWS_MSG(ss::shamirRecoverD(bad, 2), std::logic_error, "error detected during recovery");
  bad = five;
  bad[4] += G7(1);
  REQUIRE(ss::shamirRecoverD(bad, 2) == G7(5));
  REQUIRE_THROWS_MSG(ss::shamirRecoverD(five.subVector(3), 2), std::logic_error, "not enough shares provided to detect errors");
  const auto him = math::Matrix<G7>::hyperInvertible(2, 2);
  REQUIRE(him.rows() == 2 && !(him(0, 0) == him(1, 0) && him(0, 1) == him(1, 1)));
  REQUIRE(math::Matrix<G7>::vandermonde(3, 3).multiply(math::Matrix<G7>::vandermonde(3, 3).invert()).isIdentity());
}

TEST_CASE(gf7_berlekamp_welch, "user-defined field GF(7): the Wikipedia Berlekamp-Welch case (test_shamir.cc:144-160)", HOST) {
  // https://en.wikipedia.org/wiki/Berlekamp%E2%80%93Welch_algorithm#Example: seven shares of a degree-2 sharing,
  // the ones at nodes 2 and 5 wrong
  const math::Vector<G7> received = {G7(1), G7(5), G7(3), G7(6), G7(3), G7(2), G7(2)};
  const math::Vector<G7> corrected = {G7(1), G7(6), G7(3), G7(6), G7(1), G7(2), G7(2)};
  const auto s = ss::shamirRecoverC(received);
  REQUIRE(s.err.evaluate(G7(2)) == G7::zero());
  REQUIRE(s.err.evaluate(G7(5)) == G7::zero());
  REQUIRE(s.err.degree() == 2 && s.f.degree() == 2);
  for (std::size_t i = 0; i < received.size(); ++i) REQUIRE(s.f.evaluate(G7((int)i + 1)) == corrected[i]);
  // nothing to correct: the locator is 1
  const auto clean = ss::shamirRecoverC(corrected);
  REQUIRE(clean.err.degree() == 0 && clean.err.constantTerm() == G7(1) && clean.f.constantTerm() == s.f.constantTerm());
  // three errors are one too many for t = 2
  math::Vector<G7> broken = corrected;
  broken[0] = G7(0), broken[3] = G7(0), broken[6] = G7(0);
  REQUIRE_THROWS_MSG(ss::shamirRecoverC(broken), std::logic_error, "could not correct shares");
  // the same algebra through explicit nodes (shamir.h:202-250)
  const auto s2 = ss::shamirRecoverC(received, math::Vector<G7>::range(1, 8));
  REQUIRE(s2.f.constantTerm() == s.f.constantTerm());
}

// the text forms and argument checks the reference's tests pin: "Matrix ToString" (test_matrix.cc:110-126), "Vector to string"
// (test_vector.cc:102-110), "Polynomial to string" (test_poly.cc:73-85), "PRG invalid calls" / "PRG truncate seed on create"
// (test_prg.cc:106-125), "Z2k truncation" (test_z2k.cc:167-190)
TEST_CASE(text_and_checks, "text forms and argument checks pinned by the reference's tests", HOST) {
  using Matrix = math::Matrix<F61>;
  Matrix m(3, 2);
  const int vals[6] = {1, 2, 44444, 5, 6, 7};
  for (int i = 0; i < 6; ++i) m(i / 2, i % 2) = F61(vals[i]);
  const std::string expected = "\n[    1  2 ]\n[ ad9c  5 ]\n[    6  7 ]";
  REQUIRE(m.toString() == expected);
  std::stringstream ss;
  ss << m;
  REQUIRE(ss.str() == expected);
  REQUIRE(Matrix().toString() == "[ EMPTY MATRIX ]");
  const math::Vector<F61> v0 = {F61(1), F61(2), F61(3)}, v1 = {F61(2), F61(123), F61(5)};
  REQUIRE(v0.toString() == "[1, 2, 3]" && v1.toString() == "[2, 7b, 5]");
  std::stringstream sv;
  sv << v0;
  REQUIRE(sv.str() == "[1, 2, 3]" && math::Vector<F61>().toString() == "[ EMPTY VECTOR ]");
  const auto p = math::Polynomial<F61>::create(math::Vector<F61>{F61(4), F61(5), F61(1)});
  REQUIRE(p.toString() == "f(x) = 4 + 5x + 1x^2" && p.toString("g", "y") == "g(y) = 4 + 5y + 1y^2");
  std::stringstream sp;
  sp << p;
  REQUIRE(sp.str() == "f(x) = 4 + 5x + 1x^2");
  auto prg = util::PRG::create();
  std::vector<unsigned char> buf(10);
  REQUIRE_THROWS_MSG(prg.next(buf, 11), std::invalid_argument, "n exceeds buffer.size()");
  auto prg0 = util::PRG::create("0123456789abcdef_bar"), prg1 = util::PRG::create("0123456789abcdef_foo");
  REQUIRE(prg0.next(100) == prg1.next(100));  // seeds are cut to 16 bytes
  using Z32 = math::Z2k<32>;
  const Z32 a(0x34abcdef11), b(0x00abcdef11);
  REQUIRE(a == b);
  unsigned char ba[Z32::byteSize() + 2] = {0}, bb[Z32::byteSize() + 2] = {0};
  ba[4] = ba[5] = bb[4] = bb[5] = 0xff;
  a.write(ba);
  b.write(bb);
  REQUIRE(std::memcmp(ba, bb, sizeof ba) == 0 && ba[4] == 0xff && ba[5] == 0xff);  // write touches byteSize() bytes only
}

// seri::Serializer (test/scl/serialization/test_serializer.cc:60-123 for the types of the path; ff.h:355-391, vector.h:595-629,
// matrix.h:910-963, array.h:424-455): sizes, byte layout, round trips
TEST_CASE(serializer_host, "seri::Serializer: trivially copyable / std::vector / FF / Vector / Matrix / Array images", HOST) {
  using Si = seri::Serializer<int>;
  unsigned char ibuf[sizeof(int)];
  REQUIRE(Si::sizeOf(1234) == sizeof(int) && Si::write(1234, ibuf) == sizeof(int));
  int back = 0;
  REQUIRE(Si::read(back, ibuf) == sizeof(int) && back == 1234);
  // "Serialization vector of vectors"-style nesting: every level carries its u32 count
  using Svv = seri::Serializer<std::vector<std::vector<int>>>;
  const std::vector<std::vector<int>> vv = {{1, 2, 3}, {}, {7}};
  REQUIRE(Svv::sizeOf(vv) == 4 + (4 + 12) + 4 + (4 + 4));
  std::vector<unsigned char> vbuf(Svv::sizeOf(vv));
  REQUIRE(Svv::write(vv, vbuf.data()) == vbuf.size());
  std::vector<std::vector<int>> ww;
  REQUIRE(Svv::read(ww, vbuf.data()) == vbuf.size() && ww == vv);
  // "Serialization Vec": std::vector<Fp<61>> = count + 3 x 8 bytes, canonical little-endian values
  using Sv = seri::Serializer<std::vector<F61>>;
  const std::vector<F61> v = {F61(1), F61(2), F61(-1)};
  REQUIRE(Sv::sizeOf(v) == 4 + 3 * F61::byteSize());
  unsigned char buf[28];
  REQUIRE(Sv::write(v, buf) == 28);
  const unsigned char want[28] = {3, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0xfe, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0x1f};
  REQUIRE(std::memcmp(buf, want, 28) == 0);  // p - 1 = 0x1ffffffffffffffe
  std::vector<F61> w;
  REQUIRE(Sv::read(w, buf) == 28 && w == v);
  // Vector<FF> is its std::vector; Matrix adds u32 rows, u32 cols in front
  using SV = seri::Serializer<math::Vector<F61>>;
  const math::Vector<F61> mv(v);
  unsigned char buf2[28];
  REQUIRE(SV::sizeOf(mv) == 28 && SV::write(mv, buf2) == 28 && std::memcmp(buf2, want, 28) == 0);
  math::Vector<F61> mv2;
  REQUIRE(SV::read(mv2, buf2) == 28 && mv2 == mv);
  using SM = seri::Serializer<math::Matrix<F127>>;
  auto prg = util::PRG::create("seri");
  const auto M = math::Matrix<F127>::random(3, 2, prg);
  REQUIRE(SM::sizeOf(M) == 8 + 4 + 6 * 16);
  std::vector<unsigned char> mbuf(SM::sizeOf(M));
  REQUIRE(SM::write(M, mbuf.data()) == mbuf.size());
  REQUIRE(mbuf[0] == 3 && mbuf[4] == 2 && mbuf[8] == 6);
  math::Matrix<F127> M2;
  REQUIRE(SM::read(M2, mbuf.data()) == mbuf.size() && M2 == M);
  // "Array serialization" (test_array.cc:64-80) through the Serializer itself; no count: N is in the type
  using F = math::FF<math::ff::Secp256k1Scalar>;
  using SA = seri::Serializer<math::Array<F, 3>>;
  const auto prod = math::Array<F, 3>::random(prg);
  unsigned char abuf[96];
  REQUIRE(SA::sizeOf(prod) == 96 && SA::write(prod, abuf) == 96);
  math::Array<F, 3> p2;
  REQUIRE(p2 != prod && SA::read(p2, abuf) == 96 && p2 == prod);
  // a ring element: Z2k<K> writes byteSize() = ceil(K / 8) bytes
  using Z = math::Z2k<62>;
  using SZ = seri::Serializer<Z>;
  const Z z(0x123456789abcdefll);
  unsigned char zbuf[8];
  Z z2;
  REQUIRE(SZ::sizeOf(z) == Z::byteSize() && SZ::write(z, zbuf) == Z::byteSize() && SZ::read(z2, zbuf) == Z::byteSize() && z2 == z);
}

TEST_CASE(serializer_gpu, "seri::Serializer<Vector / Matrix> == the device wire kernels, byte for byte", GPU) {
  // what scl_hip_wire_pack / _pack_matrix write from SoA rows in HBM is what the host Serializer writes, for every field width
  auto check = [](auto tag, std::size_t n, const char* seed) {
    using FF = decltype(tag);
    auto prg = util::PRG::create(seed);
    const auto v = math::Vector<FF>::random(n, prg);
    using SV = seri::Serializer<math::Vector<FF>>;
    std::vector<unsigned char> host(SV::sizeOf(v));
    SV::write(v, host.data());
    const hip::DeviceVector<FF> dv(v.toStlVector());
    REQUIRE(scl_hip_wire_size(FF::Field::TAG, n) == host.size());
    hip::DeviceBuffer raw(host.size());
    hip::check(scl_hip_wire_pack(FF::Field::TAG, static_cast<unsigned char*>(raw.get()), dv.data(), n, nullptr));
    std::vector<unsigned char> dev(host.size());
    hip::check(scl_hip_memcpy_d2h(dev.data(), raw.get(), dev.size(), nullptr));
    REQUIRE(dev == host);
    // and back through the device unpack
    hip::DeviceVector<FF> du(n);
    std::size_t got = 0;
    hip::check(scl_hip_wire_unpack(FF::Field::TAG, du.data(), n, static_cast<const unsigned char*>(raw.get()), dev.size(), &got, nullptr));
    REQUIRE(got == n && math::Vector<FF>(du.toHost()) == v);
  };
  check(F61{}, 1001, "seri-61");
  check(F127{}, 333, "seri-127");
  check(math::FF<math::ff::Secp256k1Scalar>{}, 77, "seri-secp");
}

// test/scl/math/test_la.cc restated over GF(7): the helpers of solveLinearSystem (matrix.h:585-828) keep the reference's answers,
// degenerate inputs included
TEST_CASE(linalg_gf7, "LinAlg: getPivotInColumn / findFirstNonZeroRow / extractSolution / solve / hasSolution (test_la.cc)", HOST) {
  using FF = math::FF<usr::Gf7>;
  using Matrix = math::Matrix<FF>;
  using Vector = math::Vector<FF>;
  const FF zero = FF::zero(), one = FF::one();
  {  // "LinAlg GetPivot"
    Matrix A = Matrix::fromVector(3, 3, {one, zero, one, zero, one, zero, zero, zero, zero});
    REQUIRE(math::getPivotInColumn(A, 2) == -1);
    REQUIRE(math::getPivotInColumn(A, 1) == 1);
    REQUIRE(math::getPivotInColumn(A, 0) == 0);
    A(2, 2) = one;
    REQUIRE(math::getPivotInColumn(A, 2) == 2);
    Matrix B(2, 2);
    REQUIRE(math::getPivotInColumn(B, 0) == -1);
  }
  {  // "LinAlg FindFirstNonZeroRow"
    Matrix A = Matrix::fromVector(3, 3, {one, zero, one, zero, one, zero, zero, zero, zero});
    REQUIRE(math::findFirstNonZeroRow(A) == 1);
    A(2, 1) = one;
    REQUIRE(math::findFirstNonZeroRow(A) == 2);
  }
  {  // "LinAlg ExtractSolution"
    Matrix A = Matrix::fromVector(3, 4, {one, zero, zero, FF(3), zero, one, zero, FF(5), zero, zero, one, FF(2)});
    REQUIRE(math::extractSolution(A).equals(Vector{FF(3), FF(5), FF(2)}));
    Matrix B = Matrix::fromVector(3, 4, {FF(1), FF(3), FF(1), FF(2), FF(0), FF(0), FF(1), FF(4), FF(0), FF(0), FF(0), FF(0)});
    REQUIRE(math::extractSolution(B).equals(Vector{FF(4), FF(4), FF(0)}));
    Matrix C(3, 4);
    C(1, 0) = FF(2);
    REQUIRE(math::extractSolution(C).equals(Vector{zero, one, zero}));
  }
  {  // "LinAlg Solve random" (over GF(7) a random 10 x 10 matrix is singular about one time in seven: seeds until one is not)
    int solved = 0;
    for (int seed = 0; seed < 12 && solved < 3; ++seed) {
      auto prg = util::PRG::create("la-" + std::to_string(seed));
      Matrix A = Matrix::random(10, 10, prg);
      Vector b = Vector::random(10, prg), x(10);
      if (!math::solveLinearSystem(x, A, b)) continue;
      ++solved;
      REQUIRE(A.multiply(x.toColumnMatrix()).equals(b.toColumnMatrix()));
      REQUIRE(A.multiply(A.invert()).isIdentity());
    }
    REQUIRE(solved == 3);
  }
  {  // "LinAlg malformed systems"
    Vector x;
    Matrix A(2, 2);
    Vector b(3);
    REQUIRE_THROWS_MSG(math::solveLinearSystem(x, A, b), std::invalid_argument, "malformed system of equations");
  }
  {  // "LinAlg HasSolution"
    Matrix A(2, 3);
    REQUIRE(!math::hasSolution(A, true));   // an all-zero row: no unique solution
    REQUIRE(math::hasSolution(A, false));   // .. but a free variable: many
    A(0, 2) = FF(1);
    REQUIRE(!math::hasSolution(A, false));  // 0 = 1
  }
  {  // rowReduceInPlace + createAugmentedMatrix: [A | I] -> [I | A^-1], as Matrix::invert does it there (matrix.h:830-850)
    Matrix A = Matrix::fromVector(3, 3, {FF(2), FF(1), FF(0), FF(1), FF(3), FF(1), FF(0), FF(1), FF(4)});
    auto aug = math::createAugmentedMatrix(A, Matrix::identity(3));
    math::rowReduceInPlace(aug);
    Matrix inv(3, 3);
    for (std::size_t i = 0; i < 3; ++i)
      for (std::size_t j = 0; j < 3; ++j) {
        REQUIRE(aug(i, j) == (i == j ? one : zero));
        inv(i, j) = aug(i, 3 + j);
      }
    REQUIRE(inv == A.invert() && A.multiply(inv).isIdentity());
    Matrix R = A;
    math::swapRows(R, 0, 2);
    math::multiplyRow(R, 1, FF(2));
    math::addRows(R, 0, 1, FF(3));   // row 0 = old row 2 + 3 * (2 * old row 1)
    REQUIRE(R(1, 0) == FF(2) * A(1, 0) && R(0, 2) == A(2, 2) + FF(6) * A(1, 2) && R(2, 1) == A(0, 1));
    REQUIRE(A.byteSize() == 9 * FF::byteSize());
    REQUIRE_THROWS_MSG(Matrix(2, 3).resize(4, 2), std::invalid_argument, "cannot resize matrix");
    REQUIRE(Matrix(2, 3).resize(3, 2).rows() == 3);
  }
}

TEST_CASE(prg_gpu, "util::PRG stream", BOTH) {
  // test/scl/util/test_prg.cc:49-125 (determinism, reset, seed truncation) + known answers
  auto prg = util::PRG::create("shamir passive");
  const auto b = prg.next(32);
  const unsigned char want[32] = {0x96, 0x5e, 0xf3, 0x3d, 0x3c, 0x2c, 0xdc, 0x34, 0x67, 0x66, 0x2f, 0xf2, 0x20, 0x77, 0xce, 0xde,
                                  0x04, 0x8c, 0xd9, 0xa3, 0x69, 0xc0, 0xf7, 0xec, 0x8b, 0x1e, 0x76, 0xf3, 0x6b, 0xae, 0xc7, 0x1a};
  REQUIRE(std::memcmp(b.data(), want, 32) == 0);
  auto z = util::PRG::create();
  const auto zb = z.next(16);
  const unsigned char zw[16] = {0x77, 0x27, 0xa8, 0x00, 0x4e, 0xa0, 0xc9, 0x70, 0x84, 0x41, 0x89, 0x3d, 0x28, 0x08, 0xca, 0x94};
  REQUIRE(std::memcmp(zb.data(), zw, 16) == 0);
  prg.reset();
  REQUIRE(prg.next(32) == b);
  // next(n) burns whole blocks: 8 bytes then 8 bytes come from blocks 0 and 1
  prg.reset();
  const auto p0 = prg.next(8), p1 = prg.next(8);
  REQUIRE(std::memcmp(p0.data(), want, 8) == 0);
  REQUIRE(std::memcmp(p1.data(), want + 16, 8) == 0);
  // seeds longer than 16 bytes are truncated (test_prg.cc:115-125)
  auto l1 = util::PRG::create("0123456789abcdefXYZ"), l2 = util::PRG::create("0123456789abcdef");
  REQUIRE(l1.next(64) == l2.next(64));
  std::vector<unsigned char> small(4);
  REQUIRE_THROWS_MSG(prg.next(small, 5), std::invalid_argument, "n exceeds buffer.size()");
}

TEST_CASE(vector_gpu, "Vector<FF> members", BOTH) {
  using Vec = math::Vector<F61>;
  // test/scl/math/test_vector.cc:39-203
  const Vec v0{F61(1), F61(2), F61(3)}, v1{F61(2), F61(123), F61(5)};
  REQUIRE(v0.dot(v1) == F61(263));  // :81-84
  REQUIRE(v0.add(v1) == Vec{F61(3), F61(125), F61(8)});
  REQUIRE(v1.subtract(v0) == Vec{F61(1), F61(121), F61(2)});
  REQUIRE(v0.multiplyEntryWise(v1) == Vec{F61(2), F61(246), F61(15)});
  REQUIRE(v0.scalarMultiply(F61(2)) == Vec{F61(2), F61(4), F61(6)});
  REQUIRE(v0.sum() == F61(6));
  REQUIRE(v0 != v1);
  REQUIRE(!(v0 == Vec{F61(1), F61(2)}));
  REQUIRE_THROWS_MSG(v0.add(Vec{F61(1)}), std::invalid_argument, "Vec sizes mismatch");
  REQUIRE_THROWS_MSG(v0.dot(Vec{F61(1)}), std::invalid_argument, "Vec sizes mismatch");
  REQUIRE_THROWS_MSG(v0.subVector(2, 1), std::logic_error, "invalid range");
  REQUIRE(v0.subVector(1, 3) == Vec{F61(2), F61(3)});
  REQUIRE(Vec::range(1, 4) == v0);
  REQUIRE(v0.toString() == "[1, 2, 3]");
  REQUIRE(Vec{}.toString() == "[ EMPTY VECTOR ]");
  Vec w = v0;
  w.addInPlace(v1).subtractInPlace(v1);
  REQUIRE(w == v0);
  // Vector::random: known answer from the reference
  auto prg = util::PRG::create("shamir passive");
  const auto r = Vec::random(4, prg);
  REQUIRE(r[0] == F61::fromString("14dc2c3c3df35e97"));
  REQUIRE(r[1] == F61::fromString("1ece7720f22f666d"));
  REQUIRE(r[2] == F61::fromString("0cf7c069a3d98c0b"));
  REQUIRE(r[3] == F61::fromString("1ac7ae6bf3761e8b"));
  REQUIRE(prg.counter() == 2);  // 32 bytes = 2 blocks
  // a bigger vector: sum/dot against scalar loops
  auto prg2 = util::PRG::create("big");
  const auto a = Vec::random(5000, prg2), b = Vec::random(5000, prg2);
  F61 s, d;
  for (std::size_t i = 0; i < a.size(); ++i) {
    s += a[i];
    d += a[i] * b[i];
  }
  REQUIRE(a.sum() == s);
  REQUIRE(a.dot(b) == d);
  REQUIRE((math::innerProd<F61>(a.begin(), a.end(), b.begin())) == d);
}

TEST_CASE(matrix_gpu, "Matrix multiply / vandermonde", BOTH) {
  using M = math::Matrix<F61>;
  // test/scl/math/test_matrix.cc:367-395
  const auto v = M::vandermonde(3, 3);
  const int want[9] = {1, 1, 1, 1, 2, 4, 1, 3, 9};
  for (int i = 0; i < 9; ++i) REQUIRE(v(i / 3, i % 3) == F61(want[i]));
  REQUIRE_THROWS_MSG(M::vandermonde(3, 3, math::Vector<F61>{F61(1)}), std::invalid_argument, "|xs| != number of rows");
  // :175-205 style small integer products
  const auto a = M::fromVector(2, 2, {F61(1), F61(2), F61(3), F61(4)});
  const auto b = M::fromVector(2, 2, {F61(5), F61(6), F61(7), F61(8)});
  const auto c = a.multiply(b);
  REQUIRE(c == M::fromVector(2, 2, {F61(19), F61(22), F61(43), F61(50)}));
  REQUIRE_THROWS_MSG(a.multiply(M(3, 2)), std::invalid_argument, "matmul: this->cols() != that->rows()");
  REQUIRE(a.multiply(a.invert()).isIdentity());  // :326-331
  REQUIRE(a.multiply(math::Vector<F61>{F61(1), F61(1)}) == math::Vector<F61>{F61(3), F61(7)});
  REQUIRE(a.add(b) == M::fromVector(2, 2, {F61(6), F61(8), F61(10), F61(12)}));
  // :342-365 -- Vandermonde evaluation then interpolation through the inverse
  auto prg = util::PRG::create("vdm");
  const std::size_t n = 10, t = 3;
  const auto coeff = math::Vector<F61>::random(t + 1, prg);
  const auto points = M::vandermonde(n, t + 1).multiply(coeff);
  const auto poly = math::Polynomial<F61>::create(coeff);
  for (std::size_t i = 0; i < n; ++i) REQUIRE(points[i] == poly.evaluate(F61((int)i + 1)));
  const auto back = M::vandermonde(t + 1, t + 1).invert().multiply(points.subVector(t + 1));
  REQUIRE(back == coeff);
  // hyper-invertible: every square sub-matrix invertible; spot-check one
  const auto him = M::hyperInvertible(3, 3);
  REQUIRE(him.multiply(him.invert()).isIdentity());
}

template <typename FF>
static void matmul_without_bounds(const char* seed) {
  // matrix.h:477-513 has no bound on the inner dimension: (200 x 7000)(7000 x 300) and a 257 x 10 000 matrix times a vector go
  // through the device (k_matmul_tiled / k_matvec) and equal the reference's own loops, here restated with scalars on entries
  // that take in every corner
  using M = math::Matrix<FF>;
  auto prg = util::PRG::create(seed);
  const std::size_t m = 200, k = 7000, n = 300;
  const M a = M::random(m, k, prg), b = M::random(k, n, prg);
  const M c = a.multiply(b);
  REQUIRE(c.rows() == m && c.cols() == n);
  for (std::size_t i : {std::size_t(0), std::size_t(63), std::size_t(64), m - 1})
    for (std::size_t j : {std::size_t(0), std::size_t(16), std::size_t(255), n - 1}) {
      FF want;
      for (std::size_t l = 0; l < k; ++l) want += a(i, l) * b(l, j);
      REQUIRE(c(i, j) == want);
    }
  const M w = M::random(257, 10000, prg);
  const auto x = math::Vector<FF>::random(10000, prg);
  const auto y = w.multiply(x);
  REQUIRE(y.size() == 257);
  for (std::size_t i : {std::size_t(0), std::size_t(128), std::size_t(256)}) {
    FF want;
    for (std::size_t l = 0; l < 10000; ++l) want += w(i, l) * x[l];
    REQUIRE(y[i] == want);
  }
}

TEST_CASE(matrix_unbounded, "Matrix::multiply has no bound on its shape (matrix.h:477-513)", GPU) {
  matmul_without_bounds<F61>("mm-unbounded-61");
  matmul_without_bounds<F127>("mm-unbounded-127");
  matmul_without_bounds<math::FF<math::ff::Secp256k1Scalar>>("mm-unbounded-secp");
}

TEST_CASE(baseline_names, "scl::Vec / ss::ShamirShare / ShamirReconstruct / AdditiveSS (the names of BASELINE.json and of shamir.h:94,167)", BOTH) {
  // the same calls as test/scl/ss/test_shamir.cc:34-40 and test_additive.cc:26-41 under the other spelling: same shares, same secret
  {
    auto prg = util::PRG::create("shamir passive");
    const Vec<F61> shares = ss::ShamirShare(F61(123), 3, 4, prg);
    REQUIRE(shares.size() == 4 && shares[0] == F61::fromString("068de5f6897f1180") && shares[3] == F61::fromString("1ca17e1ae3ddfdde"));
    REQUIRE(ss::ShamirReconstruct(shares) == F61(123) && ss::ShamirRecoverP(shares) == F61(123));
    auto prg2 = util::PRG::create("shamir passive");
    REQUIRE(ss::shamirSecretShare(F61(123), 3, 4, prg2) == shares);
    static_assert(std::is_same_v<Vec<F61>, math::Vector<F61>> && std::is_same_v<math::Vec<F61>, math::Vector<F61>> &&
                  std::is_same_v<Mat<F127>, math::Matrix<F127>>);
  }
  {
    auto prg = util::PRG::create();
    const auto shares = ss::AdditiveSS(F61(12345), 3, prg);
    REQUIRE(shares[0] == F61::fromString("10c9a04e00a8277a") && shares[1] == F61::fromString("0e0c7bcabdee0f5b"));
    REQUIRE(ss::AdditiveReconstruct(shares) == F61(12345) && shares.sum() == F61(12345));
  }
  {  // the batch forms through the same names (kernels above the host threshold)
    std::vector<F61> secrets;
    for (int s = 0; s < 3000; ++s) secrets.emplace_back(s * 7 + 1);
    auto prg = util::PRG::create("names-batch");
    if (hip::hostThreshold() == 0) {
      const hip::DeviceVector<F61> d(secrets);
      const auto sh = ss::ShamirShare(d, 3, 10, prg);
      REQUIRE(math::Vector<F61>(ss::ShamirReconstruct(sh).toHost()) == math::Vector<F61>(secrets));
      const auto ad = ss::AdditiveSS(d, 3, prg);
      REQUIRE(math::Vector<F61>(ss::AdditiveReconstruct(ad).toHost()) == math::Vector<F61>(secrets));
    }
  }
}

TEST_CASE(shamir_mirror, "ss::shamir* per secret (reference signatures)", BOTH) {
  // test/scl/ss/test_shamir.cc:34-40
  {
    auto prg = util::PRG::create("shamir passive");
    const auto shares = ss::shamirSecretShare(F61(123), 3, 4, prg);
    REQUIRE(shares.size() == 4);
    REQUIRE(shares[0] == F61::fromString("068de5f6897f1180"));  // values from the reference
    REQUIRE(shares[1] == F61::fromString("07b963480f75f1e3"));
    REQUIRE(shares[2] == F61::fromString("04308e7c46a958eb"));
    REQUIRE(shares[3] == F61::fromString("1ca17e1ae3ddfdde"));
    REQUIRE(ss::shamirRecoverP(shares) == F61(123));
  }
  {
    auto prg = util::PRG::create("shamir passive");
    const auto shares = ss::shamirSecretShare(F127(123), 3, 4, prg);
    REQUIRE(shares[0] == F127::fromString("68cb89a3b5d99924eddf871da5480cd8"));
    REQUIRE(shares[3] == F127::fromString("7009444c16f85b3b6d51f54c483ec2f8"));
    REQUIRE(ss::shamirRecoverP(shares) == F127(123));
  }
  {
    // the field Feldman / Pedersen share over; share values from the reference (tests/golden, secp256k1_order)
    using FS = math::FF<math::ff::Secp256k1Scalar>;
    auto prg = util::PRG::create("shamir passive");
    const auto shares = ss::shamirSecretShare(FS(123), 3, 4, prg);
    REQUIRE(shares[0].toString() == "42a24b13e7a4eb634759f5de02b04555926a1896a2bdf2792f2b3d8df56b2fa3");
    REQUIRE(shares[3].toString() == "b0bfa8fdf1b92683c3bb5984a733091c6aa2df05a498f9eca488ea175c4bc281");
    REQUIRE(ss::shamirRecoverP(shares) == FS(123));
    REQUIRE(prg.counter() == 8);  // Vector::random(4) of 32-byte elements = 8 blocks
  }
  // :42-66 t=5, n=100, nodes 4..9 at x=0 and x=27
  {
    auto prg = util::PRG::create("shamir recons");
    const auto shares = ss::shamirSecretShare(F61(123), 5, 100, prg);
    REQUIRE(shares.size() == 100);
    const math::Vector<F61> nodes{F61(4), F61(5), F61(6), F61(7), F61(8), F61(9)};
    const auto lb_0 = math::computeLagrangeBasis(nodes, 0);
    const auto r_0 = math::innerProd<F61>(shares.begin() + 3, shares.begin() + 9, lb_0.begin());
    REQUIRE(r_0 == F61(123));
    REQUIRE(shares.subVector(3, 9).dot(lb_0) == r_0);
    const auto lb_27 = math::computeLagrangeBasis(nodes, 27);
    REQUIRE(math::innerProd<F61>(shares.begin() + 3, shares.begin() + 9, lb_27.begin()) == shares[26]);
    REQUIRE(ss::shamirRecoverP(shares.subVector(3, 9), nodes, F61(27)) == shares[26]);
  }
  // :68-79 detection
  {
    auto prg = util::PRG::create("shamir detect");
    auto shares = ss::shamirSecretShare(F61(123), 4, 9, prg);
    REQUIRE(ss::shamirRecoverD(shares, 4) == F61(123));
    shares[2] = F61(4);
    REQUIRE_THROWS_MSG(ss::shamirRecoverD(shares, 4), std::logic_error, "error detected during recovery");
    REQUIRE_THROWS_MSG(ss::shamirRecoverD(shares.subVector(5), 4), std::logic_error,
                       "not enough shares provided to detect errors");
  }
  // :81-109 other nodes and evaluation points
  {
    auto prg = util::PRG::create("shamir detect2");
    auto c = math::Vector<F61>::random(4, prg);
    c[0] = F61(123);
    const auto p = math::Polynomial<F61>::create(c);
    std::vector<F61> sh;
    for (int i = 0; i < 7; ++i) sh.push_back(p.evaluate(F61(i + 42)));
    const math::Vector<F61> shares(sh);
    const auto alphas = math::Vector<F61>::range(42, 50);
    REQUIRE(ss::shamirRecoverD(shares, alphas, 3, 3, F61(0)) == F61(123));
    REQUIRE(ss::shamirRecoverD(shares, alphas, 3, 3, alphas[0]) == shares[0]);
  }
}

TEST_CASE(shamir_gpu, "ss::shamir*", GPU) {
  // batch == the per-secret calls on one PRG, in order
  {
    const std::size_t N = 1000, n = 10, t = 3;
    std::vector<F61> secrets;
    for (std::size_t s = 0; s < N; ++s) secrets.emplace_back((int)(s * 7919 + 1));
    auto prg_b = util::PRG::create("batch"), prg_s = util::PRG::create("batch");
    const hip::DeviceVector<F61> dsec(secrets);
    const auto m = ss::shamirSecretShare(dsec, t, n, prg_b);
    for (std::size_t s : {std::size_t(0), std::size_t(1), std::size_t(499), N - 1}) {
      // replay the sequential PRG up to secret s
      auto prg = util::PRG::create("batch");
      prg.advance(2 * s);
      const auto one = ss::shamirSecretShare(secrets[s], t, n, prg);
      REQUIRE(math::Vector<F61>(m.sharesOf(s)) == one);
    }
    REQUIRE(prg_b.counter() == 2 * N);
    const auto rec = ss::shamirRecoverP(m).toHost();
    REQUIRE(math::Vector<F61>(rec) == math::Vector<F61>(secrets));
    std::vector<std::size_t> bad;
    (void)prg_s;
    auto prg_d = util::PRG::create("batch-d");
    const auto md = ss::shamirSecretShare(dsec, 4, 9, prg_d);
    const auto recd = ss::shamirRecoverD(md, 4, &bad).toHost();
    REQUIRE(bad.empty());
    REQUIRE(math::Vector<F61>(recd) == math::Vector<F61>(secrets));
  }
}

TEST_CASE(recover_c_mirror, "ss::shamirRecoverC per secret (Berlekamp-Welch, reference signatures)", BOTH) {
  // test/scl/ss/test_shamir.cc:111-126
  {
    auto prg = util::PRG::create("shamir correct");
    auto shares = ss::shamirSecretShare(F61(123), 2, 7, prg);
    REQUIRE(ss::shamirRecoverC(shares).f.evaluate(F61{0}) == F61(123));
    shares[0] = F61(22);
    shares[1] = F61(23);
    const auto r = ss::shamirRecoverC(shares);
    REQUIRE(r.f.evaluate(F61{0}) == F61(123));
    REQUIRE(r.err.degree() == 2);
    REQUIRE(r.err.evaluate(F61(1)) == F61(0));
    REQUIRE(r.err.evaluate(F61(2)) == F61(0));
    shares[2] = F61(24);
    REQUIRE_THROWS_MSG(ss::shamirRecoverC(shares), std::logic_error, "could not correct shares");
  }
  // test_shamir.cc:128-142: nodes 42..49 (one more node than shares, as there)
  {
    auto prg = util::PRG::create("shamir correct2");
    const auto alphas = math::Vector<F61>::range(42, 50);
    const auto coeffs = math::Vector<F61>::random(3, prg);
    std::vector<F61> c = {F61(123), coeffs[1], coeffs[2]};
    const auto p = math::Polynomial<F61>::create(math::Vector<F61>(c));
    std::vector<F61> sv;
    for (int i = 0; i < 7; ++i) sv.push_back(p.evaluate(alphas[i]));
    math::Vector<F61> shares(sv);
    REQUIRE(ss::shamirRecoverC(shares, alphas).f.constantTerm() == F61(123));
    shares[4] = F61(5555);
    const auto r = ss::shamirRecoverC(shares, alphas);
    REQUIRE(r.f.constantTerm() == F61(123));
    REQUIRE(r.err.evaluate(alphas[4]) == F61(0));
  }
}

TEST_CASE(recover_c_gpu, "ss::shamirRecoverC (Berlekamp-Welch)", GPU) {
  // batch: 500 secrets, every third one with up to t corrupted shares
  {
    std::vector<F61> secrets;
    for (int s = 0; s < 500; ++s) secrets.emplace_back(s * 7 + 1);
    auto prg = util::PRG::create("bw-batch");
    const std::size_t t = 3, n = 10;
    const auto m = ss::shamirSecretShare(hip::DeviceVector<F61>(secrets), t, n, prg);
    // pull to the host, corrupt, push back
    std::vector<std::vector<F61>> rows;
    for (std::size_t i = 0; i < n; ++i) {
      std::vector<std::uint64_t> l(500);
      hip::check(scl_hip_memcpy_d2h(l.data(), m.row(i), 500 * 8, nullptr));
      hip::check(scl_hip_stream_sync(nullptr));
      std::vector<F61> r;
      for (auto v : l) r.push_back(F61::fromLimbs(&v));
      rows.push_back(r);
    }
    for (int s = 0; s < 500; s += 3)
      for (int k = 0; k < (s / 3) % 4; ++k) rows[(s + 3 * k) % n][s] = F61(99 + k);
    hip::ShareMatrix<F61> bad(n, 500);
    for (std::size_t i = 0; i < n; ++i) {
      std::vector<std::uint64_t> l(500);
      for (int s = 0; s < 500; ++s) rows[i][s].toLimbs(&l[s]);
      hip::check(scl_hip_memcpy_h2d(bad.data() + i * 500, l.data(), 500 * 8, nullptr));
      hip::check(scl_hip_stream_sync(nullptr));  // l dies with this iteration
    }
    const auto r = ss::shamirRecoverC(bad);
    REQUIRE(r.failed == 0);
    REQUIRE(r.solved > 0 && r.solved <= 167);
    REQUIRE(math::Vector<F61>(r.f.sharesOf(0)).size() == 10);
    bool all = true;
    for (int s = 0; s < 500; ++s) all = all && r.at(s).f.evaluate(F61{0}) == secrets[s];
    if (!all) {
      const auto dbg = bad.sharesOf(6);
      std::printf("    shares of s=6:");
      for (const auto& v : dbg) std::printf(" %s", v.toString().c_str());
      std::printf("\n    solved=%zu failed=%zu\n", r.solved, r.failed);
      for (int s = 0; s < 13; ++s)
        std::printf("    s=%d status=%d errors=%u f0=%s want=%s\n", s, (int)r.status[s], r.errors[s],
                    r.status[s] ? "-" : r.at(s).f.evaluate(F61{0}).toString().c_str(), secrets[s].toString().c_str());
    }
    REQUIRE(all);
    REQUIRE(r.errors[3] == 1 && r.errors[6] == 2 && r.errors[9] == 3 && r.errors[1] == 0);
  }
}

TEST_CASE(additive_mirror, "ss::additiveShare per secret (reference signature)", BOTH) {
  // test/scl/ss/test_additive.cc:26-41
  auto prg = util::PRG::create();
  const auto shares = ss::additiveShare(F61(12345), 3, prg);
  REQUIRE(shares.size() == 3);
  REQUIRE(shares[0] == F61::fromString("10c9a04e00a8277a"));  // from the reference
  REQUIRE(shares[1] == F61::fromString("0e0c7bcabdee0f5b"));
  REQUIRE(shares[2] == F61::fromString("0129e3e74169f963"));
  REQUIRE(shares.sum() == F61(12345));
  auto prg2 = util::PRG::create("add");
  const auto x = ss::additiveShare(F61(55), 10, prg2), y = ss::additiveShare(F61(11), 10, prg2);
  REQUIRE(x.size() == 10);
  REQUIRE(x.sum() == F61(55));
  REQUIRE(x.add(y).sum() == F61(66));  // additive homomorphism
  REQUIRE(ss::additiveShare(F61(9), 1, prg2) == math::Vector<F61>{F61(9)});
}

TEST_CASE(additive_gpu, "ss::additiveShare", GPU) {
  // batch
  std::vector<F61> secrets;
  for (int s = 0; s < 300; ++s) secrets.emplace_back(s * 31 + 5);
  auto prg3 = util::PRG::create("add-batch");
  const auto m = ss::additiveShare(hip::DeviceVector<F61>(secrets), 3, prg3);
  REQUIRE(prg3.counter() == 600);
  REQUIRE(math::Vector<F61>(ss::additiveRecover(m).toHost()) == math::Vector<F61>(secrets));
  auto prg4 = util::PRG::create("add-batch");
  prg4.advance(2 * 7);
  REQUIRE(math::Vector<F61>(m.sharesOf(7)) == ss::additiveShare(secrets[7], 3, prg4));
}

// ---------------------------------------------------------------------------- rings Z2k<K>
template <typename Z>
static void ring_identities() {
  // test/scl/math/test_z2k.cc:34-166 restated
  REQUIRE(std::string(Z::name()) == "Z2k");
  const Z a = Z::fromString("9e3779b97f4a7c15f39cc0605cedc834"), b = Z::fromString("1082276bf3a27251f86c6a11d0c18e95");
  const Z zero = Z::zero(), one = Z::one();
  REQUIRE(a + b == b + a);
  REQUIRE(a + zero == a);
  REQUIRE(a - a == zero);
  REQUIRE(a + (-a) == zero);
  REQUIRE(a * one == a);
  REQUIRE(a * b == b * a);
  REQUIRE((a + b) * a == a * a + b * a);
  Z big = Z::zero() - Z::one();  // 2^K - 1
  REQUIRE(big + one == zero);
  REQUIRE(big * big == one);
  Z odd = a;
  if (!odd.lsb()) odd += one;
  REQUIRE(odd * odd.inverse() == one);
  REQUIRE((b / odd) * odd == b);
  REQUIRE_THROWS_MSG(Z(2).inverse(), std::invalid_argument, "value not invertible modulo 2^K");
  unsigned char buf[17] = {0};
  a.write(buf);
  REQUIRE(Z::read(buf) == a);
  REQUIRE(Z::fromString(a.toString().size() % 2 ? "0" + a.toString() : a.toString()) == a || Z::bitSize() > 64);
}

TEST_CASE(z2k_host, "Z2k<62> / Z2k<123> / Z2k<32> identities (host scalars)", HOST) {
  ring_identities<math::Z2k<62>>();
  ring_identities<math::Z2k<123>>();
  ring_identities<math::Z2k<32>>();
  ring_identities<math::Z2k<64>>();
  ring_identities<math::Z2k<128>>();
  using Z = math::Z2k<32>;
  REQUIRE(Z::byteSize() == 4 && math::Z2k<62>::byteSize() == 8 && math::Z2k<123>::byteSize() == 16);
  REQUIRE(Z(0xFFFFFFFFull) + Z(1) == Z(0));
  REQUIRE(Z(3).inverse() == Z(0xAAAAAAABull));  // 3 * 0xAAAAAAAB = 2^33 + 1
  REQUIRE(Z(0x1234567890ull).toString() == "34567890");
}

TEST_CASE(z2k_mirror, "Vector<Z2k>, additive sharing over a ring per secret", BOTH) {
  using Z = math::Z2k<62>;
  using Zb = math::Z2k<123>;
  {
    auto p0 = util::PRG::create("shamir passive");
    REQUIRE(Zb::random(p0).toString() == "6ce7720f22f666734dc2c3c3df35e96");  // reference golden, first element
    REQUIRE(p0.counter() == 1);
  }
  auto prg = util::PRG::create("shamir passive");
  const auto v = math::Vector<Zb>::random(4, prg);
  REQUIRE(prg.counter() == 4);
  REQUIRE(v[0].toString() == "6ce7720f22f666734dc2c3c3df35e96");  // reference golden
  REQUIRE(v[1].toString() == "2c7ae6bf3761e8becf7c069a3d98c04");
  auto prg2 = util::PRG::create("ring");
  const auto x = math::Vector<Z>::random(100, prg2), y = math::Vector<Z>::random(100, prg2);
  REQUIRE(prg2.counter() == 100);
  const auto s = x.add(y), p = x.multiplyEntryWise(y);
  bool ok = true;
  Z dot = Z::zero(), sum = Z::zero();
  for (std::size_t i = 0; i < 100; ++i) {
    ok = ok && s[i] == x[i] + y[i] && p[i] == x[i] * y[i];
    dot += x[i] * y[i];
    sum += x[i];
  }
  REQUIRE(ok);
  REQUIRE(x.dot(y) == dot);
  REQUIRE(x.sum() == sum);
  // additive sharing over the ring: shares sum to the secret modulo 2^K
  auto prg3 = util::PRG::create();
  const auto shares = ss::additiveShare(Z(12345), 3, prg3);
  REQUIRE(shares.size() == 3 && prg3.counter() == 2);
  REQUIRE(shares.sum() == Z(12345));
}

TEST_CASE(z2k_gpu, "additive sharing over a ring, batch", GPU) {
  using Zb = math::Z2k<123>;
  std::vector<Zb> secrets;
  for (int i = 0; i < 257; ++i) secrets.emplace_back(Zb((__uint128_t)i * 0x9E3779B97F4A7C15ull));
  auto prg4 = util::PRG::create("ring-batch");
  const auto m = ss::additiveShare(hip::DeviceVector<Zb>(secrets), 5, prg4);
  REQUIRE(prg4.counter() == 257 * 4);
  REQUIRE(math::Vector<Zb>(ss::additiveRecover(m).toHost()) == math::Vector<Zb>(secrets));
  auto prg5 = util::PRG::create("ring-batch");
  prg5.advance(9 * 4);
  REQUIRE(math::Vector<Zb>(m.sharesOf(9)) == ss::additiveShare(secrets[9], 5, prg5));
}

// ---------------------------------------------------------------------------- the open step (RCCL)
template <typename F>
static void open_one_rank(std::size_t N, std::size_t n, std::size_t t, std::size_t chunk, const char* seed) {
  // one rank holds every party: the all-gather is RCCL's one-rank copy, everything else (row order, chunking, the two
  // streams and their events, the partial-sum form) is the code that runs on eight ranks
  hip::Communicator comm(1, 0, hip::Communicator::uniqueId());
  REQUIRE(comm.world() == 1 && comm.rank() == 0 && comm.partiesPerRank(n) == n);
  hip::check(scl_hip_set_tuning("open_gather_always", 1));  // (one rank would otherwise reconstruct straight from its slab)
  std::vector<F> secrets;
  for (std::size_t s = 0; s < N; ++s) secrets.emplace_back((int)(s * 2654435761u >> 1));
  auto prg = util::PRG::create(seed);
  const hip::DeviceVector<F> dsec(secrets);
  const auto shares = ss::shamirSecretShare(dsec, t, n, prg);
  const auto want = ss::shamirRecoverP(shares).toHost();
  REQUIRE(math::Vector<F>(want) == math::Vector<F>(secrets));
  const auto opened = hip::open(comm, shares, n, chunk).toHost();
  REQUIRE(math::Vector<F>(opened) == math::Vector<F>(secrets));
  const auto lambda = math::computeLagrangeBasis(math::Vector<F>::range(1, n + 1), F{});
  const auto partial = hip::openByPartialSums(comm, shares, n, lambda, chunk).toHost();
  REQUIRE(math::Vector<F>(partial) == math::Vector<F>(secrets));
}

// The same two opens with world > 1: the ranks are host threads of this process on one device, RCCL is the test-only stand-in
// tests/cxx/fake_rccl.cc (the library binds it when SCL_HIP_RCCL_LIBRARY names it; tests/test_cxx_api.py runs this binary
// with it and `--open-world`).  Each rank holds its block of parties' rows (plus padding rows of ones), every rank must end
// with every secret -- through hip::Communicator / hip::open / hip::openByPartialSums, the C++ mirror of the C ABI's open.
template <typename F>
static void open_world(int world, std::size_t N, std::size_t n, std::size_t t, std::size_t chunk, const char* seed) {
  constexpr std::size_t L = hip::limbsOf<F>();
  std::vector<F> secrets;
  for (std::size_t s = 0; s < N; ++s) secrets.emplace_back((int)(s * 2246822519u >> 1));
  auto prg = util::PRG::create(seed);
  const hip::DeviceVector<F> dsec(secrets);
  const auto full = ss::shamirSecretShare(dsec, t, n, prg);  // [n][N] on the device
  std::vector<std::uint64_t> host(n * N * L);                // .. and on the host: the ranks upload their rows from it
  hip::check(scl_hip_memcpy_d2h(host.data(), full.data(), host.size() * 8, nullptr));
  const auto lambda = math::computeLagrangeBasis(math::Vector<F>::range(1, n + 1), F{});
  const auto id = hip::Communicator::uniqueId();
  std::vector<int> ok(world, 0);
  std::vector<std::string> errs(world);
  std::vector<std::thread> ranks;
  for (int r = 0; r < world; ++r)
    ranks.emplace_back([&, r] {
      try {
        hip::Communicator comm(world, r, id);
        const std::size_t per = comm.partiesPerRank(n);
        const auto [first, count] = comm.partySlab(n);
        hip::ShareMatrix<F> slab(per, N), mine(count, N);
        hip::check(scl_hip_memset(slab.data(), 0xFF, per * N * L * 8, nullptr));  // padding rows: never read, never sent
        if (count) {
          hip::check(scl_hip_memcpy_h2d(slab.data(), host.data() + first * N * L, count * N * L * 8, nullptr));
          hip::check(scl_hip_memcpy_h2d(mine.data(), host.data() + first * N * L, count * N * L * 8, nullptr));
        }
        hip::check(scl_hip_stream_sync(nullptr));
        const auto a = hip::open(comm, slab, n, chunk).toHost();
        const auto b = hip::openByPartialSums(comm, mine, n, lambda, chunk).toHost();
        ok[r] = math::Vector<F>(a) == math::Vector<F>(secrets) && math::Vector<F>(b) == math::Vector<F>(secrets);
        if constexpr (std::is_same_v<F, F61>) {  // Mersenne61: the partial sums through ncclReduceScatter as 64-bit sums
          const auto c = hip::openByReduceScatter(comm, mine, n, lambda, chunk).toHost();
          ok[r] = ok[r] && math::Vector<F>(c) == math::Vector<F>(secrets);
        }
      } catch (const std::exception& e) {
        errs[r] = e.what();
      }
    });
  for (auto& th : ranks) th.join();
  for (int r = 0; r < world; ++r) {
    if (!errs[r].empty()) std::printf("  rank %d: %s\n", r, errs[r].c_str());
    REQUIRE(ok[r] == 1);
  }
}

TEST_CASE(open_world_threads, "hip::open / openByPartialSums / openByReduceScatter over a world of host threads (fake RCCL)", GPU) {
  if (!std::getenv("SCL_HIP_RCCL_LIBRARY")) {
    std::printf("  (skipped: SCL_HIP_RCCL_LIBRARY does not name the stand-in; tests/test_cxx_api.py runs it)\n");
    return;
  }
  open_world<F61>(2, 5001, 10, 3, 2048, "world61");                              // three chunks, a ragged odd last one
  open_world<F61>(4, 3001, 7, 2, 1000, "world61b");                              // 7 parties on 4 ranks: a padding row
  open_world<F127>(3, 2001, 10, 3, 512, "world127");
  open_world<math::FF<math::ff::GF2_128>>(8, 1501, 40, 13, 400, "worldgf");       // five parties per rank
  open_world<F61>(8, 1001, 10, 3, 300, "world61c");                              // ranks 5..7 hold nothing
}

// ---------------------------------------------------------------------------- math::Array<T, N>
// test/scl/math/test_array.cc restated (the reference runs it over EC points and their scalar field; curve points are out of
// scope here, so the value types are the scalar field FF<Secp256k1Scalar> and Fp<61>) and the sharing step of
// pedersenSecretShare (include/scl/ss/pedersen.h:127-140): shamirSecretShare over Array<FF, 2>{{secret, randomness}}.
TEST_CASE(array_host, "math::Array: default init / operations / text / read-write (test_array.cc)", HOST) {
  using F = math::FF<math::ff::Secp256k1Scalar>;
  const auto zero = F::zero();
  math::Array<F, 3> q0;  // "Array default init"
  REQUIRE(q0 == math::Array<F, 3>{{zero, zero, zero}});
  math::Array<F, 3> p = {{F(1), F(2), F(4)}};  // "Array operations"
  math::Array<F, 3> q = {{F(4), F(2), F(1)}};
  REQUIRE(p + q == math::Array<F, 3>{{F(5), F(4), F(5)}});
  REQUIRE(p - q == math::Array<F, 3>{{F(-3), F(0), F(3)}});
  REQUIRE(p * q == math::Array<F, 3>{{F(4), F(4), F(4)}});
  REQUIRE(q * p == math::Array<F, 3>{{F(4), F(4), F(4)}});
  // scalar forms, increments, inverses, division (array.h:173-345)
  REQUIRE(p * F(3) == math::Array<F, 3>{{F(3), F(6), F(12)}});
  auto r = p;
  r *= q;
  REQUIRE(r == math::Array<F, 3>(F(4)));
  REQUIRE((r / q) == p && (p.Inverse() * p) == math::Array<F, 3>::one());
  auto inc = p;
  REQUIRE((inc++) == p && inc == math::Array<F, 3>{{F(2), F(3), F(5)}} && (--inc) == p);
  REQUIRE(math::Array<F, 3>(7) == math::Array<F, 3>{{F(7), F(7), F(7)}});
  REQUIRE(math::Array<F, 3>(p).negate() + p == math::Array<F, 3>::zero());
  REQUIRE(p != q && p[2] == F(4));
  using A3F = math::Array<F, 3>;
  REQUIRE_THROWS_MSG(A3F::zero().Inverse(), std::logic_error, "0 not invertible modulo prime");
  // "Array to string": P{v0, v1}
  math::Array<F61, 2> two = {{F61(10), F61(255)}};
  REQUIRE(two.toString() == "P{a, ff}");
  // "Array serialization": write / read round trip of a random Array (the reference goes through Serializer<Array>, which is
  // these two members, array.h:439-455)
  auto prg = util::PRG::create("prod seri");
  const auto prod = math::Array<F, 3>::random(prg);
  static_assert(math::Array<F, 3>::byteSize() == 96);
  unsigned char buf[96];
  prod.write(buf);
  math::Array<F, 3> back;
  REQUIRE(back != prod);
  back = math::Array<F, 3>::read(buf);
  REQUIRE(back == prod);
  // Array::random draws its components one T::random at a time: for an FF that is one AES block each (array.h:93-101)
  auto prg2 = util::PRG::create("prod seri");
  REQUIRE(F::random(prg2) == prod[0] && F::random(prg2) == prod[1] && F::random(prg2) == prod[2]);
}

static const char* const kArray3Secrets[9] = {"064f6919aafaccbc", "0a629d5ac3357b6b", "0c74b1711771d2ae", "13dd036d5aec9b11", "13c6d3e817308642",
                                              "03cdba09cbbe4853", "0f62b7361096e632", "1fa19d614c0ebb84", "15db71edd4bfd95b"};
static const char* const kArray3Shares[45] = {
    "14de80fe0bbb0370", "099cde9b097aec65", "09215287bd5784bd", "0df56060348d0b99", "10b65eee9a0e300f", "17996d10c5d924a3", "119407402570e536",
    "1faf1e5574ef4669", "17dd010c30f6b261", "1fba759dde669047", "16871ccf9a1e2f74", "09ec0e79feb02df7", "1868ab795f6e0ccd", "153e5a5d099aeb2f",
    "0dc6955a2f059764", "1f802f196ba120cf", "10510148e052db9f", "1a3bbbce52843fa9", "10b2adc223cbcb88", "1993121789c5868c", "030fe12528cf54d5",
    "07747f67836c9b3b", "0f8d06541388870a", "1e4a2a0e4e9f87d4", "03c5a4098a838fe8", "123eddfe7d9bdd18", "0bea9689c3f4d8a9", "05a61ba83910a98f",
    "01a89916c7ff88b7", "0bf1269788cf4752", "0ab21234697894c4", "0d365d9cda97bc28", "08fba206a5183b46", "1c66e10e8193a8b2", "0ee7907b22f6c374",
    "1e2f0393673c1862", "048123c458e821fe", "04b535fc252bd169", "157596941b2b70b1", "0300da55ef7600a6", "0e9f4e1fe136e606", "0ecf5b08c0e64432",
    "17e604c3453d44aa", "0ca5d8e65718014c", "0a3c50f1586c92e5"};

TEST_CASE(array_sharing, "ss::shamirSecretShare over math::Array (the sharing step of pedersenSecretShare)", BOTH) {
  // known answer from the reference itself (tests/golden/golden_v1.json, Mersenne61 shamir_packed[2]): W = 3, n = 5, t = 2,
  // three secrets shared one after the other on PRG::create("array3")
  using A3 = math::Array<F61, 3>;
  auto prg = util::PRG::create("array3");
  for (int s = 0; s < 3; ++s) {
    const A3 secret = {{F61::fromString(kArray3Secrets[3 * s]), F61::fromString(kArray3Secrets[3 * s + 1]), F61::fromString(kArray3Secrets[3 * s + 2])}};
    const auto shares = ss::shamirSecretShare(secret, 2, 5, prg);
    REQUIRE(shares.size() == 5);
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 3; ++j) REQUIRE(shares[i][j] == F61::fromString(kArray3Shares[(s * 5 + i) * 3 + j]));
    REQUIRE(ss::shamirRecoverP(shares) == secret);  // interpolation is component-wise too
  }
  // pedersen.h:137-138 as written there: {secret, randomness} over the curve's scalar field
  using F = math::FF<math::ff::Secp256k1Scalar>;
  auto prg2 = util::PRG::create("pedersen");
  const F secret(123), randomness(456);
  const math::Array<F, 2> sr = {{secret, randomness}};
  const auto shares = ss::shamirSecretShare(sr, 3, 10, prg2);
  REQUIRE(shares.size() == 10);
  math::Vector<F> first(10), second(10);
  for (std::size_t i = 0; i < 10; ++i) {
    first[i] = shares[i][0];
    second[i] = shares[i][1];
  }
  REQUIRE(ss::shamirRecoverP(first) == secret && ss::shamirRecoverP(second) == randomness);
  REQUIRE(ss::shamirRecoverD(first.subVector(0, 7), 3) == secret);
}

TEST_CASE(array_sharing_gpu, "ss::shamirSecretShare over math::Array, batch (scl_hip_shamir_share_prg_packed)", GPU) {
  // the batch form against the per-secret one on the same PRG, every secret, both components: Mersenne61 and secp256k1's order
  auto run = [](auto tag, std::size_t N, std::size_t t, std::size_t n, const char* seed) {
    using F = decltype(tag);
    using A2 = math::Array<F, 2>;
    auto src = util::PRG::create(std::string(seed) + "-secrets");
    std::vector<A2> host;
    for (std::size_t s = 0; s < N; ++s) host.push_back(A2::random(src));
    auto prg_host = util::PRG::create(seed), prg_dev = util::PRG::create(seed);
    const ss::ArrayVector<F, 2> dev{math::Vector<A2>(host)};
    const auto batch = ss::shamirSecretShare(dev, t, n, prg_dev);
    REQUIRE(batch.parties == n && batch.rows.parties() == 2 * n && batch.rows.secrets() == N);
    std::size_t bad = 0;
    for (std::size_t s = 0; s < N; ++s) {
      const auto want = ss::shamirSecretShare(host[s], t, n, prg_host);
      if (s % 37 == 0 || s + 1 == N) bad += !batch.sharesOf(s).equals(want);
    }
    REQUIRE(bad == 0);
    REQUIRE(prg_dev.counter() == prg_host.counter());  // the batch advanced the PRG exactly as the N calls did
  };
  run(F61{}, 1501, 3, 10, "array-batch-61");
  run(math::FF<math::ff::Secp256k1Scalar>{}, 301, 2, 7, "array-batch-secp");
}

TEST_CASE(open_rccl, "hip::open / openByPartialSums over a one-rank RCCL communicator", GPU) {
  // test/scl/protocol/beaver.h:43-55 opens by send-to-all / recv-from-all; the batch form over RCCL
  open_one_rank<F61>(5001, 10, 3, 2048, "open61");          // three chunks, the last one ragged and odd
  open_one_rank<F61>(1000, 10, 3, 0, "open61b");            // one chunk
  open_one_rank<F127>(3000, 7, 2, 1024, "open127");
  open_one_rank<math::FF<math::ff::GF2_128>>(4097, 40, 13, 2048, "opengf");
  // the row order of a gathered chunk on 8 ranks, 40 parties: row j * 8 + r is party 5 r + j
  long order[40];
  hip::check(scl_hip_open_row_order(40, 8, order));
  REQUIRE(order[0] == 0 && order[1] == 5 && order[8] == 1 && order[39] == 39);
  long ragged[12];  // 10 parties on 4 ranks: 3 per rank, the last rank holds one party and two padding rows
  hip::check(scl_hip_open_row_order(10, 4, ragged));
  REQUIRE(ragged[3] == 9 && ragged[7] == -1 && ragged[11] == -1 && ragged[4] == 1);
}

// ---------------------------------------------------------------------------- hip::DeviceVector element-wise, lazy zero flag
// FF::operator/ throws std::logic_error("0 not invertible modulo prime") at the first zero divisor (ff.h:203-205 through
// small_ff.h:61-70; pinned by test_ff.cc:168-171).  Over device-resident vectors the same exception arrives LATE, from
// hip::ZeroFlag::check_for, after any number of asynchronous calls; the flag-less forms throw before they return.
template <typename FF>
static void device_elementwise(const char* seed) {
  auto prg = util::PRG::create(seed);
  const std::size_t N = 5000;
  std::vector<FF> a(N), b(N);
  for (std::size_t i = 0; i < N; ++i) {
    a[i] = FF::random(prg);
    b[i] = FF::random(prg);
    if (b[i] == FF::zero()) b[i] = FF::one();
  }
  hip::DeviceVector<FF> da(a), db(b), out(N);
  hip::add(out, da, db);
  auto h = out.toHost();
  bool ok = true;
  for (std::size_t i = 0; i < N; ++i) ok = ok && h[i] == a[i] + b[i];
  REQUIRE(ok);
  hip::multiplyEntryWise(out, da, db);
  h = out.toHost();
  for (std::size_t i = 0; i < N; ++i) ok = ok && h[i] == a[i] * b[i];
  REQUIRE(ok);
  hip::ZeroFlag flag;
  hip::divide(out, da, db, flag);
  hip::inverse(db, db, flag);           // in place; several calls on one flag
  hip::multiplyEntryWise(db, db, da);   // a * b^-1 once more
  REQUIRE(!flag.raised());
  flag.template check_for<FF>();        // nothing to throw
  h = out.toHost();
  auto h2 = db.toHost();
  for (std::size_t i = 0; i < N; ++i) ok = ok && h[i] == a[i] / b[i] && h2[i] == h[i];
  REQUIRE(ok);
  // a zero among the divisors: the asynchronous call returns, the other slots are computed, the zero's slot holds 0, and the
  // reference's exception comes out of check_for -- once
  b[N / 2] = FF::zero();
  hip::DeviceVector<FF> dz(b);
  hip::divide(out, da, dz, flag);
  REQUIRE(flag.raised());
  REQUIRE_THROWS_MSG(flag.template check_for<FF>(), std::logic_error, "0 not invertible modulo prime");
  REQUIRE(!flag.raised());              // cleared by the throw
  h = out.toHost();
  REQUIRE(h[N / 2] == FF::zero());
  REQUIRE(h[0] == a[0] / b[0] && h[N - 1] == a[N - 1] / b[N - 1]);
  // the synchronous forms throw before they return
  REQUIRE_THROWS_MSG(hip::divide(out, da, dz), std::logic_error, "0 not invertible modulo prime");
  REQUIRE_THROWS_MSG(hip::inverse(out, dz), std::logic_error, "0 not invertible modulo prime");
  hip::DeviceVector<FF> shorter(N - 1);
  REQUIRE_THROWS_MSG(hip::add(shorter, da, db), std::invalid_argument, "Vec sizes mismatch");
}
TEST_CASE(device_elementwise_m61, "hip::add / multiplyEntryWise / divide / inverse over DeviceVector, lazy ZeroFlag (Mersenne61)", GPU) {
  device_elementwise<F61>("device ew 61");
}
TEST_CASE(device_elementwise_m127, "hip::divide / inverse over DeviceVector, lazy ZeroFlag (Mersenne127)", GPU) {
  device_elementwise<F127>("device ew 127");
}
TEST_CASE(device_elementwise_ring, "hip::inverse over DeviceVector<Z2k<64>>: an even element raises the ring's exception late", GPU) {
  using R = math::Z2k<64>;
  auto prg = util::PRG::create("device ew ring");
  const std::size_t N = 300;
  std::vector<R> a(N);
  for (auto& x : a) x = R::random(prg) * R(2) + R(1);   // odd: invertible modulo 2^64
  hip::DeviceVector<R> da(a), out(N);
  hip::ZeroFlag flag;
  hip::inverse(out, da, flag);
  flag.check_for<R>();
  auto h = out.toHost();
  bool ok = true;
  for (std::size_t i = 0; i < N; ++i) ok = ok && h[i] * a[i] == R(1);
  REQUIRE(ok);
  a[7] = R(10);
  hip::DeviceVector<R> de(a);
  hip::inverse(out, de, flag);
  REQUIRE_THROWS_MSG(flag.check_for<R>(), std::invalid_argument, "value not invertible modulo 2^K");
}

int main(int argc, char** argv) {
  const bool host_only = argc > 1 && std::string(argv[1]) == "--host-only";
  int ran = 0;
  auto run = [&](const Case& c, const char* suffix) {
    const int before = g_fail;
    try {
      c.fn();
    } catch (const std::exception& e) {
      ++g_fail;
      std::printf("  EXCEPTION in '%s': %s\n", c.name, e.what());
    }
    std::printf("[%s] %s%s\n", g_fail == before ? " ok " : "FAIL", c.name, suffix);
    ++ran;
  };
  const bool only_world = argc > 1 && std::string(argv[1]) == "--open-world";
  for (const auto& c : cases()) {
    if (host_only && c.where == GPU) continue;
    if (only_world && std::string(c.name).find("world of host threads") == std::string::npos) continue;
    run(c, "");
  }
  if (only_world) {
    std::printf("%d cases, %d checks, %d failures\n", ran, g_checks, g_fail);
    return g_fail ? 1 : 0;
  }
  if (!host_only) {  // the same host-container cases with every Vector / Matrix / PRG member on its kernel
    const std::size_t keep = hip::hostThreshold();
    hip::setHostThreshold(0);
    for (const auto& c : cases())
      if (c.where == BOTH) run(c, " [host threshold 0: through the kernels]");
    hip::setHostThreshold(keep);
  }
  std::printf("%d cases, %d checks, %d failures\n", ran, g_checks, g_fail);
  return g_fail ? 1 : 0;
}
