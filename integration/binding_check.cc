// integration/binding_check.cc -- compiles integration/include/scl/hip/binding.h against the REAL reference
// (/root/reference/include + its translation units, oracle/Makefile `binding`) and runs it on a GPU: reference Vector<FF> in
// -> C ABI -> reference Vector<FF> out, compared with what the reference itself computes for the same inputs
// (Vector::multiplyEntryWise, ss::shamirSecretShare on one PRG, ss::shamirRecoverP, FF::inverse element by element,
// Matrix::multiply, Vector::dot / sum, ss::additiveShare on one PRG, ss::shamirRecoverD).  Exit code 0 iff everything agreed.
#include <cstdio>
#include <string>

#include "scl/hip/binding.h"
#include "scl/math/fp.h"
#include "scl/math/matrix.h"
#include "scl/ss/additive.h"
#include "scl/ss/shamir.h"
#include "scl/util/prg.h"

using namespace scl;

static int g_fail = 0;
#define EXPECT(cond)                                                 \
  do {                                                               \
    if (!(cond)) {                                                   \
      ++g_fail;                                                      \
      std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
    }                                                                \
  } while (0)

template <typename F>
static void run(const char* name, std::size_t N, std::size_t t, std::size_t n) {
  using FF = math::FF<F>;
  auto prg = util::PRG::create("binding-check-" + std::string(name));
  // (1) element-wise product: the binding against Vector::multiplyEntryWise (vector.h:539-550)
  const auto a = math::Vector<FF>::random(N, prg);
  const auto b = math::Vector<FF>::random(N, prg);
  const auto want = a.multiplyEntryWise(b);
  const auto got = hip::multiplyEntryWise<F>(a, b);
  EXPECT(got.equals(want));
  bool threw = false;
  try {
    (void)hip::multiplyEntryWise<F>(a, math::Vector<FF>(N + 1));
  } catch (const std::invalid_argument& e) {
    threw = std::string(e.what()) == "Vec sizes mismatch";  // the reference's own text (vector.h:483)
  }
  EXPECT(threw);
  // (2) sharing N secrets on ONE fresh PRG: the batch call against N sequential reference calls (shamir.h:52-68)
  const unsigned char seed_bytes[16] = {'b', 'i', 'n', 'd', 'i', 'n', 'g', '-', 's', 'e', 'e', 'd', 0, 1, 2, 3};
  std::array<unsigned char, 16> seed;
  for (int i = 0; i < 16; ++i) seed[i] = seed_bytes[i];
  auto share_prg = util::PRG::create(seed.data(), seed.size());
  std::vector<math::Vector<FF>> ref_shares;
  for (std::size_t s = 0; s < N; ++s) ref_shares.push_back(ss::shamirSecretShare(a[s], t, n, share_prg));
  const auto dev_shares = hip::shamirSecretShare<F>(a, t, n, seed, /*counter=*/0);
  EXPECT(dev_shares.size() == N);
  std::size_t bad = 0;
  for (std::size_t s = 0; s < N && s < dev_shares.size(); ++s) bad += !dev_shares[s].equals(ref_shares[s]);
  EXPECT(bad == 0);
  // (3) reconstruction: the batch call on the reference's shares against ss::shamirRecoverP per secret (shamir.h:99-104)
  const auto opened = hip::shamirRecoverP<F>(ref_shares);
  EXPECT(opened.equals(a));
  std::size_t bad2 = 0;
  for (std::size_t s = 0; s < N; s += 97) bad2 += !(ss::shamirRecoverP(dev_shares[s]) == a[s]);
  EXPECT(bad2 == 0);
  // (4) every element's inverse: the batch call (simultaneous inversion on the GPU) against FF::inverse one by one (ff.h:225-231)
  {
    std::vector<FF> nz;
    for (std::size_t s = 0; s < N; ++s) nz.push_back(a[s] == FF::zero() ? FF::one() : a[s]);
    const math::Vector<FF> v(nz);
    const auto inv = hip::inverseEntryWise<F>(v);
    std::size_t bad3 = 0;
    for (std::size_t s = 0; s < N; ++s) bad3 += !(inv[s] == v[s].inverse());
    EXPECT(bad3 == 0);
    bool threw0 = false;
    try {
      nz[N / 2] = FF::zero();
      (void)hip::inverseEntryWise<F>(math::Vector<FF>(nz));
    } catch (const std::logic_error& e) {
      threw0 = std::string(e.what()) == "0 not invertible modulo prime";  // test_ff.cc:168-171
    }
    EXPECT(threw0);
  }
  // (5) Matrix::multiply with an inner dimension no LDS tile bounds, against the reference's own i-k-j loop (matrix.h:477-495)
  {
    const std::size_t M = 37, K = sizeof(FF) == 8 ? 7000 : 700, Nc = 41;
    const auto A = math::Matrix<FF>::random(M, K, prg);
    const auto B = math::Matrix<FF>::random(K, Nc, prg);
    EXPECT(hip::multiply<F>(A, B).equals(A.multiply(B)));
    bool threwm = false;
    try {
      (void)hip::multiply<F>(A, A);
    } catch (const std::invalid_argument& e) {
      threwm = std::string(e.what()) == "matmul: this->cols() != that->rows()";
    }
    EXPECT(threwm);
  }
  // (6) Vector::dot and Vector::sum (vector.h:252-267)
  EXPECT(hip::dot<F>(a, b) == a.dot(b));
  EXPECT(hip::sum<F>(a) == a.sum());
  // (7) additive sharing of N secrets on one PRG against N sequential ss::additiveShare calls (additive.h:41-53); every sharing
  // sums back to its secret
  {
    auto add_prg = util::PRG::create(seed.data(), seed.size());
    const auto dev_add = hip::additiveShare<F>(a, n, seed, /*counter=*/0);
    std::size_t bad4 = 0;
    for (std::size_t s = 0; s < N; ++s) {
      const auto want_s = ss::additiveShare(a[s], n, add_prg);
      bad4 += !dev_add[s].equals(want_s);
      if (s % 89 == 0) bad4 += !(dev_add[s].sum() == a[s]);
    }
    EXPECT(dev_add.size() == N && bad4 == 0);
  }
  // (8) error detection: shares of degree t among n = 2t + 1 parties, every 53rd sharing with one share changed; the batch call
  // flags exactly the secrets at which ss::shamirRecoverD(shares, t) throws (shamir.h:116-154) and opens the others
  {
    const std::size_t nd = 2 * t + 1;
    auto d_prg = util::PRG::create("binding-check-detect");
    std::vector<math::Vector<FF>> sh;
    for (std::size_t s = 0; s < N; ++s) {
      auto v = ss::shamirSecretShare(a[s], t, nd, d_prg).toStlVector();
      if (s % 53 == 7) v[(s / 53) % nd] += FF(1);
      sh.emplace_back(v);
    }
    std::vector<unsigned char> flagged;
    const auto opened_d = hip::shamirRecoverD<F>(sh, t, &flagged);
    std::size_t bad5 = 0;
    for (std::size_t s = 0; s < N; ++s) {
      bool threw_d = false;
      FF want_d;
      try {
        want_d = ss::shamirRecoverD(sh[s], t);
      } catch (const std::logic_error& e) {
        threw_d = std::string(e.what()) == "error detected during recovery";
      }
      bad5 += (threw_d != (flagged[s] != 0));
      if (!threw_d) bad5 += !(opened_d[s] == want_d);
    }
    EXPECT(bad5 == 0);
  }
  std::printf("[%s] %s: %zu secrets, (n, t) = (%zu, %zu)\n", g_fail ? "FAIL" : " ok ", name, N, n, t);
}

int main() {
  int devices = 0;
  if (scl_hip_device_count(&devices) != SCL_OK || devices == 0) {
    std::printf("binding_check: no GPU (%s)\n", scl_hip_last_error());
    return 2;
  }
  run<math::ff::Mersenne61>("Mersenne61", 20001, 3, 10);
  run<math::ff::Mersenne127>("Mersenne127", 5001, 3, 10);
  run<math::ff::Secp256k1Scalar>("secp256k1_order", 1001, 2, 7);
  run<math::ff::Secp256k1Field>("secp256k1_field", 501, 2, 5);
  std::printf("binding_check: %d failure(s)\n", g_fail);
  return g_fail ? 1 : 0;
}
