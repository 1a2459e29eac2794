// integration/binding_check.cc -- compiles integration/include/scl/hip/binding.h against the REAL reference
// (/root/reference/include + its translation units, oracle/Makefile `binding`) and runs it on a GPU: reference Vector<FF> in
// -> C ABI -> reference Vector<FF> out, compared with what the reference itself computes for the same inputs
// (Vector::multiplyEntryWise, ss::shamirSecretShare on one PRG, ss::shamirRecoverP).  Exit code 0 iff everything agreed.
#include <cstdio>
#include <string>

#include "scl/hip/binding.h"
#include "scl/math/fp.h"
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
