// include/scl/hip/binding.h -- the header a maintainer of the REFERENCE would add to its tree to send the batch-shaped call
// sites of scl::math / scl::ss through libscl_hip.so.  It is written against the reference's own
// headers (scl/math/vector.h, scl/math/ff.h ..) and the C ABI (scl_hip.h) only.
//
// oracle/Makefile's `binding` target compiles it, with integration/binding_check.cc, against /root/reference/include and the
// reference's translation units -- the reference is not modified and nothing of it is copied; the binary runs under
// `pytest -m gpu` (tests/test_gpu_parity.py::test_reference_binding_compiled_against_the_reference).
//
// What makes it a drop-in: FF<FIELD> is a standard-layout class whose only member is FIELD::ValueType m_value
// (include/scl/math/ff.h:314), so a std::vector<FF<F>>::data() IS the ABI's uint64_t limb array: one limb for Mersenne61,
// two little-endian limbs (a 16-byte aligned __uint128_t) for Mersenne127, four for the secp256k1 fields, whose
// std::array<mp_limb_t, 4> holds the Montgomery residue the ABI takes for SCL_SECP256K1_SCALAR / _FIELD
// (src/scl/math/fields/secp256k1_scalar.cc:47-135).  The static_asserts below are those claims.
#ifndef SCL_HIP_BINDING_H
#define SCL_HIP_BINDING_H

#include <scl_hip.h>

#include <array>
#include <cstdint>
#include <stdexcept>
#include <type_traits>
#include <vector>

#include "scl/math/fields/mersenne127.h"
#include "scl/math/fields/mersenne61.h"
#include "scl/math/fields/secp256k1_field.h"
#include "scl/math/fields/secp256k1_scalar.h"
#include "scl/math/ff.h"
#include "scl/math/matrix.h"
#include "scl/math/vector.h"

namespace scl::hip {

template <typename F>
struct FieldTag;
template <>
struct FieldTag<math::ff::Mersenne61> {
  static constexpr int value = SCL_M61;
  static constexpr std::size_t limbs = 1;
};
template <>
struct FieldTag<math::ff::Mersenne127> {
  static constexpr int value = SCL_M127;
  static constexpr std::size_t limbs = 2;
};
template <>
struct FieldTag<math::ff::Secp256k1Scalar> {
  static constexpr int value = SCL_SECP256K1_SCALAR;
  static constexpr std::size_t limbs = 4;
};
template <>
struct FieldTag<math::ff::Secp256k1Field> {
  static constexpr int value = SCL_SECP256K1_FIELD;
  static constexpr std::size_t limbs = 4;
};

// the layout the binding relies on (include/scl/math/ff.h:314): checked where the binding is compiled
template <typename F>
constexpr bool kLayoutOk = std::is_standard_layout_v<math::FF<F>> && sizeof(math::FF<F>) == 8 * FieldTag<F>::limbs &&
                           sizeof(math::FF<F>) == math::FF<F>::byteSize() && alignof(math::FF<F>) >= 8 &&
                           sizeof(typename F::ValueType) == sizeof(math::FF<F>);
static_assert(kLayoutOk<math::ff::Mersenne61>, "FF<Mersenne61> is one uint64_t");
static_assert(kLayoutOk<math::ff::Mersenne127> && alignof(math::FF<math::ff::Mersenne127>) == 16, "FF<Mersenne127> is a 16-byte aligned pair of limbs");
static_assert(kLayoutOk<math::ff::Secp256k1Scalar>, "FF<Secp256k1Scalar> is four limbs");
static_assert(kLayoutOk<math::ff::Secp256k1Field>, "FF<Secp256k1Field> is four limbs");
static_assert(sizeof(mp_limb_t) == 8, "64-bit GMP limbs");

inline void check(int st) {
  if (st == SCL_OK) return;
  const char* msg = scl_hip_status_message(st);
  if (st == SCL_ERR_SIZE_MISMATCH || st == SCL_ERR_MATMUL_DIMS || st == SCL_ERR_VANDERMONDE_XS)
    throw std::invalid_argument(msg);
  if (st == SCL_ERR_ZERO_INVERSE || st == SCL_ERR_ERROR_DETECTED || st == SCL_ERR_NOT_ENOUGH_SHARES)
    throw std::logic_error(msg);
  throw std::runtime_error(scl_hip_last_error());
}

struct DevBuf {  // RAII device buffer
  void* p = nullptr;
  explicit DevBuf(std::size_t bytes) { check(scl_hip_malloc(&p, bytes)); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { scl_hip_free(p); }
  uint64_t* u64() const { return static_cast<uint64_t*>(p); }
};

// Vector<FF<F>>::multiplyEntryWise (include/scl/math/vector.h:539-550) on the GPU
template <typename F>
math::Vector<math::FF<F>> multiplyEntryWise(const math::Vector<math::FF<F>>& a, const math::Vector<math::FF<F>>& b) {
  if (a.size() != b.size()) throw std::invalid_argument("Vec sizes mismatch");
  const std::size_t bytes = a.byteSize();
  DevBuf da(bytes), db(bytes), dc(bytes);
  check(scl_hip_memcpy_h2d(da.p, a.toStlVector().data(), bytes, nullptr));
  check(scl_hip_memcpy_h2d(db.p, b.toStlVector().data(), bytes, nullptr));
  check(scl_hip_ew(FieldTag<F>::value, SCL_OP_MUL, dc.u64(), da.u64(), db.u64(), a.size(), nullptr));
  std::vector<math::FF<F>> out(a.size());
  check(scl_hip_memcpy_d2h(out.data(), dc.p, bytes, nullptr));
  return out;
}

// every element's FF::inverse (include/scl/math/ff.h:225-231; throws like the reference when an element is zero) -- on the GPU one
// inversion per chain of elements instead of one per element, the same unique inverses
template <typename F>
math::Vector<math::FF<F>> inverseEntryWise(const math::Vector<math::FF<F>>& a) {
  const std::size_t bytes = a.byteSize();
  DevBuf da(bytes), dc(bytes);
  check(scl_hip_memcpy_h2d(da.p, a.toStlVector().data(), bytes, nullptr));
  check(scl_hip_ew(FieldTag<F>::value, SCL_OP_INV, dc.u64(), da.u64(), nullptr, a.size(), nullptr));
  std::vector<math::FF<F>> out(a.size());
  check(scl_hip_memcpy_d2h(out.data(), dc.p, bytes, nullptr));
  return out;
}

// Matrix<FF<F>>::multiply (include/scl/math/matrix.h:477-495) on the GPU: row-major FF elements are the C ABI's limbs as they are
template <typename F>
math::Matrix<math::FF<F>> multiply(const math::Matrix<math::FF<F>>& a, const math::Matrix<math::FF<F>>& b) {
  using FF = math::FF<F>;
  if (a.cols() != b.rows()) throw std::invalid_argument("matmul: this->cols() != that->rows()");
  const std::size_t M = a.rows(), K = a.cols(), N = b.cols(), E = sizeof(FF);
  std::vector<FF> ha, hb;
  ha.reserve(M * K);
  hb.reserve(K * N);
  for (std::size_t i = 0; i < M; ++i)
    for (std::size_t k = 0; k < K; ++k) ha.push_back(a(i, k));
  for (std::size_t k = 0; k < K; ++k)
    for (std::size_t j = 0; j < N; ++j) hb.push_back(b(k, j));
  DevBuf da(M * K * E), db(K * N * E), dc(M * N * E);
  check(scl_hip_memcpy_h2d(da.p, ha.data(), M * K * E, nullptr));
  check(scl_hip_memcpy_h2d(db.p, hb.data(), K * N * E, nullptr));
  check(scl_hip_matmul(FieldTag<F>::value, dc.u64(), N, da.u64(), K, db.u64(), N, M, K, N, nullptr));
  std::vector<FF> hc(M * N);
  check(scl_hip_memcpy_d2h(hc.data(), dc.p, M * N * E, nullptr));
  return math::Matrix<FF>::fromVector(M, N, hc);
}

// ss::shamirSecretShare for N secrets on ONE prg, bit-identical to N sequential calls of include/scl/ss/shamir.h:52-68.
// `counter` is the number of AES blocks the PRG has produced so far (util::PRG would expose m_counter,
// include/scl/util/prg.h:168; 0 for a fresh PRG); `shares_dev` is SoA [n][N].
template <typename F>
void shamirSecretShareBatch(const uint64_t* secrets_dev, std::size_t N, std::size_t t, std::size_t n,
                            const std::array<unsigned char, 16>& seed, std::uint64_t counter, uint64_t* shares_dev) {
  check(scl_hip_shamir_share_prg(FieldTag<F>::value, shares_dev, N, secrets_dev, N, t, n, seed.data(), seed.size(), counter,
                                 nullptr));
}

// ss::shamirRecoverP (include/scl/ss/shamir.h:81-104) for N secrets, basis computed once
template <typename F>
void shamirRecoverBatch(const uint64_t* shares_dev, std::size_t n, std::size_t N, uint64_t* out_dev) {
  std::vector<uint64_t> lambda(n * FieldTag<F>::limbs);
  check(scl_hip_lagrange_basis(FieldTag<F>::value, lambda.data(), nullptr, n, nullptr));  // nodes 1..n, x = 0
  check(scl_hip_shamir_recover(FieldTag<F>::value, out_dev, shares_dev, N, lambda.data(), n, N, nullptr));
}

// The two above on host vectors, as a reference call site would use them: secrets in, one Vector of n shares per secret
// out (the reference's AoS shape, through scl_hip_soa_to_aos) -- and back.
template <typename F>
std::vector<math::Vector<math::FF<F>>> shamirSecretShare(const math::Vector<math::FF<F>>& secrets, std::size_t t,
                                                         std::size_t n, const std::array<unsigned char, 16>& seed,
                                                         std::uint64_t counter = 0) {
  using FF = math::FF<F>;
  const std::size_t N = secrets.size(), E = sizeof(FF);
  DevBuf ds(N * E), dsh(n * N * E), daos(n * N * E);
  check(scl_hip_memcpy_h2d(ds.p, secrets.toStlVector().data(), N * E, nullptr));
  shamirSecretShareBatch<F>(ds.u64(), N, t, n, seed, counter, dsh.u64());
  check(scl_hip_soa_to_aos(FieldTag<F>::value, daos.u64(), dsh.u64(), N, N, n, nullptr));
  std::vector<FF> flat(n * N);
  check(scl_hip_memcpy_d2h(flat.data(), daos.p, n * N * E, nullptr));
  std::vector<math::Vector<FF>> out;
  out.reserve(N);
  for (std::size_t s = 0; s < N; ++s) out.emplace_back(flat.begin() + s * n, flat.begin() + (s + 1) * n);
  return out;
}

template <typename F>
math::Vector<math::FF<F>> shamirRecoverP(const std::vector<math::Vector<math::FF<F>>>& shares) {
  using FF = math::FF<F>;
  const std::size_t N = shares.size(), n = N ? shares[0].size() : 0, E = sizeof(FF);
  std::vector<FF> flat;
  flat.reserve(n * N);
  for (const auto& v : shares) {
    if (v.size() != n) throw std::invalid_argument("Vec sizes mismatch");
    flat.insert(flat.end(), v.begin(), v.end());
  }
  DevBuf daos(n * N * E), dsoa(n * N * E), dout(N * E);
  check(scl_hip_memcpy_h2d(daos.p, flat.data(), n * N * E, nullptr));
  check(scl_hip_aos_to_soa(FieldTag<F>::value, dsoa.u64(), N, daos.u64(), N, n, nullptr));
  shamirRecoverBatch<F>(dsoa.u64(), n, N, dout.u64());
  std::vector<FF> out(N);
  check(scl_hip_memcpy_d2h(out.data(), dout.p, N * E, nullptr));
  return out;
}

// Vector::dot / innerProd (include/scl/math/vector.h:45-52,252-255) and Vector::sum (:261-267) on the GPU
template <typename F>
math::FF<F> dot(const math::Vector<math::FF<F>>& a, const math::Vector<math::FF<F>>& b) {
  if (a.size() != b.size()) throw std::invalid_argument("Vec sizes mismatch");
  const std::size_t bytes = a.byteSize();
  DevBuf da(bytes), db(bytes);
  check(scl_hip_memcpy_h2d(da.p, a.toStlVector().data(), bytes, nullptr));
  check(scl_hip_memcpy_h2d(db.p, b.toStlVector().data(), bytes, nullptr));
  math::FF<F> out;
  check(scl_hip_dot(FieldTag<F>::value, reinterpret_cast<uint64_t*>(&out), da.u64(), db.u64(), a.size(), nullptr));
  return out;
}

template <typename F>
math::FF<F> sum(const math::Vector<math::FF<F>>& a) {
  DevBuf da(a.byteSize());
  check(scl_hip_memcpy_h2d(da.p, a.toStlVector().data(), a.byteSize(), nullptr));
  math::FF<F> out;
  check(scl_hip_sum(FieldTag<F>::value, reinterpret_cast<uint64_t*>(&out), da.u64(), a.size(), nullptr));
  return out;
}

// ss::additiveShare (include/scl/ss/additive.h:41-53) for N secrets on ONE prg, bit-identical to N sequential calls (each of the
// n - 1 random shares is one FF::random = one AES block, ff.h:72-76); one Vector of n shares per secret out
template <typename F>
std::vector<math::Vector<math::FF<F>>> additiveShare(const math::Vector<math::FF<F>>& secrets, std::size_t n,
                                                     const std::array<unsigned char, 16>& seed, std::uint64_t counter = 0) {
  using FF = math::FF<F>;
  const std::size_t N = secrets.size(), E = sizeof(FF);
  DevBuf ds(N * E), dsh(n * N * E), daos(n * N * E);
  check(scl_hip_memcpy_h2d(ds.p, secrets.toStlVector().data(), N * E, nullptr));
  check(scl_hip_additive_share_prg(FieldTag<F>::value, dsh.u64(), N, ds.u64(), N, n, seed.data(), seed.size(), counter, nullptr));
  check(scl_hip_soa_to_aos(FieldTag<F>::value, daos.u64(), dsh.u64(), N, N, n, nullptr));
  std::vector<FF> flat(n * N);
  check(scl_hip_memcpy_d2h(flat.data(), daos.p, n * N * E, nullptr));
  std::vector<math::Vector<FF>> out;
  out.reserve(N);
  for (std::size_t s = 0; s < N; ++s) out.emplace_back(flat.begin() + s * n, flat.begin() + (s + 1) * n);
  return out;
}

// ss::shamirRecoverD(shares, t) (include/scl/ss/shamir.h:116-154) for N secrets: the batch call reports the secrets whose shares
// are inconsistent in `bad` (the reference throws "error detected during recovery" at each of them) and opens the others
template <typename F>
math::Vector<math::FF<F>> shamirRecoverD(const std::vector<math::Vector<math::FF<F>>>& shares, std::size_t t,
                                         std::vector<unsigned char>* bad) {
  using FF = math::FF<F>;
  const std::size_t N = shares.size(), n = N ? shares[0].size() : 0, E = sizeof(FF);
  std::vector<FF> flat;
  flat.reserve(n * N);
  for (const auto& v : shares) {
    if (v.size() != n) throw std::invalid_argument("Vec sizes mismatch");
    flat.insert(flat.end(), v.begin(), v.end());
  }
  DevBuf daos(n * N * E), dsoa(n * N * E), dout(N * E), dstatus(N);
  check(scl_hip_memcpy_h2d(daos.p, flat.data(), n * N * E, nullptr));
  check(scl_hip_aos_to_soa(FieldTag<F>::value, dsoa.u64(), N, daos.u64(), N, n, nullptr));
  std::size_t nbad = 0;
  const int st = scl_hip_shamir_recover_detect(FieldTag<F>::value, dout.u64(), static_cast<unsigned char*>(dstatus.p), dsoa.u64(), N, n,
                                               N, t, t, nullptr, nullptr, &nbad, nullptr);
  if (st != SCL_OK && st != SCL_ERR_ERROR_DETECTED) check(st);
  std::vector<FF> out(N);
  check(scl_hip_memcpy_d2h(out.data(), dout.p, N * E, nullptr));
  if (bad) {
    bad->resize(N);
    check(scl_hip_memcpy_d2h(bad->data(), dstatus.p, N, nullptr));
  } else if (nbad) {
    throw std::logic_error("error detected during recovery");
  }
  return out;
}

}  // namespace scl::hip

#endif  // SCL_HIP_BINDING_H
