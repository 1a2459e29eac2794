// include/scl_hip/ss/shamir.h -- scl::ss Shamir sharing (include/scl/ss/shamir.h:51-155) on the GPU.
//
// The reference works one secret per call and returns a heap Vector of n shares; those signatures
// are kept and run on the host, over FF's operators (see "per-secret" below).  The batch forms in scl::ss
// take many secrets at once, keep the shares in HBM as a ShareMatrix (SoA [party][secret]) and run the kernels.
// Results are bit-identical to the per-secret reference calls driven by the same PRG.
#ifndef SCL_HIP_SS_SHAMIR_H
#define SCL_HIP_SS_SHAMIR_H

#include <cstdint>
#include <vector>

#include "../hip/device.h"
#include "../math/array.h"
#include "../math/lagrange.h"
#include "../math/matrix.h"
#include "../math/poly.h"
#include "../math/vector.h"
#include "../util/prg.h"

namespace scl::ss {

namespace shamir_detail {
template <typename T>
constexpr std::size_t limbs() {
  return hip::limbsOf<T>();
}
/// AES blocks one shamirSecretShare call draws: Vector::random(t+1) = ceil((t+1)*byteSize/16)
template <typename T>
std::uint64_t blocksPerSecret(std::size_t t) {
  return ((t + 1) * T::byteSize() + 15) / 16;
}
template <typename T>
std::vector<std::uint64_t> toLimbs(const math::Vector<T>& v) {
  std::vector<std::uint64_t> out(v.size() * limbs<T>() + 1);
  for (std::size_t i = 0; i < v.size(); ++i) v[i].toLimbs(out.data() + i * limbs<T>());
  return out;
}
}  // namespace shamir_detail

// ------------------------------------------------------------------------------------------- batch
/// shamirSecretShare for a whole batch: secret s is shared exactly as the reference's
/// shamirSecretShare(secrets[s], t, n, prg) would on this PRG, in order.  The PRG is advanced.
template <typename T>
hip::ShareMatrix<T> shamirSecretShare(const hip::DeviceVector<T>& secrets, std::size_t t, std::size_t n,
                                      util::PRG& prg) {
  hip::ShareMatrix<T> shares(n, secrets.size());
  const auto seed = prg.Seed();
  hip::check(scl_hip_shamir_share_prg(T::Field::TAG, shares.data(), shares.stride(), secrets.data(), secrets.size(), t,
                                      n, seed.data(), seed.size(), prg.counter(), nullptr));
  prg.advance(secrets.size() * shamir_detail::blocksPerSecret<T>(t));
  return shares;
}

/// the same with caller-provided coefficient vectors c_1..c_t (each of secrets.size() elements)
template <typename T>
hip::ShareMatrix<T> shamirSecretShare(const hip::DeviceVector<T>& secrets,
                                      const std::vector<hip::DeviceVector<T>>& coefficients, std::size_t n) {
  const std::size_t N = secrets.size(), t = coefficients.size();
  hip::DeviceVector<T> packed(t * N);
  for (std::size_t k = 0; k < t; ++k) {
    if (coefficients[k].size() != N) hip::detail::raise(SCL_ERR_SIZE_MISMATCH);
    hip::check(scl_hip_stream_copy(packed.data() + k * N * shamir_detail::limbs<T>(), coefficients[k].data(),
                                   N * T::byteSize(), nullptr));
  }
  hip::ShareMatrix<T> shares(n, N);
  hip::check(scl_hip_shamir_share(T::Field::TAG, shares.data(), shares.stride(), secrets.data(), packed.data(), N, N, t,
                                  n, nullptr, nullptr));
  hip::check(scl_hip_stream_sync(nullptr));
  return shares;
}

// ---- sharing over math::Array<FF, W> for a whole batch (what pedersenSecretShare shares: {secret, blinding}) ----------
/// W components of N secrets each in HBM, component-major: component j of secret s is element j * N + s.
template <typename F, std::size_t W>
struct ArrayVector {
  hip::DeviceVector<F> components;  ///< W * N elements
  std::size_t secrets = 0;

  ArrayVector() = default;
  explicit ArrayVector(const math::Vector<math::Array<F, W>>& host) : secrets(host.size()) {
    std::vector<F> flat(W * secrets);
    for (std::size_t s = 0; s < secrets; ++s)
      for (std::size_t j = 0; j < W; ++j) flat[j * secrets + s] = host[s][j];
    components = hip::DeviceVector<F>(flat);
  }
};

/// the shares of a batch over Array<F, W>: component j of party i's share of secret s is row j * n + i, column s
template <typename F, std::size_t W>
struct ArrayShares {
  hip::ShareMatrix<F> rows;  ///< [W * n][N]
  std::size_t parties = 0;

  /// the shares of secret s as shamirSecretShare(Array, t, n, prg) returns them: n Arrays
  math::Vector<math::Array<F, W>> sharesOf(std::size_t s) const {
    const auto col = rows.sharesOf(s);  // W * n elements, row order
    std::vector<math::Array<F, W>> out(parties);
    for (std::size_t i = 0; i < parties; ++i)
      for (std::size_t j = 0; j < W; ++j) out[i][j] = col[j * parties + i];
    return math::Vector<math::Array<F, W>>(std::move(out));
  }
};

/// shamirSecretShare(Array<F, W>{..}, t, n, prg) (shamir.h:51-68 with T = Array, pedersen.h:137-138) for every secret of
/// the batch on ONE prg, in order: Vector<Array>::random(t + 1) is one draw of (t + 1) * W elements per secret, component j
/// of coefficient k is element k * W + j, arithmetic is component-wise, nodes 1..n.  The PRG is advanced.
template <typename F, std::size_t W>
ArrayShares<F, W> shamirSecretShare(const ArrayVector<F, W>& secrets, std::size_t t, std::size_t n, util::PRG& prg) {
  const std::size_t N = secrets.secrets;
  ArrayShares<F, W> out{hip::ShareMatrix<F>(W * n, N), n};
  const auto seed = prg.Seed();
  hip::check(scl_hip_shamir_share_prg_packed(F::Field::TAG, out.rows.data(), out.rows.stride(), secrets.components.data(), N, N,
                                             t, n, W, seed.data(), seed.size(), prg.counter(), nullptr));
  prg.advance(N * (((t + 1) * W * F::byteSize() + 15) / 16));
  return out;
}

namespace shamir_detail {
/// out[s] = sum_{i < parties} lambda[i] * shares[i][s]
template <typename T>
hip::DeviceVector<T> recoverWithBasis(const hip::ShareMatrix<T>& shares, const math::Vector<T>& lambda) {
  if (lambda.size() < shares.parties()) hip::detail::raise(SCL_ERR_SIZE_MISMATCH);
  const auto lam = toLimbs(lambda);
  hip::DeviceVector<T> out(shares.secrets());
  hip::check(scl_hip_shamir_recover(T::Field::TAG, out.data(), shares.data(), shares.stride(), lam.data(),
                                    shares.parties(), shares.secrets(), nullptr));
  return out;
}
}  // namespace shamir_detail

/// shamirRecoverP for every secret of a share matrix: basis for `alphas` at `x`, computed once
/// (the reference recomputes it -- n(n-1) field inversions -- for every secret, shamir.h:85)
template <typename T>
hip::DeviceVector<T> shamirRecoverP(const hip::ShareMatrix<T>& shares, const math::Vector<T>& alphas, const T& x) {
  return shamir_detail::recoverWithBasis(shares, math::computeLagrangeBasis(alphas, x));
}

template <typename T>
hip::DeviceVector<T> shamirRecoverP(const hip::ShareMatrix<T>& shares) {
  return shamirRecoverP(shares, math::Vector<T>::range(1, shares.parties() + 1), T{});
}

/// batch shamirRecoverD(shares, t): returns the values; `bad` (optional) receives the indices of the
/// secrets for which the reference would throw "error detected during recovery"
template <typename T>
hip::DeviceVector<T> shamirRecoverD(const hip::ShareMatrix<T>& shares, std::size_t t,
                                    std::vector<std::size_t>* bad = nullptr) {
  hip::DeviceVector<T> out(shares.secrets());
  hip::DeviceBuffer status(shares.secrets() ? shares.secrets() : 1);
  std::size_t nbad = 0;
  const int st = scl_hip_shamir_recover_detect(T::Field::TAG, out.data(), static_cast<unsigned char*>(status.get()),
                                               shares.data(), shares.stride(), shares.parties(), shares.secrets(), t,
                                               t, nullptr, nullptr, &nbad, nullptr);
  if (st != SCL_OK && st != SCL_ERR_ERROR_DETECTED) hip::detail::raise(st);
  if (st == SCL_ERR_ERROR_DETECTED && bad == nullptr) hip::detail::raise(st);
  if (bad) {
    bad->clear();
    std::vector<unsigned char> h(shares.secrets());
    if (!h.empty()) hip::check(scl_hip_memcpy_d2h(h.data(), status.get(), h.size(), nullptr));
    for (std::size_t i = 0; i < h.size(); ++i)
      if (h[i]) bad->push_back(i);
  }
  return out;
}

// ------------------------------------------------------------------------------- per-secret (reference)
// The reference's own signatures: ONE secret per call, values in host memory.  They run the reference's own steps on the
// host through FF's operators (detail/field.hpp: the arithmetic the kernels compile) -- a device round trip per call
// would cost two orders of magnitude more than the reference's 0.5 us (hip/device.h, "where a host-resident operand is
// worked on").  Bit-identical to the batch kernels on the same PRG (every operation returns the canonical representative).
namespace shamir_detail {
/// the i-th evaluation point, i = 1, 2, ..: x++ on T::one() as in shamir.h:62-65 -- except over GF(2^128), where that walk
/// cycles 1, 0, 1, .. and the nodes are the bit patterns of 1, 2, .. (T(int)), as in every batch path
template <typename T>
T nextNode(const T& x, std::size_t i) {
  if constexpr (requires { T::Field::TAG; }) {
    if constexpr (T::Field::TAG == SCL_GF2_128) return T(static_cast<int>(i + 1));
  }
  T y = x;
  return ++y;
}
}  // namespace shamir_detail

/// shamirSecretShare(secret, t, n, prg) (shamir.h:51-68)
template <typename T>
math::Vector<T> shamirSecretShare(const T& secret, std::size_t t, std::size_t n, util::PRG& prg) {
  // random coefficients, c_0 = secret, evaluate at 1, 2, .. produced by x++ on T::one()
  auto c = math::Vector<T>::random(t + 1, prg);
  c[0] = secret;
  const auto p = math::Polynomial<T>::create(c);
  std::vector<T> shares;
  shares.reserve(n);
  T x = T::one();
  for (std::size_t i = 1; i <= n; ++i) {
    shares.emplace_back(p.evaluate(x));
    x = shamir_detail::nextNode(x, i);
  }
  return math::Vector<T>(std::move(shares));
}

/// shamirRecoverP(shares, alphas, x) (shamir.h:81-87): the basis is taken over ALL alphas and paired
/// with the shares in order, as innerProd does there
template <typename T>
T shamirRecoverP(const math::Vector<T>& shares, const math::Vector<T>& alphas, const T& x) {
  const auto lb = math::computeLagrangeBasis(alphas, x);
  return math::innerProd<T>(shares.begin(), shares.end(), lb.begin());
}

/// shamirRecoverP(shares) (shamir.h:99-104): nodes 1..size, x = 0
template <typename T>
T shamirRecoverP(const math::Vector<T>& shares) {
  return shamirRecoverP(shares, math::Vector<T>::range(1, shares.size() + 1), T{});
}

/// shamirRecoverD(shares, alphas, t, d, x) (shamir.h:116-139)
template <typename T>
T shamirRecoverD(const math::Vector<T>& shares, const math::Vector<T>& alphas, std::size_t t, std::size_t d,
                 const T& x) {
  if (shares.size() < d + t || alphas.size() < d + t) hip::detail::raise(SCL_ERR_NOT_ENOUGH_SHARES);
  if (shares.size() < d + 1 || alphas.size() < d + 1) hip::detail::raise(SCL_ERR_INVALID_RANGE);  // t = 0 with d shares
  const std::size_t m1 = d + 1;
  const auto ns = alphas.subVector(m1);
  for (std::size_t i = m1; i < d + t; ++i) {
    const auto lb = math::computeLagrangeBasis(ns, alphas[i]);
    if (math::innerProd<T>(shares.begin(), shares.begin() + static_cast<std::ptrdiff_t>(m1), lb.begin()) != shares[i])
      hip::detail::raise(SCL_ERR_ERROR_DETECTED);
  }
  const auto lb = math::computeLagrangeBasis(ns, x);
  return math::innerProd<T>(shares.begin(), shares.begin() + static_cast<std::ptrdiff_t>(m1), lb.begin());
}

/// shamirRecoverD(shares, t) (shamir.h:150-155): n = 2t+1 nodes 1..n, d = t, x = 0
template <typename T>
T shamirRecoverD(const math::Vector<T>& shares, std::size_t t) {
  const std::size_t n = 2 * t + 1;
  return shamirRecoverD(shares, math::Vector<T>::range(1, n + 1), t, t, T{});
}

// ------------------------------------------------------------------------------- error correction
/// what shamirRecoverC returns per secret (shamir.h:160-180): the corrected polynomial (the secret is
/// f.evaluate(0)) and the monic error locator, whose roots are the nodes of the corrupted shares
template <typename T>
struct ErrorCorrectedSecret {
  math::Polynomial<T> f;
  math::Polynomial<T> err;
};

/// batch shamirRecoverC (Berlekamp-Welch, shamir.h:202-259) over device-resident shares: t = (parties-1)/3, the
/// first 3t+1 share vectors are used.  Row k of `f` / `err` holds coefficient k for every secret (zero padded).
template <typename T>
struct ErrorCorrectedBatch {
  hip::ShareMatrix<T> f;      ///< [3t+1][N]; row 0 = the secrets
  hip::ShareMatrix<T> err;    ///< [t+1][N]
  std::vector<unsigned char> status;  ///< 1 where the reference throws "could not correct shares"
  std::vector<unsigned> errors;       ///< degree of the locator = number of shares corrected
  std::size_t solved = 0;     ///< secrets that needed the linear systems (the others were already consistent)
  std::size_t failed = 0;

  ErrorCorrectedSecret<T> at(std::size_t s) const {
    if (status[s]) throw std::logic_error("could not correct shares");
    return {math::Polynomial<T>::create(math::Vector<T>(f.sharesOf(s))),
            math::Polynomial<T>::create(math::Vector<T>(err.sharesOf(s)))};
  }
};

template <typename T>
ErrorCorrectedBatch<T> shamirRecoverC(const hip::ShareMatrix<T>& shares, const math::Vector<T>* alphas = nullptr) {
  const std::size_t m = shares.parties(), N = shares.secrets();
  if (m == 0) throw std::invalid_argument("no shares");
  const std::size_t t = (m - 1) / 3, n = 3 * t + 1;
  ErrorCorrectedBatch<T> out{hip::ShareMatrix<T>(n, N), hip::ShareMatrix<T>(t + 1, N), {}, {}, 0, 0};
  hip::DeviceBuffer status(N ? N : 1), nerr((N ? N : 1) * sizeof(unsigned));
  std::vector<std::uint64_t> al;
  if (alphas) al = shamir_detail::toLimbs(alphas->subVector(n));
  hip::check(scl_hip_shamir_recover_correct(T::Field::TAG, out.f.data(), out.f.stride(), out.err.data(), out.err.stride(),
                                            static_cast<unsigned char*>(status.get()),
                                            static_cast<unsigned*>(nerr.get()), shares.data(), shares.stride(), m, N,
                                            alphas ? al.data() : nullptr, &out.solved, &out.failed, nullptr));
  out.status.resize(N);
  out.errors.resize(N);
  if (N) {
    hip::check(scl_hip_memcpy_d2h(out.status.data(), status.get(), N, nullptr));
    hip::check(scl_hip_memcpy_d2h(out.errors.data(), nerr.get(), N * sizeof(unsigned), nullptr));
  }
  return out;
}

namespace shamir_detail {
/// The unique solution of the n x n system A x = b, or false when A is singular (what the reference's
/// solveLinearSystem answers for "unique solutions only", matrix.h:811-828).  Gauss-Jordan with row swaps; A and b are
/// consumed.
template <typename T>
bool solveUnique(std::vector<std::vector<T>>& A, std::vector<T>& b, std::vector<T>& x) {
  const std::size_t n = b.size();
  for (std::size_t c = 0; c < n; ++c) {
    std::size_t piv = c;
    while (piv < n && A[piv][c] == T{}) ++piv;
    if (piv == n) return false;
    std::swap(A[piv], A[c]);
    std::swap(b[piv], b[c]);
    const T inv = A[c][c].inverse();
    for (std::size_t j = c; j < n; ++j) A[c][j] *= inv;
    b[c] *= inv;
    for (std::size_t r = 0; r < n; ++r) {
      if (r == c || A[r][c] == T{}) continue;
      const T f = A[r][c];
      for (std::size_t j = c; j < n; ++j) A[r][j] -= f * A[c][j];
      b[r] -= f * b[c];
    }
  }
  x = b;
  return true;
}

/// Berlekamp-Welch on the host for element types without kernels (shamir.h:202-250): for e = t, t-1, .. 0 the system
/// s_i E(a_i) = Q(a_i) with E monic of degree e and deg Q <= n-1-e; the first e whose system has a unique solution is
/// taken, f = Q / E, and a non-zero remainder is "could not correct shares".
template <typename T>
ErrorCorrectedSecret<T> recoverCHost(const math::Vector<T>& shares, const math::Vector<T>& alphas) {
  if (shares.empty()) throw std::invalid_argument("no shares");
  const std::size_t t = (shares.size() - 1) / 3, n = 3 * t + 1;
  std::vector<T> x;
  std::size_t e = t;
  for (;; --e) {
    std::vector<std::vector<T>> A(n, std::vector<T>(n));
    std::vector<T> b(n);
    for (std::size_t i = 0; i < n; ++i) {
      T pw = shares[i];  // s_i a_i^j for the e locator coefficients, then -a_i^(j-e) for Q's
      for (std::size_t j = 0; j < e; ++j) {
        A[i][j] = pw;
        pw *= alphas[i];
      }
      b[i] = -pw;  // - s_i a_i^e: the monic term moved to the right-hand side
      pw = -T(1);
      for (std::size_t j = e; j < n; ++j) {
        A[i][j] = pw;
        pw *= alphas[i];
      }
    }
    if (solveUnique(A, b, x)) break;
    if (e == 0) throw std::logic_error("could not correct shares");  // duplicate nodes: not even interpolation is unique
  }
  std::vector<T> cE(x.begin(), x.begin() + static_cast<std::ptrdiff_t>(e));
  cE.emplace_back(T(1));
  const auto E = math::Polynomial<T>::create(math::Vector<T>(std::move(cE)));
  const auto Q = math::Polynomial<T>::create(math::Vector<T>(x.begin() + static_cast<std::ptrdiff_t>(e), x.end()));
  const auto qr = Q.divide(E);
  if (!qr[1].isZero()) throw std::logic_error("could not correct shares");
  return {qr[0], E};
}
}  // namespace shamir_detail

/// shamirRecoverC(shares, alphas) (shamir.h:202-250)
template <typename T>
ErrorCorrectedSecret<T> shamirRecoverC(const math::Vector<T>& shares, const math::Vector<T>& alphas) {
  return shamir_detail::recoverCHost(shares, alphas);
}

/// shamirRecoverC(shares) (shamir.h:255-259): nodes 1..size
template <typename T>
ErrorCorrectedSecret<T> shamirRecoverC(const math::Vector<T>& shares) {
  return shamir_detail::recoverCHost(shares, math::Vector<T>::range(1, shares.size() + 1));
}

}  // namespace scl::ss

#endif
