// include/scl_hip/ss/additive.h -- scl::ss::additiveShare (include/scl/ss/additive.h:41-53) on the GPU;
// reconstruction is the per-secret sum (Vector::sum, vector.h:261-267).
#ifndef SCL_HIP_SS_ADDITIVE_H
#define SCL_HIP_SS_ADDITIVE_H

#include <stdexcept>
#include <vector>

#include "../hip/device.h"
#include "../math/vector.h"
#include "../util/prg.h"

namespace scl::ss {

/// batch: shares of secret s are what additiveShare(secrets[s], n, prg) returns on this PRG, in order
template <typename T>
hip::ShareMatrix<T> additiveShare(const hip::DeviceVector<T>& secrets, std::size_t n, util::PRG& prg) {
  if (n == 0) throw std::invalid_argument("cannot create shares for 0 people");
  hip::ShareMatrix<T> shares(n, secrets.size());
  const auto seed = prg.Seed();
  hip::check(scl_hip_additive_share_prg(T::Field::TAG, shares.data(), shares.stride(), secrets.data(), secrets.size(),
                                        n, seed.data(), seed.size(), prg.counter(), nullptr));
  prg.advance(secrets.size() * (n - 1) * ((T::byteSize() + 15) / 16));  // whole blocks per T::random (ff.h:72-76)
  return shares;
}

/// batch reconstruction: out[s] = sum_i shares[i][s]
template <typename T>
hip::DeviceVector<T> additiveRecover(const hip::ShareMatrix<T>& shares) {
  hip::DeviceVector<T> out(shares.secrets());
  hip::check(scl_hip_additive_recover(T::Field::TAG, out.data(), shares.data(), shares.stride(), shares.parties(),
                                      shares.secrets(), nullptr));
  return out;
}

/// additiveShare(secret, n, prg) (additive.h:41-53); reconstruct with shares.sum()
template <typename T>
math::Vector<T> additiveShare(const T& secret, std::size_t n, util::PRG& prg) {
  // n - 1 random elements, the last = secret - their sum; on the host like every per-secret signature (ss/shamir.h)
  if (n == 0) throw std::invalid_argument("cannot create shares for 0 people");
  std::vector<T> shares;
  shares.reserve(n);
  T rest = secret;
  for (std::size_t i = 0; i + 1 < n; ++i) {
    shares.emplace_back(T::random(prg));
    rest -= shares.back();
  }
  shares.emplace_back(rest);
  return math::Vector<T>(std::move(shares));
}

}  // namespace scl::ss

#endif
