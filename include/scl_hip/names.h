// include/scl_hip/names.h -- the spellings BASELINE.json's north star and the reference's own doc comments use for the hot
// path's API, as aliases of the mirror's (= the reference's current) names:
//
//   scl::Vec<T>, scl::math::Vec<T>          math::Vector<T>      ("a math::Vec object", include/scl/ss/additive.h:38)
//   scl::Mat<T>, scl::math::Mat<T>          math::Matrix<T>
//   scl::ss::ShamirShare(..)                ss::shamirSecretShare(..)   ("as obtained from ss::ShamirShare", ss/shamir.h:94,167)
//   scl::ss::ShamirReconstruct(..)          ss::shamirRecoverP(..)      (BASELINE.json: "ShamirShare/Reconstruct")
//   scl::ss::ShamirRecoverP / D / C(..)     ss::shamirRecoverP / D / C  ("identical to ss::ShamirRecoverP", ss/shamir.h:94)
//   scl::ss::AdditiveSS(..)                 ss::additiveShare(..)       (BASELINE.json: "AdditiveSS")
//   scl::ss::AdditiveReconstruct(shares)    shares.sum() per secret / ss::additiveRecover(share matrix)
//
// Every overload of the target is reachable: the per-secret reference signatures (host) and the batch forms over
// hip::DeviceVector / hip::ShareMatrix (kernels).  Nothing here adds behaviour.
#ifndef SCL_HIP_NAMES_H
#define SCL_HIP_NAMES_H

#include <utility>

#include "math/matrix.h"
#include "math/vector.h"
#include "ss/additive.h"
#include "ss/shamir.h"

namespace scl {

namespace math {
template <typename T>
using Vec = Vector<T>;
template <typename T>
using Mat = Matrix<T>;
}  // namespace math

template <typename T>
using Vec = math::Vector<T>;
template <typename T>
using Mat = math::Matrix<T>;

namespace ss {

#define SCL_HIP_FORWARD(alias, target)                                                          \
  template <typename... Args>                                                                   \
  auto alias(Args&&... args) -> decltype(target(std::forward<Args>(args)...)) {                 \
    return target(std::forward<Args>(args)...);                                                 \
  }
SCL_HIP_FORWARD(ShamirShare, shamirSecretShare)
SCL_HIP_FORWARD(ShamirReconstruct, shamirRecoverP)
SCL_HIP_FORWARD(ShamirRecoverP, shamirRecoverP)
SCL_HIP_FORWARD(ShamirRecoverD, shamirRecoverD)
SCL_HIP_FORWARD(ShamirRecoverC, shamirRecoverC)
SCL_HIP_FORWARD(AdditiveSS, additiveShare)
#undef SCL_HIP_FORWARD

/// reconstruct one additively shared secret: the sum of its shares (vector.h:261-267)
template <typename T>
T AdditiveReconstruct(const math::Vector<T>& shares) {
  return shares.sum();
}
/// .. and a batch of them
template <typename T>
hip::DeviceVector<T> AdditiveReconstruct(const hip::ShareMatrix<T>& shares) {
  return additiveRecover(shares);
}

}  // namespace ss
}  // namespace scl

#endif
