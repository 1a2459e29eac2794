// include/scl_hip/math/lagrange.h -- computeLagrangeBasis (include/scl/math/lagrange.h:54-82).
// O(n^2) table work done once on the host by the library (scl_hip_lagrange_basis); duplicate nodes
// raise std::logic_error("0 not invertible modulo prime") exactly like the reference's division.
#ifndef SCL_HIP_MATH_LAGRANGE_H
#define SCL_HIP_MATH_LAGRANGE_H

#include <vector>

#include "vector.h"

namespace scl::math {

template <typename T>
Vector<T> computeLagrangeBasis(const Vector<T>& nodes, const T& x) {
  if constexpr (!OnDevice<T>) {  // ell_i = prod_{j != i} (x - x_j) / (x_i - x_j), one division per factor (lagrange.h:54-71)
    const std::size_t n = nodes.size();
    std::vector<T> b;
    b.reserve(n);
    for (std::size_t i = 0; i < n; ++i) {
      T ell(1);
      for (std::size_t j = 0; j < n; ++j)
        if (i != j) ell *= (x - nodes[j]) / (nodes[i] - nodes[j]);
      b.emplace_back(ell);
    }
    return Vector<T>(std::move(b));
  } else {
  constexpr std::size_t L = hip::limbsOf<T>();
  const std::size_t n = nodes.size();
  std::vector<std::uint64_t> nd(n * L + 1), out(n * L + 1);
  std::uint64_t xl[L];
  for (std::size_t i = 0; i < n; ++i) nodes[i].toLimbs(nd.data() + i * L);
  x.toLimbs(xl);
  hip::check(scl_hip_lagrange_basis(T::Field::TAG, out.data(), nd.data(), n, xl));
  std::vector<T> b;
  b.reserve(n);
  for (std::size_t i = 0; i < n; ++i) b.emplace_back(T::fromLimbs(out.data() + i * L));
  return Vector<T>(std::move(b));
  }
}

template <typename T>
Vector<T> computeLagrangeBasis(const Vector<T>& nodes, int x) {
  return computeLagrangeBasis(nodes, T{x});
}

}  // namespace scl::math

#endif
