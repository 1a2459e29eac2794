// include/scl_hip/math/poly.h -- scl::math::Polynomial<T> (include/scl/math/poly.h:31-296): coefficient
// vector, create() trims zero top coefficients, Horner evaluate().  One polynomial at one point is
// scalar host work (as in the reference); evaluating many polynomials at many points is the share
// kernel (scl::ss / scl_hip_shamir_share).
#ifndef SCL_HIP_MATH_POLY_H
#define SCL_HIP_MATH_POLY_H

#include <algorithm>
#include <array>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "vector.h"

namespace scl::math {

template <typename RING>
class Polynomial {
 public:
  /// drops zero coefficients from the top; the zero polynomial keeps one zero coefficient (poly.h:178-198)
  static Polynomial create(const Vector<RING>& coefficients) {
    std::size_t keep = coefficients.size();
    while (keep > 0 && coefficients[keep - 1] == RING{}) --keep;
    if (keep == 0) return Polynomial{};
    return Polynomial(coefficients.subVector(keep));
  }

  Polynomial() : m_coefficients(1) {}
  Polynomial(const RING& constant) : m_coefficients({constant}) {}

  /// Horner from the top coefficient: y = c_k + y * x (poly.h:56-64)
  RING evaluate(const RING& x) const {
    std::size_t k = m_coefficients.size();
    RING y = m_coefficients[--k];
    while (k > 0) y = m_coefficients[--k] + y * x;
    return y;
  }

  RING& operator[](std::size_t idx) { return m_coefficients[idx]; }
  RING operator[](std::size_t idx) const { return m_coefficients[idx]; }
  Vector<RING> coefficients() const { return m_coefficients; }
  RING constantTerm() const { return m_coefficients[0]; }
  RING leadingTerm() const { return m_coefficients[degree()]; }
  std::size_t degree() const { return m_coefficients.size() - 1; }
  bool isZero() const { return degree() == 0 && m_coefficients[0] == RING{}; }

  Polynomial add(const Polynomial& q) const { return combine(q, false); }
  Polynomial subtract(const Polynomial& q) const { return combine(q, true); }

  /// schoolbook product (poly.h:237-255)
  Polynomial multiply(const Polynomial& q) const {
    std::vector<RING> c(degree() + q.degree() + 1);
    for (std::size_t i = 0; i <= degree(); ++i)
      for (std::size_t j = 0; j <= q.degree(); ++j) c[i + j] += m_coefficients[i] * q.m_coefficients[j];
    return create(Vector<RING>(std::move(c)));
  }

  /// long division: {quotient, remainder}; division by zero throws (poly.h:261-278)
  std::array<Polynomial, 2> divide(const Polynomial& q) const {
    if (q.isZero()) throw std::invalid_argument("division by 0");
    Polynomial quo, rem = *this;
    const RING lead_inv = q.leadingTerm().inverse();
    while (!rem.isZero() && rem.degree() >= q.degree()) {
      const std::size_t shift = rem.degree() - q.degree();
      const RING f = rem.leadingTerm() * lead_inv;
      std::vector<RING> mono(shift + 1);
      mono[shift] = f;
      const Polynomial term = create(Vector<RING>(std::move(mono)));
      quo = quo.add(term);
      rem = rem.subtract(term.multiply(q));
    }
    return {quo, rem};
  }

  std::string toString(const char* polynomial_name = "f", const char* variable_name = "x") const {
    std::stringstream ss;
    ss << polynomial_name << "(" << variable_name << ") = " << m_coefficients[0];
    for (std::size_t i = 1; i < m_coefficients.size(); ++i) {
      ss << " + " << m_coefficients[i] << variable_name;
      if (i > 1) ss << "^" << i;
    }
    return ss.str();
  }
  friend std::ostream& operator<<(std::ostream& os, const Polynomial& p) { return os << p.toString(); }

 private:
  explicit Polynomial(const Vector<RING>& c) : m_coefficients(c) {}

  Polynomial combine(const Polynomial& q, bool minus) const {
    const std::size_t n = std::max(m_coefficients.size(), q.m_coefficients.size());
    std::vector<RING> c(n);
    for (std::size_t i = 0; i < n; ++i) {
      const RING a = i < m_coefficients.size() ? m_coefficients[i] : RING{};
      const RING b = i < q.m_coefficients.size() ? q.m_coefficients[i] : RING{};
      c[i] = minus ? a - b : a + b;
    }
    return create(Vector<RING>(std::move(c)));
  }

  Vector<RING> m_coefficients;
};

}  // namespace scl::math

#endif
