// include/scl_hip/math/array.h -- scl::math::Array<T, N> (include/scl/math/array.h:69-415): N values of one type handled as
// ONE element of the N-fold direct product -- every operator works component by component.  It is what the reference
// shares when a secret travels with its blinding value: pedersenSecretShare runs shamirSecretShare over
// Array<FF, 2>{{secret, randomness}} (include/scl/ss/pedersen.h:127-140).  Host type, like FF: the per-secret scl::ss
// templates take it as their T; the batch form (ss/shamir.h, shamirSecretShare over hip::ArrayVector) keeps the components
// in HBM and runs scl_hip_shamir_share_prg_packed.
#ifndef SCL_HIP_MATH_ARRAY_H
#define SCL_HIP_MATH_ARRAY_H

#include <array>
#include <cstddef>
#include <ostream>
#include <sstream>
#include <string>
#include <utility>

#include "../util/prg.h"

namespace scl::math {

template <typename T, std::size_t N>
class Array final {
  static_assert(N > 0, "an Array holds at least one value");

 public:
  /// bytes of the wire image: the N components one after the other (array.h:75-77)
  constexpr static std::size_t byteSize() { return N * T::byteSize(); }

  /// array.h:82-88: component i from src + i * T::byteSize()
  static Array read(const unsigned char* src) {
    Array a;
    std::size_t at = 0;
    for (T& v : a.m_values) {
      v = T::read(src + at);
      at += T::byteSize();
    }
    return a;
  }

  /// array.h:93-101: N draws of T::random, in order -- each its own prg.next (for an FF: one AES block per component)
  static Array random(util::PRG& prg)
    requires requires { T::random(prg); }
  {
    Array a;
    for (T& v : a.m_values) v = T::random(prg);
    return a;
  }

  static Array one()
    requires requires { T::one(); }
  {
    return Array(T::one());
  }
  static Array zero() { return Array(T::zero()); }

  Array() : m_values{} {}                      ///< N default-constructed values (FF(): zero)
  Array(const T& element) { m_values.fill(element); }   ///< the same value in every component (array.h:128-130)
  explicit Array(int value) : Array(T{value}) {}
  Array(const std::array<T, N>& values) : m_values(values) {}
  Array(std::array<T, N>&& values) : m_values(std::move(values)) {}

  Array& operator+=(const Array& o) { return zip(o, [](T& a, const T& b) { a += b; }); }
  Array& operator-=(const Array& o) { return zip(o, [](T& a, const T& b) { a -= b; }); }
  friend Array operator+(const Array& a, const Array& b) { return Array(a) += b; }
  friend Array operator-(const Array& a, const Array& b) { return Array(a) -= b; }

  Array& operator++()
    requires requires(T v) { ++v; }
  {
    for (T& v : m_values) ++v;
    return *this;
  }
  friend Array operator++(Array& a, int)
    requires requires(T v) { ++v; }
  {
    const Array before(a);
    ++a;
    return before;
  }
  Array& operator--()
    requires requires(T v) { --v; }
  {
    for (T& v : m_values) --v;
    return *this;
  }
  friend Array operator--(Array& a, int)
    requires requires(T v) { --v; }
  {
    const Array before(a);
    --a;
    return before;
  }

  Array& negate() {
    for (T& v : m_values) v.negate();
    return *this;
  }

  /// every component times one scalar (array.h:251-259)
  template <typename S>
  Array& operator*=(const S& scalar)
    requires requires(T v, const S& s) { v *= s; }
  {
    for (T& v : m_values) v *= scalar;
    return *this;
  }
  /// component i times other[i] (array.h:264-272)
  template <typename S>
  Array& operator*=(const Array<S, N>& other)
    requires requires(T v, const S& s) { v *= s; }
  {
    for (std::size_t i = 0; i < N; ++i) m_values[i] *= other[i];
    return *this;
  }
  template <typename S>
  Array operator*(const S& scalar) const
    requires requires(T v, const S& s) { v *= s; }
  {
    return Array(*this) *= scalar;
  }
  /// component-wise product of two arrays, possibly of different value types (a group element times a scalar:
  /// array.h:288-298); the result holds whatever T * V yields
  template <typename V>
  friend auto operator*(const Array& lhs, const Array<V, N>& rhs)
    requires requires(const T& a, const V& b) { a * b; }
  {
    Array<decltype(std::declval<T>() * std::declval<V>()), N> out;
    for (std::size_t i = 0; i < N; ++i) out[i] = lhs[i] * rhs[i];
    return out;
  }

  Array& invert()
    requires requires(T v) { v.invert(); }
  {
    for (T& v : m_values) v.invert();
    return *this;
  }
  Array Inverse() const
    requires requires(T v) { v.invert(); }
  {
    return Array(*this).invert();
  }
  Array operator/=(const Array& o)
    requires requires(T v, const T& w) { v /= w; }
  {
    return zip(o, [](T& a, const T& b) { a /= b; });
  }
  Array operator/(const Array& o) const
    requires requires(T v, const T& w) { v /= w; }
  {
    return Array(*this) /= o;
  }

  T& operator[](std::size_t i) { return m_values[i]; }
  T operator[](std::size_t i) const { return m_values[i]; }

  /// array.h:362-368: every pair is compared (no early exit)
  bool equal(const Array& o) const {
    bool same = true;
    for (std::size_t i = 0; i < N; ++i) same = (m_values[i] == o.m_values[i]) && same;
    return same;
  }
  friend bool operator==(const Array& a, const Array& b) { return a.equal(b); }
  friend bool operator!=(const Array& a, const Array& b) { return !a.equal(b); }

  /// "P{v0, v1, ..}" (array.h:387-395)
  std::string toString() const {
    std::ostringstream os;
    os << "P{";
    for (std::size_t i = 0; i < N; ++i) os << (i ? ", " : "") << m_values[i];
    os << "}";
    return os.str();
  }
  friend std::ostream& operator<<(std::ostream& os, const Array& a) { return os << a.toString(); }

  void write(unsigned char* dest) const {
    for (const T& v : m_values) {
      v.write(dest);
      dest += T::byteSize();
    }
  }

  const std::array<T, N>& values() const { return m_values; }

 private:
  template <typename Fn>
  Array& zip(const Array& o, Fn&& fn) {
    for (std::size_t i = 0; i < N; ++i) fn(m_values[i], o.m_values[i]);
    return *this;
  }

  std::array<T, N> m_values;
};

}  // namespace scl::math

#endif  // SCL_HIP_MATH_ARRAY_H
