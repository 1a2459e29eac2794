// include/scl_hip/math/vector.h -- scl::math::Vector<T> with its batch members on the GPU.
//
// Surface of include/scl/math/vector.h:45-586 (names, SizeType, error text).  Storage is a host
// std::vector like the reference's; for element types with kernels (math::OnDevice: the built-in fields and rings)
// an element-wise / reduction member of a vector at or above hip::hostThreshold() uploads its operands, runs the HIP
// kernel behind the C ABI and downloads the result -- a drop-in, PCIe-bound for large vectors; a shorter vector (the
// reference's typical one: the n shares of a secret) is worked on where it lives, by FF's operators (hip/device.h).  Code that wants the data to stay in HBM uses
// scl::hip::DeviceVector with the free functions of scl::hip (same kernels, no transfers).  An element type over a
// user-defined field (fields/ff_ops.h) takes the reference's per-element loops on the host instead.
#ifndef SCL_HIP_MATH_VECTOR_H
#define SCL_HIP_MATH_VECTOR_H

#include <cstdint>
#include <initializer_list>
#include <iterator>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../hip/device.h"
#include "../util/prg.h"
#include "ff.h"

namespace scl::math {

template <typename T>
class Matrix;

namespace vec_detail {

template <typename T>
constexpr int tag() {
  static_assert(OnDevice<T>, "this path needs a field or ring with kernels (math::OnDevice)");
  return T::Field::TAG;
}

template <typename T>
std::vector<T> runEw(int op, const std::vector<T>& a, const std::vector<T>* b) {
  hip::DeviceVector<T> da(a), out(a.size());
  if (b) {
    hip::DeviceVector<T> db(*b);
    hip::check(scl_hip_ew(tag<T>(), op, out.data(), da.data(), db.data(), a.size(), nullptr));
    return out.toHost();
  }
  hip::check(scl_hip_ew(tag<T>(), op, out.data(), da.data(), nullptr, a.size(), nullptr));
  return out.toHost();
}

}  // namespace vec_detail

/// innerProd (vector.h:45-52): sum of x_k * y_k over [xb, xe)
template <typename T, typename IT0, typename IT1>
T innerProd(IT0 xb, IT0 xe, IT1 yb) {
  const std::size_t n = static_cast<std::size_t>(std::distance(xb, xe));
  if (hip::onHost<T>(n)) {  // v = 0; v += x_k * y_k in order (vector.h:45-52)
    T v{};
    for (; xb != xe; ++xb, ++yb) v += *xb * *yb;
    return v;
  }
  if constexpr (OnDevice<T>) {
    std::vector<T> x(xb, xe);
    std::vector<T> y(yb, yb + static_cast<std::ptrdiff_t>(x.size()));
    hip::DeviceVector<T> dx(x), dy(y);
    std::uint64_t limbs[hip::limbsOf<T>()];
    hip::check(scl_hip_dot(vec_detail::tag<T>(), limbs, dx.data(), dy.data(), x.size(), nullptr));
    return T::fromLimbs(limbs);
  } else {
    hip::detail::unreachable();
  }
}

template <typename ELEMENT>
class Vector final {
 public:
  using ValueType = ELEMENT;
  using SizeType = std::uint32_t;  // vector.h:73
  using iterator = typename std::vector<ELEMENT>::iterator;
  using const_iterator = typename std::vector<ELEMENT>::const_iterator;
  using reverse_iterator = typename std::vector<ELEMENT>::reverse_iterator;
  using const_reverse_iterator = typename std::vector<ELEMENT>::const_reverse_iterator;

  /// Vector::random (vector.h:507-519): ONE prg.next(n * byteSize) then FF::read per element.  Runs
  /// as the device kernel on the PRG's counter; the PRG is advanced by ceil(n*byteSize/16) blocks.
  static Vector random(std::size_t n, util::PRG& prg) {
    if (n == 0) return Vector{};
    if (hip::onHost<ELEMENT>(n)) {  // one prg.next(n * byteSize), then ELEMENT::read per element
      std::vector<unsigned char> buf(n * ELEMENT::byteSize());
      prg.next(buf.data(), buf.size());
      std::vector<ELEMENT> v;
      v.reserve(n);
      for (std::size_t i = 0; i < n; ++i) v.emplace_back(ELEMENT::read(buf.data() + i * ELEMENT::byteSize()));
      return Vector(std::move(v));
    }
    if constexpr (OnDevice<ELEMENT>) {
      hip::DeviceVector<ELEMENT> d(n);
      const auto seed = prg.Seed();
      hip::check(scl_hip_vector_random(vec_detail::tag<ELEMENT>(), d.data(), n, seed.data(), seed.size(), prg.counter(),
                                       nullptr));
      prg.advance((n * ELEMENT::byteSize() + 15) / 16);
      return Vector(d.toHost());
    } else {
      hip::detail::unreachable();
    }
  }

  /// Vector::range (vector.h:490-505): FF(int i) for i in [start, end)
  static Vector range(std::size_t start, std::size_t end) {
    if (start > end) hip::detail::raise(SCL_ERR_INVALID_RANGE);
    std::vector<ELEMENT> v;
    v.reserve(end - start);
    for (std::size_t i = start; i < end; ++i) v.emplace_back(ELEMENT{(int)i});
    return Vector(std::move(v));
  }
  static Vector range(std::size_t end) { return range(0, end); }

  Vector() {}
  explicit Vector(std::size_t n) : m_values(n) {}
  Vector(std::initializer_list<ELEMENT> values) : m_values(values) {}
  Vector(const std::vector<ELEMENT>& values) : m_values(values) {}
  Vector(std::vector<ELEMENT>&& values) : m_values(std::move(values)) {}
  template <typename IT>
  explicit Vector(IT first, IT last) : m_values(first, last) {}

  SizeType size() const { return static_cast<SizeType>(m_values.size()); }
  bool empty() const { return m_values.empty(); }
  ELEMENT& operator[](std::size_t idx) { return m_values[idx]; }
  ELEMENT operator[](std::size_t idx) const { return m_values[idx]; }

  Vector add(const Vector& o) const { return binary(SCL_OP_ADD, o); }
  Vector subtract(const Vector& o) const { return binary(SCL_OP_SUB, o); }
  Vector multiplyEntryWise(const Vector& o) const { return binary(SCL_OP_MUL, o); }
  Vector& addInPlace(const Vector& o) { return *this = add(o); }
  Vector& subtractInPlace(const Vector& o) { return *this = subtract(o); }
  Vector& multiplyEntryWiseInPlace(const Vector& o) { return *this = multiplyEntryWise(o); }

  ELEMENT dot(const Vector& o) const {
    ensureCompatible(o);
    return innerProd<ELEMENT>(begin(), end(), o.begin());
  }

  ELEMENT sum() const {
    if (empty()) return ELEMENT{};
    if (hip::onHost<ELEMENT>(size())) {
      ELEMENT v{};
      for (const auto& e : m_values) v += e;
      return v;
    }
    if constexpr (OnDevice<ELEMENT>) {
      hip::DeviceVector<ELEMENT> d(m_values);
      std::uint64_t limbs[hip::limbsOf<ELEMENT>()];
      hip::check(scl_hip_sum(vec_detail::tag<ELEMENT>(), limbs, d.data(), m_values.size(), nullptr));
      return ELEMENT::fromLimbs(limbs);
    } else {
      hip::detail::unreachable();
    }
  }

  Vector scalarMultiply(const ELEMENT& scalar) const {
    if (empty()) return Vector{};
    if (hip::onHost<ELEMENT>(size())) {
      std::vector<ELEMENT> r;
      r.reserve(size());
      for (const auto& e : m_values) r.emplace_back(scalar * e);
      return Vector(std::move(r));
    }
    if constexpr (OnDevice<ELEMENT>) {
      hip::DeviceVector<ELEMENT> d(m_values), out(m_values.size());
      std::uint64_t limbs[hip::limbsOf<ELEMENT>()];
      scalar.toLimbs(limbs);
      hip::check(scl_hip_scalar_mul(vec_detail::tag<ELEMENT>(), out.data(), d.data(), limbs, m_values.size(), nullptr));
      return Vector(out.toHost());
    } else {
      hip::detail::unreachable();
    }
  }
  Vector& scalarMultiplyInPlace(const ELEMENT& scalar) { return *this = scalarMultiply(scalar); }

  /// Vector::equals (vector.h:558-570): no early exit; different sizes are simply unequal
  bool equals(const Vector& o) const {
    if (size() != o.size()) return false;
    if (empty()) return true;
    if (hip::onHost<ELEMENT>(size())) {
      bool eq = true;
      for (std::size_t i = 0; i < m_values.size(); ++i) eq &= m_values[i] == o.m_values[i];
      return eq;
    }
    if constexpr (OnDevice<ELEMENT>) {
      hip::DeviceVector<ELEMENT> a(m_values), b(o.m_values);
      int eq = 0;
      hip::check(scl_hip_equals(vec_detail::tag<ELEMENT>(), &eq, a.data(), b.data(), m_values.size(), nullptr));
      return eq != 0;
    } else {
      hip::detail::unreachable();
    }
  }
  friend bool operator==(const Vector& l, const Vector& r) { return l.equals(r); }
  friend bool operator!=(const Vector& l, const Vector& r) { return !l.equals(r); }

  Matrix<ELEMENT> toRowMatrix() const { return Matrix<ELEMENT>::fromVector(1, size(), m_values); }
  Matrix<ELEMENT> toColumnMatrix() const { return Matrix<ELEMENT>::fromVector(size(), 1, m_values); }

  std::vector<ELEMENT>& toStlVector() { return m_values; }
  const std::vector<ELEMENT>& toStlVector() const { return m_values; }

  /// subVector (vector.h:358-375): throws std::logic_error("invalid range") when start > end
  Vector subVector(std::size_t start, std::size_t end) const {
    if (start > end) throw std::logic_error("invalid range");
    return Vector(begin() + static_cast<std::ptrdiff_t>(start), begin() + static_cast<std::ptrdiff_t>(end));
  }
  Vector subVector(std::size_t end) const { return subVector(0, end); }

  std::string toString() const {
    if (empty()) return "[ EMPTY VECTOR ]";
    std::stringstream ss;
    ss << "[";
    for (std::size_t i = 0; i + 1 < m_values.size(); ++i) ss << m_values[i] << ", ";
    ss << m_values.back() << "]";
    return ss.str();
  }
  friend std::ostream& operator<<(std::ostream& os, const Vector& v) { return os << v.toString(); }

  std::size_t byteSize() const { return size() * ELEMENT::byteSize(); }

  iterator begin() { return m_values.begin(); }
  const_iterator begin() const { return m_values.begin(); }
  const_iterator cbegin() const { return m_values.cbegin(); }
  iterator end() { return m_values.end(); }
  const_iterator end() const { return m_values.end(); }
  const_iterator cend() const { return m_values.cend(); }
  reverse_iterator rbegin() { return m_values.rbegin(); }
  const_reverse_iterator rbegin() const { return m_values.rbegin(); }
  const_reverse_iterator crbegin() const { return m_values.crbegin(); }
  reverse_iterator rend() { return m_values.rend(); }
  const_reverse_iterator rend() const { return m_values.rend(); }
  const_reverse_iterator crend() const { return m_values.crend(); }

 private:
  void ensureCompatible(const Vector& o) const {
    if (size() != o.size()) hip::detail::raise(SCL_ERR_SIZE_MISMATCH);  // "Vec sizes mismatch"
  }
  Vector binary(int op, const Vector& o) const {
    ensureCompatible(o);
    if (empty()) return Vector{};
    if (hip::onHost<ELEMENT>(size())) {
      std::vector<ELEMENT> r;
      r.reserve(size());
      for (std::size_t i = 0; i < m_values.size(); ++i)
        r.emplace_back(op == SCL_OP_ADD ? m_values[i] + o.m_values[i]
                                        : op == SCL_OP_SUB ? m_values[i] - o.m_values[i] : m_values[i] * o.m_values[i]);
      return Vector(std::move(r));
    }
    if constexpr (OnDevice<ELEMENT>) {
      return Vector(vec_detail::runEw<ELEMENT>(op, m_values, &o.m_values));
    } else {
      hip::detail::unreachable();
    }
  }

  std::vector<ELEMENT> m_values;
};

}  // namespace scl::math

#endif
