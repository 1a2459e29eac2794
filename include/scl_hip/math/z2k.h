// include/scl_hip/math/z2k.h -- scl::math::Z2k<BITS> (include/scl/math/z2k.h:39-320, z2k/z2k_ops.h:32-150):
// the ring of integers modulo 2^BITS on one 64-bit word (BITS <= 64) or one 128-bit word.  Scalars run on the
// host with the arithmetic of include/scl_hip/detail/field.hpp (the source the kernels compile); Vector<Z2k<K>>,
// Matrix<Z2k<K>> and ss::additiveShare over DeviceVector<Z2k<K>> run on the GPU under the tag SCL_Z2K(BITS).
#ifndef SCL_HIP_MATH_Z2K_H
#define SCL_HIP_MATH_Z2K_H

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "../detail/field.hpp"
#include "../util/prg.h"
#include "ff.h"
#include "scl_hip.h"

namespace scl::math {

template <std::size_t BITS>
class Z2k final {
  static_assert(BITS >= 1 && BITS <= 128, "Z2k: 1 <= BITS <= 128");

 public:
  using ValueType = std::conditional_t<(BITS <= 64), std::uint64_t, __uint128_t>;
  /// what the batch entry points of libscl_hip.so know this ring as
  struct Field {
    static constexpr int TAG = SCL_Z2K(BITS);
  };
  static constexpr std::size_t kLimbs = BITS <= 64 ? 1 : 2;

  constexpr static std::size_t byteSize() { return (BITS - 1) / 8 + 1; }
  constexpr static std::size_t bitSize() { return BITS; }
  constexpr static const char* name() { return "Z2k"; }

  /// Z2k::read (z2k.h:71-75, z2k_ops.h:107-112); only the first byteSize() bytes can matter after the mask
  static Z2k read(const unsigned char* src) {
    ValueType v = 0;
    std::memcpy(&v, src, byteSize());
    return Z2k(v & mask());
  }
  /// Z2k::random (z2k.h:81-85): one prg.next(byteSize()) -- a whole AES block
  static Z2k random(util::PRG& prg) {
    unsigned char buffer[byteSize()];
    prg.next(buffer, byteSize());
    return read(buffer);
  }
  static Z2k fromString(const std::string& str) { return Z2k(ff::detail::parseHex<ValueType>(str) & mask()); }
  static Z2k zero() { return Z2k(); }
  static Z2k one() { return Z2k(1); }

  explicit constexpr Z2k(const ValueType& value) : m_value(value) {}
  constexpr Z2k() : m_value(0) {}

  Z2k& operator+=(const Z2k& o) { m_value += o.m_value; return *this; }
  Z2k& operator-=(const Z2k& o) { m_value -= o.m_value; return *this; }
  Z2k& operator*=(const Z2k& o) { m_value *= o.m_value; return *this; }
  Z2k& operator/=(const Z2k& o) { m_value *= o.inverse().m_value; return *this; }
  friend Z2k operator+(const Z2k& a, const Z2k& b) { Z2k t(a); return t += b; }
  friend Z2k operator-(const Z2k& a, const Z2k& b) { Z2k t(a); return t -= b; }
  friend Z2k operator*(const Z2k& a, const Z2k& b) { Z2k t(a); return t *= b; }
  friend Z2k operator/(const Z2k& a, const Z2k& b) { Z2k t(a); return t /= b; }
  Z2k& operator++() { return *this += one(); }
  friend Z2k operator++(Z2k& e, int) { Z2k t(e); ++e; return t; }
  Z2k& operator--() { return *this -= one(); }
  friend Z2k operator--(Z2k& e, int) { Z2k t(e); --e; return t; }
  Z2k& negate() { m_value = (ValueType)0 - m_value; return *this; }
  Z2k negated() const { Z2k c(m_value); return c.negate(); }
  friend Z2k operator-(const Z2k& e) { return e.negated(); }

  /// z2k_ops.h:80-93: only odd values are invertible
  Z2k& invert() {
    if (!lsb()) throw std::invalid_argument("value not invertible modulo 2^K");
    m_value = Impl::inv(Impl::make_ctx((int)BITS), m_value);
    return *this;
  }
  Z2k inverse() const { Z2k c(m_value); return c.invert(); }
  unsigned lsb() const { return (unsigned)(m_value & 1); }

  bool equal(const Z2k& o) const { return ((m_value ^ o.m_value) & mask()) == 0; }
  friend bool operator==(const Z2k& a, const Z2k& b) { return a.equal(b); }
  friend bool operator!=(const Z2k& a, const Z2k& b) { return !(a == b); }

  /// hex of the masked value; the 128-bit form prints the two words back to back like the reference does
  /// (src/scl/util/str.cc:23-39)
  std::string toString() const {
    const ValueType w = m_value & mask();
    std::stringstream ss;
    ss << std::hex;
    if constexpr (BITS <= 64) {
      ss << w;
    } else {
      if (w == 0) return "0";
      const auto top = static_cast<std::uint64_t>(w >> 64);
      if (top > 0) ss << top;
      ss << static_cast<std::uint64_t>(w);
    }
    return ss.str();
  }
  friend std::ostream& operator<<(std::ostream& os, const Z2k& e) { return os << e.toString(); }

  /// Z2k::write (z2k_ops.h:117-124): the masked value, byteSize() bytes
  void write(unsigned char* dest) const {
    const ValueType w = m_value & mask();
    std::memcpy(dest, &w, byteSize());
  }

  /// element image on the C ABI: the masked word as little-endian 64-bit limbs
  void toLimbs(std::uint64_t* dest) const {
    const ValueType w = m_value & mask();
    std::memcpy(dest, &w, sizeof w);
  }
  static Z2k fromLimbs(const std::uint64_t* src) {
    ValueType w;
    std::memcpy(&w, src, sizeof w);
    return Z2k(w);
  }

 private:
  using Impl = std::conditional_t<(BITS <= 64), sclhip::Z2k64, sclhip::Z2k128>;
  static constexpr ValueType mask() {
    return BITS >= 8 * sizeof(ValueType) ? ~(ValueType)0 : (ValueType)(((ValueType)1 << BITS) - 1);
  }
  ValueType m_value;
};

}  // namespace scl::math

#endif
