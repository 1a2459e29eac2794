// tests/cxx/gf7_field.h -- a user-defined field written the REFERENCE's way: a traits struct with nothing but
// { ValueType, NAME, BYTE_SIZE, BIT_SIZE } and the eleven scl::math::ff:: specialisations of
// include/scl_hip/math/fields/ff_ops.h (reference interface: include/scl/math/fields/ff_ops.h:35-118; the reference's
// own example of such a field is test/scl/gf7.h:26-31 + test/scl/gf7.cc:26-103, the integers modulo 7).  This file
// is this repo's own GF(7), written from that interface: no TAG, no Impl -- so it has no kernels and every
// Vector / Matrix / Polynomial / scl::ss member takes its generic host path for it.
#ifndef TESTS_CXX_GF7_FIELD_H
#define TESTS_CXX_GF7_FIELD_H

#include <cstddef>
#include <stdexcept>
#include <string>

#include "scl_hip/math/fields/ff_ops.h"

namespace usr {

struct Gf7 {
  using ValueType = unsigned char;  // the residue 0..6
  constexpr static const char* NAME = "GF(7)";
  constexpr static const std::size_t BYTE_SIZE = 1;
  constexpr static const std::size_t BIT_SIZE = 8;
};

}  // namespace usr

namespace scl::math::ff {

template <>
inline void convertTo<usr::Gf7>(unsigned char& out, int value) {
  const int r = value % 7;  // C++ remainder: sign of the dividend
  out = static_cast<unsigned char>(r < 0 ? r + 7 : r);
}

template <>
inline void convertTo<usr::Gf7>(unsigned char& out, const std::string& src) {
  // the digits of a hexadecimal number, folded modulo 7 as they come (16 = 2 mod 7)
  if (src.size() % 2) throw std::invalid_argument("odd-length hex string");
  unsigned acc = 0;
  for (char ch : src) {
    unsigned d;
    if (ch >= '0' && ch <= '9') d = static_cast<unsigned>(ch - '0');
    else if (ch >= 'a' && ch <= 'f') d = static_cast<unsigned>(ch - 'a') + 10;
    else if (ch >= 'A' && ch <= 'F') d = static_cast<unsigned>(ch - 'A') + 10;
    else throw std::invalid_argument("encountered invalid hex character");
    acc = (acc * 16 + d) % 7;
  }
  out = static_cast<unsigned char>(acc);
}

template <>
inline void add<usr::Gf7>(unsigned char& out, const unsigned char& op) {
  const unsigned s = static_cast<unsigned>(out) + op;
  out = static_cast<unsigned char>(s >= 7 ? s - 7 : s);
}

template <>
inline void subtract<usr::Gf7>(unsigned char& out, const unsigned char& op) {
  out = static_cast<unsigned char>((static_cast<unsigned>(out) + 7 - op) % 7);
}

template <>
inline void multiply<usr::Gf7>(unsigned char& out, const unsigned char& op) {
  out = static_cast<unsigned char>((static_cast<unsigned>(out) * op) % 7);
}

template <>
inline void negate<usr::Gf7>(unsigned char& out) {
  out = static_cast<unsigned char>(out ? 7 - out : 0);
}

template <>
inline void invert<usr::Gf7>(unsigned char& out) {
  if (out == 0) throw std::logic_error("0 not invertible modulo prime");
  const unsigned a = out, a2 = a * a % 7, a4 = a2 * a2 % 7;
  out = static_cast<unsigned char>(a4 * a % 7);  // Fermat: a^(7-2)
}

template <>
inline bool equal<usr::Gf7>(const unsigned char& in1, const unsigned char& in2) {
  return in1 == in2;
}

template <>
inline void toBytes<usr::Gf7>(unsigned char* dest, const unsigned char& src) {
  dest[0] = src;
}

template <>
inline void fromBytes<usr::Gf7>(unsigned char& dest, const unsigned char* src) {
  dest = static_cast<unsigned char>(src[0] % 7);
}

template <>
inline std::string toString<usr::Gf7>(const unsigned char& in) {
  return std::to_string(static_cast<int>(in));
}

}  // namespace scl::math::ff

#endif
