// include/scl_hip/math/ff.h -- scl::math::FF<FIELD> / Fp<BITS>, the scalar value type.
//
// Same surface as the reference's include/scl/math/ff.h:36-346 and fp.h:34-64 (names, operators,
// exception text).  A single element is host data in the reference and stays host data here; the
// per-element arithmetic comes from the one field implementation shared with the device kernels
// (detail/field.hpp), so a scalar computed on the host and a lane computed on the GPU agree by
// construction.  Everything batch-shaped (Vector, Matrix, ss::*) goes through the C ABI.
//
// Field plug-in boundary: exactly the reference's -- a traits struct { ValueType, NAME, BYTE_SIZE, BIT_SIZE } plus the
// eleven scl::math::ff:: specialisations of fields/ff_ops.h (include/scl/math/fields/ff_ops.h:35-118, worked example
// test/scl/gf7.cc:26-103); FF<FIELD> below calls nothing else.  The five fields that have kernels are defined here
// the same way: their traits additionally name TAG (scl_field) and Impl (detail/field.hpp), and their eleven
// specialisations are stamped out by SCL_HIP_BUILTIN_FIELD_OPS over that Impl.
#ifndef SCL_HIP_MATH_FF_H
#define SCL_HIP_MATH_FF_H

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <string>

#include "../detail/call.h"
#include "../detail/field.hpp"
#include "../util/prg.h"
#include "fields/ff_ops.h"

namespace scl::math {

namespace ff {

struct Mersenne61 {  // include/scl/math/fields/mersenne61.h:29-49
  using ValueType = std::uint64_t;
  using Impl = sclhip::M61;
  constexpr static const char* NAME = "Mersenne61";
  constexpr static std::size_t BYTE_SIZE = 8;
  constexpr static std::size_t BIT_SIZE = 61;
  constexpr static int TAG = SCL_M61;
};

struct Mersenne127 {  // include/scl/math/fields/mersenne127.h:29-49
  using ValueType = __uint128_t;
  using Impl = sclhip::M127;
  constexpr static const char* NAME = "Mersenne127";
  constexpr static std::size_t BYTE_SIZE = 16;
  constexpr static std::size_t BIT_SIZE = 127;
  constexpr static int TAG = SCL_M127;
};

struct Secp256k1Scalar {  // include/scl/math/fields/secp256k1_scalar.h: the order of the secp256k1 group
  using ValueType = sclhip::U256;  // 4 x 64-bit limbs, Montgomery form like the reference's std::array<mp_limb_t, 4>
  using Impl = sclhip::Secp256k1Scalar;
  constexpr static const char* NAME = "secp256k1_order";
  constexpr static std::size_t BYTE_SIZE = 32;
  constexpr static std::size_t BIT_SIZE = 256;
  constexpr static int TAG = SCL_SECP256K1_SCALAR;
};

struct Secp256k1Field {  // include/scl/math/fields/secp256k1_field.h: the prime the secp256k1 curve is defined over
  using ValueType = sclhip::U256;
  using Impl = sclhip::Secp256k1Field;
  constexpr static const char* NAME = "secp256k1_field";
  constexpr static std::size_t BYTE_SIZE = 32;
  constexpr static std::size_t BIT_SIZE = 256;
  constexpr static int TAG = SCL_SECP256K1_FIELD;
};

struct Mont128 {  // 128-bit prime, Montgomery form: the N = 2 instance of the reference's monty*<N> family (ff_ops_gmp.h:44-392;
                  // the reference itself instantiates N = 4 only -- pinned by tests/golden/golden_mont128.json), run-time modulus
  using ValueType = __uint128_t;
  using Impl = sclhip::Mont128;
  constexpr static const char* NAME = "Mont128";
  constexpr static std::size_t BYTE_SIZE = 16;
  constexpr static std::size_t BIT_SIZE = 128;
  constexpr static int TAG = SCL_MONT128;
};

struct GF2_128 {  // plug-in: GF(2^128) (not in the reference)
  using ValueType = __uint128_t;
  using Impl = sclhip::Gf128;
  constexpr static const char* NAME = "GF(2^128)";
  constexpr static std::size_t BYTE_SIZE = 16;
  constexpr static std::size_t BIT_SIZE = 128;
  constexpr static int TAG = SCL_GF2_128;
};

namespace detail {

template <typename FIELD>
inline typename FIELD::Impl::Ctx context() {
  if constexpr (FIELD::TAG == SCL_MONT128) {
    std::uint64_t p[2];
    scl::hip::detail::check(scl_hip_mont128_get_prime(p));
    static thread_local __uint128_t cached_p = 0;
    static thread_local sclhip::Mont128::Ctx cached;
    const __uint128_t pp = ((__uint128_t)p[1] << 64) | p[0];
    if (pp != cached_p) {
      cached = sclhip::Mont128::make_ctx(pp);
      cached_p = pp;
    }
    return cached;
  } else {
    return typename FIELD::Impl::Ctx{};
  }
}

// big-endian hex, no "0x"; bits shifted past the value type are dropped (include/scl/util/str.h:49-76)
template <typename V>
inline V parseHex(const std::string& s) {
  if (s.size() % 2) throw std::invalid_argument("odd-length hex string");
  V t = 0;
  for (char c : s) {
    unsigned d;
    if (c >= '0' && c <= '9') d = c - '0';
    else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
    else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
    else throw std::invalid_argument("encountered invalid hex character");
    t = (V)(t << 4) + d;
  }
  return t;
}

// montyFromString (include/scl/math/fields/ff_ops_gmp.h:370-398): an odd-length string gets a leading "0",
// more than 64 digits is an error, and the digits are cut into 16-character limbs FROM THE LEFT (first
// chunk = top limb (n-1)/16; a short last chunk becomes limb 0 as it stands).  Kept as the reference does it.
// `limbs` = N of the instance: 4 for the secp256k1 fields; 2 for Mont128, where the reference's template would write limbs
// 2 and 3 of a two-limb value for 33..64 digits (its bound is 64 whatever N is) -- undefined there, refused here.
inline sclhip::U256 parseHexLimbs(const std::string& str, std::size_t limbs = 4) {
  sclhip::U256 out = sclhip::Secp256k1Scalar::zero();
  if (str.empty()) return out;
  if (str.size() > 16 * limbs) throw std::invalid_argument("hex string too large to parse");
  const std::string s = str.size() % 2 ? "0" + str : str;
  int c = (int)((s.size() - 1) / 16);
  for (std::size_t i = 0; i < s.size() && c >= 0; i += 16)
    out.w[c--] = parseHex<std::uint64_t>(s.substr(i, 16).size() % 2 ? "0" + s.substr(i, 16) : s.substr(i, 16));
  return out;
}

inline std::string hex64(std::uint64_t v) {
  std::stringstream ss;
  ss << std::hex << v;
  return ss.str();
}


// ---- the eleven operations of a field that has an Impl (detail/field.hpp) ----------------------------------
template <typename FIELD>
inline void fromInt(typename FIELD::ValueType& out, int value) {  // negative: p - |v| (mersenne61.cc:37-40)
  using Impl = typename FIELD::Impl;
  const auto c = context<FIELD>();
  const std::uint64_t mag = value < 0 ? (std::uint64_t)(-(std::int64_t)value) : (std::uint64_t)value;
  out = Impl::from_u64(c, mag);
  if (value < 0) out = Impl::neg(c, out);
}

template <typename FIELD>
inline void fromString(typename FIELD::ValueType& out, const std::string& hexstr) {  // hex, reduced mod p (mersenne61.cc:42-46)
  using Impl = typename FIELD::Impl;
  using V = typename FIELD::ValueType;
  const auto c = context<FIELD>();
  if constexpr (FIELD::Impl::LIMBS == 4) {
    const V limbs = parseHexLimbs(hexstr);
    out = hexstr.empty() ? limbs : Impl::to_mont(c, limbs);
  } else if constexpr (FIELD::TAG == SCL_MONT128) {
    // the Montgomery family's montyFromString at N = 2 (pinned by the reference's template compiled at two limbs,
    // oracle/ref_harness.cc): limbs cut from the left, an odd length padded, the empty string 0
    const sclhip::U256 limbs = parseHexLimbs(hexstr, 2);
    const V raw = ((V)limbs.w[1] << 64) | limbs.w[0];
    out = hexstr.empty() ? raw : Impl::to_mont(c, raw);
  } else {
    out = Impl::from_le_word(c, parseHex<V>(hexstr));
  }
}

/// Mersenne61: std::hex of the word; Mersenne127: the top word (if non-zero) then the low word
/// without zero padding -- the reference's formatting (src/scl/util/str.cc:23-39), kept as is.
template <typename FIELD>
inline std::string toStringOf(const typename FIELD::ValueType& value) {
  using Impl = typename FIELD::Impl;
  using V = typename FIELD::ValueType;
  const auto c = context<FIELD>();
  if constexpr (FIELD::TAG == SCL_M61) {
    return hex64(value);
  } else if constexpr (FIELD::Impl::LIMBS == 4) {
    // montyToString: the value out of Montgomery form, hex without leading zeros
    const V v = Impl::from_mont(c, value);
    std::string out;
    for (int i = 3; i >= 0; --i) {
      if (out.empty()) {
        if (v.w[i] || i == 0) out = hex64(v.w[i]);
      } else {
        const std::string part = hex64(v.w[i]);
        out += std::string(16 - part.size(), '0') + part;
      }
    }
    return out;
  } else {
    V v = value;
    if constexpr (FIELD::TAG == SCL_MONT128) v = Impl::from_mont(c, v);
    const auto top = (std::uint64_t)(v >> 64), bot = (std::uint64_t)v;
    if (v == 0) return "0";
    std::string s;
    if (top) s = hex64(top);
    if (FIELD::TAG == SCL_M127 || !top) return s + hex64(bot);
    std::string b = hex64(bot);
    return s + std::string(16 - b.size(), '0') + b;
  }
}

/// FF::write (ff.h:300-302).  Mersenne fields: the canonical little-endian word.  Mont128 follows
/// the reference's Montgomery family (ff_ops_gmp.h:298-314): out of Montgomery form, big-endian.
template <typename FIELD>
inline void toBytesOf(unsigned char* dest, const typename FIELD::ValueType& value) {
  using Impl = typename FIELD::Impl;
  using V = typename FIELD::ValueType;
  if constexpr (FIELD::Impl::LIMBS == 4) {
    const V v = Impl::to_be_image(context<FIELD>(), value);  // montyToBytes: value, big-endian
    std::memcpy(dest, &v, sizeof v);
  } else if constexpr (FIELD::TAG == SCL_MONT128) {
    const V v = sclhip::bswap128(Impl::from_mont(context<FIELD>(), value));
    std::memcpy(dest, &v, sizeof v);
  } else {
    std::memcpy(dest, &value, sizeof value);
  }
}

/// FF::read (ff.h:63-67): byteSize() bytes, reduced into the field
template <typename FIELD>
inline void fromBytesOf(typename FIELD::ValueType& dest, const unsigned char* src) {
  typename FIELD::ValueType raw;
  std::memcpy(&raw, src, sizeof raw);
  dest = FIELD::Impl::from_le_word(context<FIELD>(), raw);
}

template <typename FIELD>
inline void invertOf(typename FIELD::ValueType& out) {
  if (FIELD::Impl::is_zero(out)) scl::hip::detail::raise(SCL_ERR_ZERO_INVERSE);
  out = FIELD::Impl::inv(context<FIELD>(), out);
}

}  // namespace detail

// the eleven specialisations (fields/ff_ops.h) of a built-in field, over its Impl
#define SCL_HIP_BUILTIN_FIELD_OPS(F)                                                                                      \
  template <> inline void convertTo<F>(F::ValueType & out, int value) { detail::fromInt<F>(out, value); }                 \
  template <> inline void convertTo<F>(F::ValueType & out, const std::string& src) { detail::fromString<F>(out, src); }   \
  template <> inline void add<F>(F::ValueType & out, const F::ValueType& op) { out = F::Impl::add(detail::context<F>(), out, op); } \
  template <> inline void subtract<F>(F::ValueType & out, const F::ValueType& op) { out = F::Impl::sub(detail::context<F>(), out, op); } \
  template <> inline void multiply<F>(F::ValueType & out, const F::ValueType& op) { out = F::Impl::mul(detail::context<F>(), out, op); } \
  template <> inline void negate<F>(F::ValueType & out) { out = F::Impl::neg(detail::context<F>(), out); }                \
  template <> inline void invert<F>(F::ValueType & out) { detail::invertOf<F>(out); }                                     \
  template <> inline bool equal<F>(const F::ValueType& in1, const F::ValueType& in2) { return F::Impl::eq(in1, in2); }    \
  template <> inline void toBytes<F>(unsigned char* dest, const F::ValueType& src) { detail::toBytesOf<F>(dest, src); }   \
  template <> inline void fromBytes<F>(F::ValueType & dest, const unsigned char* src) { detail::fromBytesOf<F>(dest, src); } \
  template <> inline std::string toString<F>(const F::ValueType& in) { return detail::toStringOf<F>(in); }

SCL_HIP_BUILTIN_FIELD_OPS(Mersenne61)
SCL_HIP_BUILTIN_FIELD_OPS(Mersenne127)
SCL_HIP_BUILTIN_FIELD_OPS(Secp256k1Scalar)
SCL_HIP_BUILTIN_FIELD_OPS(Secp256k1Field)
SCL_HIP_BUILTIN_FIELD_OPS(Mont128)
SCL_HIP_BUILTIN_FIELD_OPS(GF2_128)
#undef SCL_HIP_BUILTIN_FIELD_OPS

}  // namespace ff

/// true for element types whose field (or ring) has kernels behind the C ABI: the batch-shaped members of Vector,
/// Matrix and scl::ss go to the GPU for them and take the generic per-element host path otherwise
template <typename T>
concept OnDevice = requires { T::Field::TAG; };

template <typename FIELD>
class FF final {
  using V = typename FIELD::ValueType;

 public:
  using Field = FIELD;

  constexpr static std::size_t byteSize() { return FIELD::BYTE_SIZE; }
  constexpr static std::size_t bitSize() { return FIELD::BIT_SIZE; }
  constexpr static const char* name() { return FIELD::NAME; }

  /// FF::read (ff.h:63-67): byteSize() bytes, reduced into the field
  static FF read(const unsigned char* src) {
    FF e;
    ff::fromBytes<FIELD>(e.m_value, src);
    return e;
  }

  /// FF::random (ff.h:72-76): one prg.next(byteSize()) -- a whole AES block per element
  static FF random(util::PRG& prg) {
    unsigned char buffer[FIELD::BYTE_SIZE];
    prg.next(buffer, FIELD::BYTE_SIZE);
    return read(buffer);
  }

  /// hex string, reduced mod p (mersenne61.cc:42-46)
  static FF fromString(const std::string& hexstr) {
    FF e;
    ff::convertTo<FIELD>(e.m_value, hexstr);
    return e;
  }

  static FF zero() { return FF(); }
  static FF one() { return FF(1); }

  /// FF(int): negative values wrap to p - |v| (mersenne61.cc:37-40)
  explicit FF(int value) { ff::convertTo<FIELD>(m_value, value); }
  FF() : FF(0) {}

  FF& operator+=(const FF& o) { ff::add<FIELD>(m_value, o.m_value); return *this; }
  FF& operator-=(const FF& o) { ff::subtract<FIELD>(m_value, o.m_value); return *this; }
  FF& operator*=(const FF& o) { ff::multiply<FIELD>(m_value, o.m_value); return *this; }
  FF& operator/=(const FF& o) { return *this *= o.inverse(); }  // ff.h:203-205
  friend FF operator+(FF a, const FF& b) { return a += b; }
  friend FF operator-(FF a, const FF& b) { return a -= b; }
  friend FF operator*(FF a, const FF& b) { return a *= b; }
  friend FF operator/(FF a, const FF& b) { return a /= b; }
  FF& operator++() { return *this += one(); }
  friend FF operator++(FF& e, int) { FF t(e); ++e; return t; }
  FF& operator--() { return *this -= one(); }
  friend FF operator--(FF& e, int) { FF t(e); --e; return t; }

  FF& negate() { ff::negate<FIELD>(m_value); return *this; }
  FF negated() const { FF r(*this); return r.negate(); }
  friend FF operator-(const FF& e) { return e.negated(); }

  /// throws std::logic_error("0 not invertible modulo prime") on zero (test_ff.cc:168-171)
  FF& invert() { ff::invert<FIELD>(m_value); return *this; }
  FF inverse() const { FF r(*this); return r.invert(); }

  bool equal(const FF& o) const { return ff::equal<FIELD>(m_value, o.m_value); }
  friend bool operator==(const FF& a, const FF& b) { return a.equal(b); }
  friend bool operator!=(const FF& a, const FF& b) { return !a.equal(b); }

  std::string toString() const { return ff::toString<FIELD>(m_value); }
  friend std::ostream& operator<<(std::ostream& os, const FF& e) { return os << e.toString(); }

  /// FF::write (ff.h:300-302)
  void write(unsigned char* dest) const { ff::toBytes<FIELD>(dest, m_value); }

  /// the limb image the C ABI and the kernels use (internal representation, little-endian limbs); fields with kernels only
  void toLimbs(std::uint64_t* dest) const
    requires requires { FIELD::TAG; }
  {
    std::memcpy(dest, &m_value, sizeof m_value);
  }
  static FF fromLimbs(const std::uint64_t* src)
    requires requires { FIELD::TAG; }
  {
    FF e;
    std::memcpy(&e.m_value, src, sizeof e.m_value);
    return e;
  }

  V value() const { return m_value; }
  V& value() { return m_value; }

 private:
  V m_value;
};

/// exp(base, e): square-and-multiply over the bits of e (ff.h:329-346)
template <typename F>
FF<F> exp(const FF<F>& base, std::size_t e) {
  FF<F> r = FF<F>::one();
  if (e == 0) return r;
  for (int i = 63 - __builtin_clzll((unsigned long long)e); i >= 0; --i) {
    r *= r;
    if ((e >> i) & 1) r *= base;
  }
  return r;
}

namespace ff_detail {
template <std::size_t BITS, bool SMALL = (BITS <= 61)>
struct Select {
  using Field = ff::Mersenne61;
};
template <std::size_t BITS>
struct Select<BITS, false> {
  using Field = ff::Mersenne127;
};
}  // namespace ff_detail

/// Fp<BITS>: 1..61 bits -> Mersenne61, 62..127 -> Mersenne127 (fp.h:34-64)
template <std::size_t BITS>
  requires(BITS > 0 && BITS < 128)
using Fp = FF<typename ff_detail::Select<BITS>::Field>;

}  // namespace scl::math

#endif
