// include/scl_hip/serialization/serializer.h -- scl::seri::Serializer<T> (include/scl/serialization/serializer.h:35-210 and the
// specialisations next to the types: ff.h:355-391, vector.h:595-629, matrix.h:910-963, array.h:424-455): how the values of
// the path leave and reach a party.  The images are
//   trivially copyable T     its bytes
//   FF<FIELD>                byteSize() bytes: the canonical value, little-endian (Montgomery fields: big-endian, out of Montgomery
//                            form -- whatever FF::write does, as in the reference)
//   std::vector<T>, Vector   u32 count, then the elements' images
//   Matrix                   u32 rows, u32 cols, then the row-major elements as a vector (u32 count first)
//   Array<T, N>              the N elements' images, no count
// Host code over the elements' own read / write.  The same images are produced and consumed on the device, straight from and
// into SoA rows, by scl_hip_wire_pack / _unpack (kernels k_wire_pack / k_wire_unpack); tests/cxx/test_scl_api.cc compares the
// two byte for byte, and tests/golden pins both to the bytes the reference's own Serializer wrote.
#ifndef SCL_HIP_SERIALIZATION_SERIALIZER_H
#define SCL_HIP_SERIALIZATION_SERIALIZER_H

#include <concepts>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#include "../math/array.h"
#include "../math/ff.h"
#include "../math/matrix.h"
#include "../math/vector.h"

namespace scl::seri {

template <typename T, typename = void>
struct Serializer;

/// a type that defines its own image: byteSize(), write(buf), read(buf) -- FF, Z2k, Array
template <typename T>
concept HasWireImage = requires(const T& v, unsigned char* out, const unsigned char* in) {
  { T::byteSize() } -> std::convertible_to<std::size_t>;
  v.write(out);
  { T::read(in) } -> std::convertible_to<T>;
};

/// anything memcpy may move (and that does not define an image of its own): its object representation
template <typename T>
struct Serializer<T, std::enable_if_t<std::is_trivially_copyable<T>::value && !HasWireImage<T>>> {
  static constexpr std::size_t sizeOf(const T&) { return sizeof(T); }
  static std::size_t write(const T& obj, unsigned char* buf) {
    std::memcpy(buf, &obj, sizeof(T));
    return sizeof(T);
  }
  static std::size_t read(T& obj, const unsigned char* buf) {
    std::memcpy(&obj, buf, sizeof(T));
    return sizeof(T);
  }
};

/// the count in front of every sequence (serializer.h: StlVecSizeType)
using StlVecSizeType = std::uint32_t;

template <typename T>
struct Serializer<std::vector<T>> {
  static std::size_t sizeOf(const std::vector<T>& vec) {
    std::size_t total = sizeof(StlVecSizeType);
    for (const T& v : vec) total += Serializer<T>::sizeOf(v);
    return total;
  }
  static std::size_t write(const std::vector<T>& vec, unsigned char* buf) {
    std::size_t at = Serializer<StlVecSizeType>::write(static_cast<StlVecSizeType>(vec.size()), buf);
    for (const T& v : vec) at += Serializer<T>::write(v, buf + at);
    return at;
  }
  static std::size_t read(std::vector<T>& vec, const unsigned char* buf) {
    StlVecSizeType count = 0;
    std::size_t at = Serializer<StlVecSizeType>::read(count, buf);
    vec.clear();
    vec.reserve(count);
    for (StlVecSizeType i = 0; i < count; ++i) {
      T v;
      at += Serializer<T>::read(v, buf + at);
      vec.push_back(std::move(v));
    }
    return at;
  }
};

/// FF (ff.h:355-391), Z2k, Array (array.h:424-455): the image is what the type's own write puts out.  (The reference's FF is
/// not trivially copyable -- a user-provided destructor -- and so never takes the memcpy form; the mirror's types are kept
/// off it by HasWireImage.)
template <typename T>
struct Serializer<T, std::enable_if_t<HasWireImage<T>>> {
  static constexpr std::size_t sizeOf(const T&) { return T::byteSize(); }
  static std::size_t write(const T& e, unsigned char* buf) {
    e.write(buf);
    return T::byteSize();
  }
  static std::size_t read(T& e, const unsigned char* buf) {
    e = T::read(buf);
    return T::byteSize();
  }
};

template <typename ELEMENT>
struct Serializer<math::Vector<ELEMENT>, void> {
  using Inner = Serializer<std::vector<ELEMENT>>;
  static std::size_t sizeOf(const math::Vector<ELEMENT>& v) { return Inner::sizeOf(v.toStlVector()); }
  static std::size_t write(const math::Vector<ELEMENT>& v, unsigned char* buf) { return Inner::write(v.toStlVector(), buf); }
  static std::size_t read(math::Vector<ELEMENT>& v, const unsigned char* buf) {
    std::vector<ELEMENT> elements;
    const std::size_t used = Inner::read(elements, buf);
    v = math::Vector<ELEMENT>(std::move(elements));
    return used;
  }
};

template <typename ELEMENT>
struct Serializer<math::Matrix<ELEMENT>, void> {
  using DimType = std::uint32_t;
  using Inner = Serializer<std::vector<ELEMENT>>;
  static std::size_t sizeOf(const math::Matrix<ELEMENT>& m) { return 2 * sizeof(DimType) + Inner::sizeOf(m.values()); }
  static std::size_t write(const math::Matrix<ELEMENT>& m, unsigned char* buf) {
    std::size_t at = Serializer<DimType>::write(static_cast<DimType>(m.rows()), buf);
    at += Serializer<DimType>::write(static_cast<DimType>(m.cols()), buf + at);
    return at + Inner::write(m.values(), buf + at);
  }
  static std::size_t read(math::Matrix<ELEMENT>& m, const unsigned char* buf) {
    DimType rows = 0, cols = 0;
    std::size_t at = Serializer<DimType>::read(rows, buf);
    at += Serializer<DimType>::read(cols, buf + at);
    std::vector<ELEMENT> elements;
    at += Inner::read(elements, buf + at);
    m = math::Matrix<ELEMENT>::fromVector(rows, cols, elements);
    return at;
  }
};

}  // namespace scl::seri

#endif  // SCL_HIP_SERIALIZATION_SERIALIZER_H
