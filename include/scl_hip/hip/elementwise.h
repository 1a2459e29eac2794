// include/scl_hip/hip/elementwise.h -- Vector<T>'s element-wise members for operands that STAY in HBM.
//
// math::Vector (math/vector.h) is the reference's host container: its members upload, run the kernel and download.  Code that
// keeps its data on the GPU uses hip::DeviceVector with the functions below -- the same kernels (scl_hip_ew / scl_hip_ew_status),
// no transfers, every call asynchronous on the stream it is given.
//
// What the reference throws for a zero divisor (std::logic_error("0 not invertible modulo prime"): small_ff.h:61-70,
// ff_ops_gmp.h:250-260, reached through operator/ and inverse(), ff.h:203-246; std::invalid_argument("value not invertible
// modulo 2^K") for an even ring element, z2k_ops.h:81-83) is raised LAZILY here: divide() and inverse() take a ZeroFlag, a
// 32-bit word in device memory the kernels OR into, and ZeroFlag::check() -- called when the caller synchronises anyway, or
// by the destructor-free end of a sequence -- throws exactly that exception if any call since the last clear() met a
// non-invertible operand (every other element is computed all the same; the offending slot holds 0).  A sequence of small
// batches then pays no stream round trip per call, and the calls can be captured into a hipGraph.  The forms without a flag
// keep the reference's immediacy: they synchronise and throw before they return.
#ifndef SCL_HIP_HIP_ELEMENTWISE_H
#define SCL_HIP_HIP_ELEMENTWISE_H

#include <cstddef>
#include <stdexcept>

#include "device.h"

namespace scl::hip {

/// The device word behind the lazy "not invertible" report of divide() / inverse().
class ZeroFlag {
 public:
  ZeroFlag() : m_word(sizeof(unsigned)) { clear(); }
  /// forget what was raised (asynchronous on `stream`)
  void clear(void* stream = nullptr) { check(scl_hip_memset(m_word.get(), 0, sizeof(unsigned), stream)); }
  unsigned* device() const { return static_cast<unsigned*>(m_word.get()); }
  /// waits for `stream`; true if a call since the last clear() met a non-invertible operand
  bool raised(void* stream = nullptr) const {
    unsigned h = 0;
    check(scl_hip_memcpy_d2h(&h, m_word.get(), sizeof h, stream));
    check(scl_hip_stream_sync(stream));
    return h != 0;
  }
  /// the reference's exception for element type T, late: throws if raised(); the flag is cleared either way
  template <typename T>
  void check_for(void* stream = nullptr) {
    if (!raised(stream)) return;
    clear(stream);
    detail::raise(isRing<T>() ? SCL_ERR_NOT_INVERTIBLE_2K : SCL_ERR_ZERO_INVERSE);
  }

 private:
  template <typename T>
  static constexpr bool isRing() {
    return T::Field::TAG > 0x100;
  }
  DeviceBuffer m_word;
};

namespace ew_detail {
template <typename T>
void run(int op, DeviceVector<T>& out, const DeviceVector<T>& a, const DeviceVector<T>* b, unsigned* status, void* stream) {
  if (out.size() != a.size() || (b && b->size() != a.size())) detail::raise(SCL_ERR_SIZE_MISMATCH);  // Vector::ensureCompatible
  if (status)
    check(scl_hip_ew_status(DeviceVector<T>::FIELD_TAG, op, out.data(), a.data(), b ? b->data() : nullptr, a.size(), status, stream));
  else
    check(scl_hip_ew(DeviceVector<T>::FIELD_TAG, op, out.data(), a.data(), b ? b->data() : nullptr, a.size(), stream));
}
}  // namespace ew_detail

/// Vector::add / subtract / multiplyEntryWise (vector.h:199-245) over device-resident operands; out may alias a or b
template <typename T>
void add(DeviceVector<T>& out, const DeviceVector<T>& a, const DeviceVector<T>& b, void* stream = nullptr) {
  ew_detail::run(SCL_OP_ADD, out, a, &b, nullptr, stream);
}
template <typename T>
void subtract(DeviceVector<T>& out, const DeviceVector<T>& a, const DeviceVector<T>& b, void* stream = nullptr) {
  ew_detail::run(SCL_OP_SUB, out, a, &b, nullptr, stream);
}
template <typename T>
void multiplyEntryWise(DeviceVector<T>& out, const DeviceVector<T>& a, const DeviceVector<T>& b, void* stream = nullptr) {
  ew_detail::run(SCL_OP_MUL, out, a, &b, nullptr, stream);
}
template <typename T>
void negate(DeviceVector<T>& out, const DeviceVector<T>& a, void* stream = nullptr) {
  ew_detail::run<T>(SCL_OP_NEG, out, a, nullptr, nullptr, stream);
}

/// out[i] = a[i] / b[i] (FF::operator/, ff.h:203-205), asynchronous: a zero b[i] raises `flag` (see ZeroFlag::check_for)
template <typename T>
void divide(DeviceVector<T>& out, const DeviceVector<T>& a, const DeviceVector<T>& b, ZeroFlag& flag, void* stream = nullptr) {
  ew_detail::run(SCL_OP_DIV, out, a, &b, flag.device(), stream);
}
/// out[i] = a[i]^-1 (FF::inverse, ff.h:243-246), asynchronous
template <typename T>
void inverse(DeviceVector<T>& out, const DeviceVector<T>& a, ZeroFlag& flag, void* stream = nullptr) {
  ew_detail::run<T>(SCL_OP_INV, out, a, nullptr, flag.device(), stream);
}
/// the same with the reference's immediacy: synchronises `stream`, throws before returning
template <typename T>
void divide(DeviceVector<T>& out, const DeviceVector<T>& a, const DeviceVector<T>& b, void* stream = nullptr) {
  ew_detail::run(SCL_OP_DIV, out, a, &b, nullptr, stream);
}
template <typename T>
void inverse(DeviceVector<T>& out, const DeviceVector<T>& a, void* stream = nullptr) {
  ew_detail::run<T>(SCL_OP_INV, out, a, nullptr, nullptr, stream);
}

}  // namespace scl::hip

#endif
