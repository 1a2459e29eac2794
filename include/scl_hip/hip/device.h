// include/scl_hip/hip/device.h -- device-resident containers for the batch API.
//
// The reference keeps every value in host std::vector and works one secret per call
// (include/scl/ss/shamir.h:52-68).  At 10^8 secrets the data has to live in HBM, so the batch API
// is expressed over two RAII buffers: DeviceVector<T> (N elements) and ShareMatrix<T> (SoA
// [party][secret]).  Both talk to libscl_hip.so through the C ABI only.
#ifndef SCL_HIP_HIP_DEVICE_H
#define SCL_HIP_HIP_DEVICE_H

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <utility>
#include <vector>

#include "../detail/call.h"

namespace scl::hip {

using detail::check;

/// Raw device allocation (move-only).
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  explicit DeviceBuffer(std::size_t bytes) : m_bytes(bytes) {
    if (bytes) check(scl_hip_malloc(&m_ptr, bytes));
  }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  DeviceBuffer(DeviceBuffer&& o) noexcept : m_ptr(o.m_ptr), m_bytes(o.m_bytes) {
    o.m_ptr = nullptr;
    o.m_bytes = 0;
  }
  DeviceBuffer& operator=(DeviceBuffer&& o) noexcept {
    if (this != &o) {
      release();
      m_ptr = o.m_ptr;
      m_bytes = o.m_bytes;
      o.m_ptr = nullptr;
      o.m_bytes = 0;
    }
    return *this;
  }
  ~DeviceBuffer() { release(); }

  void* get() const { return m_ptr; }
  std::size_t bytes() const { return m_bytes; }

 private:
  void release() {
    if (m_ptr) (void)scl_hip_free(m_ptr);
    m_ptr = nullptr;
  }
  void* m_ptr = nullptr;
  std::size_t m_bytes = 0;
};

/// 64-bit limbs of T's element image on the C ABI: byteSize() / 8 for the fields, T::kLimbs where the type says
/// otherwise (Z2k<K>: one word for K <= 64, two above, whatever its byteSize())
template <typename T>
constexpr std::size_t limbsOf() {
  if constexpr (requires { T::kLimbs; }) return T::kLimbs;
  else return T::byteSize() / 8;
}

// ---- where a HOST-resident operand is worked on -----------------------------------------------------------------
// The reference's containers (math::Vector, math::Matrix) and its per-secret scl::ss calls keep their values in host
// memory.  Sending such an operand through a kernel costs two device allocations, two PCIe copies, a launch and a
// synchronisation -- tens to hundreds of microseconds -- where the reference spends half a microsecond on a (10,3) sharing
// (BASELINE.md section 3).  So a host-resident operand below a work threshold is computed on the host, by FF<FIELD>'s own
// operators, i.e. by detail/field.hpp: the arithmetic source the kernels are compiled from (per-element arithmetic is
// host work in the reference's plug-in boundary too, SURVEY.md section 8 b-i).  At and above the threshold, and for everything
// that already lives in HBM (hip::DeviceVector, hip::ShareMatrix: the batch API), the kernels run and there is no host path.
// The threshold counts 64-bit limb products: elements x limbs^2 (a 256-bit multiplication costs ~16 Mersenne61 ones).  Its
// default, 2^20, is read off tests/cxx/bench_threshold.cc on the GPU box (profiles/r3_host_threshold.txt): a round trip costs
// ~55 us plus the per-element limb conversions on both sides, so an element-wise member of a HOST vector is faster on the host
// at every size measured (Mersenne61: 20 us against 91 us at 16 384 elements, 1.1 ms against 3.6 ms at 2^20), and dot only
// pays off on the device for the wide fields (secp256k1 from ~4 096 elements, Mersenne127 from ~65 536).  Code that wants the
// GPU for its vectors keeps them there (hip::DeviceVector); a host container is the reference's convenience type.
inline std::atomic<std::size_t>& hostThresholdRef() {
  static std::atomic<std::size_t> v{std::size_t(1) << 20};
  return v;
}
/// 0 sends every host-resident operand of a field with kernels to the GPU (what tests of the kernels behind
/// math::Vector / math::Matrix set)
inline void setHostThreshold(std::size_t limb_products) { hostThresholdRef().store(limb_products); }
inline std::size_t hostThreshold() { return hostThresholdRef().load(); }
/// true: compute on the host.  Always for an element type without kernels (a user-defined field).
template <typename T>
bool onHost(std::size_t elems) {
  if constexpr (!requires { T::Field::TAG; }) return true;
  else return elems * limbsOf<T>() * limbsOf<T>() < hostThreshold();
}
/// PRG draws up to this many bytes are made by detail/aes_host.hpp instead of k_prg_blocks (64 KiB; none with the threshold at 0)
inline std::size_t prgHostBytes() { return hostThreshold() ? std::size_t(64) << 10 : 0; }
namespace detail {
[[noreturn]] inline void unreachable() { std::abort(); }
}  // namespace detail

/// N elements in HBM.  T is an scl::math::FF<FIELD> or an scl::math::Z2k<K>.
template <typename T>
class DeviceVector {
 public:
  static constexpr int FIELD_TAG = T::Field::TAG;
  static constexpr std::size_t LIMBS = limbsOf<T>();

  DeviceVector() = default;
  explicit DeviceVector(std::size_t n) : m_buf(n * LIMBS * 8), m_size(n) {}

  /// upload (element image = the C ABI's little-endian limbs)
  explicit DeviceVector(const std::vector<T>& host) : DeviceVector(host.size()) {
    if (!host.empty()) {
      std::vector<std::uint64_t> limbs(host.size() * LIMBS);
      for (std::size_t i = 0; i < host.size(); ++i)
        host[i].toLimbs(limbs.data() + i * LIMBS);
      check(scl_hip_memcpy_h2d(m_buf.get(), limbs.data(), limbs.size() * 8, nullptr));
      check(scl_hip_stream_sync(nullptr));
    }
  }

  std::vector<T> toHost() const {
    std::vector<std::uint64_t> limbs(m_size * LIMBS);
    if (m_size) check(scl_hip_memcpy_d2h(limbs.data(), m_buf.get(), limbs.size() * 8, nullptr));
    std::vector<T> out;
    out.reserve(m_size);
    for (std::size_t i = 0; i < m_size; ++i)
      out.emplace_back(T::fromLimbs(limbs.data() + i * LIMBS));
    return out;
  }

  std::size_t size() const { return m_size; }
  std::uint64_t* data() { return static_cast<std::uint64_t*>(m_buf.get()); }
  const std::uint64_t* data() const { return static_cast<const std::uint64_t*>(m_buf.get()); }

 private:
  DeviceBuffer m_buf;
  std::size_t m_size = 0;
};

/// n share vectors of N secrets each, SoA [party][secret], row stride = N.
template <typename T>
class ShareMatrix {
 public:
  static constexpr std::size_t LIMBS = limbsOf<T>();

  ShareMatrix() = default;
  ShareMatrix(std::size_t parties, std::size_t secrets)
      : m_buf(parties * secrets * LIMBS * 8), m_parties(parties), m_secrets(secrets) {}

  std::size_t parties() const { return m_parties; }
  std::size_t secrets() const { return m_secrets; }
  std::size_t stride() const { return m_secrets; }
  std::uint64_t* data() { return static_cast<std::uint64_t*>(m_buf.get()); }
  const std::uint64_t* data() const { return static_cast<const std::uint64_t*>(m_buf.get()); }
  /// party i's share vector (device pointer)
  const std::uint64_t* row(std::size_t i) const { return data() + i * m_secrets * LIMBS; }

  /// the shares of secret s as the reference returns them: one host Vector of n elements
  std::vector<T> sharesOf(std::size_t s) const {
    std::vector<T> out;
    out.reserve(m_parties);
    std::uint64_t limbs[LIMBS];
    for (std::size_t i = 0; i < m_parties; ++i) {
      check(scl_hip_memcpy_d2h(limbs, row(i) + s * LIMBS, sizeof limbs, nullptr));
      out.emplace_back(T::fromLimbs(limbs));
    }
    return out;
  }

 private:
  DeviceBuffer m_buf;
  std::size_t m_parties = 0, m_secrets = 0;
};

}  // namespace scl::hip

#endif
