// include/scl_hip/util/prg.h -- scl::util::PRG over the device AES-128-CTR kernel (draws of more than
// hip::prgHostBytes() bytes) and the host AES of detail/aes_host.hpp (smaller ones).
//
// Mirrors include/scl/util/prg.h:64-173 / src/scl/util/prg.cc:88-146: key = seed zero-padded or
// truncated to 16 bytes, block i = AES(LE64(counter) || LE64(0x0123456789ABCDEF)), next(buf, n)
// consumes ceil(n/16) whole blocks and buffers nothing.  The stream is counter-addressable, which is
// what lets the batch kernels reproduce a sequential run (see counter()).
#ifndef SCL_HIP_UTIL_PRG_H
#define SCL_HIP_UTIL_PRG_H

#include <algorithm>
#include <array>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../detail/aes_host.hpp"
#include "../hip/device.h"

namespace scl::util {

class PRG {
 public:
  static constexpr std::size_t seedSize() { return 16; }

  static PRG create() { return PRG(nullptr, 0); }
  static PRG create(const unsigned char* seed, std::size_t seed_len) { return PRG(seed, seed_len); }
  static PRG create(const std::string& seed) {
    return PRG(reinterpret_cast<const unsigned char*>(seed.data()), seed.size());
  }

  void reset() { m_counter = 0; }

  void next(unsigned char* buffer, std::size_t n) {
    if (n == 0) return;
    const std::size_t nblocks = (n + 15) / 16;
    if (n <= hip::prgHostBytes()) {  // a few blocks: AES on the host (detail/aes_host.hpp), no device round trip
      const detail::Aes128Host& aes = m_aes;
      if (n % 16 == 0) {
        aes.prgBlocks(buffer, nblocks, m_counter);
      } else {
        std::vector<unsigned char> tmp(nblocks * 16);
        aes.prgBlocks(tmp.data(), nblocks, m_counter);
        std::copy(tmp.begin(), tmp.begin() + static_cast<std::ptrdiff_t>(n), buffer);
      }
      m_counter += nblocks;
      return;
    }
    hip::DeviceBuffer dev(nblocks * 16);
    hip::check(scl_hip_prg_blocks(static_cast<unsigned char*>(dev.get()), nblocks, m_seed.data(), m_seed.size(),
                                  m_counter, nullptr));
    std::vector<unsigned char> tmp(nblocks * 16);
    hip::check(scl_hip_memcpy_d2h(tmp.data(), dev.get(), tmp.size(), nullptr));
    std::copy(tmp.begin(), tmp.begin() + static_cast<std::ptrdiff_t>(n), buffer);
    m_counter += nblocks;
  }

  void next(std::vector<unsigned char>& buffer) { next(buffer.data(), buffer.size()); }

  template <std::size_t N>
  void next(std::array<unsigned char, N>& buffer) {
    next(buffer.data(), N);
  }

  void next(std::vector<unsigned char>& buffer, std::size_t n) {
    if (buffer.size() < n) throw std::invalid_argument("n exceeds buffer.size()");
    next(buffer.data(), n);
  }

  std::vector<unsigned char> next(std::size_t n) {
    std::vector<unsigned char> out(n);
    next(out.data(), n);
    return out;
  }

  std::array<unsigned char, 16> Seed() const { return m_seed; }

  /// blocks consumed so far; batch kernels take (seed, counter) instead of a PRG object
  std::uint64_t counter() const { return m_counter; }
  /// account for blocks consumed by a batch kernel on this PRG's behalf
  void advance(std::uint64_t blocks) { m_counter += blocks; }

 private:
  PRG(const unsigned char* seed, std::size_t seed_len) {
    m_seed.fill(0);
    if (seed != nullptr) std::copy(seed, seed + std::min<std::size_t>(seed_len, 16), m_seed.begin());
    m_aes = detail::Aes128Host(m_seed.data());
  }

  std::array<unsigned char, 16> m_seed;
  detail::Aes128Host m_aes{std::array<unsigned char, 16>{}.data()};  // round keys of the seed, for the host-side draws
  std::uint64_t m_counter = 0;
};

}  // namespace scl::util

#endif
