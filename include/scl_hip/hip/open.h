// include/scl_hip/hip/open.h -- the MPC "open" step for a batch: all parties' shares of N secrets come together on every
// rank over RCCL / xGMI and every rank reconstructs (C ABI: scl_hip_comm_*, scl_hip_open_*).
//
// Reference: Network::send to every party + Network::recv from every party (include/scl/net/network.h:148-152,178-185;
// the pattern of test/scl/protocol/beaver.h:43-55), then shamirRecoverP per secret (include/scl/ss/shamir.h:81-104).
// Here the n parties are dealt to the ranks of a communicator in contiguous blocks of ceil(n / world); one process per GPU.
#ifndef SCL_HIP_HIP_OPEN_H
#define SCL_HIP_HIP_OPEN_H

#include <algorithm>
#include <array>
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

#include "../math/lagrange.h"
#include "../math/vector.h"
#include "device.h"

namespace scl::hip {

/// An RCCL communicator with the stream, events and gather buffers of the open step (move-only).
class Communicator {
 public:
  using UniqueId = std::array<unsigned char, 128>;

  /// on ONE rank; the caller hands the bytes to the others (MPI_Bcast, a file, a socket)
  static UniqueId uniqueId() {
    UniqueId id{};
    check(scl_hip_comm_unique_id(id.data()));
    return id;
  }
  /// ncclCommInitRank on every rank; the current HIP device is the rank's
  Communicator(int world, int rank, const UniqueId& id) { check(scl_hip_comm_init_rank(&m_comm, world, rank, id.data())); }
  /// wrap an ncclComm_t the caller made (it stays the caller's)
  explicit Communicator(void* nccl_comm) { check(scl_hip_comm_adopt(&m_comm, nccl_comm)); }
  Communicator(const Communicator&) = delete;
  Communicator& operator=(const Communicator&) = delete;
  Communicator(Communicator&& o) noexcept : m_comm(o.m_comm) { o.m_comm = nullptr; }
  ~Communicator() {
    if (m_comm) (void)scl_hip_comm_destroy(m_comm);
  }

  int world() const {
    int w = 0;
    check(scl_hip_comm_info(m_comm, &w, nullptr));
    return w;
  }
  int rank() const {
    int r = 0;
    check(scl_hip_comm_info(m_comm, nullptr, &r));
    return r;
  }
  /// parties per rank when n parties are dealt in contiguous blocks
  std::size_t partiesPerRank(std::size_t n) const { return (n + (std::size_t)world() - 1) / (std::size_t)world(); }
  /// [first, first + count): the parties of this rank
  std::pair<std::size_t, std::size_t> partySlab(std::size_t n) const {
    const std::size_t per = partiesPerRank(n), first = std::min((std::size_t)rank() * per, n);
    return {first, std::min(per, n - first)};
  }
  void* get() const { return m_comm; }

 private:
  void* m_comm = nullptr;
};

namespace open_detail {
template <typename T>
std::vector<std::uint64_t> limbsOfVector(const math::Vector<T>& v, std::size_t first, std::size_t count) {
  constexpr std::size_t L = limbsOf<T>();
  std::vector<std::uint64_t> out(count * L + 1);
  for (std::size_t i = 0; i < count; ++i) v[first + i].toLimbs(out.data() + i * L);
  return out;
}
}  // namespace open_detail

/// Open N secrets shared among n parties: `local` holds this rank's ceil(n / world) party rows (rows past its party count
/// are padding), `lambda` the n Lagrange coefficients in party order.  Every rank returns every secret.
template <typename T>
DeviceVector<T> open(Communicator& comm, const ShareMatrix<T>& local, std::size_t n, const math::Vector<T>& lambda,
                     std::size_t chunk = 0) {
  if (lambda.size() < n || local.parties() != comm.partiesPerRank(n)) detail::raise(SCL_ERR_SIZE_MISMATCH);
  const auto lam = open_detail::limbsOfVector(lambda, 0, n);
  DeviceVector<T> out(local.secrets());
  check(scl_hip_open_all_gather(comm.get(), T::Field::TAG, out.data(), local.data(), local.stride(), n, lam.data(),
                                local.secrets(), chunk, nullptr));
  check(scl_hip_stream_sync(nullptr));
  return out;
}
/// nodes 1..n, x = 0 (shamirRecoverP(shares), shamir.h:99-104)
template <typename T>
DeviceVector<T> open(Communicator& comm, const ShareMatrix<T>& local, std::size_t n, std::size_t chunk = 0) {
  return open(comm, local, n, math::computeLagrangeBasis(math::Vector<T>::range(1, n + 1), T{}), chunk);
}

/// The same secrets from 1 / ceil(n / world) of the traffic: each rank reduces its OWN parties' rows (`mine`, no padding) to
/// one partial sum per secret, the ranks all-gather and add the partials.  `lambda` in party order, all n of them.
template <typename T>
DeviceVector<T> openByPartialSums(Communicator& comm, const ShareMatrix<T>& mine, std::size_t n,
                                  const math::Vector<T>& lambda, std::size_t chunk = 0) {
  const auto [first, count] = comm.partySlab(n);
  if (lambda.size() < n || mine.parties() != count) detail::raise(SCL_ERR_SIZE_MISMATCH);
  const auto lam = open_detail::limbsOfVector(lambda, first, count);
  DeviceVector<T> out(mine.secrets());
  check(scl_hip_open_partial_gather(comm.get(), T::Field::TAG, out.data(), mine.data(), mine.stride(), count, lam.data(),
                                    mine.secrets(), chunk, nullptr));
  check(scl_hip_stream_sync(nullptr));
  return out;
}

/// Mersenne61 only, at most 8 ranks: the partial sums meet in an ncclReduceScatter of plain 64-bit sums (canonical partials
/// cannot wrap), each rank folds its slice mod p and an all-gather hands every rank every secret -- the same bits as open().
/// all_ranks = false leaves each secret on the one rank that owns its slice (scl_hip.h, scl_hip_open_reduce_scatter).
template <typename T>
DeviceVector<T> openByReduceScatter(Communicator& comm, const ShareMatrix<T>& mine, std::size_t n, const math::Vector<T>& lambda,
                                    std::size_t chunk = 0, bool all_ranks = true) {
  const auto [first, count] = comm.partySlab(n);
  if (lambda.size() < n || mine.parties() != count) detail::raise(SCL_ERR_SIZE_MISMATCH);
  const auto lam = open_detail::limbsOfVector(lambda, first, count);
  DeviceVector<T> out(mine.secrets());
  check(scl_hip_open_reduce_scatter(comm.get(), T::Field::TAG, out.data(), mine.data(), mine.stride(), count, lam.data(),
                                    mine.secrets(), chunk, all_ranks ? 1 : 0, nullptr));
  check(scl_hip_stream_sync(nullptr));
  return out;
}

}  // namespace scl::hip

#endif
