// tests/cxx/fake_rccl.cc -- a TEST-ONLY stand-in for RCCL: the ten entry points csrc/open_rccl.inc binds, for a world
// whose ranks are host threads of ONE process on ONE device.  It exists so that the multi-rank code of the open step
// (scl_hip_open_all_gather / scl_hip_open_partial_gather: permuted lambda, grouped per-row gathers, the two streams and
// their events, padding rows, a rank without parties; scl_hip_open_reduce_scatter: the 64-bit sums, the slices) executes with
// world > 1 on a one-GPU box; the library binds it when SCL_HIP_RCCL_LIBRARY names it.  Not a transport: an all-gather is a
// device-to-device copy per peer, a reduce-scatter (ncclSum over ncclUint64 only) one small kernel that adds the peers' slices.
// Compiled with hipcc for that kernel.
//
// Semantics kept from the real thing: every rank calls the same collectives in the same order; a collective is enqueued on
// the caller's stream and is asynchronous for the host apart from a rendezvous with the other ranks' calls (NCCL may block
// there too); it reads a peer's send buffer only after the work that rank had enqueued before ITS call, and a rank's
// stream passes the collective only once every peer has read that rank's send buffer -- so a rank that overwrites its send
// buffer right after the call races with nobody, exactly as with RCCL.  ncclGroupStart / ncclGroupEnd defer the calls in
// between to the outermost ncclGroupEnd.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr int MAX_RANKS = 64;

struct World {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int joined = 0, left = 0;
  // barrier
  int arrived = 0;
  unsigned long generation = 0;
  bool broken = false;
  // what the ranks publish for the collective in flight
  const void* send[MAX_RANKS] = {};
  hipEvent_t ready[MAX_RANKS] = {}, done[MAX_RANKS] = {};
  size_t count[MAX_RANKS] = {};

  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const unsigned long gen = generation;
    if (++arrived == n) {
      arrived = 0;
      ++generation;
      cv.notify_all();
      return !broken;
    }
    cv.wait(lk, [&] { return generation != gen || broken; });
    return !broken;
  }
  void fail() {
    std::lock_guard<std::mutex> lk(mu);
    broken = true;
    cv.notify_all();
  }
};

struct Comm {
  std::shared_ptr<World> world;
  int rank = 0;
  std::vector<hipEvent_t> events;  // destroyed with the communicator, after the device has drained
};

std::mutex g_reg_mu;
std::map<std::string, std::shared_ptr<World>> g_worlds;
unsigned long g_next_id = 1;

struct Pending {
  const void* send;
  void* recv;
  size_t count;  // all-gather: elements sent; reduce-scatter: elements received
  ncclDataType_t type;
  Comm* comm;
  hipStream_t stream;
  bool reduce_scatter = false;
};

struct Peers {
  const unsigned long long* send[MAX_RANKS];
};
// out[i] = sum over the ranks q of send_q[slice * count + i], wrapping 64-bit adds (ncclSum on ncclUint64)
__global__ void k_fake_reduce_slice(unsigned long long* out, Peers peers, int n, size_t slice, size_t count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long acc = 0;
    for (int q = 0; q < n; ++q) acc += peers.send[q][slice * count + i];
    out[i] = acc;
  }
}
thread_local int t_group_depth = 0;
thread_local std::vector<Pending> t_group;

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

#define FAKE_HIP(expr)                  \
  do {                                  \
    if ((expr) != hipSuccess) {         \
      W.fail();                         \
      return ncclUnhandledCudaError;    \
    }                                   \
  } while (0)

ncclResult_t all_gather_now(const Pending& p) {
  Comm* c = p.comm;
  World& W = *c->world;
  const int r = c->rank, n = W.n;
  const size_t bytes = p.count * type_bytes(p.type);
  if (bytes == 0 && p.count) return ncclInvalidArgument;
  hipEvent_t ready = nullptr, done = nullptr;
  FAKE_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
  FAKE_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  c->events.push_back(ready);
  c->events.push_back(done);
  // 1. my send buffer is ready at this point of my stream
  FAKE_HIP(hipEventRecord(ready, p.stream));
  W.send[r] = p.send;
  W.ready[r] = ready;
  W.count[r] = p.count;
  if (!W.barrier()) return ncclSystemError;
  // 2. pull every rank's contribution (my own included) into my receive buffer, on my stream
  for (int q = 0; q < n; ++q) {
    if (W.count[q] != p.count) {  // the ranks disagree about the collective: a bug in the caller
      W.fail();
      return ncclInvalidArgument;
    }
    if (q != r) FAKE_HIP(hipStreamWaitEvent(p.stream, W.ready[q], 0));
    if (p.reduce_scatter) continue;
    char* dst = static_cast<char*>(p.recv) + (size_t)q * bytes;
    if (dst != W.send[q] && bytes) FAKE_HIP(hipMemcpyAsync(dst, W.send[q], bytes, hipMemcpyDeviceToDevice, p.stream));
  }
  if (p.reduce_scatter && p.count) {  // my slice of the element-wise sum of everybody's send buffer
    Peers peers;
    for (int q = 0; q < n; ++q) peers.send[q] = static_cast<const unsigned long long*>(W.send[q]);
    const unsigned grid = (unsigned)((p.count + 255) / 256 < 4096 ? (p.count + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_fake_reduce_slice, dim3(grid), dim3(256), 0, p.stream, static_cast<unsigned long long*>(p.recv), peers, n,
                       (size_t)r, p.count);
    FAKE_HIP(hipGetLastError());
  }
  FAKE_HIP(hipEventRecord(done, p.stream));
  W.done[r] = done;
  if (!W.barrier()) return ncclSystemError;
  // 3. my stream passes the collective only after every peer has read my send buffer
  for (int q = 0; q < n; ++q)
    if (q != r) FAKE_HIP(hipStreamWaitEvent(p.stream, W.done[q], 0));
  if (!W.barrier()) return ncclSystemError;  // the slots are free for the next collective
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  std::memset(id, 0, sizeof *id);
  std::lock_guard<std::mutex> lk(g_reg_mu);
  const unsigned long k = g_next_id++;
  std::memcpy(id->internal, "fake-rccl", 9);
  std::memcpy(id->internal + 16, &k, sizeof k);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  const std::string key(id.internal, sizeof id.internal);
  std::shared_ptr<World> w;
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto& slot = g_worlds[key];
    if (!slot) {
      slot = std::make_shared<World>();
      slot->n = nranks;
    }
    w = slot;
  }
  if (w->n != nranks) return ncclInvalidArgument;
  {
    std::unique_lock<std::mutex> lk(w->mu);
    ++w->joined;
    w->cv.notify_all();
    w->cv.wait(lk, [&] { return w->joined >= w->n; });  // as ncclCommInitRank: returns once every rank has joined
  }
  auto* c = new Comm;
  c->world = w;
  c->rank = rank;
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (!comm) return ncclSuccess;
  auto* c = reinterpret_cast<Comm*>(comm);
  (void)hipDeviceSynchronize();
  for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    std::lock_guard<std::mutex> lk2(c->world->mu);
    if (++c->world->left == c->world->n)
      for (auto it = g_worlds.begin(); it != g_worlds.end(); ++it)
        if (it->second == c->world) {
          g_worlds.erase(it);
          break;
        }
  }
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
  if (!comm || !count) return ncclInvalidArgument;
  *count = reinterpret_cast<const Comm*>(comm)->world->n;
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
  if (!comm || !rank) return ncclInvalidArgument;
  *rank = reinterpret_cast<const Comm*>(comm)->rank;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
  if (!comm || (sendcount && (!sendbuff || !recvbuff))) return ncclInvalidArgument;
  const Pending p{sendbuff, recvbuff, sendcount, datatype, reinterpret_cast<Comm*>(comm), stream, false};
  if (t_group_depth > 0) {
    t_group.push_back(p);
    return ncclSuccess;
  }
  return all_gather_now(p);
}

ncclResult_t ncclReduceScatter(const void* sendbuff, void* recvbuff, size_t recvcount, ncclDataType_t datatype, ncclRedOp_t op,
                               ncclComm_t comm, hipStream_t stream) {
  if (!comm || (recvcount && (!sendbuff || !recvbuff))) return ncclInvalidArgument;
  if (datatype != ncclUint64 || op != ncclSum) return ncclInvalidArgument;  // all the open step asks for
  Pending p{sendbuff, recvbuff, recvcount, datatype, reinterpret_cast<Comm*>(comm), stream};
  p.reduce_scatter = true;
  if (t_group_depth > 0) {
    t_group.push_back(p);
    return ncclSuccess;
  }
  return all_gather_now(p);
}

ncclResult_t ncclGroupStart() {
  ++t_group_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (t_group_depth <= 0) return ncclInvalidUsage;
  if (--t_group_depth > 0) return ncclSuccess;
  std::vector<Pending> ops;
  ops.swap(t_group);
  ncclResult_t rc = ncclSuccess;
  for (const Pending& p : ops) {
    const ncclResult_t r = all_gather_now(p);
    if (r != ncclSuccess && rc == ncclSuccess) rc = r;
  }
  return rc;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake rccl: HIP call failed";
    case ncclSystemError: return "fake rccl: another rank failed";
    case ncclInvalidArgument: return "fake rccl: invalid argument (or the ranks disagree about a collective)";
    case ncclInvalidUsage: return "fake rccl: invalid usage";
    default: return "fake rccl: error";
  }
}

}  // extern "C"
