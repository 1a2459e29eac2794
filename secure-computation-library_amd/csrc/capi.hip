// csrc/capi.hip -- the extern "C" boundary of libscl_hip.so (include/scl_hip.h).
//
// Host-side work done here is table building only: AES key schedule, Lagrange
// basis / alpha / Vandermonde tables (O(n^2) field ops, hoisted out of the
// per-secret path exactly once per batch) and the final fold of per-block
// reduction partials.  There is NO CPU fallback for any batch operation: if no
// HIP device is usable the calls fail with SCL_ERR_NO_DEVICE / SCL_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "tu_config.hpp"
#if SCL_TU_FIELDS != 0xff
#include "capi_names.inc"  // this unit's names for the field-taking entry points
#endif
#include "../../include/scl_hip.h"
#include "../../include/scl_hip/detail/field.hpp"
#include "kernels.hpp"
#include "share_mfma.hpp"

// the general matrix-core product's kernels live in gemm_unit.hip, compiled without -amdgpu-mfma-vgpr-form (see there)
namespace sclhip {
hipError_t gemm_launch_planes_a(u64x2* planes, const u64* A, size_t lda, size_t M, size_t K, size_t ktiles, hipStream_t st);
hipError_t gemm_launch_planes_b(u64x2* planes, const u64* B, size_t ldb, size_t K, size_t N, size_t ktiles, hipStream_t st);
hipError_t gemm_launch_main(u64* C, size_t ldc, const u64x2* Ap, const u64x2* Bp, size_t M, size_t N, size_t ktiles, size_t kslice,
                            size_t cslice, size_t splits, hipStream_t st);
}  // namespace sclhip

using namespace sclhip;

// ---- state shared by the translation units of the library ------------------------------------------------
// The library is this file compiled once per field family (tu_config.hpp, Makefile); the unit built with SCL_TU_COMMON
// defines the state below, the others see it as extern.  Everything mutable is per host thread except the Mont128 default
// (behind its mutex).
struct Knob {
  long v;
  long load() const { return v; }
  Knob& operator=(long x) {
    v = x;
    return *this;
  }
};
struct Scratch {
  int device = -1;
  void* dev = nullptr;
  size_t bytes = 0;
};
struct TempArena {
  int device = -1;
  void* dev = nullptr;
  size_t bytes = 0;
  hipEvent_t done = nullptr;
  bool pending = false;
};
namespace sclhip_state {
#if SCL_TU_COMMON
#define SCL_STATE(decl, ...) decl __VA_ARGS__
#else
#define SCL_STATE(decl, ...) extern decl
#endif
SCL_STATE(thread_local std::string g_err);
// Experiment knobs (scl_hip_set_tuning).  Per host thread, like the Mont128 modulus below: the library keeps no mutable
// state that two host threads share, so concurrent callers with different settings cannot race (scl_hip.h, Conventions).
SCL_STATE(thread_local Knob g_max_blocks, {0});
SCL_STATE(thread_local Knob g_nontemporal, {1});
SCL_STATE(thread_local Knob g_force_scalar, {0});
SCL_STATE(thread_local Knob g_force_table, {0});
SCL_STATE(thread_local Knob g_mfma_tpb, {0});
// 4-row-tile shapes: 2 = two pipelined waves per SIMD on 16x16x64 tiles (33..64 coefficient rows; else as 1), 1 = one
// pipelined wave per SIMD, 0 = burst kernel ("mfma_pipe")
SCL_STATE(thread_local Knob g_mfma_pipe, {2});
SCL_STATE(thread_local Knob g_mfma_areg, {1});  // register-resident V fragments for the 4-row-tile shapes ("mfma_areg")
SCL_STATE(thread_local Knob g_mfma, {0});       // 0 auto, 1 always (where applicable), -1 never
// GF(2^128) sharing at the default nodes: 1 = eight nodes per Horner loop (k_share_gf_tiles), 0 = one node at a time
// (k_share_gf_nodes) ("gf_tiles")
SCL_STATE(thread_local Knob g_gf_tiles, {1});
SCL_STATE(thread_local Knob g_open_gather_always, {0});  // the open step on ONE rank: 1 = still through RCCL's all-gather
SCL_STATE(thread_local Knob g_prg_t3, {1});  // PRG-driven sharing at t = 3 over the Mersenne fields: 1 = threshold compiled in
SCL_STATE(thread_local Knob g_prg_two_pass, {0});  // PRG-driven sharing: 0 auto, 1 always two passes, -1 always fused
// Headline streaming kernels (k_recover_fixed, k_share_small): workgroup size and resident waves per CU
// (kernels.hpp, "Launch geometry"); "stream_block" 64 | 256, "stream_waves" 0 = no cap
SCL_STATE(thread_local Knob g_stream_block, {64});
// -1 = by element size: 10 for one-word elements, 12 for wider ones (profiles/r2_probe_cap_rec.txt, r2_probe_c3_waves.txt)
SCL_STATE(thread_local Knob g_stream_waves, {-1});
// the same cap for the Mersenne61 small-node share kernel ("share_waves"; 0 = the 256-thread kernel without a cap)
SCL_STATE(thread_local Knob g_share_waves, {9});
SCL_STATE(thread_local Knob g_share_waves128, {12});  // .. and for the 16-byte fields' small-node share kernel ("share_waves128")
SCL_STATE(thread_local Knob g_aes_blocks, {0});
// element-wise inverse / divide: 0 auto = Montgomery's simultaneous inversion with the chain length chosen by the batch, N > 0 =
// that chain length (8, 16, 32, 64 or 128; Mersenne61: any N = its register kernel), -1 = one Fermat chain per element (k_ew, the kernels
// of rounds 1-4); GF(2^128) multiply: -1 = the register-only product ("inv_batch")
SCL_STATE(thread_local Knob g_inv_batch, {0});
SCL_STATE(thread_local Knob g_inv_two_level, {0});
SCL_STATE(thread_local Knob g_gemm_slab_mib, {0});  // digit planes per factor and launch of the general matrix-core product, MiB (0 = 1024) ("gemm_slab_mib")
SCL_STATE(thread_local Knob g_matmul_lds_min, {0});  // columns from which k_matmul (left factor in LDS, a thread per column) is taken; thin kernel likewise (0 = default) ("matmul_lds_min")
SCL_STATE(thread_local Knob g_transpose_tile, {0});  // secrets per LDS tile of the 16-byte layout bridge (0 = 512 within 40 KiB) ("transpose_tile")
// Mont128 modulus: a process-wide default, latched per host thread at its first use (mont_ctx below)
SCL_STATE(thread_local Mont128::Ctx g_mont, = {0, 0, 0, 0});  // p == 0: this thread has not latched a modulus yet
SCL_STATE(std::mutex g_mont_default_mu);
SCL_STATE(Mont128::Ctx g_mont_default, = {0, 0, 0, 0});
// how often the default has been set, and where this thread stands: the generation it latched (or set its own modulus) at
SCL_STATE(unsigned long g_mont_default_gen, = 0);
SCL_STATE(thread_local unsigned long g_mont_gen, = 0);
SCL_STATE(thread_local bool g_mont_own, = false);
// per-thread device scratch (only used by calls that synchronise before returning)
SCL_STATE(thread_local Scratch g_scratch);
// the "a zero was inverted" flag of scl_hip_ew in pinned host memory the device writes straight into: the call then needs no
// memset launch and no copy back, only the stream synchronisation it owes its caller anyway.  host == nullptr: not allocated
// (yet, or the allocation failed once: then the flag lives in the device scratch as before)
struct HostFlag {
  unsigned* host = nullptr;
  unsigned* dev = nullptr;
  bool failed = false;
};
SCL_STATE(thread_local HostFlag g_hflag);
// arena 0: tables, queues and products of one call; arena 1: the coefficient rows of a two-pass PRG sharing (whose second
// pass may take arena 0 itself)
SCL_STATE(thread_local TempArena g_temps[2]);
#undef SCL_STATE
}  // namespace sclhip_state
using namespace sclhip_state;

namespace {

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(e_ == hipErrorNoDevice ? SCL_ERR_NO_DEVICE : SCL_ERR_HIP,                    \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                          \
  } while (0)

#define SCL_TRY(expr)        \
  do {                       \
    int s_ = (expr);         \
    if (s_ != SCL_OK) return s_; \
  } while (0)

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

// dynamic LDS bytes that cap the residency of a kernel with `static_lds` bytes of its own at `waves` waves of
// `block` threads per CU.  A CU has 160 KiB of LDS and hands it out in granules; the granule of gfx950 is not documented
// (512 B on earlier parts, 1280 B = 1/128 of the array is the other candidate), so the request is chosen to hold for BOTH:
// `groups` allocations fit, `groups + 1` do not, whichever granule rounds it up.  No such size exists for every count
// (24 and more groups per CU): then no cap is applied.
size_t residency_pad(long waves, int block, size_t static_lds) {
  if (waves <= 0) return 0;
  const size_t groups = (size_t)(waves * 64 / block);
  if (groups < 1) return 0;
  constexpr size_t LDS = 160 * 1024;
  auto holds = [&](size_t total, size_t granule) {
    const size_t alloc = (total + granule - 1) / granule * granule;
    return groups * alloc <= LDS && (groups + 1) * alloc > LDS;
  };
  for (size_t total = LDS / groups / 128 * 128; total > static_lds && total >= 1024; total -= 128)
    if (holds(total, 512) && holds(total, 1280)) return total <= 64 * 1024 ? total - static_lds : 0;
  return 0;
}

unsigned grid_for(size_t work_items) {
  size_t blocks = (work_items + BLOCK - 1) / BLOCK;
  // Measured on MI355X (profiles/r1_tune_m61.txt): one pack per thread with no grid-stride wrap is the
  // fastest geometry for every streaming kernel here; "max_blocks" caps it for experiments.
  long cap = g_max_blocks.load();
  if (cap <= 0) cap = 0x7fffffff;
  if (blocks > (size_t)cap) blocks = (size_t)cap;
  if (blocks == 0) blocks = 1;
  return (unsigned)blocks;
}

unsigned grid_for_block(size_t work_items, int block) {
  size_t blocks = (work_items + block - 1) / block;
  long cap = g_max_blocks.load();
  if (cap <= 0) cap = 0x7fffffff;
  if (blocks > (size_t)cap) blocks = (size_t)cap;
  if (blocks == 0) blocks = 1;
  return (unsigned)blocks;
}

// PRG kernels carry a 32 KiB replicated AES table per block: a fixed grid of resident blocks that
// grid-strides amortises filling it.
unsigned grid_aes(size_t work_items) {
  size_t blocks = (work_items + BLOCK - 1) / BLOCK;
  long cap = g_aes_blocks.load();
  if (cap <= 0) cap = AES_GRID_CAP;
  if (blocks > (size_t)cap) blocks = (size_t)cap;
  if (blocks == 0) blocks = 1;
  return (unsigned)blocks;
}

// The four-table AES kernels: one 1024-thread workgroup per CU (128 KiB of dynamic LDS), grid-strided.
unsigned grid_aes4(size_t work_items) {
  size_t blocks = (work_items + ABLOCK - 1) / ABLOCK;
  long cap = g_aes_blocks.load();
  if (cap <= 0) cap = AES4_GRID_CAP;
  if (blocks > (size_t)cap) blocks = (size_t)cap;
  if (blocks == 0) blocks = 1;
  return (unsigned)blocks;
}
// launch KERN (a four-table AES kernel) with its 128 KiB of dynamic LDS
#define AES4_LAUNCH(KERN, WORK, ST, ...)                                                                            \
  do {                                                                                                              \
    auto kern_ = &KERN;                                                                                             \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                AES4_LDS_BYTES));                                                                   \
    hipLaunchKernelGGL(kern_, dim3(grid_aes4(WORK)), dim3(ABLOCK), AES4_LDS_BYTES, ST, __VA_ARGS__);                \
  } while (0)

// ---- Mont128 modulus: a process-wide default, latched per host thread ---------------------------------------
// scl_hip_mont128_set_prime sets the CALLING thread's modulus and the process-wide default.  A thread that has set its own
// keeps it whatever other threads do; a thread that never set one LATCHES the default at its first use (the modulus set last
// by any thread before that, 2^128 - 159 if none was) and keeps that value: a worker in the middle of a share-then-recover
// sequence does not switch modulus because some other thread set a different prime.  What it must not do either is go on
// SILENTLY: once the default has changed after a thread latched it, that thread's next Mont128 call fails with
// SCL_ERR_BAD_ARG (its data may be residues of either modulus -- only the caller knows) until the thread says which it means:
// scl_hip_mont128_set_prime (its own) or scl_hip_mont128_relatch (the current default).
int mont_set(u128 p) {
  if (!(p & 1) || p < 3) return fail(SCL_ERR_BAD_ARG, "mont128: modulus must be odd and >= 3");
  g_mont = Mont128::make_ctx(p);
  g_mont_own = true;
  std::lock_guard<std::mutex> lk(g_mont_default_mu);
  // the generation moves only when the default's VALUE does: pool workers that each set the same prime at start-up (or the
  // built-in 2^128 - 159) change nothing for the threads that latched it
  if (g_mont_default.p != g_mont.p) {
    g_mont_default = g_mont;
    ++g_mont_default_gen;
  }
  g_mont_gen = g_mont_default_gen;
  return SCL_OK;
}

void mont_latch_locked() {  // (g_mont_default_mu held)
  if (!g_mont_default.p)
    g_mont_default = Mont128::make_ctx((((u128)0xFFFFFFFFFFFFFFFFull) << 64) | (u128)0xFFFFFFFFFFFFFF61ull);  // 2^128 - 159
  g_mont = g_mont_default;
  g_mont_gen = g_mont_default_gen;
  g_mont_own = false;
}

// SCL_OK, or the stale-latch error described above
int mont_check() {
  if (!g_mont.p || g_mont_own) return SCL_OK;
  std::lock_guard<std::mutex> lk(g_mont_default_mu);
  if (g_mont_gen == g_mont_default_gen) return SCL_OK;
  return fail(SCL_ERR_BAD_ARG,
              "mont128: this thread latched the process-wide default modulus at its first Mont128 call and the default has been "
              "changed since (scl_hip_mont128_set_prime on another thread); call scl_hip_mont128_set_prime or "
              "scl_hip_mont128_relatch on this thread to say which modulus it computes over");
}

Mont128::Ctx mont_ctx() {
  if (g_mont.p) return g_mont;
  std::lock_guard<std::mutex> lk(g_mont_default_mu);
  mont_latch_locked();
  return g_mont;
}

// (tu_config.hpp: the fields THIS translation unit instantiates kernels for)
int not_in_this_unit() { return fail(SCL_ERR_BAD_ARG, "field family not built into this translation unit"); }

template <class Fn>
int with_field(int field, Fn&& fn) {
  switch (field) {
    case SCL_M61:
      if constexpr (SCL_TU_HAS(0)) return fn(M61{}, M61::Ctx{});
      else return not_in_this_unit();
    case SCL_M127:
      if constexpr (SCL_TU_HAS(1)) return fn(M127{}, M127::Ctx{});
      else return not_in_this_unit();
    case SCL_MONT128:
      if constexpr (SCL_TU_HAS(2)) {
        SCL_TRY(mont_check());
        return fn(Mont128{}, mont_ctx());
      } else {
        return not_in_this_unit();
      }
    case SCL_GF2_128:
      if constexpr (SCL_TU_HAS(3)) return fn(Gf128{}, Gf128::Ctx{});
      else return not_in_this_unit();
    case SCL_SECP256K1_SCALAR:
      if constexpr (SCL_TU_HAS(4)) return fn(Secp256k1Scalar{}, Secp256k1Scalar::Ctx{});
      else return not_in_this_unit();
    case SCL_SECP256K1_FIELD:
      if constexpr (SCL_TU_HAS(5)) return fn(Secp256k1Field{}, Secp256k1Field::Ctx{});
      else return not_in_this_unit();
    default: return fail(SCL_ERR_BAD_ARG, "unknown field tag");
  }
}

// Rings Z2k<K> travel under the tag SCL_Z2K(K) = 0x100 + K.  Only the entry points that make sense in a ring
// dispatch through here (element-wise, reductions, randomness, additive sharing, matmul); the Shamir and
// Lagrange entry points stay on with_field and refuse ring tags, as division by node differences would.
inline bool is_ring(int field) { return field > 0x100 && field <= 0x100 + 128; }

template <class Fn>
int with_ring_or_field(int field, Fn&& fn) {
  if (is_ring(field)) {
    const int K = field - 0x100;
    if constexpr (SCL_TU_HAS(6)) {
      if (K <= 64) return fn(Z2k64{}, Z2k64::make_ctx(K));
      return fn(Z2k128{}, Z2k128::make_ctx(K));
    } else {
      return not_in_this_unit();
    }
  }
  return with_field(field, fn);
}

// ---- per-thread device scratch (only used by calls that synchronise before returning) -------------

int scratch(size_t bytes, void** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (g_scratch.dev && (g_scratch.device != dev || g_scratch.bytes < bytes)) {
    (void)hipFree(g_scratch.dev);
    g_scratch = Scratch{};
  }
  if (!g_scratch.dev) {
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    HIP_TRY(hipMalloc(&g_scratch.dev, want));
    g_scratch.device = dev;
    g_scratch.bytes = want;
  }
  *out = g_scratch.dev;
  return SCL_OK;
}

// ---- per-thread device temporary for asynchronous calls ---------------------------------------------------
// A kernel-written temporary that outlives the call (the call returns before its kernels ran).  One buffer per
// host thread, kept and grown; an event recorded after the last use makes the next user -- possibly on another
// stream -- wait for it.  (hipMallocAsync / hipFreeAsync on the null stream proved unreliable under the ROCm 7.2
// runtime: a queue allocated that way lost writes between two kernels of one call.)

int temp_acquire(size_t bytes, hipStream_t st, void** out, int which = 0) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  TempArena& a = g_temps[which];
  if (a.dev && (a.device != dev || a.bytes < bytes)) {
    if (a.pending) HIP_TRY(hipEventSynchronize(a.done));
    (void)hipFree(a.dev);
    (void)hipEventDestroy(a.done);
    a = TempArena{};
  }
  if (!a.dev) {
    HIP_TRY(hipMalloc(&a.dev, bytes < (1u << 20) ? (1u << 20) : bytes));
    HIP_TRY(hipEventCreateWithFlags(&a.done, hipEventDisableTiming));
    a.device = dev;
    a.bytes = bytes < (1u << 20) ? (1u << 20) : bytes;
  }
  if (a.pending) HIP_TRY(hipStreamWaitEvent(st, a.done, 0));
  *out = a.dev;
  return SCL_OK;
}

// An arena is kept for the thread's next call up to TEMP_RETAIN_BYTES; a larger one (the digit planes of a matrix product
// with both factors beyond ~4096 x 4096, or with K beyond 2^21: the planes grow with K, 512 bytes per 32-row tile and inner
// column) goes back when the call that grew it ends -- hipFree waits for that call's kernels, which run for milliseconds
// at such sizes; the retained arenas are what scl_hip_thread_cleanup frees.
constexpr size_t TEMP_RETAIN_BYTES = (size_t)1 << 30;

int temp_release(hipStream_t st, int which = 0) {
  TempArena& a = g_temps[which];
  if (a.dev && a.bytes > TEMP_RETAIN_BYTES) {
    HIP_TRY(hipStreamSynchronize(st));
    (void)hipFree(a.dev);
    (void)hipEventDestroy(a.done);
    a = TempArena{};
    return SCL_OK;
  }
  HIP_TRY(hipEventRecord(a.done, st));
  a.pending = true;
  return SCL_OK;
}

// ---- host tables ---------------------------------------------------------------------------------
template <class F>
void default_nodes(const typename F::Ctx& ctx, size_t n, std::vector<typename F::E>& out) {
  // Vector::range(1, n+1): FF(int i) (vector.h:490-505) -- equal to the x++ walk of shamir.h:62-65 in
  // every prime field; for GF(2^128) the nodes are the bit patterns of 1..n.
  out.resize(n);
  for (size_t i = 0; i < n; ++i) out[i] = F::from_u64(ctx, (u64)(i + 1));
}

template <class F>
void load_host(const u64* src, size_t n, std::vector<typename F::E>& out) {
  out.resize(n);
  for (size_t i = 0; i < n; ++i) out[i] = F::ld(src + i * F::LIMBS);
}

// computeLagrangeBasis (lagrange.h:54-71): ell_i = prod_{j != i} (x - x_j) / (x_i - x_j).  The reference divides factor by
// factor (n (n - 1) inversions); here each ell_i is one numerator times the inverse of one denominator, and the n
// denominators are inverted together (prefix products, ONE field inversion, back-substitution): the per-secret
// shamirRecoverP of the C++ mirror spends its time here (tests/cxx/bench_per_secret.cc).  Same canonical values.
template <class F>
int lagrange(const typename F::Ctx& ctx, const std::vector<typename F::E>& nodes, typename F::E x,
             typename F::E* out) {
  using E = typename F::E;
  const size_t m = nodes.size();
  if (m == 0) return SCL_OK;
  std::vector<E> num(m), den(m), pre(m);
  for (size_t i = 0; i < m; ++i) {
    E nu = F::one(ctx), de = F::one(ctx);
    for (size_t j = 0; j < m; ++j) {
      if (i == j) continue;
      const E d = F::sub(ctx, nodes[i], nodes[j]);
      if (F::is_zero(d)) return fail(SCL_ERR_ZERO_INVERSE, scl_hip_status_message(SCL_ERR_ZERO_INVERSE));
      nu = F::mul(ctx, nu, F::sub(ctx, x, nodes[j]));
      de = F::mul(ctx, de, d);
    }
    num[i] = nu;
    den[i] = de;
    pre[i] = i ? F::mul(ctx, pre[i - 1], de) : de;  // den_0 * .. * den_i (no factor is zero)
  }
  E inv = F::inv(ctx, pre[m - 1]);  // 1 / (den_0 * .. * den_(m-1))
  for (size_t i = m; i-- > 0;) {
    const E inv_i = i ? F::mul(ctx, inv, pre[i - 1]) : inv;  // 1 / den_i
    out[i] = F::mul(ctx, num[i], inv_i);
    inv = F::mul(ctx, inv, den[i]);
  }
  return SCL_OK;
}

// ---- AES-128 key schedule + T-table (FIPS-197), host side ----------------------------------------------
struct AesHost {
  unsigned char sbox[256];
  AesHost() {
    auto mul = [](unsigned a, unsigned b) {
      unsigned r = 0;
      while (b) {
        if (b & 1) r ^= a;
        a = ((a << 1) ^ ((a & 0x80) ? 0x11b : 0)) & 0xff;
        b >>= 1;
      }
      return r;
    };
    // inverse table via generator 3: log/antilog
    unsigned char exp[256], log[256] = {0};
    unsigned v = 1;
    for (int i = 0; i < 255; ++i) {
      exp[i] = (unsigned char)v;
      log[v] = (unsigned char)i;
      v = mul(v, 3);
    }
    for (int x = 0; x < 256; ++x) {
      unsigned inv = x ? exp[(255 - log[x]) % 255] : 0;
      unsigned s = inv, r = inv;
      for (int i = 0; i < 4; ++i) {
        s = ((s << 1) | (s >> 7)) & 0xff;
        r ^= s;
      }
      sbox[x] = (unsigned char)(r ^ 0x63);
    }
  }
};

const AesHost& aes_host() {
  static const AesHost h;
  return h;
}

// key = seed zero-padded / truncated to 16 bytes (prg.cc:88-101)
void make_aes_key(const unsigned char* seed, size_t seed_len, AesKey& k) {
  const AesHost& h = aes_host();
  unsigned char rk[176] = {0};
  if (seed) std::memcpy(rk, seed, seed_len > 16 ? 16 : seed_len);
  unsigned rcon = 1;
  for (int i = 16; i < 176; i += 4) {
    unsigned char t[4] = {rk[i - 4], rk[i - 3], rk[i - 2], rk[i - 1]};
    if (i % 16 == 0) {
      const unsigned char t0 = t[0];
      t[0] = (unsigned char)(h.sbox[t[1]] ^ rcon);
      t[1] = h.sbox[t[2]];
      t[2] = h.sbox[t[3]];
      t[3] = h.sbox[t0];
      rcon = ((rcon << 1) ^ ((rcon & 0x80) ? 0x11b : 0)) & 0xff;
    }
    for (int j = 0; j < 4; ++j) rk[i + j] = (unsigned char)(rk[i - 16 + j] ^ t[j]);
  }
  for (int w = 0; w < 44; ++w)
    k.rk[w] = (u32)rk[4 * w] | ((u32)rk[4 * w + 1] << 8) | ((u32)rk[4 * w + 2] << 16) | ((u32)rk[4 * w + 3] << 24);
  for (int x = 0; x < 256; ++x) {
    const unsigned s = h.sbox[x];
    const unsigned s2 = ((s << 1) ^ ((s & 0x80) ? 0x11b : 0)) & 0xff;
    const unsigned s3 = s2 ^ s;
    k.te0[x] = s2 | (s << 8) | (s << 16) | (s3 << 24);
  }
  aes_key_round1(k);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Decide the per-lane vector width for M61 rows: 2 needs 16-byte aligned bases and even strides.
template <class F>
int vec_width(std::initializer_list<const void*> ptrs, std::initializer_list<size_t> strides) {
  if (F::LIMBS != 1) return 1;
  if (g_force_scalar.load()) return 1;
  for (const void* p : ptrs)
    if (p && !aligned16(p)) return 1;
  for (size_t s : strides)
    if (s & 1) return 1;
  return 2;
}

template <class F>
int check_align(std::initializer_list<const void*> ptrs) {
  const uintptr_t mask = F::LIMBS == 1 ? 7 : 15;
  for (const void* p : ptrs)
    if (p && (reinterpret_cast<uintptr_t>(p) & mask))
      return fail(SCL_ERR_BAD_ARG, F::LIMBS == 1 ? "pointer not 8-byte aligned" : "pointer not 16-byte aligned");
  return SCL_OK;
}

// The plain integer a node stands for, if it is one that fits 64 bits: the element itself for the Mersenne fields, the
// value out of Montgomery form for the 256-bit primes.
template <class F>
bool node_value(const typename F::Ctx& ctx, const typename F::E& e, u128& out) {
  if constexpr (F::LIMBS == 4) {
    const typename F::E v = F::from_mont(ctx, e);
    if (v.w[1] | v.w[2] | v.w[3]) return false;
    out = v.w[0];
    return true;
  } else if constexpr (F::TAG == 2) {  // Mont128: only a full-width modulus has the small-node reduction (Ctx::bmu)
    if (!F::small_nodes_ok(ctx)) return false;
    out = F::from_mont(ctx, e);
    return true;
  } else {
    out = (u128)e;
    return true;
  }
}

// every power a^k, k <= t, as an integer below 2^29: the one criterion of the small-node kernels, used by small_vandermonde
// (node by node) and by share_prg_impl's fused-or-two-pass choice (on the largest default node) -- so the two cannot drift
inline bool small_powers_fit(u128 a, size_t t) {
  const u128 lim = (u128)1 << 29;
  u128 pw = 1;
  for (size_t k = 0; k <= t; ++k) {
    if (pw >= lim) return false;
    if (k < t) {
      if (a >= lim) return false;
      pw *= a;  // < 2^58
    }
  }
  return true;
}

// Small-node test: every power alpha_i^k, k <= t, as an integer (no reduction) stays below 2^29.
template <class F>
bool small_vandermonde(const typename F::Ctx& ctx, const BigTable<F>& al, size_t n, size_t t, SmallVdm& sv) {
  if (g_force_table.load() || t > (size_t)SmallVdm::TMAX || n * (t + 1) > (size_t)SmallVdm::CAP) return false;
  for (size_t i = 0; i < n; ++i) {
    u128 a;
    if (!node_value<F>(ctx, al.v[i], a)) return false;
    if (!small_powers_fit(a, t)) return false;
    u128 pw = 1;
    for (size_t k = 0; k <= t; ++k) {
      sv.v[i * (t + 1) + k] = (u32)pw;
      pw *= a;  // (a^t < 2^29 and a < 2^29 checked above: < 2^58)
    }
  }
  return true;
}

// Blocked small-node form (k_share_blocked): the largest group size G in {8, 6, 4} with every alpha_i^(G-1) < 2^29
// and every alpha_i^G < 2^32 as integers; 0 if none fits.
template <class F>
int blocked_vandermonde(const typename F::Ctx& ctx, const BigTable<F>& al, size_t n, BlockVdm& bv) {
  if (g_force_table.load()) return 0;
  for (int G : {8, 6, 4}) {
    if (n * (size_t)(G + 1) > (size_t)BlockVdm::CAP) continue;
    bool ok = true;
    for (size_t i = 0; i < n && ok; ++i) {
      u128 a;
      if (!node_value<F>(ctx, al.v[i], a) || (a >> 29)) {
        ok = false;
        break;
      }
      u128 pw = 1;  // a < 2^29 and pw < 2^32 before each product: no overflow
      for (int r = 0; r <= G; ++r) {
        if (r < G) {
          if (pw >> 29) ok = false;
          else bv.v[i * G + r] = (u32)pw;
        } else {
          if (pw >> 32) ok = false;
          else bv.v[n * G + i] = (u32)pw;
        }
        if (!ok) break;
        pw *= a;
      }
    }
    if (ok) return G;
  }
  return 0;
}

// every node below 2^SMALL_BITS as an integer / bit pattern -> the Horner kernels' small-constant form.  `plain` receives
// what those kernels read the nodes from: the table itself for the Mersenne fields and GF(2^128), the nodes' plain integer
// values (out of Montgomery form) for the Montgomery fields.
template <class F>
bool small_nodes(const typename F::Ctx& ctx, const BigTable<F>& al, size_t n, BigTable<F>& plain) {
  plain = al;
  if constexpr (F::SMALL_BITS == 0) {
    return false;
  } else {
    if (g_force_table.load() > 1) return false;
    if constexpr (F::TAG == 2 || F::LIMBS == 4) {
      if (!F::small_nodes_ok(ctx)) return false;
      for (size_t i = 0; i < n; ++i) {
        u128 a;
        if (!node_value<F>(ctx, al.v[i], a) || (a >> F::SMALL_BITS)) return false;
        u64 w[F::LIMBS] = {(u64)a};
        plain.v[i] = F::ld(w);
      }
      return true;
    } else {
      for (size_t i = 0; i < n; ++i)
        if ((u128)al.v[i] >> F::SMALL_BITS) return false;
      return true;
    }
  }
}

template <class F>
int alpha_table(const typename F::Ctx& ctx, const u64* alphas_host, size_t n, BigTable<F>& tab) {
  if (n > (size_t)BigTable<F>::CAP)
    return fail(SCL_ERR_BAD_ARG, "share: internal party block too large");
  std::vector<typename F::E> nodes;
  if (alphas_host) load_host<F>(alphas_host, n, nodes);
  else default_nodes<F>(ctx, n, nodes);
  for (size_t i = 0; i < n; ++i) tab.v[i] = nodes[i];
  return SCL_OK;
}


#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

// ---- MFMA share path (Mersenne61): limb planes of the Vandermonde matrix, cached per device ----------
// A cached device table is owned through a shared pointer: the cache holds one reference, every caller holds another (its
// "pin") from the lookup until its kernel is enqueued.  An entry evicted in between -- by another host thread that misses
// while 16 entries are cached -- is therefore freed only when the last pin drops, i.e. after the launch; hipFree then waits
// for the kernels already enqueued, the pinning one included.
using DevPin = std::shared_ptr<void>;
inline DevPin make_dev_pin(void* dev) {
  return DevPin(dev, [](void* p) { (void)hipFree(p); });
}
struct MfmaTable {
  int device, n, t, KS, MT;
  std::vector<u64> alphas;
  DevPin dev;
};
std::mutex g_mfma_mu;
std::vector<MfmaTable> g_mfma_tables;  // immutable once built; most recently used last, at most TABLE_CACHE_CAP entries
constexpr size_t TABLE_CACHE_CAP = 16;

// Keeps a table cache bounded: a hit moves its entry to the back, a miss past the cap drops the cache's reference to the
// front (least recently used) entry; the table itself goes when no caller pins it any more (DevPin above).
template <class Entry>
void cache_touch(std::vector<Entry>& cache, size_t hit) {
  if (hit + 1 != cache.size()) std::rotate(cache.begin() + hit, cache.begin() + hit + 1, cache.end());
}
template <class Entry>
void cache_make_room(std::vector<Entry>& cache) {
  while (cache.size() >= TABLE_CACHE_CAP) cache.erase(cache.begin());
}

template <class FieldG>
int mfma_table(const BigTable<M61>& al, size_t n, size_t t, int KS, int MT, DevPin* pin, const unsigned char** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mfma_mu);
  for (size_t k = 0; k < g_mfma_tables.size(); ++k) {
    const MfmaTable& e = g_mfma_tables[k];
    if (e.device == dev && e.n == (int)n && e.t == (int)t && e.KS == KS && e.MT == MT &&
        std::equal(e.alphas.begin(), e.alphas.end(), al.v)) {
      *pin = e.dev;
      *out = static_cast<const unsigned char*>(e.dev.get());
      cache_touch(g_mfma_tables, k);
      return SCL_OK;
    }
  }
  cache_make_room(g_mfma_tables);
  const int ROWB = mf_rowb(KS);
  std::vector<unsigned char> host(mf_a_bytes(KS, MT), 0);
  const M61::Ctx ctx{};
  for (size_t i = 0; i < n; ++i) {
    u64 v = 1;  // alpha_i^k, Matrix::vandermonde (matrix.h:444-460)
    for (size_t k = 0; k <= t; ++k) {
      if (k) v = M61::mul(ctx, v, al.v[i]);
      const u64 digits = mf_recode(v);  // 8 signed base-256 digits, one per byte
      for (int l = 0; l < MF_LIMBS; ++l)
        host[((size_t)(l * MT + (int)(i / 32)) * 32 + (i % 32)) * ROWB + k] = (unsigned char)(digits >> (8 * l));
    }
  }
  void* raw = nullptr;
  HIP_TRY(hipMalloc(&raw, host.size()));
  MfmaTable e{dev, (int)n, (int)t, KS, MT, std::vector<u64>(al.v, al.v + n), make_dev_pin(raw)};
  HIP_TRY(hipMemcpy(raw, host.data(), host.size(), hipMemcpyHostToDevice));
  g_mfma_tables.push_back(e);
  *pin = e.dev;
  *out = static_cast<const unsigned char*>(raw);
  return SCL_OK;
}

template <int KS, int MT>
int launch_share_mfma(u64* shares, size_t stride, const u64* secrets, const u64* coeffs, size_t cstride,
                      const unsigned char* tab, int t, int n, size_t N, hipStream_t st, bool accumulate = false) {
  if (accumulate) {
    // Matrix::multiply's k-chunks after the first: the same kernels with their stores adding to what the rows hold
    if constexpr (MT == 4 && KS == 2) {
      auto kern = &k_share_mfma_m61_p16<M61, true>;
      const size_t shmem = 2 * mf_b_bytes(KS, MT, 1);
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
      const size_t nblocks = (N + 31) / 32;
      hipLaunchKernelGGL(kern, dim3((unsigned)(nblocks < 256 ? nblocks : 256)), dim3(512), shmem, st, shares, stride, secrets, coeffs,
                         cstride, tab, t, n, N);
    } else {
      auto kern = &k_share_mfma_m61<KS, MT, false, 512, true>;
      const size_t shmem = mf_a_bytes(KS, MT) + mf_b_bytes(KS, MT);
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
      const size_t cols = (size_t)2 * (4 / MT) * 32, nblocks = (N + cols - 1) / cols;
      hipLaunchKernelGGL(kern, dim3((unsigned)(nblocks < 256 ? nblocks : 256)), dim3(512), shmem, st, shares, stride, secrets, coeffs,
                         cstride, tab, t, n, N);
    }
    HIP_TRY(hipGetLastError());
    return SCL_OK;
  }
  const bool areg = g_mfma_areg.load() != 0 && MT == 4;
  const long tpb_mode = g_mfma_tpb.load();
  if constexpr (MT == 4 && KS == 2) {
    if (g_mfma_pipe.load() >= 2) {
      // k_share_mfma_m61_p16: one 8-wave workgroup per CU, 32 secrets per trip, two LDS images of the recoded block
      const size_t shmem = 2 * mf_b_bytes(KS, MT, 1);
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_share_mfma_m61_p16<>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
      const size_t nblocks = (N + 31) / 32;
      const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);
      hipLaunchKernelGGL(k_share_mfma_m61_p16<>, dim3(grid), dim3(512), shmem, st, shares, stride, secrets, coeffs, cstride, tab,
                         t, n, N);
      HIP_TRY(hipGetLastError());
      return SCL_OK;
    }
  }
  if constexpr (MT == 4) {
    if (g_mfma_pipe.load() != 0) {
      // k_share_mfma_m61_pipe: one 4-wave workgroup per CU, 32 secrets per trip
      const size_t shmem = mf_b_bytes(KS, MT, 1);
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_share_mfma_m61_pipe<KS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
      const size_t nblocks = (N + 31) / 32;
      const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);
      hipLaunchKernelGGL((k_share_mfma_m61_pipe<KS>), dim3(grid), dim3(256), shmem, st, shares, stride, secrets, coeffs,
                         cstride, tab, t, n, N);
      HIP_TRY(hipGetLastError());
      return SCL_OK;
    }
  }
  if (areg && tpb_mode == 256) {
    constexpr int TPB = 256;
    const size_t shmem = mf_b_bytes(KS, MT, 1);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_share_mfma_m61<KS, MT, (MT == 4), TPB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const size_t cols = (size_t)(4 / MT) * 32;
    const size_t nblocks = (N + cols - 1) / cols;
    const unsigned grid = (unsigned)(nblocks < 512 ? nblocks : 512);  // two 4-wave workgroups per CU
    hipLaunchKernelGGL((k_share_mfma_m61<KS, MT, (MT == 4), TPB>), dim3(grid), dim3(TPB), shmem, st, shares, stride,
                       secrets, coeffs, cstride, tab, t, n, N);
    HIP_TRY(hipGetLastError());
    return SCL_OK;
  }
  const size_t shmem = (areg ? 0 : mf_a_bytes(KS, MT)) + mf_b_bytes(KS, MT);
  // per device and cheap: set on every call so that multi-device processes are covered
  if (areg)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_share_mfma_m61<KS, MT, (MT == 4)>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  else
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_share_mfma_m61<KS, MT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  const size_t cols = (size_t)2 * (4 / MT) * 32;
  const size_t nblocks = (N + cols - 1) / cols;
  const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);  // one 8-wave workgroup per CU, grid-strided
  if (areg)
    hipLaunchKernelGGL((k_share_mfma_m61<KS, MT, (MT == 4)>), dim3(grid), dim3(512), shmem, st, shares, stride, secrets,
                       coeffs, cstride, tab, t, n, N);
  else
    hipLaunchKernelGGL((k_share_mfma_m61<KS, MT>), dim3(grid), dim3(512), shmem, st, shares, stride, secrets, coeffs,
                       cstride, tab, t, n, N);
  HIP_TRY(hipGetLastError());
  return SCL_OK;
}

// C[M x N] = A[M x K] * B[K x N] over Mersenne61 on the matrix cores: A's digit planes are built on the
// device into a stream-ordered temporary, B's rows play the role of the coefficient rows.
// the digit planes of A into `tab` (mf_a_bytes(KS, MT) bytes of device memory the caller owns)
template <int KS, int MT>
int mfma_planes_of(void* tab, const u64* A, size_t lda, size_t M, size_t K, hipStream_t st) {
  HIP_TRY(hipMemsetAsync(tab, 0, mf_a_bytes(KS, MT), st));
  hipLaunchKernelGGL((k_mfma_planes_from_matrix<KS, MT>), dim3((unsigned)((M * K + 255) / 256)), dim3(256), 0, st,
                     static_cast<unsigned char*>(tab), A, lda, (int)M, (int)K);
  return hipGetLastError() == hipSuccess ? SCL_OK : fail(SCL_ERR_HIP, "k_mfma_planes_from_matrix launch failed");
}

template <int KS, int MT>
int matmul_mfma_impl(u64* C, size_t ldc, const u64* A, size_t lda, const u64* B, size_t ldb, size_t M, size_t K,
                     size_t N, hipStream_t st, bool accumulate) {
  void* tab = nullptr;
  SCL_TRY(temp_acquire(mf_a_bytes(KS, MT), st, &tab));
  int rc = mfma_planes_of<KS, MT>(tab, A, lda, M, K, st);
  if (rc == SCL_OK)
    rc = launch_share_mfma<KS, MT>(C, ldc, B, B + ldb, ldb, static_cast<const unsigned char*>(tab), (int)K - 1, (int)M, N,
                                   st, accumulate);
  (void)temp_release(st);
  return rc;
}

template <class FieldG>
int matmul_mfma(u64* C, size_t ldc, const u64* A, size_t lda, const u64* B, size_t ldb, size_t M, size_t K, size_t N,
                hipStream_t st, bool accumulate = false) {
  const int KS = K <= 32 ? 1 : 2;
  // (one row tile with two k-steps would need 160 KiB for the staged right factor: such shapes take two row tiles)
  const int MT = (M <= 32 && KS == 1) ? 1 : M <= 64 ? 2 : 4;
#define MM_CASE(ks, mt) \
  if (KS == ks && MT == mt) return matmul_mfma_impl<ks, mt>(C, ldc, A, lda, B, ldb, M, K, N, st, accumulate);
  MM_CASE(1, 1) MM_CASE(1, 2) MM_CASE(1, 4) MM_CASE(2, 2) MM_CASE(2, 4)
#undef MM_CASE
  return fail(SCL_ERR_BAD_ARG, "matmul_mfma: unsupported shape");
}

// Any M and K on the matrix cores: row blocks of at most 128 rows of A, column chunks of at most 64 (the kernel's K).  The first
// chunk of a row block writes C, every later one ADDS its product to C in the kernel's own epilogue (the ACC instantiations of
// k_share_mfma_m61 / _p16: canonical partial products add exactly, SURVEY 8a note C; a temporary and a separate modular add
// cost as much as a chunk: 10 against 17 T multiply-adds/s).  Each (block, chunk) builds its digit planes in the call's
// temporary arena; the stream orders the launches.
template <class FieldG>
int matmul_mfma_blocks(const typename FieldG::Ctx&, u64* C, size_t ldc, const u64* A, size_t lda, const u64* B, size_t ldb, size_t M,
                       size_t K, size_t N, hipStream_t st) {
  for (size_t r0 = 0; r0 < M; r0 += 128) {
    const size_t mr = std::min<size_t>(128, M - r0);
    for (size_t k0 = 0; k0 < K; k0 += 64) {
      const size_t kc = std::min<size_t>(64, K - k0);
      SCL_TRY(matmul_mfma<FieldG>(C + r0 * ldc, ldc, A + r0 * lda + k0, lda, B + k0 * ldb, ldb, mr, kc, N, st, k0 != 0));
    }
  }
  return SCL_OK;
}

// General shapes on the matrix cores (gemm_mfma.hpp): both factors recoded into digit planes in fragment order (a temporary), then
// one kernel whose inner loop is matrix instructions only, K in super-steps of 8192.  The planes of a launch are kept within
// 1 GiB per factor: a longer factor goes slab by slab (rows of A outside, columns of B inside), each slab a product of its own.
template <class FieldG>
int gemm_mfma_slab(u64* C, size_t ldc, const u64x2* Ap, const u64* B, size_t ldb, u64x2* Bp, size_t M, size_t K, size_t N, hipStream_t st) {
  const size_t ktiles = (K + 31) / 32, mtiles = (M + 31) / 32, ntiles = (N + 31) / 32;
  HIP_TRY(gemm_launch_planes_b(Bp, B, ldb, K, N, ktiles, st));
  const size_t wgs = ((mtiles + 1) / 2) * ((ntiles + 1) / 2);
  // fewer workgroups than CUs and a long inner dimension: slices of the k-steps to workgroups of their own (>= 16 k-steps each),
  // the partial products into arena 1, one Vector::sum per entry over the slices
  size_t split = 1;
  if (wgs < 256 && ktiles >= 32 && ldc == N) split = std::min<size_t>(ktiles / 16, (512 + wgs - 1) / wgs);
  if (split > 1) {
    const size_t kslice = (ktiles + split - 1) / split;
    split = (ktiles + kslice - 1) / kslice;
    const size_t slice_elems = (M * N + 1) & ~(size_t)1;
    void* part = nullptr;
    SCL_TRY(temp_acquire(split * slice_elems * 8, st, &part, 1));
    int rc2 = gemm_launch_main(static_cast<u64*>(part), N, Ap, Bp, M, N, ktiles, kslice, slice_elems, split, st) == hipSuccess
                  ? SCL_OK
                  : fail(SCL_ERR_HIP, "matmul: launch failed");
    if (rc2 == SCL_OK) rc2 = scl_hip_additive_recover(SCL_M61, C, static_cast<u64*>(part), slice_elems, split, M * N, st);
    (void)temp_release(st, 1);
    return rc2;
  }
  HIP_TRY(gemm_launch_main(C, ldc, Ap, Bp, M, N, ktiles, ktiles, 0, 1, st));
  return SCL_OK;
}

template <class FieldG>
int matmul_gemm_mfma(u64* C, size_t ldc, const u64* A, size_t lda, const u64* B, size_t ldb, size_t M, size_t K, size_t N, hipStream_t st) {
  const long slab_mib = g_gemm_slab_mib.load();
  const size_t ktiles = (K + 31) / 32, budget = ((size_t)(slab_mib > 0 ? slab_mib : 1024) << 20) / 16;  // 16-byte units per factor and launch
  const size_t tile_units = ktiles * MF_LIMBS * 64;                                    // one 32-row (32-column) tile over all of K
  const size_t max_tiles = std::max<size_t>(1, budget / tile_units);
  const size_t Ms = std::min<size_t>(M, max_tiles * 32), Ns = std::min<size_t>(N, max_tiles * 32);
  const size_t a_units = (Ms + 31) / 32 * tile_units, b_units = (Ns + 31) / 32 * tile_units;
  void* tmp = nullptr;
  SCL_TRY(temp_acquire((a_units + b_units) * 16, st, &tmp));
  u64x2* Ap = static_cast<u64x2*>(tmp);
  u64x2* Bp = Ap + a_units;
  auto body = [&]() -> int {
    for (size_t r0 = 0; r0 < M; r0 += Ms) {
      const size_t mr = std::min(Ms, M - r0);
      HIP_TRY(gemm_launch_planes_a(Ap, A + r0 * lda, lda, mr, K, ktiles, st));
      for (size_t c0 = 0; c0 < N; c0 += Ns)
        SCL_TRY(gemm_mfma_slab<FieldG>(C + r0 * ldc + c0, ldc, Ap, B + c0, ldb, Bp, mr, K, std::min(Ns, N - c0), st));
    }
    return SCL_OK;
  };
  const int rc = body();
  (void)temp_release(st);
  return rc;
}

// shamirRecoverD as a contraction (k_detect_compare): L [rows x d1] host elements, rows = nchk + 1.  The product Y is kept
// for a slab of secrets at a time in the per-thread temporary, next to L and its digit planes.
template <int KS, int MT>
int detect_mfma_impl(u64* out, unsigned char* status, const u64* shares, size_t stride, const std::vector<u64>& L,
                     size_t rows, size_t d1, size_t N, unsigned long long* cnt, hipStream_t st) {
  const size_t slab = std::min<size_t>(N, (size_t)1 << 22);
  const size_t ldy = (slab + 63) / 64 * 64;
  const size_t abytes = (mf_a_bytes(KS, MT) + 255) / 256 * 256, lbytes = (rows * d1 * 8 + 255) / 256 * 256;
  void* tmp = nullptr;
  SCL_TRY(temp_acquire(abytes + lbytes + rows * ldy * 8, st, &tmp));
  unsigned char* base = static_cast<unsigned char*>(tmp);
  u64* L_dev = reinterpret_cast<u64*>(base + abytes);
  u64* Y = reinterpret_cast<u64*>(base + abytes + lbytes);
  auto body = [&]() -> int {
    HIP_TRY(hipMemcpyAsync(L_dev, L.data(), rows * d1 * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // L is a host temporary of the caller
    SCL_TRY((mfma_planes_of<KS, MT>(base, L_dev, d1, rows, d1, st)));
    for (size_t s0 = 0; s0 < N; s0 += slab) {
      const size_t ns = std::min(slab, N - s0);
      SCL_TRY((launch_share_mfma<KS, MT>(Y, ldy, shares + s0, shares + stride + s0, stride, base, (int)d1 - 1, (int)rows, ns, st)));
      hipLaunchKernelGGL((k_detect_compare<M61>), dim3(grid_for(ns)), dim3(BLOCK), 0, st, out + s0, status + s0, Y, ldy,
                         shares + d1 * stride + s0, stride, (int)rows - 1, ns, cnt);
      HIP_TRY(hipGetLastError());
    }
    return SCL_OK;
  };
  const int rc = body();
  (void)temp_release(st);
  return rc;
}

template <class FieldG>
int detect_mfma(u64* out, unsigned char* status, const u64* shares, size_t stride, const std::vector<u64>& L, size_t rows,
                size_t d1, size_t N, unsigned long long* cnt, hipStream_t st) {
  const int KS = d1 <= 32 ? 1 : 2;
  const int MT = (rows <= 32 && KS == 1) ? 1 : rows <= 64 ? 2 : 4;
#define DM_CASE(ks, mt) \
  if (KS == ks && MT == mt) return detect_mfma_impl<ks, mt>(out, status, shares, stride, L, rows, d1, N, cnt, st);
  DM_CASE(1, 1) DM_CASE(1, 2) DM_CASE(1, 4) DM_CASE(2, 2) DM_CASE(2, 4)
#undef DM_CASE
  return fail(SCL_ERR_BAD_ARG, "detect_mfma: unsupported shape");
}

template <class FieldG>
int share_mfma(const BigTable<M61>& al, u64* shares, size_t stride, const u64* secrets, const u64* coeffs,
               size_t cstride, size_t N, size_t t, size_t n, hipStream_t st) {
  const int KS = t + 1 <= 32 ? 1 : 2;
  const int MT = (n <= 32 && KS == 1) ? 1 : n <= 64 ? 2 : 4;  // (as matmul_mfma: no one-row-tile form with two k-steps)
  const unsigned char* tab = nullptr;
  DevPin pin;  // held until the launch below is enqueued
  SCL_TRY(mfma_table<FieldG>(al, n, t, KS, MT, &pin, &tab));
#define MF_CASE(ks, mt) \
  if (KS == ks && MT == mt) return launch_share_mfma<ks, mt>(shares, stride, secrets, coeffs, cstride, tab, (int)t, (int)n, N, st);
  MF_CASE(1, 1) MF_CASE(1, 2) MF_CASE(1, 4) MF_CASE(2, 2) MF_CASE(2, 4)
#undef MF_CASE
  return fail(SCL_ERR_BAD_ARG, "share_mfma: unsupported shape");
}

// Device-resident Vandermonde power tables of the Montgomery-field share kernels (k_share_vdm): row i holds
// table_scale(alpha_i^k), k = 1..t.  Cached per (device, field parameters, nodes, t) like the MFMA tables.
struct VdmTable {
  int device, tag, n, t;
  std::vector<u64> key;  // raw_coeffs flag, field parameters (Mont128: the prime), then the nodes
  DevPin dev;
};
std::mutex g_vdm_mu;
std::vector<VdmTable> g_vdm_tables;  // immutable once built; bounded like g_mfma_tables

template <class F>
bool vdm_eligible(size_t n, size_t t) {
  return (F::TAG == 2 || F::LIMBS == 4) && t >= 1 && t <= 16 && n * t * F::LIMBS <= (size_t)VdmLds::WORDS;
}

template <class F>
int vdm_table(const typename F::Ctx& ctx, const BigTable<F>& al, size_t n, size_t t, bool raw_coeffs, DevPin* pin, const u64** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::vector<u64> key{raw_coeffs ? 1u : 0u};
  if constexpr (F::TAG == 2) {
    key.push_back((u64)ctx.p);
    key.push_back((u64)(ctx.p >> 64));
  }
  for (size_t i = 0; i < n; ++i) {
    u64 w[F::LIMBS];
    F::st(w, al.v[i]);
    key.insert(key.end(), w, w + F::LIMBS);
  }
  std::lock_guard<std::mutex> lk(g_vdm_mu);
  for (size_t k = 0; k < g_vdm_tables.size(); ++k) {
    const VdmTable& e = g_vdm_tables[k];
    if (e.device == dev && e.tag == (int)F::TAG && e.n == (int)n && e.t == (int)t && e.key == key) {
      *pin = e.dev;
      *out = static_cast<const u64*>(e.dev.get());
      cache_touch(g_vdm_tables, k);
      return SCL_OK;
    }
  }
  cache_make_room(g_vdm_tables);
  std::vector<u64> host(n * t * F::LIMBS);
  for (size_t i = 0; i < n; ++i) {
    typename F::E v = al.v[i];  // alpha_i^k, Matrix::vandermonde (matrix.h:444-460)
    for (size_t k = 1; k <= t; ++k) {
      // raw_coeffs: the kernel multiplies by plain integers x, not residues xR, so the row carries the R
      const typename F::E entry = raw_coeffs ? F::to_mont(ctx, v) : v;
      F::st(host.data() + (i * t + (k - 1)) * F::LIMBS, F::table_scale(ctx, entry));
      v = F::mul(ctx, v, al.v[i]);
    }
  }
  void* raw = nullptr;
  HIP_TRY(hipMalloc(&raw, host.size() * sizeof(u64)));
  VdmTable e{dev, (int)F::TAG, (int)n, (int)t, key, make_dev_pin(raw)};
  HIP_TRY(hipMemcpy(raw, host.data(), host.size() * sizeof(u64), hipMemcpyHostToDevice));
  g_vdm_tables.push_back(e);
  *pin = e.dev;
  *out = static_cast<const u64*>(raw);
  return SCL_OK;
}

// Runs body(VEC-tag, first_element, npacks) for the vectorisable head and the scalar tail of [0,N).
template <class F, class Body>
int split_vec(int vec, size_t N, Body&& body) {
  if constexpr (F::LIMBS == 1) {
    if (vec == 2) {
      const size_t head = N & ~(size_t)1;
      if (head) SCL_TRY(body(std::integral_constant<int, 2>{}, (size_t)0, head / 2));
      if (N & 1) SCL_TRY(body(std::integral_constant<int, 1>{}, head, (size_t)1));
      return SCL_OK;
    }
  }
  if (N) SCL_TRY(body(std::integral_constant<int, 1>{}, (size_t)0, N));
  return SCL_OK;
}

// The Lagrange coefficients as small signed integers, when they are (k_recover_small): every lambda_i or p - lambda_i below
// 2^32 as a plain integer and both signed sums below 2^32 (the lazy limb sums then cannot wrap and stay below p * 2^32).
template <class F>
bool small_lambda(const typename F::Ctx& ctx, const u64* lambda_host, size_t m, SmallLam& sl) {
  if constexpr (F::TAG == 2 || F::LIMBS == 4) {
    if (m > (size_t)FIXED_M_MAX || g_force_table.load()) return false;
    u64 sum_pos = 0, sum_neg = 0;
    sl.neg = 0;
    for (size_t i = 0; i < m; ++i) {
      const typename F::E e = F::ld(lambda_host + i * F::LIMBS);
      u128 v;
      if (node_value<F>(ctx, e, v) && v < ((u128)1 << 32)) {
        sum_pos += (u64)v;
      } else if (node_value<F>(ctx, F::neg(ctx, e), v) && v < ((u128)1 << 32)) {
        sum_neg += (u64)v;
        sl.neg |= 1u << i;
      } else {
        return false;
      }
      sl.v[i] = (u32)v;
    }
    for (size_t i = m; i < (size_t)FIXED_M_MAX; ++i) sl.v[i] = 0;
    return sum_pos < ((u64)1 << 32) && sum_neg < ((u64)1 << 32);
  } else {
    (void)ctx; (void)lambda_host; (void)m; (void)sl;
    return false;
  }
}

template <class F, int M>
struct RecoverSmall {
  static int run(const typename F::Ctx& ctx, u64* out, const u64* shares, size_t stride, const SmallLam& lam, int m, size_t n,
                 hipStream_t st) {
    if (m == M) {
      // single-wave workgroups under the residency cap of the (m <= 16) stream kernels (kernels.hpp, "Launch geometry")
      const long sw = g_stream_waves.load();
      const size_t pad = residency_pad(sw < 0 ? (F::LIMBS == 4 ? 10 : 12) : sw, 64, 0);  // (profiles/r5_probe_f3_waves.txt)
      const size_t lanes = F::LIMBS == 4 ? 2 * n : n;  // 32-byte elements: a pair of lanes per secret
      hipLaunchKernelGGL((k_recover_small<F, M, true, 64>), dim3(grid_for_block(lanes, 64)), dim3(64), pad, st, ctx, out, shares, stride,
                         lam, n);
      HIP_TRY(hipGetLastError());
      return SCL_OK;
    }
    if constexpr (M > 1) return RecoverSmall<F, M - 1>::run(ctx, out, shares, stride, lam, m, n, st);
    return fail(SCL_ERR_BAD_ARG, "recover: internal");
  }
};

template <class F, int M>
struct RecoverFixed {
  template <int VEC>
  static int run(const typename F::Ctx& ctx, u64* out, const u64* shares, size_t stride, const Table<F>& lam, int m,
                 size_t npacks, hipStream_t st) {
    if (m == M) {
      const bool wave_groups = g_stream_block.load() == 64;
      const int blk = wave_groups ? 64 : BLOCK;
      const long sw = g_stream_waves.load();
      const size_t pad = residency_pad(sw < 0 ? (F::LIMBS == 1 ? 10 : 12) : sw, blk, 0);
      const dim3 g(grid_for_block(npacks, blk));
      if (!g_nontemporal.load())
        hipLaunchKernelGGL((k_recover_fixed<F, VEC, M, false>), dim3(grid_for(npacks)), dim3(BLOCK), 0, st, ctx, out,
                           shares, stride, lam, npacks);
      else if (wave_groups)
        hipLaunchKernelGGL((k_recover_fixed<F, VEC, M, true, 64>), g, dim3(64), pad, st, ctx, out, shares, stride, lam, npacks);
      else
        hipLaunchKernelGGL((k_recover_fixed<F, VEC, M, true>), g, dim3(BLOCK), pad, st, ctx, out, shares, stride, lam, npacks);
      return SCL_OK;
    }
    if constexpr (M > 1) return RecoverFixed<F, M - 1>::template run<VEC>(ctx, out, shares, stride, lam, m, npacks, st);
    return fail(SCL_ERR_BAD_ARG, "recover: m out of range");
  }
};


// the n nodes first_party+1 .. (default: FF(int) images, vector.h:490-505) or the caller's, as host limbs
static int nodes_to_host(int field, const uint64_t* alphas_host, size_t n, size_t first_party, u64* out) {
  return with_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    for (size_t i = 0; i < n; ++i) {
      if (alphas_host) F::st(out + i * F::LIMBS, F::ld(alphas_host + i * F::LIMBS));
      else F::st(out + i * F::LIMBS, F::from_u64(ctx, (u64)(first_party + i + 1)));
    }
    return SCL_OK;
  });
}

// Thresholds above 48 (k_share_chunk): Horner over chunks of the coefficient rows, top chunk first.  n <= the node table.
// The node / node-power tables sit in the calling thread's temporary arena; the call synchronises (rarely used path).
static int share_chunked(int field, uint64_t* shares, size_t share_stride, const uint64_t* secrets, const uint64_t* coeffs,
                         size_t coeff_stride, size_t N, size_t t, size_t n, const uint64_t* alphas_host, size_t first_party,
                         void* stream) {
  return with_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    typedef typename F::E E;
    SCL_TRY(check_align<F>({shares, secrets, coeffs}));
    if (n > (size_t)BigTable<F>::CAP) return fail(SCL_ERR_BAD_ARG, "share: internal party block too large");
    constexpr size_t TC = (size_t)share_chunk_t<F>(), C = TC + 1;
    std::vector<u64> nodes(n * F::LIMBS), tab(2 * n * F::LIMBS);
    SCL_TRY(nodes_to_host(field, alphas_host, n, first_party, nodes.data()));
    BigTable<F> al, alx;
    for (size_t i = 0; i < n; ++i) al.v[i] = F::ld(nodes.data() + i * F::LIMBS);
    const bool smallx = small_nodes<F>(ctx, al, n, alx);  // nodes as small integers: the fields' small-constant Horner step
    for (size_t i = 0; i < n; ++i) {
      const E a = al.v[i];
      E pw = F::one(ctx);
      for (size_t k = 0; k < C; ++k) pw = F::mul(ctx, pw, a);
      F::st(tab.data() + i * F::LIMBS, smallx ? alx.v[i] : a);
      F::st(tab.data() + (n + i) * F::LIMBS, pw);
    }
    void* dev = nullptr;
    SCL_TRY(temp_acquire(tab.size() * 8, S(stream), &dev));
    auto body = [&]() -> int {
      HIP_TRY(hipMemcpyAsync(dev, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, S(stream)));
      // chunks [0, TC], [C, C + TC], ..: the top one may be short and goes first
      const size_t nchunks = t / C + 1;
      for (size_t j = nchunks; j-- > 0;) {
        const size_t k_lo = j * C, k_hi = (k_lo + TC < t) ? k_lo + TC : t;
        const u64* c0 = k_lo == 0 ? secrets : coeffs + (k_lo - 1) * coeff_stride * F::LIMBS;
        const u64* crest = coeffs + k_lo * coeff_stride * F::LIMBS;
        if (smallx)
          hipLaunchKernelGGL((k_share_chunk<F, true>), dim3(grid_for(N)), dim3(BLOCK), 0, S(stream), ctx, shares, share_stride, c0,
                             crest, coeff_stride, static_cast<const u64*>(dev), (int)(k_hi - k_lo), (int)n, N,
                             j + 1 != nchunks ? 1 : 0);
        else
          hipLaunchKernelGGL((k_share_chunk<F>), dim3(grid_for(N)), dim3(BLOCK), 0, S(stream), ctx, shares, share_stride, c0, crest,
                             coeff_stride, static_cast<const u64*>(dev), (int)(k_hi - k_lo), (int)n, N, j + 1 != nchunks ? 1 : 0);
        LAUNCH_CHECK();
      }
      HIP_TRY(hipStreamSynchronize(S(stream)));  // the host tables die with this frame
      return SCL_OK;
    };
    const int rc = body();
    (void)temp_release(S(stream));
    return rc;
  });
}

// Inverse / divide by Montgomery's simultaneous inversion (kernels.hpp, k_ew_inv / k_ew_inv_rolled).  Chain length: Mersenne61
// keeps 32 elements per lane in registers (16-byte packs); the other fields keep the chain in memory and take 8 / 16 / 32 / 64 / 128
// by the batch size (ew_inverse_rolled; the inversion's share of a chain is I / L products per element, I = 138 for Mersenne127 up
// to ~330 for secp256k1, against 3 for the walk).
template <class F, bool DIV, int L>
int launch_inv_rolled(const typename F::Ctx& ctx, u64* dst, const u64* a, const u64* b, size_t n, unsigned* flag, hipStream_t st) {
  // single-wave workgroups; GF(2^128): 256 threads around a 32 KiB window table (3-bit windows; 16.0 against 14.1 G inversions/s with 64,
  // profiles/r5_ew_bench.txt)
  constexpr int BLK = F::TAG == 3 ? 256 : 64;
  using ARITH = std::conditional_t<F::TAG == 3, GfLdsArith<BLK, 3>, FieldArith<F>>;
  auto kern = &k_ew_inv_rolled<F, ARITH, DIV, L, BLK>;
  const size_t lds = (size_t)BLK * ARITH::LDS_PER_LANE;
  if (lds) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const size_t tiles = (n + (size_t)BLK * L - 1) / ((size_t)BLK * L);
  hipLaunchKernelGGL(kern, dim3(grid_for_block(tiles, 1)), dim3(BLK), lds, st, ctx, dst, a, b, n, flag);
  LAUNCH_CHECK();
  return SCL_OK;
}

// the two-level form (k_ew_inv_blocked): the 16-byte prime fields only (their products are the field's own; L2 = 8 operands and
// 8 prefix products in registers)
// Block length taken when "inv_two_level" is 0, from tools/probe_inv_two_level.py (profiles/r6_probe_inv_two_level.txt, _b.txt): the
// form pays where the rolled inversion is bound by its scratch traffic.  Mersenne127: from chains of 64 on (10^7 elements: 0.186 ->
// 0.170 ms with blocks of 4; 3 * 10^7 and 10^8 at chains of 128 with blocks of 8: 0.49 -> 0.40 ms and 1.55 -> 1.14 ms; divide
// 1.15-1.28 x), a loss below (chains of 32: 0.84-1.0 x).  Mont128, once its product had been halved (Mont128::mul, product
// scanning): chains of 128 with blocks of 8, 1.66 -> 1.50 ms at 10^8 (1.04-1.10 x); level or behind at chains of 64 and below.
template <class F>
constexpr int inv_two_level_default(int L) {
  return F::TAG == 1 ? (L >= 128 ? 8 : L == 64 ? 4 : 0) : F::TAG == 2 ? (L >= 128 ? 8 : 0) : 0;
}
template <class F, bool DIV, int L, int L2>
int launch_inv_blocked(const typename F::Ctx& ctx, u64* dst, const u64* a, const u64* b, size_t n, unsigned* flag, hipStream_t st) {
  constexpr int BLK = 64, L1 = L / L2;
  auto kern = &k_ew_inv_blocked<F, FieldArith<F>, DIV, L1, L2, BLK>;
  const size_t tiles = (n + (size_t)BLK * L - 1) / ((size_t)BLK * L);
  hipLaunchKernelGGL(kern, dim3(grid_for_block(tiles, 1)), dim3(BLK), 0, st, ctx, dst, a, b, n, flag);
  LAUNCH_CHECK();
  return SCL_OK;
}
template <class F, bool DIV, int L2>
int launch_inv_blocked_by_length(const typename F::Ctx& ctx, u64* dst, const u64* a, const u64* b, size_t n, int L, unsigned* flag, hipStream_t st) {
  if (L == 256) return launch_inv_blocked<F, DIV, 256, L2>(ctx, dst, a, b, n, flag, st);
  if (L == 128) return launch_inv_blocked<F, DIV, 128, L2>(ctx, dst, a, b, n, flag, st);
  if (L == 64) return launch_inv_blocked<F, DIV, 64, L2>(ctx, dst, a, b, n, flag, st);
  return launch_inv_blocked<F, DIV, 32, L2>(ctx, dst, a, b, n, flag, st);
}

template <class F, bool DIV>
int ew_inverse_rolled(const typename F::Ctx& ctx, u64* dst, const u64* a, const u64* b, size_t n, long want, unsigned* flag, hipStream_t st) {
  // The chain length by batch size, from the sweeps of tools/probe_inv_chain.py (profiles/r5_probe_inv_chain.txt): the one
  // inversion of a chain is 50-65 % of a lane's work at L = 32, so a longer chain wins as soon as about 500 (L = 16) to 3000
  // (L = 128) waves remain -- one to three per SIMD, not four.  Elements from which a length is taken, for 16 / 32 / 64 / 128:
  // the 16-byte prime fields; secp256k1 (the costliest inversion: longer chains sooner); GF(2^128) (256-thread workgroups around
  // their LDS tables: later).
  static const size_t from_plain[4] = {500000, 2050000, 6000000, 24600000}, from_wide[4] = {500000, 1800000, 4000000, 15000000},
                      from_gf[4] = {500000, 6000000, 15000000, 50000000};
  const size_t* from = F::TAG == 3 ? from_gf : F::LIMBS == 4 ? from_wide : from_plain;
  int L = 8;
  for (int i = 0; i < 4; ++i)
    if (n >= from[i]) L = 16 << i;
  if (want > 0) L = want >= 256 ? 256 : want >= 128 ? 128 : want >= 64 ? 64 : want >= 32 ? 32 : want >= 16 ? 16 : 8;
  if constexpr (F::TAG == 1 || F::TAG == 2) {
    // two levels for the field whose rolled inversion is bound by its scratch traffic: Mersenne127 from chains of 32 on
    // ("inv_two_level": 4 | 8 = that block length, also for Mont128; -1 never)
    const long knob = g_inv_two_level.load();
    const long two = knob > 0 ? knob : knob == 0 ? inv_two_level_default<F>(L) : 0;   // block length: 4 or 8; 0 = rolled
    if (L >= 32 && two >= 8) return launch_inv_blocked_by_length<F, DIV, 8>(ctx, dst, a, b, n, L, flag, st);
    if (L >= 32 && two >= 1) return launch_inv_blocked_by_length<F, DIV, 4>(ctx, dst, a, b, n, L, flag, st);
  }
  if (L >= 128) return launch_inv_rolled<F, DIV, 128>(ctx, dst, a, b, n, flag, st);   // (256: the two-level form only)
  if (L == 64) return launch_inv_rolled<F, DIV, 64>(ctx, dst, a, b, n, flag, st);
  if (L == 32) return launch_inv_rolled<F, DIV, 32>(ctx, dst, a, b, n, flag, st);
  if (L == 16) return launch_inv_rolled<F, DIV, 16>(ctx, dst, a, b, n, flag, st);
  return launch_inv_rolled<F, DIV, 8>(ctx, dst, a, b, n, flag, st);
}

template <class F>
int ew_inverse(const typename F::Ctx& ctx, bool div, u64* dst, const u64* a, const u64* b, size_t n, int vec, long want, unsigned* flag,
               hipStream_t st) {
  if constexpr (F::LIMBS == 1) {
    if (vec == 2 && n >= 2) {
      const size_t npacks = n / 2, tiles = (npacks + (size_t)BLOCK * 16 - 1) / ((size_t)BLOCK * 16);
      const dim3 g(grid_for_block(tiles, 1)), blk(BLOCK);
      if (div) hipLaunchKernelGGL((k_ew_inv<F, true, 2, 16, 4, 3>), g, blk, 0, st, ctx, dst, a, b, npacks, flag);
      else hipLaunchKernelGGL((k_ew_inv<F, false, 2, 16, 4, 3>), g, blk, 0, st, ctx, dst, a, b, npacks, flag);
      LAUNCH_CHECK();
      if (n & 1) {  // the odd element out
        const size_t o = n - 1;
        if (div) hipLaunchKernelGGL((k_ew<F, 5, 1, true>), dim3(1), dim3(BLOCK), 0, st, ctx, dst + o, a + o, b + o, (size_t)1, flag);
        else hipLaunchKernelGGL((k_ew<F, 4, 1, true>), dim3(1), dim3(BLOCK), 0, st, ctx, dst + o, a + o, b, (size_t)1, flag);
        LAUNCH_CHECK();
      }
      return SCL_OK;
    }
  }
  return div ? ew_inverse_rolled<F, true>(ctx, dst, a, b, n, want, flag, st) : ew_inverse_rolled<F, false>(ctx, dst, a, b, n, want, flag, st);
}

}  // namespace

// =====================================================================================================
extern "C" {

#if SCL_TU_COMMON
int scl_hip_abi_version(void) { return SCL_HIP_ABI_VERSION; }

const char* scl_hip_last_error(void) { return g_err.c_str(); }

const char* scl_hip_status_message(int status) {
  switch (status) {
    case SCL_OK: return "";
    case SCL_ERR_SIZE_MISMATCH: return "Vec sizes mismatch";
    case SCL_ERR_ZERO_INVERSE: return "0 not invertible modulo prime";
    case SCL_ERR_BAD_ARG: return "bad argument";
    case SCL_ERR_HIP: return "HIP runtime error";
    case SCL_ERR_NO_DEVICE: return "no HIP device";
    case SCL_ERR_ERROR_DETECTED: return "error detected during recovery";
    case SCL_ERR_NOT_ENOUGH_SHARES: return "not enough shares provided to detect errors";
    case SCL_ERR_MATMUL_DIMS: return "matmul: this->cols() != that->rows()";
    case SCL_ERR_VANDERMONDE_XS: return "|xs| != number of rows";
    case SCL_ERR_INVALID_RANGE: return "invalid range";
    case SCL_ERR_NOT_INVERTIBLE_2K: return "value not invertible modulo 2^K";
    default: return "unknown status";
  }
}

int scl_hip_limbs(int field) {
  if (is_ring(field)) return field - 0x100 <= 64 ? 1 : 2;
  return field == SCL_M61 ? 1 : (field >= 1 && field <= 3) ? 2 : (field == SCL_SECP256K1_SCALAR || field == SCL_SECP256K1_FIELD) ? 4 : -1;
}

const char* scl_hip_field_name(int field) {
  if (is_ring(field)) return "Z2k";  // z2k.h:64-66
  switch (field) {
    case SCL_M61: return "Mersenne61";
    case SCL_M127: return "Mersenne127";
    case SCL_MONT128: return "Mont128";
    case SCL_GF2_128: return "GF(2^128)";
    case SCL_SECP256K1_SCALAR: return "secp256k1_order";  // secp256k1_scalar.h NAME
    case SCL_SECP256K1_FIELD: return "secp256k1_field";   // secp256k1_field.h NAME
    default: return "";
  }
}

// ---- plumbing ------------------------------------------------------------------------------------------
int scl_hip_device_count(int* count) {
  if (!count) return fail(SCL_ERR_BAD_ARG, "count is NULL");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return fail(SCL_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  return SCL_OK;
}
int scl_hip_set_device(int device) {
  HIP_TRY(hipSetDevice(device));
  return SCL_OK;
}
int scl_hip_malloc(void** dev, size_t bytes) {
  if (!dev) return fail(SCL_ERR_BAD_ARG, "dev is NULL");
  HIP_TRY(hipMalloc(dev, bytes ? bytes : 16));
  return SCL_OK;
}
int scl_hip_free(void* dev) {
  HIP_TRY(hipFree(dev));
  return SCL_OK;
}
int scl_hip_memcpy_h2d(void* dev, const void* host, size_t bytes, void* stream) {
  HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, S(stream)));
  return SCL_OK;
}
int scl_hip_memcpy_d2h(void* host, const void* dev, size_t bytes, void* stream) {
  HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, S(stream)));
  HIP_TRY(hipStreamSynchronize(S(stream)));
  return SCL_OK;
}
int scl_hip_memset(void* dev, int value, size_t bytes, void* stream) {
  HIP_TRY(hipMemsetAsync(dev, value, bytes, S(stream)));
  return SCL_OK;
}
int scl_hip_stream_create(void** stream) {
  if (!stream) return fail(SCL_ERR_BAD_ARG, "stream is NULL");
  hipStream_t s;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return SCL_OK;
}
int scl_hip_stream_destroy(void* stream) {
  HIP_TRY(hipStreamDestroy(S(stream)));
  return SCL_OK;
}
int scl_hip_stream_sync(void* stream) {
  HIP_TRY(hipStreamSynchronize(S(stream)));
  return SCL_OK;
}

struct SclTimer {
  hipEvent_t a, b;
};
int scl_hip_timer_create(void** timer) {
  if (!timer) return fail(SCL_ERR_BAD_ARG, "timer is NULL");
  SclTimer* t = new SclTimer;
  hipError_t e = hipEventCreate(&t->a);
  if (e == hipSuccess) e = hipEventCreate(&t->b);
  if (e != hipSuccess) {
    delete t;
    return fail(SCL_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e));
  }
  *timer = t;
  return SCL_OK;
}
int scl_hip_timer_destroy(void* timer) {
  SclTimer* t = static_cast<SclTimer*>(timer);
  if (!t) return SCL_OK;
  (void)hipEventDestroy(t->a);
  (void)hipEventDestroy(t->b);
  delete t;
  return SCL_OK;
}
int scl_hip_timer_start(void* timer, void* stream) {
  if (!timer) return fail(SCL_ERR_BAD_ARG, "timer is NULL");
  HIP_TRY(hipEventRecord(static_cast<SclTimer*>(timer)->a, S(stream)));
  return SCL_OK;
}
int scl_hip_timer_stop(void* timer, void* stream) {
  if (!timer) return fail(SCL_ERR_BAD_ARG, "timer is NULL");
  HIP_TRY(hipEventRecord(static_cast<SclTimer*>(timer)->b, S(stream)));
  return SCL_OK;
}
int scl_hip_timer_elapsed_ms(void* timer, float* ms) {
  SclTimer* t = static_cast<SclTimer*>(timer);
  if (!t || !ms) return fail(SCL_ERR_BAD_ARG, "timer or ms is NULL");
  HIP_TRY(hipEventSynchronize(t->b));
  HIP_TRY(hipEventElapsedTime(ms, t->a, t->b));
  return SCL_OK;
}

// Frees what the CALLING thread holds on the device: its scratch buffer and its temporary arena (both are kept and
// grown across calls).  A host thread that is about to exit calls this; the buffers cannot be released from a
// thread_local destructor, which may run after the HIP runtime has shut down.
int scl_hip_thread_cleanup(void) {
  if (g_scratch.dev) {
    HIP_TRY(hipFree(g_scratch.dev));
    g_scratch = Scratch{};
  }
  if (g_hflag.host) {
    HIP_TRY(hipHostFree(g_hflag.host));
    g_hflag = HostFlag{};
  }
  for (TempArena& g_temp : g_temps) {
    if (!g_temp.dev) continue;
    if (g_temp.pending) HIP_TRY(hipEventSynchronize(g_temp.done));
    HIP_TRY(hipFree(g_temp.dev));
    (void)hipEventDestroy(g_temp.done);
    g_temp = TempArena{};
  }
  return SCL_OK;
}

int scl_hip_set_tuning(const char* key, long value) {
  if (!key) return fail(SCL_ERR_BAD_ARG, "key is NULL");
  const std::string k(key);
  if (k == "max_blocks") g_max_blocks = value;
  else if (k == "nontemporal") g_nontemporal = value;
  else if (k == "force_scalar") g_force_scalar = value;
  else if (k == "force_table") g_force_table = value;
  else if (k == "mfma_areg") g_mfma_areg = value;
  else if (k == "prg_two_pass") g_prg_two_pass = value;
  else if (k == "mfma_pipe") g_mfma_pipe = value;
  else if (k == "mfma_tpb") g_mfma_tpb = value;
  else if (k == "aes_blocks") g_aes_blocks = value;
  else if (k == "inv_batch") g_inv_batch = value;
  else if (k == "inv_two_level") g_inv_two_level = value;
  else if (k == "transpose_tile") g_transpose_tile = value;
  else if (k == "gemm_slab_mib") g_gemm_slab_mib = value;
  else if (k == "matmul_lds_min") g_matmul_lds_min = value;
  else if (k == "stream_block") g_stream_block = (value == 256 ? 256 : 64);
  else if (k == "stream_waves") g_stream_waves = value;
  else if (k == "share_waves") g_share_waves = value;
  else if (k == "share_waves128") g_share_waves128 = value;
  else if (k == "mfma") g_mfma = value;
  else if (k == "gf_tiles") g_gf_tiles = value;
  else if (k == "prg_t3") g_prg_t3 = value;
  else if (k == "open_gather_always") g_open_gather_always = value;
  else return fail(SCL_ERR_BAD_ARG, "unknown tuning key " + k);
  return SCL_OK;
}

int scl_hip_mont128_set_prime(const uint64_t p[2]) {
  if (!p) return fail(SCL_ERR_BAD_ARG, "p is NULL");
  return mont_set(((u128)p[1] << 64) | p[0]);
}
int scl_hip_mont128_relatch(void) {
  std::lock_guard<std::mutex> lk(g_mont_default_mu);
  mont_latch_locked();
  return SCL_OK;
}
int scl_hip_mont128_get_prime(uint64_t p[2]) {
  if (!p) return fail(SCL_ERR_BAD_ARG, "p is NULL");
  const Mont128::Ctx c = mont_ctx();
  p[0] = (u64)c.p;
  p[1] = (u64)(c.p >> 64);
  return SCL_OK;
}
#endif  // SCL_TU_COMMON

// ---- element-wise ---------------------------------------------------------------------------------------
// status_dev == nullptr: scl_hip_ew, which reports a zero operand of INV / DIV itself (and therefore synchronises);
// otherwise scl_hip_ew_status: the flag is the caller's device word, nothing waits.
static int ew_impl(int field, int op, uint64_t* dst, const uint64_t* a, const uint64_t* b, size_t n, unsigned* status_dev,
                   bool async, void* stream) {
  if (op < 0 || op > 5) return fail(SCL_ERR_BAD_ARG, "unknown element-wise op");
  const bool binary = (op == SCL_OP_ADD || op == SCL_OP_SUB || op == SCL_OP_MUL || op == SCL_OP_DIV);
  const bool needs_flag = (op == SCL_OP_INV || op == SCL_OP_DIV);
  if (async && needs_flag && !status_dev) return fail(SCL_ERR_BAD_ARG, "status_dev is NULL (INV / DIV report through it)");
  if (async && status_dev && (reinterpret_cast<uintptr_t>(status_dev) & 3)) return fail(SCL_ERR_BAD_ARG, "status_dev is not 4-byte aligned");
  if (n == 0) return SCL_OK;
  if (!dst || !a || (binary && !b)) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  return with_ring_or_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({dst, a, binary ? b : nullptr}));
    unsigned* flag = nullptr;
    bool host_flag = false;
    if (needs_flag && async) {
      flag = status_dev;
    } else if (needs_flag) {
      if (!g_hflag.host && !g_hflag.failed) {
        void* hp = nullptr;
        void* dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
          g_hflag.host = static_cast<unsigned*>(hp);
          g_hflag.dev = static_cast<unsigned*>(dp);
        } else {
          if (hp) (void)hipHostFree(hp);
          (void)hipGetLastError();
          g_hflag.failed = true;
        }
      }
      if (g_hflag.host) {  // (the previous call on this thread synchronised before it returned: nobody writes the word now)
        *g_hflag.host = 0;
        flag = g_hflag.dev;
        host_flag = true;
      } else {
        void* sc;
        SCL_TRY(scratch(64, &sc));
        flag = static_cast<unsigned*>(sc);
        HIP_TRY(hipMemsetAsync(flag, 0, 4, S(stream)));
      }
    }
    const int vec = vec_width<F>({dst, a, binary ? b : nullptr}, {});
    const bool nt = g_nontemporal.load() != 0;
    const long batch = g_inv_batch.load();
    int rc_ = SCL_OK;
    constexpr bool is_field = F::TAG != 5 && F::TAG != 6;
    if (needs_flag && batch >= 0 && is_field) {
      // FF::invert / operator/ (ff.h:203-246) over the batch by simultaneous inversion -- fields only: a ring's Newton inverse
      // (z2k_ops.h:80-93) is a handful of products already
      if constexpr (is_field) rc_ = ew_inverse<F>(ctx, op == SCL_OP_DIV, dst, a, b, n, vec, batch, flag, S(stream));
    } else if (op == SCL_OP_MUL && F::TAG == 3 && batch >= 0) {
      if constexpr (F::TAG == 3) {  // GF(2^128) products on the window table in LDS
        auto kern = &k_ew_gf128_mul<64, 3>;  // 3-bit windows: 128 bytes of table per lane
        hipLaunchKernelGGL(kern, dim3(grid_for_block(n, 64)), dim3(64), 64 * 128, S(stream), dst, a, b, n);
        LAUNCH_CHECK();
      }
    } else {
      rc_ = split_vec<F>(vec, n, [&](auto V, size_t first, size_t npacks) -> int {
        constexpr int VEC = decltype(V)::value;
        u64* d = dst + first * F::LIMBS;
        const u64* pa = a + first * F::LIMBS;
        const u64* pb = binary ? b + first * F::LIMBS : nullptr;
        const dim3 g(grid_for(npacks)), blk(BLOCK);
#define EW_CASE(OP)                                                                                             \
  case OP:                                                                                                      \
    if (nt) hipLaunchKernelGGL((k_ew<F, OP, VEC, true>), g, blk, 0, S(stream), ctx, d, pa, pb, npacks, flag);   \
    else hipLaunchKernelGGL((k_ew<F, OP, VEC, false>), g, blk, 0, S(stream), ctx, d, pa, pb, npacks, flag);     \
    break;
        switch (op) {
          EW_CASE(0) EW_CASE(1) EW_CASE(2) EW_CASE(3) EW_CASE(4) EW_CASE(5)
        }
#undef EW_CASE
        LAUNCH_CHECK();
        return SCL_OK;
      });
    }
    if (rc_ != SCL_OK) {
      // some launches of the call may be in flight with the flag in their arguments: nothing of this call runs on once it has
      // returned (the next call on this thread resets the host-mapped word)
      if (needs_flag && !async) (void)hipStreamSynchronize(S(stream));
      return rc_;
    }
    if (needs_flag && !async) {
      unsigned h = 0;
      if (!host_flag) HIP_TRY(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, S(stream)));
      HIP_TRY(hipStreamSynchronize(S(stream)));
      if (host_flag) h = *static_cast<volatile unsigned*>(g_hflag.host);
      if (h) {
        const int code = (F::TAG == 5 || F::TAG == 6) ? SCL_ERR_NOT_INVERTIBLE_2K : SCL_ERR_ZERO_INVERSE;
        return fail(code, scl_hip_status_message(code));
      }
    }
    return SCL_OK;
  });
}

int scl_hip_ew(int field, int op, uint64_t* dst, const uint64_t* a, const uint64_t* b, size_t n, void* stream) {
  return ew_impl(field, op, dst, a, b, n, nullptr, false, stream);
}
int scl_hip_ew_status(int field, int op, uint64_t* dst, const uint64_t* a, const uint64_t* b, size_t n, unsigned* status_dev,
                      void* stream) {
  return ew_impl(field, op, dst, a, b, n, status_dev, true, stream);
}

int scl_hip_scalar_mul(int field, uint64_t* dst, const uint64_t* a, const uint64_t* scalar_host, size_t n,
                       void* stream) {
  if (n == 0) return SCL_OK;
  if (!dst || !a || !scalar_host) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  return with_ring_or_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({dst, a}));
    Table<F> sc;
    sc.v[0] = F::ld(scalar_host);
    const int vec = vec_width<F>({dst, a}, {});
    if constexpr (F::TAG == 3) {
      if (g_inv_batch.load() >= 0) {  // GF(2^128): one shared window table of the scalar in LDS
        hipLaunchKernelGGL(k_scalar_mul_gf128<BLOCK>, dim3(grid_for(n)), dim3(BLOCK), 0, S(stream), dst, a, sc, n);
        LAUNCH_CHECK();
        return SCL_OK;
      }
    }
    return split_vec<F>(vec, n, [&](auto V, size_t first, size_t npacks) -> int {
      constexpr int VEC = decltype(V)::value;
      hipLaunchKernelGGL((k_scalar_mul<F, VEC, true>), dim3(grid_for(npacks)), dim3(BLOCK), 0, S(stream), ctx,
                         dst + first * F::LIMBS, a + first * F::LIMBS, sc, npacks);
      LAUNCH_CHECK();
      return SCL_OK;
    });
  });
}

static int reduce_impl(int field, uint64_t* out_host, const uint64_t* a, const uint64_t* b, size_t n, void* stream,
                       bool is_dot) {
  if (!out_host) return fail(SCL_ERR_BAD_ARG, "out is NULL");
  return with_ring_or_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    if (n == 0) {
      F::st(out_host, F::zero());
      return SCL_OK;
    }
    if (!a || (is_dot && !b)) return fail(SCL_ERR_BAD_ARG, "NULL operand");
    SCL_TRY(check_align<F>({a, b}));
    const int vec = vec_width<F>({a, b}, {});
    // stage 1: one trip per thread where the batch allows (at most 2^20 workgroups, then they grid-stride); stage 2: the
    // same kernel over the partials with at most 1024 workgroups; the host folds those
    const unsigned max1 = 1u << 20, max2 = 1024;
    void* sc;
    SCL_TRY(scratch(((size_t)max1 + 2 + 2 * (size_t)max2) * F::LIMBS * 8 + 64, &sc));
    u64* part1 = static_cast<u64*>(sc);
    unsigned used1 = 0;
    bool staged = false;
    if constexpr (F::TAG == 3) {
      if (is_dot && g_inv_batch.load() >= 0) {  // GF(2^128): the products on per-lane window tables in LDS (k_dot_gf128)
        constexpr int BLK = 256;
        auto kern = &k_dot_gf128<BLK, 3>;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, BLK * 128));
        size_t g = (n + BLK - 1) / BLK;
        if (g > 4096) g = 4096;  // resident workgroups that stride: each partial costs a host-side add
        hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(BLK), BLK * 128, S(stream), part1, a, b, n);
        LAUNCH_CHECK();
        used1 = (unsigned)g;
        staged = true;
      }
    }
    if (!staged) SCL_TRY((split_vec<F>(vec, n, [&](auto V, size_t first, size_t npacks) -> int {
      constexpr int VEC = decltype(V)::value;
      size_t g = (npacks + (size_t)BLOCK * RED_UNROLL - 1) / ((size_t)BLOCK * RED_UNROLL);
      const long cap = g_max_blocks.load();
      if (cap > 0 && g > (size_t)cap) g = (size_t)cap;
      if (g > max1) g = max1;
      if (g == 0) g = 1;
      if (is_dot)
        hipLaunchKernelGGL((k_dot<F, VEC>), dim3((unsigned)g), dim3(BLOCK), 0, S(stream), ctx, part1 + (size_t)used1 * F::LIMBS,
                           a + first * F::LIMBS, b + first * F::LIMBS, npacks);
      else
        hipLaunchKernelGGL((k_sum<F, VEC>), dim3((unsigned)g), dim3(BLOCK), 0, S(stream), ctx, part1 + (size_t)used1 * F::LIMBS,
                           a + first * F::LIMBS, npacks);
      LAUNCH_CHECK();
      used1 += (unsigned)g;
      return SCL_OK;
    })));
    const u64* fold_src = part1;
    unsigned fold_n = used1;
    if (used1 > 2 * max2) {  // second stage: the partials are canonical elements, summed like any vector
      u64* part2 = part1 + ((size_t)max1 + 2) * F::LIMBS;
      size_t g = ((size_t)used1 + (size_t)BLOCK * RED_UNROLL - 1) / ((size_t)BLOCK * RED_UNROLL);
      if (g > max2) g = max2;
      hipLaunchKernelGGL((k_sum<F, 1>), dim3((unsigned)g), dim3(BLOCK), 0, S(stream), ctx, part2, part1, (size_t)used1);
      LAUNCH_CHECK();
      fold_src = part2;
      fold_n = (unsigned)g;
    }
    std::vector<u64> host((size_t)fold_n * F::LIMBS);
    HIP_TRY(hipMemcpyAsync(host.data(), fold_src, host.size() * 8, hipMemcpyDeviceToHost, S(stream)));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    typename F::E tot = F::zero();
    for (unsigned i = 0; i < fold_n; ++i) tot = F::add(ctx, tot, F::ld(host.data() + (size_t)i * F::LIMBS));
    F::st(out_host, tot);
    return SCL_OK;
  });
}

int scl_hip_sum(int field, uint64_t* out_host, const uint64_t* a, size_t n, void* stream) {
  return reduce_impl(field, out_host, a, nullptr, n, stream, false);
}
int scl_hip_dot(int field, uint64_t* out_host, const uint64_t* a, const uint64_t* b, size_t n, void* stream) {
  return reduce_impl(field, out_host, a, b, n, stream, true);
}

#if SCL_TU_COMMON
int scl_hip_equals(int field, int* equal_host, const uint64_t* a, const uint64_t* b, size_t n, void* stream) {
  const int L = scl_hip_limbs(field);
  if (L < 0) return fail(SCL_ERR_BAD_ARG, "unknown field tag");
  if (!equal_host) return fail(SCL_ERR_BAD_ARG, "equal is NULL");
  if (n == 0) {
    *equal_host = 1;
    return SCL_OK;
  }
  if (!a || !b) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  void* sc;
  SCL_TRY(scratch(64, &sc));
  unsigned long long* cnt = static_cast<unsigned long long*>(sc);
  HIP_TRY(hipMemsetAsync(cnt, 0, 8, S(stream)));
  const size_t words = n * (size_t)L;
  if (is_ring(field)) {
    const int K = field - 0x100;
    const u64 lo = K >= 64 ? ~0ull : ((1ull << K) - 1), hi = K >= 128 ? ~0ull : K > 64 ? ((1ull << (K - 64)) - 1) : 0;
    hipLaunchKernelGGL(k_count_diff_masked<>, dim3(grid_for(words)), dim3(BLOCK), 0, S(stream), cnt, a, b, words, lo, hi, L);
  } else {
    hipLaunchKernelGGL(k_count_diff<>, dim3(grid_for(words)), dim3(BLOCK), 0, S(stream), cnt, a, b, words);
  }
  LAUNCH_CHECK();
  unsigned long long h = 0;
  HIP_TRY(hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, S(stream)));
  HIP_TRY(hipStreamSynchronize(S(stream)));
  *equal_host = (h == 0);
  return SCL_OK;
}
#endif  // SCL_TU_COMMON

// ---- randomness ----------------------------------------------------------------------------------------------
#if SCL_TU_COMMON
int scl_hip_prg_blocks(unsigned char* dst, size_t nblocks, const unsigned char* seed, size_t seed_len,
                       uint64_t counter0, void* stream) {
  if (nblocks == 0) return SCL_OK;
  if (!dst) return fail(SCL_ERR_BAD_ARG, "dst is NULL");
  if (!aligned16(dst)) return fail(SCL_ERR_BAD_ARG, "dst not 16-byte aligned");
  AesKey key;
  make_aes_key(seed, seed_len, key);
  aes_key_range(key, (u64)counter0, (u64)nblocks);
  AES4_LAUNCH(k_prg_blocks<>, (nblocks + 3) / 4, S(stream), reinterpret_cast<u64*>(dst), key, (u64)counter0, nblocks);
  LAUNCH_CHECK();
  return SCL_OK;
}
#endif  // SCL_TU_COMMON

// Z2k::read per element at a stride of byteSize (z2k.h:50-52,71-75)
static int ring_from_bytes(int field, uint64_t* dst, const unsigned char* src, size_t n, hipStream_t st) {
  const int bs = (field - 0x100 - 1) / 8 + 1;
  return with_ring_or_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    if constexpr (F::TAG == 5 || F::TAG == 6) {
      SCL_TRY(check_align<F>({dst}));
      hipLaunchKernelGGL((k_ring_from_bytes<F>), dim3(grid_for(n)), dim3(BLOCK), 0, st, ctx, dst, src, n, bs);
      LAUNCH_CHECK();
      return SCL_OK;
    } else {
      return fail(SCL_ERR_BAD_ARG, "not a ring tag");
    }
  });
}

int scl_hip_from_bytes(int field, uint64_t* dst, const unsigned char* src, size_t n, void* stream) {
  if (n == 0) return SCL_OK;
  if (!dst || !src) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  if (is_ring(field)) return ring_from_bytes(field, dst, src, n, S(stream));
  return with_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({dst}));
    hipLaunchKernelGGL((k_from_bytes<F>), dim3(grid_for(n)), dim3(BLOCK), 0, S(stream), ctx, dst, src, n);
    LAUNCH_CHECK();
    return SCL_OK;
  });
}

int scl_hip_vector_random(int field, uint64_t* dst, size_t n, const unsigned char* seed, size_t seed_len,
                          uint64_t counter0, void* stream) {
  if (n == 0) return SCL_OK;
  if (!dst) return fail(SCL_ERR_BAD_ARG, "dst is NULL");
  if (is_ring(field)) {
    // Vector<Z2k>::random (vector.h:507-519): one prg.next(n * byteSize) = ceil(n * byteSize / 16) blocks into a
    // per-thread temporary, then Z2k::read at a stride of byteSize
    const size_t bs = (size_t)(field - 0x100 - 1) / 8 + 1, nblocks = (n * bs + 15) / 16;
    void* tmp = nullptr;
    SCL_TRY(temp_acquire(nblocks * 16, S(stream), &tmp));
    int rc = scl_hip_prg_blocks(static_cast<unsigned char*>(tmp), nblocks, seed, seed_len, counter0, stream);
    if (rc == SCL_OK) rc = ring_from_bytes(field, dst, static_cast<const unsigned char*>(tmp), n, S(stream));
    (void)temp_release(S(stream));
    return rc;
  }
  return with_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({dst}));
    AesKey key;
    make_aes_key(seed, seed_len, key);
    aes_key_range(key, (u64)counter0, ((u64)n * F::LIMBS * 8 + 15) / 16);
    const size_t work = F::LIMBS == 4 ? (n + 1) / 2 : ((F::LIMBS == 1 ? (n + 1) / 2 : n) + 3) / 4;
    if (F::LIMBS == 1 && !aligned16(dst))  // an 8-byte aligned window of a larger vector: element-wise stores
      AES4_LAUNCH((k_vector_random<F, false>), work, S(stream), ctx, dst, key, (u64)counter0, n);
    else
      AES4_LAUNCH((k_vector_random<F, true>), work, S(stream), ctx, dst, key, (u64)counter0, n);
    LAUNCH_CHECK();
    return SCL_OK;
  });
}

// ---- Shamir -----------------------------------------------------------------------------------------------------
int scl_hip_lagrange_basis(int field, uint64_t* lambda_host, const uint64_t* alphas_host, size_t m,
                           const uint64_t* x_host) {
  if (m == 0) return SCL_OK;
  if (!lambda_host) return fail(SCL_ERR_BAD_ARG, "lambda is NULL");
  return with_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    std::vector<typename F::E> nodes, out(m);
    if (alphas_host) load_host<F>(alphas_host, m, nodes);
    else default_nodes<F>(ctx, m, nodes);
    const typename F::E x = x_host ? F::ld(x_host) : F::zero();
    SCL_TRY(lagrange<F>(ctx, nodes, x, out.data()));
    for (size_t i = 0; i < m; ++i) F::st(lambda_host + i * F::LIMBS, out[i]);
    return SCL_OK;
  });
}

static int recover_block(int field, uint64_t* out, const uint64_t* shares, size_t stride, const uint64_t* lambda_host,
                         size_t m, size_t N, const uint64_t* prev, void* stream);

int scl_hip_shamir_recover(int field, uint64_t* out, const uint64_t* shares, size_t stride,
                           const uint64_t* lambda_host, size_t m, size_t N, void* stream) {
  if (N == 0) return SCL_OK;
  if (!out || !shares || !lambda_host) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  if (m == 0) return fail(SCL_ERR_BAD_ARG, "recover: need at least one share");
  if (stride < N) return fail(SCL_ERR_SIZE_MISMATCH, "share_stride < N");
  const int L = scl_hip_limbs(field);
  if (L < 0 || is_ring(field)) return fail(SCL_ERR_BAD_ARG, "unknown field tag");
  // shamirRecoverP takes any number of shares (shamir.h:81-87); one launch holds 256 / limbs Lagrange coefficients.
  // More parties are summed over several launches, each adding its block's terms to the canonical partial sum.
  const size_t cap = (size_t)256 / (size_t)L;
  for (size_t b0 = 0; b0 < m; b0 += cap) {
    const size_t mb = m - b0 < cap ? m - b0 : cap;
    SCL_TRY(recover_block(field, out, shares + b0 * stride * (size_t)L, stride, lambda_host + b0 * (size_t)L, mb, N,
                          b0 ? out : nullptr, stream));
  }
  return SCL_OK;
}

// k_recover_gf128_pos issues its LDS reads from inline assembly one batch ahead of the `s_waitcnt lgkmcnt` that covers them
// (kernels.hpp): between the two the compiler believes the destination registers hold data.  The generated code of this
// toolchain leaves them alone, which is a property of its register allocation, not a guarantee.  So the first GF(2^128)
// reconstruct of a process runs both launch shapes of that kernel once against k_recover_gf128 (compiler-scheduled reads)
// on 4096 pseudo-random secrets; if a result differs, the position-table kernel is never used by this process
// (scl_hip_last_error says so) -- a toolchain change then costs speed, not correctness.  -1 unknown, 1 good, 0 bad.
static std::atomic<int> g_gfpos_state{-1};
static std::mutex g_gfpos_mu;

extern "C++" {
template <class FieldG>
static int launch_gfpos(bool big_block, u64* out, const u64* shares, size_t stride, const BigTable<Gf128>& big, size_t m, size_t N,
                        hipStream_t st) {
  const size_t lds = gfpos_lds_bytes(m);
  if (!big_block) {  // two 512-thread workgroups per CU
    auto kern = &k_recover_gf128_pos<512, 2>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const size_t blocks = (N + 511) / 512;
    hipLaunchKernelGGL(kern, dim3((unsigned)(blocks < 512 ? blocks : 512)), dim3(512), lds, st, out, shares, stride, big, (int)m, N);
  } else {           // one 1024-thread workgroup per CU
    auto kern = &k_recover_gf128_pos<1024, 1>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const size_t blocks = (N + 1023) / 1024;
    hipLaunchKernelGGL(kern, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(1024), lds, st, out, shares, stride, big, (int)m, N);
  }
  LAUNCH_CHECK();
  return SCL_OK;
}

template <class FieldG>
static int gfpos_usable(bool* usable) {
  int stt = g_gfpos_state.load();
  if (stt < 0) {
    std::lock_guard<std::mutex> lk(g_gfpos_mu);
    stt = g_gfpos_state.load();
    if (stt < 0) {
      constexpr size_t N = 4096, M = 7;  // 7 parties: two groups of five, the second one padded
      u64 *sh = nullptr, *o = nullptr;
      HIP_TRY(hipMalloc(&sh, M * N * 16));
      HIP_TRY(hipMalloc(&o, 3 * N * 16));
      auto body = [&]() -> int {
        const unsigned char seed[] = "gfpos self-check";
        SCL_TRY(scl_hip_prg_blocks(reinterpret_cast<unsigned char*>(sh), M * N, seed, sizeof seed - 1, 0, nullptr));
        BigTable<Gf128> big;
        for (size_t i = 0; i < M; ++i) big.v[i] = ((u128)(0x9E3779B97F4A7C15ull * (i + 1)) << 64) | (0xD1B54A32D192ED03ull * (i + 7));
        hipLaunchKernelGGL(k_recover_gf128<FieldG>, dim3(grid_for(N)), dim3(BLOCK), 0, nullptr, o, sh, N, big, (int)M, N, (const u64*)nullptr);
        LAUNCH_CHECK();
        SCL_TRY(launch_gfpos<FieldG>(false, o + 2 * N, sh, N, big, M, N, nullptr));
        SCL_TRY(launch_gfpos<FieldG>(true, o + 4 * N, sh, N, big, M, N, nullptr));
        std::vector<u64> h(6 * N);
        HIP_TRY(hipMemcpy(h.data(), o, h.size() * 8, hipMemcpyDeviceToHost));
        const bool same = std::equal(h.begin(), h.begin() + 2 * N, h.begin() + 2 * N) &&
                          std::equal(h.begin(), h.begin() + 2 * N, h.begin() + 4 * N);
        g_gfpos_state.store(same ? 1 : 0);
        if (!same)
          std::fprintf(stderr, "libscl_hip: k_recover_gf128_pos failed its self-check against k_recover_gf128 on this toolchain; "
                               "GF(2^128) reconstruction uses the shared-shift table kernel instead\n");
        return SCL_OK;
      };
      const int rc = body();
      (void)hipFree(sh);
      (void)hipFree(o);
      if (rc != SCL_OK) return rc;
      stt = g_gfpos_state.load();
    }
  }
  *usable = stt == 1;
  return SCL_OK;
}

}  // extern "C++"

static int recover_block(int field, uint64_t* out, const uint64_t* shares, size_t stride, const uint64_t* lambda_host,
                         size_t m, size_t N, const uint64_t* prev, void* stream) {
  return with_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({out, shares}));
    const int vec = vec_width<F>({out, shares}, {stride});
    const bool nt = g_nontemporal.load() != 0;
    const bool fixed = (m <= FIXED_M_MAX) && (F::TAG <= 1) && !g_force_table.load() && !prev;
    if (m > (size_t)BigTable<F>::CAP) return fail(SCL_ERR_BAD_ARG, "recover: internal party block too large");
    Table<F> lam;
    BigTable<F> big;
    if (fixed) {
      for (size_t i = 0; i < m; ++i) lam.v[i] = F::ld(lambda_host + i * F::LIMBS);
    } else {
      for (size_t i = 0; i < m; ++i) big.v[i] = F::ld(lambda_host + i * F::LIMBS);
    }
    if constexpr (F::TAG == 2 || F::LIMBS == 4) {
      // Montgomery fields: coefficients that are small signed integers (the default nodes' binomials) need no Montgomery product
      SmallLam sl;
      if (!prev && small_lambda<F>(ctx, lambda_host, m, sl))
        return RecoverSmall<F, FIXED_M_MAX>::run(ctx, out, shares, stride, sl, (int)m, N, S(stream));
    }
    if constexpr (F::TAG == 3) {
      const long ft = g_force_table.load();
      if (!ft || ft == 3) {  // GF(2^128): nibble-table kernels ("force_table" 3: the shared-shift form at any m)
        const size_t lds = gfpos_lds_bytes(m);
        bool pos_ok = false;  // the position-table kernel passed its first-use self-check (see gfpos_usable)
        if (!prev && ft != 3 && lds <= 160 * 1024) SCL_TRY(gfpos_usable<F>(&pos_ok));
        if (prev) {
          hipLaunchKernelGGL(k_recover_gf128<F>, dim3(grid_for(N)), dim3(BLOCK), 0, S(stream), out, shares, stride, big,
                             (int)m, N, prev);
        } else if (pos_ok && lds <= 80 * 1024) {  // position tables, two 512-thread workgroups per CU
          SCL_TRY(launch_gfpos<F>(false, out, shares, stride, big, m, N, S(stream)));
        } else if (pos_ok) {  // one 1024-thread workgroup per CU
          SCL_TRY(launch_gfpos<F>(true, out, shares, stride, big, m, N, S(stream)));
        } else {
          hipLaunchKernelGGL(k_recover_gf128<F>, dim3(grid_for(N)), dim3(BLOCK), 0, S(stream), out, shares, stride, big,
                             (int)m, N, (const u64*)nullptr);
        }
        LAUNCH_CHECK();
        return SCL_OK;
      }
    }
    return split_vec<F>(vec, N, [&](auto V, size_t first, size_t npacks) -> int {
      constexpr int VEC = decltype(V)::value;
      u64* o = out + first * F::LIMBS;
      const u64* sh = shares + first * F::LIMBS;
      if (fixed) {
        if constexpr (F::TAG <= 1) {
          SCL_TRY((RecoverFixed<F, FIXED_M_MAX>::template run<VEC>(ctx, o, sh, stride, lam, (int)m, npacks, S(stream))));
        }
      } else if (nt) {
        hipLaunchKernelGGL((k_recover_table<F, VEC, true>), dim3(grid_for(npacks)), dim3(BLOCK), 0, S(stream), ctx, o,
                           sh, stride, big, (int)m, npacks, prev ? prev + first * F::LIMBS : nullptr);
      } else {
        hipLaunchKernelGGL((k_recover_table<F, VEC, false>), dim3(grid_for(npacks)), dim3(BLOCK), 0, S(stream), ctx, o,
                           sh, stride, big, (int)m, npacks, prev ? prev + first * F::LIMBS : nullptr);
      }
      LAUNCH_CHECK();
      return SCL_OK;
    });
  });
}


int scl_hip_shamir_share(int field, uint64_t* shares, size_t share_stride, const uint64_t* secrets,
                         const uint64_t* coeffs, size_t coeff_stride, size_t N, size_t t, size_t n,
                         const uint64_t* alphas_host, void* stream) {
  if (N == 0 || n == 0) return SCL_OK;
  if (!shares || !secrets || (t > 0 && !coeffs)) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  if (share_stride < N || (t > 0 && coeff_stride < N)) return fail(SCL_ERR_SIZE_MISMATCH, "stride < N");
  {
    // shamirSecretShare bounds neither n nor t (shamir.h:51-68).  One launch holds 256 / limbs nodes: more parties go
    // block by block (parties are independent).  More than 48 coefficients go through the chunked Horner kernel.
    const int L = scl_hip_limbs(field);
    if (L < 0 || is_ring(field)) return fail(SCL_ERR_BAD_ARG, "unknown field tag");
    const size_t cap = (size_t)256 / (size_t)L;
    if (n > cap) {
      std::vector<u64> nodes(n * (size_t)L);
      SCL_TRY(nodes_to_host(field, alphas_host, n, 0, nodes.data()));
      for (size_t b0 = 0; b0 < n; b0 += cap)
        SCL_TRY(scl_hip_shamir_share(field, shares + b0 * share_stride * (size_t)L, share_stride, secrets, coeffs, coeff_stride, N,
                                     t, n - b0 < cap ? n - b0 : cap, nodes.data() + b0 * (size_t)L, stream));
      return SCL_OK;
    }
    // (256-bit elements: 49 register-resident coefficients would be 392 registers -- the 48-coefficient Horner kernel spills --
    // so past 16 they take the chunked kernel's 24 at a time)
    // (Mersenne61 with up to 63 coefficients and up to 128 parties is one tile of the matrix-core kernel -- 64 rows of K -- and
    // 5-7 x faster there than chunk by chunk: (128,63) 0.95 against 7.3 ms per 2 * 10^6, profiles/r5_probe_auto_choices.txt)
    const bool one_mfma_tile = field == SCL_M61 && t <= 63 && n <= 128 && n * (t + 1) >= 1024 && g_mfma.load() >= 0 &&
                               !g_force_table.load();
    if ((t > 48 && !one_mfma_tile) || (L == 4 && t > 16))
      return share_chunked(field, shares, share_stride, secrets, coeffs, coeff_stride, N, t, n, alphas_host, 0, stream);
  }
  return with_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({shares, secrets, coeffs}));
    BigTable<F> al;
    SCL_TRY(alpha_table<F>(ctx, alphas_host, n, al));
    BigTable<F> alx;  // what the SMALLX kernels read the nodes from
    const bool smallx = small_nodes<F>(ctx, al, n, alx);
    const int vec = vec_width<F>({shares, secrets, coeffs}, {share_stride, t ? coeff_stride : 0});
    if constexpr (F::TAG == 0) {
      // dense-contraction path on the matrix cores for large (n, t); "mfma" tuning: 1 forces, -1 disables
      const long mode = g_mfma.load();
      SmallVdm probe;
      const bool eligible = n <= 128 && t >= 1 && t <= 63;
      // measured on MI355X (profiles/r1_probe_mfma.txt, r1_probe_share_paths.txt): against Horner the matrix-core
      // path wins from about n*(t+1) >= 512 multiply-adds per secret ((64,21): 3.5x, (128,42): 3.6x); where the
      // blocked small-node kernel applies (t <= 16, small nodes) that one is ahead up to about 1024
      // ((40,13): 2.5 vs 4.0 ms, (64,16): 2.4 vs 2.0 ms)
      BlockVdm bprobe;
      const bool blocked = t <= (size_t)BlockVdm::TMAX && blocked_vandermonde<F>(ctx, al, n, bprobe) != 0;
      const size_t work_min = blocked ? 1024 : 512;
      if (eligible && (mode > 0 || t > 48 || (mode == 0 && n * (t + 1) >= work_min && !small_vandermonde<F>(ctx, al, n, t, probe) &&
                                              !g_force_table.load())))  // (t > 48 arrives here only as one_mfma_tile)
        return share_mfma<F>(al, shares, share_stride, secrets, coeffs, coeff_stride, N, t, n, S(stream));
    }
    if constexpr (F::TAG <= 2 || F::LIMBS == 4) {  // the Mersenne fields and the Montgomery primes (Mont128: of full width)
      SmallVdm sv;
      const bool small = t >= 1 && small_vandermonde<F>(ctx, al, n, t, sv);
      if (small) {
        return split_vec<F>(vec, N, [&](auto V, size_t first, size_t npacks) -> int {
          constexpr int VEC = decltype(V)::value;
          // The Mersenne fields: single-wave workgroups under the residency cap, threshold compiled in (k_share_small_t);
          // "share_waves" 0 or "stream_block" 256 give the 256-thread kernel with the threshold at run time
          const long sw = g_share_waves.load();
          bool launched = false;
          // (Mersenne127: 12 resident waves per CU, 0.39 -> 0.37 ms at C3's size and steadier; Mont128 -- the Barrett fold --
          // the same kernel since round 4, "share_waves128" = the cap of the 16-byte fields, 0 = the 256-thread kernel)
          if constexpr (F::TAG <= 2) {
            const long sw128 = g_share_waves128.load();
            if (sw > 0 && g_stream_block.load() == 64 && (F::LIMBS == 1 || sw128 > 0)) {
              const size_t pad = residency_pad(F::LIMBS == 1 ? sw : sw128, 64, sizeof(u32) * SmallVdm::CAP);
              const dim3 g(grid_for_block(npacks, 64));
#define SST_CASE(TT)                                                                                                     \
  case TT:                                                                                                               \
    hipLaunchKernelGGL((k_share_small_t<F, VEC, TT, 64>), g, dim3(64), pad, S(stream), ctx, shares + first * F::LIMBS, share_stride, \
                       secrets + first * F::LIMBS, coeffs + first * F::LIMBS, coeff_stride, sv, (int)n, npacks);         \
    launched = true;                                                                                                     \
    break;
              switch (t) {
                SST_CASE(1) SST_CASE(2) SST_CASE(3) SST_CASE(4) SST_CASE(5) SST_CASE(6) SST_CASE(7)
                default: break;
              }
#undef SST_CASE
            }
          }
          if constexpr (F::LIMBS == 4) {
            // 32-byte elements: a pair of lanes per secret (k_share_small_pair), single-wave workgroups under the 16-byte
            // fields' residency cap; "share_waves128" 0 = the lane-per-element kernel
            const long sw128 = g_share_waves128.load();
            if (sw128 > 0 && g_stream_block.load() == 64) {
              // (the default cap of the 16-byte fields is 12; the lane pairs run best at 14-16: profiles/r5_probe_f3_waves.txt)
              const size_t pad = residency_pad(sw128 == 12 ? 16 : sw128, 64, sizeof(u32) * SmallVdm::CAP);
              const dim3 g(grid_for_block(2 * npacks, 64));
#define SSP_CASE(TT)                                                                                                     \
  case TT:                                                                                                               \
    hipLaunchKernelGGL((k_share_small_pair<F, TT, 64>), g, dim3(64), pad, S(stream), ctx, shares + first * F::LIMBS, share_stride, \
                       secrets + first * F::LIMBS, coeffs + first * F::LIMBS, coeff_stride, sv, (int)n, npacks);         \
    launched = true;                                                                                                     \
    break;
              switch (t) {
                SSP_CASE(1) SSP_CASE(2) SSP_CASE(3) SSP_CASE(4) SSP_CASE(5) SSP_CASE(6) SSP_CASE(7)
                default: break;
              }
#undef SSP_CASE
            }
          }
          if (!launched)
            hipLaunchKernelGGL((k_share_small<F, VEC>), dim3(grid_for(npacks)), dim3(BLOCK), 0, S(stream), ctx, shares + first * F::LIMBS,
                               share_stride, secrets + first * F::LIMBS, coeffs + first * F::LIMBS, coeff_stride, sv, (int)t, (int)n,
                               npacks);
          LAUNCH_CHECK();
          return SCL_OK;
        });
      }
    }
    if constexpr (F::TAG == 0 || F::TAG == 2 || F::LIMBS == 4) {  // Mersenne127 gains nothing here (its lazy Horner step is already 4 multiplies + a fold)
      BlockVdm bv;
      const int G = (t >= 1 && t <= (size_t)BlockVdm::TMAX) ? blocked_vandermonde<F>(ctx, al, n, bv) : 0;
      if (G) {
        return split_vec<F>(vec, N, [&](auto V, size_t first, size_t npacks) -> int {
          constexpr int VEC = decltype(V)::value;
          u64* sh = shares + first * F::LIMBS;
          const u64* se = secrets + first * F::LIMBS;
          const u64* co = coeffs + first * F::LIMBS;
          const dim3 g(grid_for(npacks)), blk(BLOCK);
#define BLK_LAUNCH(GG) \
  hipLaunchKernelGGL((k_share_blocked<F, VEC, GG>), g, blk, 0, S(stream), ctx, sh, share_stride, se, co, coeff_stride, bv, (int)t, (int)n, npacks)
          if (G == 8) BLK_LAUNCH(8);
          else if (G == 6) BLK_LAUNCH(6);
          else BLK_LAUNCH(4);
#undef BLK_LAUNCH
          LAUNCH_CHECK();
          return SCL_OK;
        });
      }
    }
    if constexpr (F::TAG == 2 || F::LIMBS == 4) {
      if (vdm_eligible<F>(n, t) && !g_force_table.load()) {
        const u64* vdm = nullptr;
        DevPin pin;  // held until the launch below is enqueued
        SCL_TRY((vdm_table<F>(ctx, al, n, t, false, &pin, &vdm)));
        const dim3 g(grid_for(N)), blk(BLOCK);
        if (t <= 4)
          hipLaunchKernelGGL((k_share_vdm<F, 4>), g, blk, 0, S(stream), ctx, shares, share_stride, secrets, coeffs,
                             coeff_stride, vdm, (int)t, (int)n, N);
        else
          hipLaunchKernelGGL((k_share_vdm<F, 16>), g, blk, 0, S(stream), ctx, shares, share_stride, secrets, coeffs,
                             coeff_stride, vdm, (int)t, (int)n, N);
        LAUNCH_CHECK();
        return SCL_OK;
      }
    }
    return split_vec<F>(vec, N, [&](auto V, size_t first, size_t npacks) -> int {
      constexpr int VEC = decltype(V)::value;
      u64* sh = shares + first * F::LIMBS;
      const u64* se = secrets + first * F::LIMBS;
      const u64* co = coeffs ? coeffs + first * F::LIMBS : nullptr;
      const dim3 g(grid_for(npacks)), blk(BLOCK);
#define SHARE_LAUNCH(TREG)                                                                                        \
  do {                                                                                                            \
    if (smallx)                                                                                                   \
      hipLaunchKernelGGL((k_share<F, VEC, TREG, true>), g, blk, 0, S(stream), ctx, sh, share_stride, se, co,      \
                         coeff_stride, alx, (int)t, (int)n, npacks);                                              \
    else                                                                                                          \
      hipLaunchKernelGGL((k_share<F, VEC, TREG, false>), g, blk, 0, S(stream), ctx, sh, share_stride, se, co,     \
                         coeff_stride, al, (int)t, (int)n, npacks);                                               \
  } while (0)
      bool launched = false;
      if constexpr (F::TAG == 3) {
        if (smallx && t >= 5 && t <= 16 && !g_force_table.load()) {  // per-node Horner code ("force_table" 1: the tested-bits form)
          // a few persistent workgroups per CU keep the waves of a CU on the same few nodes' code (4.91 against 5.06 ms at
          // C4's shard size with one pack per thread; profiles/r2_gf128_node_horner.txt)
          const dim3 gg(g_max_blocks.load() > 0 ? g.x : std::min<unsigned>(g.x, 4096u));
          bool default_nodes_in_order = n <= 64 && g_gf_tiles.load() != 0;
          for (size_t i = 0; i < n && default_nodes_in_order; ++i) default_nodes_in_order = (u128)al.v[i] == (u128)(i + 1);
          if (default_nodes_in_order) {  // party i at the bit pattern of i + 1: eight nodes per Horner loop
#define GFT_CASE(TT)                                                                                                          \
  case TT:                                                                                                                    \
    hipLaunchKernelGGL(k_share_gf_tiles<TT>, gg, blk, 0, S(stream), sh, share_stride, se, co, coeff_stride, (int)n, npacks); \
    break;
            switch (t) {
              GFT_CASE(5) GFT_CASE(6) GFT_CASE(7) GFT_CASE(8) GFT_CASE(9) GFT_CASE(10) GFT_CASE(11) GFT_CASE(12) GFT_CASE(13)
              GFT_CASE(14) GFT_CASE(15) GFT_CASE(16)
              default: break;
            }
#undef GFT_CASE
          } else {
            hipLaunchKernelGGL(k_share_gf_nodes<F>, gg, blk, 0, S(stream), sh, share_stride, se, co, coeff_stride, al, (int)t, (int)n,
                               npacks);
          }
          launched = true;
        }
      }
      if (launched) {
      } else if (t <= 4) SHARE_LAUNCH(4);
      else if (t <= 16) SHARE_LAUNCH(16);
      else SHARE_LAUNCH(48);
#undef SHARE_LAUNCH
      LAUNCH_CHECK();
      return SCL_OK;
    });
  });
}

static int share_prg_impl(int field, uint64_t* shares, size_t share_stride, const uint64_t* secrets, size_t N, size_t t,
                          size_t n, const unsigned char* seed, size_t seed_len, uint64_t counter0, ArrayLane lane,
                          void* stream);

// PRG-driven sharing beyond one launch's party / coefficient capacity
// Two-pass PRG-driven sharing: the coefficient rows of a slab of secrets are drawn into arena 1 (k_prg_coeff_rows: AES at
// the rate of the four-table kernel), then shared from there by whatever explicit-coefficient kernel fits the shape.  At most
// 256 MiB of rows per slab; everything is stream-ordered, the arena is kept and grown by the calling thread.
static int share_prg_two_pass(int field, uint64_t* shares, size_t share_stride, const uint64_t* secrets, size_t N, size_t t,
                              size_t n, const unsigned char* seed, size_t seed_len, uint64_t counter0, ArrayLane lane,
                              void* stream) {
  const size_t L = (size_t)scl_hip_limbs(field), E = 8 * L;
  const u64 B = ((u64)(t + 1) * (u64)lane.W * E + 15) / 16;  // blocks per secret
  if (t == 0) return scl_hip_shamir_share(field, shares, share_stride, secrets, nullptr, 0, N, 0, n, nullptr, stream);
  // slab: at most 1 GiB of rows; a whole number of passes of the AES grid (256 workgroups of 1024 lanes: a 3.05-pass slab
  // runs at 76 % of the 4-pass rate) and 8 KiB-aligned slab origins in every row (the matrix-core kernel runs at half speed on
  // share rows that start 32 bytes into a 128-byte line; profiles/r2_prg_two_pass_trace.txt)
  size_t slab = ((size_t)1 << 30) / (t * E);
  const size_t pass = (size_t)AES4_GRID_CAP * ABLOCK;
  if (slab >= pass) slab -= slab % pass;
  else if (slab >= 4096) slab &= ~(size_t)1023;
  else slab = 4096;
  if (slab > N) slab = N;
  const size_t rstride = (slab + 1) & ~(size_t)1;
  void* rows_v = nullptr;
  SCL_TRY(temp_acquire(t * rstride * E, S(stream), &rows_v, 1));
  u64* rows = static_cast<u64*>(rows_v);
  AesKey key;
  make_aes_key(seed, seed_len, key);
  aes_key_range(key, (u64)counter0, (u64)N * B);
  int rc = SCL_OK;
  for (size_t s0 = 0; s0 < N && rc == SCL_OK; s0 += slab) {
    const size_t c = N - s0 < slab ? N - s0 : slab;
    rc = with_field(field, [&](auto f, auto ctx) -> int {
      using F = decltype(f);
      AES4_LAUNCH((k_prg_coeff_rows<F>), c, S(stream), ctx, rows, rstride, key, (u64)(counter0 + s0 * B), (int)t, c, lane);
      LAUNCH_CHECK();
      return SCL_OK;
    });
    if (rc == SCL_OK)
      rc = scl_hip_shamir_share(field, shares + s0 * L, share_stride, secrets + s0 * L, rows, rstride, c, t, n, nullptr, stream);
  }
  (void)temp_release(S(stream), 1);
  return rc;
}

int scl_hip_shamir_share_prg(int field, uint64_t* shares, size_t share_stride, const uint64_t* secrets, size_t N,
                             size_t t, size_t n, const unsigned char* seed, size_t seed_len, uint64_t counter0,
                             void* stream) {
  return share_prg_impl(field, shares, share_stride, secrets, N, t, n, seed, seed_len, counter0, ArrayLane{1, 0}, stream);
}

// shamirSecretShare over math::Array<FF, W> (pedersen.h:138): W interleaved sharings on one PRG draw
int scl_hip_shamir_share_prg_packed(int field, uint64_t* shares, size_t share_stride, const uint64_t* secrets,
                                    size_t secret_stride, size_t N, size_t t, size_t n, size_t width,
                                    const unsigned char* seed, size_t seed_len, uint64_t counter0, void* stream) {
  if (width == 0 || width > 16) return fail(SCL_ERR_BAD_ARG, "share_prg_packed: width must be in 1..16");
  if (N == 0 || n == 0) return SCL_OK;
  if (secret_stride < N) return fail(SCL_ERR_SIZE_MISMATCH, "stride < N");
  const int L = scl_hip_limbs(field);
  if (L < 0) return fail(SCL_ERR_BAD_ARG, "unknown field tag");
  for (size_t j = 0; j < width; ++j)
    SCL_TRY(share_prg_impl(field, shares + j * n * share_stride * (size_t)L, share_stride,
                           secrets + j * secret_stride * (size_t)L, N, t, n, seed, seed_len, counter0,
                           ArrayLane{(int)width, (int)j}, stream));
  return SCL_OK;
}

static int share_prg_impl(int field, uint64_t* shares, size_t share_stride, const uint64_t* secrets, size_t N, size_t t,
                          size_t n, const unsigned char* seed, size_t seed_len, uint64_t counter0, ArrayLane lane,
                          void* stream) {
  if (N == 0 || n == 0) return SCL_OK;
  if (!shares || !secrets) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  if (share_stride < N) return fail(SCL_ERR_SIZE_MISMATCH, "stride < N");
  {
    const int L = scl_hip_limbs(field);
    if (L < 0 || is_ring(field)) return fail(SCL_ERR_BAD_ARG, "unknown field tag");
    const size_t cap = (size_t)256 / (size_t)L;
    // One fused kernel (coefficients drawn into registers, evaluated in place) or two passes (rows drawn into a temporary,
    // then the explicit-coefficient kernel of the shape).  Fused wins where its kernel is small and the evaluation cheap: the
    // small-node kernels of the Mersenne fields (t <= 7) and GF(2^128) up to t = 4.  Beyond that the fused kernels unroll
    // 9 .. 25 AES blocks per lane (100 - 260 KB of code) and evaluate by plain Horner, and the Montgomery fields' fused
    // kernels multiply by full-width Vandermonde entries: (40,13) Mersenne61 3.0 -> 2.1 ms, (128,42) 6.4 -> 1.6 ms,
    // secp256k1 (10,3) 1.9 -> 1.0 ms per 5 * 10^6 (profiles/r2_probe_prg_share.txt).  "prg_two_pass": 1 always, -1 never
    // (where a fused kernel exists).
    const bool must = n > cap || t > 48;  // more parties than one node table / more coefficients than registers hold
    const long pref = g_prg_two_pass.load();
    const bool montgomery = field == SCL_MONT128 || L == 4;
    // (GF(2^128) from t = 5 where the eight-nodes-per-loop kernel shares the drawn rows (default nodes, n <= 64): 1.1-1.9 x the
    // fused kernel at (10..40, 5..11), profiles/r5_probe_auto_choices.txt)
    // (the Mersenne fields: fused only while the small-node kernel applies -- every power n^k, k <= t, below 2^29 as
    // small_vandermonde wants it, i.e. (10, <= 8), (20, <= 6), (40, <= 5) -- the generic fused Horner kernel behind it is 1.2-2.3 x
    // behind the two passes: (40,6) 0.37 against 0.23 ms per 2 * 10^6, profiles/r5_probe_auto_choices.txt)
    // (this entry point shares at the default nodes 1..n only: the largest node, n, decides for all of them)
    const bool small_nodes_fit = small_powers_fit((u128)n, t);
    const bool want = t >= 1 && (montgomery || (field == SCL_GF2_128 ? (t >= 12 || (t >= 5 && n <= 64)) : (t >= 8 || !small_nodes_fit)));
    if (must || pref > 0 || (pref == 0 && want))
      return share_prg_two_pass(field, shares, share_stride, secrets, N, t, n, seed, seed_len, counter0, lane, stream);
  }
  return with_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({shares, secrets}));
    BigTable<F> al;
    SCL_TRY(alpha_table<F>(ctx, nullptr, n, al));
    BigTable<F> alx;  // what the SMALLX kernels read the nodes from
    const bool smallx = small_nodes<F>(ctx, al, n, alx);
    AesKey key;
    make_aes_key(seed, seed_len, key);
    const int vec = vec_width<F>({shares, secrets}, {share_stride});
    const u64 blocks_per_secret = ((u64)(t + 1) * lane.W * F::LIMBS * 8 + 15) / 16;  // ceil((t+1)*W*byteSize/16)
    aes_key_range(key, (u64)counter0, (u64)N * blocks_per_secret);
    if constexpr (F::TAG <= 1) {
      SmallVdm sv;
      if (lane.W == 1 && t >= 1 && small_vandermonde<F>(ctx, al, n, t, sv)) {
        return split_vec<F>(vec, N, [&](auto V, size_t first, size_t npacks) -> int {
          constexpr int VEC = decltype(V)::value;
          const int nblk = F::LIMBS == 1 ? (int)(t / 2 + 1) : (int)t;
          if (t == 3 && g_prg_t3.load() != 0) {  // BASELINE's threshold: the term loop compiled in ("prg_t3" 0: the generic kernel)
            AES4_LAUNCH((k_share_prg_small_t<F, VEC, (F::LIMBS == 1 ? 2 : 3), 3>), npacks, S(stream), shares + first * F::LIMBS,
                        share_stride, secrets + first * F::LIMBS, key, (u64)(counter0 + first * blocks_per_secret), sv, (int)n,
                        npacks);
            LAUNCH_CHECK();
            return SCL_OK;
          }
#define SPS_CASE(NB)                                                                                          \
  case NB:                                                                                                    \
    AES4_LAUNCH((k_share_prg_small<F, VEC, NB>), npacks, S(stream), shares + first * F::LIMBS, share_stride,   \
                secrets + first * F::LIMBS, key, (u64)(counter0 + first * blocks_per_secret), sv, (int)t, (int)n,  \
                npacks);                                                                                      \
    break;
          switch (nblk) {
            SPS_CASE(1) SPS_CASE(2) SPS_CASE(3) SPS_CASE(4)
            default:
              if constexpr (F::LIMBS == 2) {
                switch (nblk) {
                  SPS_CASE(5) SPS_CASE(6) SPS_CASE(7)
                  default: return fail(SCL_ERR_BAD_ARG, "share_prg: internal block count");
                }
              } else {
                return fail(SCL_ERR_BAD_ARG, "share_prg: internal block count");
              }
          }
#undef SPS_CASE
          LAUNCH_CHECK();
          return SCL_OK;
        });
      }
    }
    if constexpr (F::TAG == 2 || F::LIMBS == 4) {
      if (vdm_eligible<F>(n, t) && !g_force_table.load()) {
        const u64* vdm = nullptr;
        DevPin pin;  // held until the launch below is enqueued
        SCL_TRY((vdm_table<F>(ctx, al, n, t, true, &pin, &vdm)));
        const dim3 g(grid_aes(N)), blk(BLOCK);
        if (t <= 4)
          hipLaunchKernelGGL((k_share_prg_vdm<F, 4>), g, blk, 0, S(stream), ctx, shares, share_stride, secrets, key,
                             (u64)counter0, vdm, (int)t, (int)n, N, lane);
        else
          hipLaunchKernelGGL((k_share_prg_vdm<F, 16>), g, blk, 0, S(stream), ctx, shares, share_stride, secrets, key,
                             (u64)counter0, vdm, (int)t, (int)n, N, lane);
        LAUNCH_CHECK();
        return SCL_OK;
      }
    }
    return split_vec<F>(vec, N, [&](auto V, size_t first, size_t npacks) -> int {
      constexpr int VEC = decltype(V)::value;
      u64* sh = shares + first * F::LIMBS;
      const u64* se = secrets + first * F::LIMBS;
#define SHAREP_LAUNCH1(TREG, SX)                                                                                 \
  do {                                                                                                           \
    const BigTable<F>& nodes_ = SX ? alx : al;                                                                   \
    if constexpr (share_prg_four_tables<F, TREG>())                                                              \
      AES4_LAUNCH((k_share_prg<F, VEC, TREG, SX>), npacks, S(stream), ctx, sh, share_stride, se, key,            \
                  (u64)(counter0 + first * blocks_per_secret), nodes_, (int)t, (int)n, npacks, lane);            \
    else                                                                                                         \
      hipLaunchKernelGGL((k_share_prg<F, VEC, TREG, SX>), dim3(grid_aes(npacks)), dim3(BLOCK), 0, S(stream), ctx, \
                         sh, share_stride, se, key, (u64)(counter0 + first * blocks_per_secret), nodes_, (int)t, \
                         (int)n, npacks, lane);                                                                  \
  } while (0)
#define SHAREP_LAUNCH(TREG)                                                                                      \
  do {                                                                                                           \
    if (smallx) SHAREP_LAUNCH1(TREG, true);                                                                      \
    else SHAREP_LAUNCH1(TREG, false);                                                                            \
  } while (0)
      if (t <= 4) SHAREP_LAUNCH(4);
      else if (t <= 16) SHAREP_LAUNCH(16);
      else SHAREP_LAUNCH(48);
#undef SHAREP_LAUNCH
#undef SHAREP_LAUNCH1
      LAUNCH_CHECK();
      return SCL_OK;
    });
  });
}

int scl_hip_shamir_recover_detect(int field, uint64_t* out, unsigned char* status, const uint64_t* shares,
                                  size_t stride, size_t m, size_t N, size_t t, size_t d, const uint64_t* alphas_host,
                                  const uint64_t* x_host, size_t* num_bad_host, void* stream) {
  // shamir.h:122-124: both the shares and the alphas must number at least d+t
  if (m < d + t) return fail(SCL_ERR_NOT_ENOUGH_SHARES, scl_hip_status_message(SCL_ERR_NOT_ENOUGH_SHARES));
  // t = 0 passes that test with m = d, yet the interpolation below takes d + 1 shares and nodes.  The reference reads
  // past the end of both vectors there (alphas.subVector(d + 1) only checks start <= end, shamir.h:127, vector.h:358-363):
  // undefined behaviour, refused here with the text of its range check.
  if (m < d + 1) return fail(SCL_ERR_INVALID_RANGE, scl_hip_status_message(SCL_ERR_INVALID_RANGE));
  if (num_bad_host) *num_bad_host = 0;
  if (N == 0) return SCL_OK;
  if (!out || !status || !shares) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  if (stride < N) return fail(SCL_ERR_SIZE_MISMATCH, "share_stride < N");
  return with_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({out, shares}));
    std::vector<typename F::E> alphas;
    if (alphas_host) load_host<F>(alphas_host, m, alphas);
    else default_nodes<F>(ctx, m, alphas);
    const size_t d1 = d + 1;
    const size_t nchk = (d + t > d1) ? d + t - d1 : 0;  // checks i = d+1 .. d+t-1 (shamir.h:129)
    std::vector<typename F::E> ns(alphas.begin(), alphas.begin() + d1), L((nchk + 1) * d1);
    for (size_t r = 0; r < nchk; ++r) SCL_TRY(lagrange<F>(ctx, ns, alphas[d1 + r], L.data() + r * d1));
    SCL_TRY(lagrange<F>(ctx, ns, x_host ? F::ld(x_host) : F::zero(), L.data() + nchk * d1));
    const size_t rows = nchk + 1;
    if constexpr (F::TAG == 0) {
      // many rows over many shares (large t and d): the rows-times-shares product on the matrix cores, then one compare
      // pass -- 1.26 G secrets/s for the vector-ALU kernel at t = d = 42 (profiles/r2_probe_detect.txt)
      const long mode = g_mfma.load();
      // (few secrets: the contraction still wins from about 30 check rows on -- 0.26 against 0.37 ms at t = 42, profiles/r5_probe_auto_choices.txt)
      if (rows <= 128 && d1 >= 2 && d1 <= 64 && (mode > 0 || (mode == 0 && rows * d1 >= 512 && (N >= 4096 || rows * d1 >= 900)))) {
        void* sc;
        SCL_TRY(scratch(64, &sc));
        unsigned long long* cnt = static_cast<unsigned long long*>(sc);
        HIP_TRY(hipMemsetAsync(cnt, 0, 8, S(stream)));
        std::vector<u64> Lw(L.begin(), L.end());
        SCL_TRY(detect_mfma<F>(out, status, shares, stride, Lw, rows, d1, N, cnt, S(stream)));
        unsigned long long h = 0;
        HIP_TRY(hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, S(stream)));
        HIP_TRY(hipStreamSynchronize(S(stream)));
        if (num_bad_host) *num_bad_host = (size_t)h;
        if (h) return fail(SCL_ERR_ERROR_DETECTED, scl_hip_status_message(SCL_ERR_ERROR_DETECTED));
        return SCL_OK;
      }
    }
    // device image of L as prepared constants: [row block][k][RB], zero rows pad the last block (k_recover_detect)
    typedef typename F::KC KC;
    // rows per pass: the 256-bit field's accumulators are 24 registers each (two rows), the others take four or eight
    // (sixteen rows per pass with one secret per lane was measured for Mersenne61: no faster -- the kernel is
    // bound by multiply issue, not by the re-reads)
    const int RBsel = F::LIMBS >= 4 ? 2 : rows <= 4 ? 4 : 8;
    const size_t nblk = (rows + RBsel - 1) / RBsel;
    std::vector<KC> Lk(nblk * d1 * RBsel, F::kc_make(ctx, F::zero()));
    for (size_t r = 0; r < rows; ++r)
      for (size_t k = 0; k < d1; ++k) Lk[((r / RBsel) * d1 + k) * RBsel + r % RBsel] = F::kc_make(ctx, L[r * d1 + k]);
    const size_t tbytes = Lk.size() * sizeof(KC), lbytes = d1 * RBsel * sizeof(KC);
    if (lbytes > 150 * 1024) return fail(SCL_ERR_BAD_ARG, "recover_detect: one row block of the (t)(d+1) table exceeds LDS");
    void* sc;
    SCL_TRY(scratch(tbytes + 64, &sc));
    unsigned long long* cnt = static_cast<unsigned long long*>(sc);
    KC* L_dev = reinterpret_cast<KC*>(static_cast<unsigned char*>(sc) + 64);
    HIP_TRY(hipMemsetAsync(cnt, 0, 8, S(stream)));
    HIP_TRY(hipMemcpyAsync(L_dev, Lk.data(), tbytes, hipMemcpyHostToDevice, S(stream)));
    const int vec = vec_width<F>({out, shares}, {stride});
    SCL_TRY((split_vec<F>(vec, N, [&](auto V, size_t first, size_t npacks) -> int {
      constexpr int VEC = decltype(V)::value;
      auto launch = [&](auto RBc) -> int {
        constexpr int RB = decltype(RBc)::value;
        auto kern = &k_recover_detect<F, VEC, RB>;
        if (lbytes > 48 * 1024)
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lbytes));
        const size_t blocks = (npacks + BLOCK - 1) / BLOCK;  // one pack per thread (the kernel has barriers)
        if (blocks > 0x7fffffffull) return fail(SCL_ERR_BAD_ARG, "recover_detect: batch too large for one launch");
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BLOCK), lbytes, S(stream), ctx, out + first * F::LIMBS,
                           status + first, shares + first * F::LIMBS, stride, L_dev, (int)d1, (int)nchk, npacks, cnt);
        LAUNCH_CHECK();
        return SCL_OK;
      };
      if constexpr (F::LIMBS >= 4) {
        return launch(std::integral_constant<int, 2>{});
      } else {
        if (RBsel == 4) return launch(std::integral_constant<int, 4>{});
        return launch(std::integral_constant<int, 8>{});
      }
    })));
    unsigned long long h = 0;
    HIP_TRY(hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, S(stream)));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    if (num_bad_host) *num_bad_host = (size_t)h;
    if (h) return fail(SCL_ERR_ERROR_DETECTED, scl_hip_status_message(SCL_ERR_ERROR_DETECTED));
    return SCL_OK;
  });
}

// Batched shamirRecoverC (Berlekamp-Welch, shamir.h:202-259).
int scl_hip_shamir_recover_correct(int field, uint64_t* f_out, size_t f_stride, uint64_t* e_out, size_t e_stride,
                                   unsigned char* status, unsigned* nerr, const uint64_t* shares, size_t stride,
                                   size_t m, size_t N, const uint64_t* alphas_host, size_t* num_queued_host,
                                   size_t* num_failed_host, void* stream) {
  if (num_queued_host) *num_queued_host = 0;
  if (num_failed_host) *num_failed_host = 0;
  if (m == 0) return fail(SCL_ERR_BAD_ARG, "recover_correct: need at least one share");
  if (N == 0) return SCL_OK;
  if (!f_out || !e_out || !status || !nerr || !shares) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  if (stride < N || f_stride < N || e_stride < N) return fail(SCL_ERR_SIZE_MISMATCH, "stride < N");
  if (N > 0xFFFFFFFFull) return fail(SCL_ERR_BAD_ARG, "recover_correct: at most 2^32 - 1 secrets per call");
  const size_t t = (m - 1) / 3, n = 3 * t + 1, d1 = t + 1, nchk = n - d1;  // shamir.h:205-206: the first 3t+1 shares
  return with_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    typedef typename F::E E;
    SCL_TRY(check_align<F>({f_out, e_out, shares}));
    std::vector<E> alphas;
    if (alphas_host) load_host<F>(alphas_host, n, alphas);
    else default_nodes<F>(ctx, n, alphas);
    alphas.resize(n);
    // L = [ nchk rows: share d1+r from the first d1 | d1 rows: coefficient k of the interpolant ]
    std::vector<E> ns(alphas.begin(), alphas.begin() + d1), L((nchk + d1) * d1);
    for (size_t r = 0; r < nchk; ++r) SCL_TRY(lagrange<F>(ctx, ns, alphas[d1 + r], L.data() + r * d1));
    {
      // coefficients of l_i(x) = prod_{j != i} (x - a_j) / (a_i - a_j): master polynomial P = prod (x - a_j), then
      // P / (x - a_i) by synthetic division, scaled by 1 / P'(a_i)
      std::vector<E> P(d1 + 1, F::zero()), q(d1);
      P[0] = F::one(ctx);
      for (size_t j = 0; j < d1; ++j) {  // multiply by (x - a_j)
        for (size_t k = j + 1; k > 0; --k) P[k] = F::sub(ctx, P[k - 1], F::mul(ctx, P[k], ns[j]));
        P[0] = F::neg(ctx, F::mul(ctx, P[0], ns[j]));
      }
      for (size_t i = 0; i < d1; ++i) {
        E den = F::one(ctx);
        for (size_t j = 0; j < d1; ++j) {
          if (j == i) continue;
          const E d = F::sub(ctx, ns[i], ns[j]);
          if (F::is_zero(d)) return fail(SCL_ERR_ZERO_INVERSE, scl_hip_status_message(SCL_ERR_ZERO_INVERSE));
          den = F::mul(ctx, den, d);
        }
        const E dinv = F::inv(ctx, den);
        E carry = P[d1];  // = 1
        for (size_t k = d1; k > 0; --k) {
          q[k - 1] = carry;
          carry = F::add(ctx, P[k - 1], F::mul(ctx, carry, ns[i]));
        }
        for (size_t k = 0; k < d1; ++k) L[(nchk + k) * d1 + i] = F::mul(ctx, q[k], dinv);
      }
    }
    // duplicates anywhere among the n nodes would make every system singular
    for (size_t i = 0; i < n; ++i)
      for (size_t j = i + 1; j < n; ++j)
        if (F::eq(alphas[i], alphas[j])) return fail(SCL_ERR_ZERO_INVERSE, scl_hip_status_message(SCL_ERR_ZERO_INVERSE));
    // device image of L as prepared constants, [row block][k][RB], zero rows padding the last block
    typedef typename F::KC KC;
    constexpr int RB = F::LIMBS >= 4 ? 2 : 8;
    const size_t rows = nchk + d1, nblk = (rows + RB - 1) / RB;
    std::vector<KC> Lk(nblk * d1 * RB, F::kc_make(ctx, F::zero()));
    for (size_t r = 0; r < rows; ++r)
      for (size_t k = 0; k < d1; ++k) Lk[((r / RB) * d1 + k) * RB + r % RB] = F::kc_make(ctx, L[r * d1 + k]);
    const size_t tbytes = (Lk.size() * sizeof(KC) + 63) / 64 * 64, lbytes = d1 * RB * sizeof(KC);
    const size_t solve_bytes = bw_lds_elems(n) * sizeof(E), nbytes = n * sizeof(E);
    if (lbytes > 150 * 1024) return fail(SCL_ERR_BAD_ARG, "recover_correct: the interpolation rows for this many shares exceed LDS");
    // the systems of one secret live in the workgroup's LDS while they fit (n <= 136 / 95 / 66 shares by element size) and
    // in a slice of device memory per workgroup beyond that
    const bool in_lds = solve_bytes <= 150 * 1024;
    const size_t max_groups = in_lds ? 8192 : std::max<size_t>(1, std::min<size_t>(1024, ((size_t)512 << 20) / solve_bytes));
    const size_t groups_cap = std::min<size_t>(N, max_groups);
    void* sc;
    SCL_TRY(scratch(tbytes + nbytes + 64, &sc));
    unsigned* counters = static_cast<unsigned*>(sc);  // [0] queued, [1] failed
    KC* L_dev = reinterpret_cast<KC*>(static_cast<unsigned char*>(sc) + 64);
    u64* nodes_dev = reinterpret_cast<u64*>(static_cast<unsigned char*>(sc) + 64 + tbytes);
    // the queue of inconsistent secrets (and the out-of-LDS work area) lives in the per-thread temporary (kept and grown):
    // a hipMalloc / hipFree pair per call costs more than the consistency pass of a few million secrets
    const size_t qbytes = (N * sizeof(unsigned) + 255) / 256 * 256;
    void* queue = nullptr;
    SCL_TRY(temp_acquire(qbytes + (in_lds ? 0 : groups_cap * solve_bytes), S(stream), &queue));
    u64* work = in_lds ? nullptr : reinterpret_cast<u64*>(static_cast<unsigned char*>(queue) + qbytes);
    auto body = [&]() -> int {
      HIP_TRY(hipMemsetAsync(counters, 0, 8, S(stream)));
      HIP_TRY(hipMemcpyAsync(L_dev, Lk.data(), Lk.size() * sizeof(KC), hipMemcpyHostToDevice, S(stream)));
      HIP_TRY(hipMemcpyAsync(nodes_dev, alphas.data(), nbytes, hipMemcpyHostToDevice, S(stream)));
      auto kern = &k_bw_consistent<F, RB>;
      if (lbytes > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lbytes));
      // one secret per thread (the kernel has barriers); N < 2^32 was checked above
      hipLaunchKernelGGL(kern, dim3((unsigned)((N + BLOCK - 1) / BLOCK)), dim3(BLOCK), lbytes, S(stream), ctx, f_out,
                         f_stride, e_out, e_stride, status, nerr, shares, stride, L_dev, (int)d1, (int)nchk, N,
                         static_cast<unsigned*>(queue), counters);
      LAUNCH_CHECK();
      unsigned h[2] = {0, 0};
      HIP_TRY(hipMemcpyAsync(h, counters, 8, hipMemcpyDeviceToHost, S(stream)));
      HIP_TRY(hipStreamSynchronize(S(stream)));
      if (num_queued_host) *num_queued_host = h[0];
      if (h[0] == 0) return SCL_OK;
      const size_t lds = in_lds ? solve_bytes : 0;
      if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bw_solve<F>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const unsigned grid = (unsigned)std::min<size_t>(h[0], groups_cap);
      const unsigned threads = (unsigned)std::min<size_t>(1024, (n + BW_WAVE - 1) / BW_WAVE * BW_WAVE);  // threads over rows
      hipLaunchKernelGGL((k_bw_solve<F>), dim3(grid), dim3(threads), lds, S(stream), ctx, f_out, f_stride, e_out, e_stride,
                         status, nerr, shares, stride, nodes_dev, (int)n, static_cast<const unsigned*>(queue), h[0],
                         counters + 1, work);
      LAUNCH_CHECK();
      HIP_TRY(hipMemcpyAsync(h, counters, 8, hipMemcpyDeviceToHost, S(stream)));
      HIP_TRY(hipStreamSynchronize(S(stream)));
      if (num_failed_host) *num_failed_host = h[1];
      return SCL_OK;
    };
    const int rc = body();
    (void)temp_release(S(stream));
    return rc;
  });
}

// ---- additive ------------------------------------------------------------------------------------------------------
int scl_hip_additive_share(int field, uint64_t* shares, size_t share_stride, const uint64_t* secrets,
                           const uint64_t* rnd, size_t rnd_stride, size_t N, size_t n, void* stream) {
  if (n == 0) return fail(SCL_ERR_BAD_ARG, "additive share: n must be >= 1");  // additive.h:46 underflows
  if (N == 0) return SCL_OK;
  if (!shares || !secrets || (n > 1 && !rnd)) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  if (share_stride < N || (n > 1 && rnd_stride < N)) return fail(SCL_ERR_SIZE_MISMATCH, "stride < N");
  return with_ring_or_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({shares, secrets, rnd}));
    const int vec = vec_width<F>({shares, secrets, rnd}, {share_stride, n > 1 ? rnd_stride : 0});
    return split_vec<F>(vec, N, [&](auto V, size_t first, size_t npacks) -> int {
      constexpr int VEC = decltype(V)::value;
      hipLaunchKernelGGL((k_additive_share<F, VEC>), dim3(grid_for(npacks)), dim3(BLOCK), 0, S(stream), ctx,
                         shares + first * F::LIMBS, share_stride, secrets + first * F::LIMBS,
                         rnd ? rnd + first * F::LIMBS : nullptr, rnd_stride, (int)n, npacks);
      LAUNCH_CHECK();
      return SCL_OK;
    });
  });
}

int scl_hip_additive_share_prg(int field, uint64_t* shares, size_t share_stride, const uint64_t* secrets, size_t N,
                               size_t n, const unsigned char* seed, size_t seed_len, uint64_t counter0,
                               void* stream) {
  if (n == 0) return fail(SCL_ERR_BAD_ARG, "additive share: n must be >= 1");
  if (N == 0) return SCL_OK;
  if (!shares || !secrets) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  if (share_stride < N) return fail(SCL_ERR_SIZE_MISMATCH, "stride < N");
  return with_ring_or_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({shares, secrets}));
    AesKey key;
    make_aes_key(seed, seed_len, key);
    aes_key_range(key, (u64)counter0, (u64)N * (n - 1) * ((F::LIMBS * 8 + 15) / 16));
    const int vec = vec_width<F>({shares, secrets}, {share_stride});
    return split_vec<F>(vec, N, [&](auto V, size_t first, size_t npacks) -> int {
      constexpr int VEC = decltype(V)::value;
      AES4_LAUNCH((k_additive_share_prg<F, VEC>), npacks, S(stream), ctx, shares + first * F::LIMBS, share_stride,
                  secrets + first * F::LIMBS, key, (u64)(counter0 + first * (n - 1) * ((F::LIMBS * 8 + 15) / 16)), (int)n,
                  npacks);
      LAUNCH_CHECK();
      return SCL_OK;
    });
  });
}

int scl_hip_additive_recover(int field, uint64_t* out, const uint64_t* shares, size_t stride, size_t n, size_t N,
                             void* stream) {
  if (N == 0) return SCL_OK;
  if (!out || !shares) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  if (stride < N) return fail(SCL_ERR_SIZE_MISMATCH, "share_stride < N");
  return with_ring_or_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({out, shares}));
    const int vec = vec_width<F>({out, shares}, {stride});
    const bool nt = g_nontemporal.load() != 0;
    return split_vec<F>(vec, N, [&](auto V, size_t first, size_t npacks) -> int {
      constexpr int VEC = decltype(V)::value;
      if (nt)
        hipLaunchKernelGGL((k_additive_recover<F, VEC, true>), dim3(grid_for(npacks)), dim3(BLOCK), 0, S(stream), ctx,
                           out + first * F::LIMBS, shares + first * F::LIMBS, stride, (int)n, npacks);
      else
        hipLaunchKernelGGL((k_additive_recover<F, VEC, false>), dim3(grid_for(npacks)), dim3(BLOCK), 0, S(stream), ctx,
                           out + first * F::LIMBS, shares + first * F::LIMBS, stride, (int)n, npacks);
      LAUNCH_CHECK();
      return SCL_OK;
    });
  });
}

// ---- matrices --------------------------------------------------------------------------------------------------------
int scl_hip_vandermonde(int field, uint64_t* V_dev, size_t n, size_t m, const uint64_t* xs_host, void* stream) {
  if (n == 0 || m == 0) return SCL_OK;
  if (!V_dev) return fail(SCL_ERR_BAD_ARG, "V is NULL");
  return with_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({V_dev}));
    std::vector<typename F::E> xs;
    if (xs_host) load_host<F>(xs_host, n, xs);
    else default_nodes<F>(ctx, n, xs);
    // V(i,0) = 1, V(i,j) = V(i,j-1) * xs[i]  (matrix.h:444-460)
    std::vector<u64> host(n * m * F::LIMBS);
    for (size_t i = 0; i < n; ++i) {
      typename F::E v = F::one(ctx);
      for (size_t j = 0; j < m; ++j) {
        if (j) v = F::mul(ctx, v, xs[i]);
        F::st(host.data() + (i * m + j) * F::LIMBS, v);
      }
    }
    HIP_TRY(hipMemcpyAsync(V_dev, host.data(), host.size() * 8, hipMemcpyHostToDevice, S(stream)));
    HIP_TRY(hipStreamSynchronize(S(stream)));  // host staging buffer dies with this frame
    return SCL_OK;
  });
}

int scl_hip_matmul(int field, uint64_t* C, size_t ldc, const uint64_t* A, size_t lda, const uint64_t* B, size_t ldb,
                   size_t M, size_t K, size_t N, void* stream) {
  if (M == 0 || N == 0) return SCL_OK;
  if (!C || (K && (!A || !B))) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  if (ldc < N || (K && (lda < K || ldb < N))) return fail(SCL_ERR_MATMUL_DIMS, scl_hip_status_message(SCL_ERR_MATMUL_DIMS));
  // No bound on M, K or N, as in matrix.h:477-513.  Paths (DESIGN.md section 3 has their rates):
  //   one column (Matrix::multiply(Vector), :497-513)           k_matvec: a wavefront per row
  //   Mersenne61, all three dimensions sizeable                  matrix cores, general kernel (gemm_mfma.hpp): digit planes of both
  //                                                             factors, K looped in the kernel
  //   Mersenne61, long right factor, M*K >= 512                 matrix cores: row blocks of 128, k-chunks of 64 (the chunks
  //                                                             after the first add to C in the kernel's epilogue)
  //   K <= 16 (8 for 32-byte elements), long right factor       k_matmul_thin: B's rows in registers, A in LDS, 16-byte accesses
  //   left factor within 48 KiB and a long right factor          k_matmul: the left factor in LDS, a thread per column
  //   anything else                                              k_matmul_tiled: LDS tiles of both factors, K in steps
  return with_ring_or_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({C, A, B}));
    if (K == 0) {  // an empty sum: the zero matrix
      for (size_t i = 0; i < M; ++i) HIP_TRY(hipMemsetAsync(C + i * ldc * F::LIMBS, 0, N * F::LIMBS * 8, S(stream)));
      return SCL_OK;
    }
    if (N == 1) {
      const size_t blocks = (M + BLOCK / 64 - 1) / (BLOCK / 64);
      hipLaunchKernelGGL((k_matvec<F>), dim3(grid_for_block(blocks, 1)), dim3(BLOCK), 0, S(stream), ctx, C, ldc, A, lda, B, ldb, M, K);
      LAUNCH_CHECK();
      return SCL_OK;
    }
    if constexpr (F::TAG == 0) {
      const long mode = g_mfma.load();
      // (a (row block, k-chunk) launch needs ~10^5 columns to outweigh its launches: 4096^3 runs at 1.5 T multiply-adds/s
      // this way and at 3.3 through k_matmul_tiled, profiles/r5_probe_matmul.txt)
      const bool one_tile = M <= 128 && K <= 64;
      // beyond one tile: the general kernel (gemm_mfma.hpp) wherever K > 64 and both outer dimensions reach a tile -- its three
      // launches take 23 us against 49 us and more for the vector-ALU kernels on every small shape tried, 2-12 x ahead from
      // 33 x 65 x 33 up (profiles/r5_probe_matmul_paths.txt); its digit planes are a temporary of at most 2 GiB, longer factors
      // go slab by slab.  One or two k-chunks against a long right factor (64 < K <= 128, N >= 24576) stay on the sharing
      // kernels (no pass over B: 24-31 % ahead there); K <= 64 is one k-chunk: row blocks on the sharing kernels, from 131072
      // columns (below that the vector-ALU kernel is ahead).  "mfma" 2 forces the (row block, k-chunk) form for A/B runs.
      const bool two_chunks_long = K > 64 && K <= 128 && N >= 24576;
      const bool gemm_ok = K > 64 && M >= 33 && N >= 33;
      if (gemm_ok && mode != 2 && (mode > 0 || (mode == 0 && !two_chunks_long)))
        return matmul_gemm_mfma<F>(C, ldc, A, lda, B, ldb, M, K, N, S(stream));
      // (one tile: two launches, 17-20 us whatever the shape, where the vector-ALU kernels need 30 us and more from 2^18
      // multiply-adds on -- (100 x 43)(43 x 3000) 0.017 against 0.33 ms)
      const bool one_tile_pays = one_tile && (N >= 4096 || M * K * N >= ((size_t)1 << 18));
      if (mode > 0 || (mode == 0 && std::min<size_t>(M, 128) * std::min<size_t>(K, 64) >= 512 &&
                       (one_tile ? one_tile_pays : N >= (two_chunks_long ? 24576u : 131072u))))
        return matmul_mfma_blocks<F>(ctx, C, ldc, A, lda, B, ldb, M, K, N, S(stream));
    }
    const size_t esz = F::LIMBS * 8;
    {
      // a thin inner dimension (a Vandermonde matrix times the coefficient rows): B's K rows in registers, 16-byte accesses
      constexpr int KMAX = F::LIMBS == 4 ? 8 : 16;
      // (a thread per pack of columns loops over the M rows: from 1024 columns for a tiny left factor, from 65536 otherwise --
      // below that the tiled kernel's workgroups fill the chip and it is 2-15 x ahead: (40 x 14)(14 x 3000) over secp256k1 0.024
      // against 0.26 ms, profiles/r5_probe_auto_choices.txt; "matmul_lds_min" pins the bound for tests and A/B runs)
      const long thin_min_knob = g_matmul_lds_min.load();
      const size_t thin_min = thin_min_knob > 0 ? (size_t)thin_min_knob : M * K <= 64 ? 1024 : 65536;
      if (K <= (size_t)KMAX && M * K * esz <= 48 * 1024 && N >= thin_min && g_force_table.load() == 0) {
        const int vec = vec_width<F>({C, B}, {ldc, ldb});
        return split_vec<F>(vec, N, [&](auto V, size_t first, size_t npacks) -> int {
          constexpr int VEC = decltype(V)::value;
          hipLaunchKernelGGL((k_matmul_thin<F, VEC, KMAX>), dim3(grid_for(npacks)), dim3(BLOCK), M * K * esz, S(stream), ctx,
                             C + first * F::LIMBS, ldc, A, lda, B + first * F::LIMBS, ldb, (int)M, (int)K, npacks);
          LAUNCH_CHECK();
          return SCL_OK;
        });
      }
    }
    const long lds_min_knob = g_matmul_lds_min.load();
    const size_t lds_min = lds_min_knob > 0 ? (size_t)lds_min_knob : 131072;  // (a thread per column: level with the tiled kernel at 10^5 columns, ahead beyond)
    if (M * K * esz <= 48 * 1024 && N >= lds_min) {
      const unsigned gx = grid_for(N);
      hipLaunchKernelGGL((k_matmul<F, 4>), dim3(gx, 1), dim3(BLOCK), M * K * esz, S(stream), ctx, C, ldc, A, lda, B, ldb, (int)M,
                         (int)K, N, (int)M);
      LAUNCH_CHECK();
      return SCL_OK;
    }
    typedef MatmulShape<F> SH;
    const size_t tiles = ((M + SH::TM - 1) / SH::TM) * ((N + SH::TN - 1) / SH::TN);
    // few output tiles and a long inner dimension: slices of K go to workgroups of their own (at least 256 columns each, about
    // 2048 workgroups in all), their partial products to a temporary that one Vector::sum per entry adds up
    size_t split = 1;
    if (tiles < 1024 && K >= 512 && ldc == N) {
      split = std::min<size_t>(K / 256, (2048 + tiles - 1) / tiles);
      if (split > 65535) split = 65535;
    }
    if (split > 1) {
      const size_t kslice = ((K + split - 1) / split + SH::TK - 1) / SH::TK * SH::TK;
      split = (K + kslice - 1) / kslice;
      const size_t slice_elems = (M * N + 1) & ~(size_t)1;  // (16-byte aligned slices)
      void* tmp = nullptr;
      SCL_TRY(temp_acquire(split * slice_elems * esz, S(stream), &tmp));
      u64* part = static_cast<u64*>(tmp);
      hipLaunchKernelGGL((k_matmul_tiled<F>), dim3(grid_for_block(tiles, 1), (unsigned)split), dim3(BLOCK), 0, S(stream), ctx, part, N, A, lda,
                         B, ldb, M, K, N, kslice, slice_elems * F::LIMBS);
      int rc = hipGetLastError() == hipSuccess ? SCL_OK : fail(SCL_ERR_HIP, "matmul: launch failed");
      if (rc == SCL_OK) rc = scl_hip_additive_recover(field, C, part, slice_elems, split, M * N, stream);
      (void)temp_release(S(stream));
      return rc;
    }
    hipLaunchKernelGGL((k_matmul_tiled<F>), dim3(grid_for_block(tiles, 1)), dim3(BLOCK), 0, S(stream), ctx, C, ldc, A, lda, B, ldb, M, K, N,
                       K, (size_t)0);
    LAUNCH_CHECK();
    return SCL_OK;
  });
}

// ---- layout ------------------------------------------------------------------------------------------------------------
static int transpose_impl(int field, uint64_t* dst, const uint64_t* src, size_t stride, size_t N, size_t n,
                          void* stream, bool to_soa) {
  const int L = scl_hip_limbs(field);
  if (L < 0) return fail(SCL_ERR_BAD_ARG, "unknown field tag");
  if (N == 0 || n == 0) return SCL_OK;
  if (!dst || !src) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  if (stride < N) return fail(SCL_ERR_SIZE_MISMATCH, "stride < N");
  // 16-byte accesses on both sides (k_transpose16) where the layout allows: aligned bases, an even row stride for one-limb
  // fields, and a tile of at least 64 secrets within 40 KiB of LDS ("force_scalar" 1: the 8-byte kernel, for A/B runs)
  {
    size_t t16 = (40 * 1024) / (n * (size_t)L * 8);
    const long want_tile = g_transpose_tile.load();
    if (t16 > (want_tile > 0 ? (size_t)want_tile : 512)) t16 = want_tile > 0 ? (size_t)want_tile : 512;
    for (size_t pw = 512; pw >= 64; pw >>= 1)  // a power of two: the kernel splits its flat (party, piece) index with a shift
      if (t16 >= pw) {
        t16 = pw;
        break;
      }
    if (t16 < 64) t16 = 0;
    if (t16 >= 64 && aligned16(dst) && aligned16(src) && (L != 1 || (stride & 1) == 0) && !g_force_scalar.load()) {
      const size_t ntiles16 = (N + t16 - 1) / t16;
      const unsigned g16 = (unsigned)(ntiles16 < (1u << 20) ? ntiles16 : (1u << 20));
      const size_t shmem16 = t16 * n * (size_t)L * 8;
#define TR16_LAUNCH(LL, SOA) \
  hipLaunchKernelGGL((k_transpose16<LL, SOA>), dim3(g16), dim3(BLOCK), shmem16, S(stream), dst, src, stride, N, (int)n, (int)t16)
      if (L == 1) {
        if (to_soa) TR16_LAUNCH(1, true);
        else TR16_LAUNCH(1, false);
      } else if (L == 2) {
        if (to_soa) TR16_LAUNCH(2, true);
        else TR16_LAUNCH(2, false);
      } else {
        if (to_soa) TR16_LAUNCH(4, true);
        else TR16_LAUNCH(4, false);
      }
#undef TR16_LAUNCH
      LAUNCH_CHECK();
      return SCL_OK;
    }
  }
  size_t tile = (32 * 1024) / (n * (size_t)L * 8);
  if (tile == 0) return fail(SCL_ERR_BAD_ARG, "transpose: n too large");
  if (tile > 1024) tile = 1024;
  if (tile >= 64) tile &= ~(size_t)63;
  const size_t ntiles = (N + tile - 1) / tile;
  const unsigned g = (unsigned)(ntiles < 8192 ? ntiles : 8192);
  const size_t shmem = tile * n * (size_t)L * 8;
#define TR_LAUNCH(LL, SOA) \
  hipLaunchKernelGGL((k_transpose<LL, SOA>), dim3(g), dim3(BLOCK), shmem, S(stream), dst, src, stride, N, (int)n, (int)tile)
  if (L == 1) {
    if (to_soa) TR_LAUNCH(1, true);
    else TR_LAUNCH(1, false);
  } else if (L == 2) {
    if (to_soa) TR_LAUNCH(2, true);
    else TR_LAUNCH(2, false);
  } else {
    if (to_soa) TR_LAUNCH(4, true);
    else TR_LAUNCH(4, false);
  }
#undef TR_LAUNCH
  LAUNCH_CHECK();
  return SCL_OK;
}

int scl_hip_aos_to_soa(int field, uint64_t* soa, size_t stride, const uint64_t* aos, size_t N, size_t n, void* stream) {
  return transpose_impl(field, soa, aos, stride, N, n, stream, true);
}
int scl_hip_soa_to_aos(int field, uint64_t* aos, const uint64_t* soa, size_t stride, size_t N, size_t n, void* stream) {
  return transpose_impl(field, aos, soa, stride, N, n, stream, false);
}

// ---- wire image ------------------------------------------------------------------------------------------------------------
#if SCL_TU_COMMON
size_t scl_hip_wire_size(int field, size_t n) {
  const int L = scl_hip_limbs(field);
  return L < 0 ? 0 : 4 + n * (size_t)L * 8;
}
#endif  // SCL_TU_COMMON

static int wire_pack_impl(int field, unsigned char* dst, const uint64_t* src, size_t n, bool framed, void* stream);
int scl_hip_wire_pack(int field, unsigned char* dst, const uint64_t* src, size_t n, void* stream) {
  return wire_pack_impl(field, dst, src, n, false, stream);
}
#if SCL_TU_COMMON
size_t scl_hip_frame_size(size_t image_bytes) { return 4 + image_bytes; }
#endif  // SCL_TU_COMMON
int scl_hip_frame_pack(int field, unsigned char* dst, const uint64_t* src, size_t n, void* stream) {
  return wire_pack_impl(field, dst, src, n, true, stream);
}
static int wire_pack_impl(int field, unsigned char* dst, const uint64_t* src, size_t n, bool framed, void* stream) {
  if (!dst || (n && !src)) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  if (reinterpret_cast<uintptr_t>(dst) & 3) return fail(SCL_ERR_BAD_ARG, "wire buffer not 4-byte aligned");
  if (n > 0xFFFFFFFFull) return fail(SCL_ERR_BAD_ARG, "vector too long for the u32 count");  // Vector::SizeType, vector.h:73
  return with_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({src}));
    // framed: TcpChannel::send's u32 packet size in front (tcp_channel.h:127-137)
    const u32 image = (u32)(4 + n * F::LIMBS * 8);
    const WireGeom g = framed ? WireGeom{{image, (u32)n, 0, 0}, 2, n, n} : WireGeom{{(u32)n, 0, 0, 0}, 1, n, n};
    if (framed && 4 + n * F::LIMBS * 8 > 0xFFFFFFFFull) return fail(SCL_ERR_BAD_ARG, "frame larger than the u32 packet size");
    hipLaunchKernelGGL((k_wire_pack<F>), dim3(grid_for(n ? n : 1)), dim3(BLOCK), 0, S(stream), ctx,
                       reinterpret_cast<u32*>(dst), src, n, g);
    LAUNCH_CHECK();
    return SCL_OK;
  });
}

#if SCL_TU_COMMON
size_t scl_hip_wire_size_matrix(int field, size_t rows, size_t cols) {
  const int L = scl_hip_limbs(field);
  return L < 0 ? 0 : 12 + rows * cols * (size_t)L * 8;
}
#endif  // SCL_TU_COMMON

static int wire_pack_matrix_impl(int field, unsigned char* dst, const uint64_t* src, size_t ld, size_t rows, size_t cols,
                                 bool framed, void* stream);
int scl_hip_wire_pack_matrix(int field, unsigned char* dst, const uint64_t* src, size_t ld, size_t rows, size_t cols,
                             void* stream) {
  return wire_pack_matrix_impl(field, dst, src, ld, rows, cols, false, stream);
}
int scl_hip_frame_pack_matrix(int field, unsigned char* dst, const uint64_t* src, size_t ld, size_t rows, size_t cols,
                              void* stream) {
  return wire_pack_matrix_impl(field, dst, src, ld, rows, cols, true, stream);
}
static int wire_pack_matrix_impl(int field, unsigned char* dst, const uint64_t* src, size_t ld, size_t rows, size_t cols,
                                 bool framed, void* stream) {
  const size_t n = rows * cols;
  if (!dst || (n && !src)) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  if (reinterpret_cast<uintptr_t>(dst) & 3) return fail(SCL_ERR_BAD_ARG, "wire buffer not 4-byte aligned");
  if (rows > 0xFFFFFFFFull || cols > 0xFFFFFFFFull || n > 0xFFFFFFFFull)
    return fail(SCL_ERR_BAD_ARG, "matrix too large for the u32 dimensions of its wire image");  // matrix.h:913
  if (n && ld < cols) return fail(SCL_ERR_SIZE_MISMATCH, "ld < cols");
  return with_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({src}));
    const u32 image = (u32)(12 + n * F::LIMBS * 8);
    if (framed && 12 + n * F::LIMBS * 8 > 0xFFFFFFFFull) return fail(SCL_ERR_BAD_ARG, "frame larger than the u32 packet size");
    const WireGeom g = framed ? WireGeom{{image, (u32)rows, (u32)cols, (u32)n}, 4, cols ? cols : 1, n ? ld : 1}
                              : WireGeom{{(u32)rows, (u32)cols, (u32)n, 0}, 3, cols ? cols : 1, n ? ld : 1};
    hipLaunchKernelGGL((k_wire_pack<F>), dim3(grid_for(n ? n : 1)), dim3(BLOCK), 0, S(stream), ctx,
                       reinterpret_cast<u32*>(dst), src, n, g);
    LAUNCH_CHECK();
    return SCL_OK;
  });
}

int scl_hip_wire_unpack_matrix(int field, uint64_t* dst, size_t ld, size_t capacity_rows, const unsigned char* src,
                               size_t nbytes, size_t* rows_host, size_t* cols_host, void* stream) {
  if (!src || !rows_host || !cols_host) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  if (reinterpret_cast<uintptr_t>(src) & 3) return fail(SCL_ERR_BAD_ARG, "wire buffer not 4-byte aligned");
  if (nbytes < 12) return fail(SCL_ERR_BAD_ARG, "wire image shorter than its header");
  u32 hdr[3] = {0, 0, 0};
  HIP_TRY(hipMemcpyAsync(hdr, src, 12, hipMemcpyDeviceToHost, S(stream)));
  HIP_TRY(hipStreamSynchronize(S(stream)));
  const int L = scl_hip_limbs(field);
  if (L < 0) return fail(SCL_ERR_BAD_ARG, "unknown field tag");
  const size_t rows = hdr[0], cols = hdr[1], cnt = hdr[2];
  // the reference builds Matrix(rows, cols, values) without looking (matrix.h:420,956-960); an image whose
  // count disagrees with its dimensions is refused here
  if (cnt != rows * cols) return fail(SCL_ERR_BAD_ARG, "wire image: count != rows * cols");
  if (12 + cnt * L * 8 > nbytes) return fail(SCL_ERR_BAD_ARG, "wire image truncated");
  if (cnt && (rows > capacity_rows || cols > ld)) return fail(SCL_ERR_SIZE_MISMATCH, "destination too small for the wire image");
  *rows_host = rows;
  *cols_host = cols;
  if (cnt == 0) return SCL_OK;
  if (!dst) return fail(SCL_ERR_BAD_ARG, "dst is NULL");
  return with_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({dst}));
    const WireGeom g{{0, 0, 0, 0}, 3, cols, ld};
    hipLaunchKernelGGL((k_wire_unpack<F>), dim3(grid_for(cnt)), dim3(BLOCK), 0, S(stream), ctx, dst,
                       reinterpret_cast<const u32*>(src), cnt, g);
    LAUNCH_CHECK();
    return SCL_OK;
  });
}

static int wire_unpack_impl(int field, uint64_t* dst, size_t capacity, const unsigned char* src, size_t nbytes,
                            size_t* n_host, bool framed, void* stream);
int scl_hip_wire_unpack(int field, uint64_t* dst, size_t capacity, const unsigned char* src, size_t nbytes,
                        size_t* n_host, void* stream) {
  return wire_unpack_impl(field, dst, capacity, src, nbytes, n_host, false, stream);
}
int scl_hip_frame_unpack(int field, uint64_t* dst, size_t capacity, const unsigned char* src, size_t nbytes,
                         size_t* n_host, void* stream) {
  return wire_unpack_impl(field, dst, capacity, src, nbytes, n_host, true, stream);
}
static int wire_unpack_impl(int field, uint64_t* dst, size_t capacity, const unsigned char* src, size_t nbytes,
                            size_t* n_host, bool framed, void* stream) {
  if (!src || !n_host) return fail(SCL_ERR_BAD_ARG, "NULL operand");
  if (reinterpret_cast<uintptr_t>(src) & 3) return fail(SCL_ERR_BAD_ARG, "wire buffer not 4-byte aligned");
  const size_t hdr_bytes = framed ? 8 : 4;
  if (nbytes < hdr_bytes) return fail(SCL_ERR_BAD_ARG, "wire image shorter than its count");
  u32 head[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(head, src, hdr_bytes, hipMemcpyDeviceToHost, S(stream)));
  HIP_TRY(hipStreamSynchronize(S(stream)));
  const u32 cnt = framed ? head[1] : head[0];
  if (framed) {  // TcpChannel::recv reads exactly packet-size bytes (tcp_channel.h:190-206)
    if ((size_t)head[0] + 4 > nbytes) return fail(SCL_ERR_BAD_ARG, "frame truncated");
    nbytes = (size_t)head[0];
    src += 4;
  }
  const int L = scl_hip_limbs(field);
  if (L < 0) return fail(SCL_ERR_BAD_ARG, "unknown field tag");
  if (4 + (size_t)cnt * L * 8 > nbytes) return fail(SCL_ERR_BAD_ARG, "wire image truncated");
  if (cnt > capacity) return fail(SCL_ERR_SIZE_MISMATCH, "destination too small for the wire image");
  *n_host = cnt;
  if (cnt == 0) return SCL_OK;
  if (!dst) return fail(SCL_ERR_BAD_ARG, "dst is NULL");
  return with_field(field, [&](auto f, auto ctx) -> int {
    using F = decltype(f);
    SCL_TRY(check_align<F>({dst}));
    const WireGeom g{{0, 0, 0, 0}, 1, cnt, cnt};
    hipLaunchKernelGGL((k_wire_unpack<F>), dim3(grid_for(cnt)), dim3(BLOCK), 0, S(stream), ctx, dst,
                       reinterpret_cast<const u32*>(src), (size_t)cnt, g);
    LAUNCH_CHECK();
    return SCL_OK;
  });
}

// ---- roofline probe -------------------------------------------------------------------------------------------------------
#if SCL_TU_COMMON
int scl_hip_stream_copy(void* dst, const void* src, size_t bytes, void* stream) {
  if (bytes == 0) return SCL_OK;
  if (!dst || !src || !aligned16(dst) || !aligned16(src) || (bytes & 15))
    return fail(SCL_ERR_BAD_ARG, "stream_copy: 16-byte aligned buffers and sizes only");
  const size_t n16 = bytes / 16;
  hipLaunchKernelGGL(k_copy16<>, dim3(grid_for(n16)), dim3(BLOCK), 0, S(stream), static_cast<u64x2*>(dst),
                     static_cast<const u64x2*>(src), n16);
  LAUNCH_CHECK();
  return SCL_OK;
}
#endif  // SCL_TU_COMMON

}  // extern "C"

// ---- the open step: RCCL all-gather + reconstruct ----------------------------------------------------------------------
#if SCL_TU_COMMON
#if SCL_TU_FIELDS != 0xff
#include "capi_names_undef.inc"  // the open step calls the PUBLIC scl_hip_shamir_recover / scl_hip_additive_recover (any field)
#endif
#include "open_rccl.inc"
#endif  // SCL_TU_COMMON
