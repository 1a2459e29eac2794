// csrc/kernels.hpp -- HIP kernels (gfx950) for the SCL field / secret-sharing hot path.
//
// Data layout in HBM: SoA share matrices, row i = party i's share vector
// (u64 limbs, little-endian), so lane s reads shares[i][s] coalesced; one thread
// handles VEC consecutive secrets through 16-byte loads (M61: 2 x u64, 128-bit
// fields: 1 element).  Lagrange / alpha tables are wave-uniform: kernel
// arguments (scalar loads) for small tables, LDS for large ones.  There is no
// cross-lane traffic on the per-secret paths; sum/dot reduce through LDS.
//
// All of these are HBM-bound streaming kernels except inverse, the PRG-driven
// kernels (AES) and matmul; DESIGN.md has the per-kernel roofline accounting.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/scl_hip/detail/field.hpp"
#include "tu_config.hpp"

namespace sclhip {

constexpr int BLOCK = 256;
constexpr int FIXED_M_MAX = 16;  // recover kernels with lambda in kernel arguments

typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// gfx950's three-input bitwise instruction (v_bitop3_b32, truth table in the immediate): a ^ b ^ c and the bit select
// (a & m) | (b & ~m) in one instruction each.  Measured at 3.0 cycles per wave-instruction with four or more waves per SIMD
// (v_xor_b32 2.7, so a three-way xor costs 3.0 instead of 5.5; profiles/r3_oprate_bitop3.txt); the compiler does not form it
// from two xors by itself.
// (-DSCL_NO_BITOP3: the two-input forms, for A/B runs of the tools)
__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SCL_NO_BITOP3)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}
__device__ __forceinline__ u32 bitsel(u32 a, u32 b, u32 m) {  // bits of a where m is set, of b elsewhere
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SCL_NO_BITOP3)
  return __builtin_amdgcn_bitop3_b32(a, b, m, 0xE4);  // table bit 4a + 2b + c (tools/bitop3_probe.hip)
#else
  return (a & m) | (b & ~m);
#endif
}


// ---- 16-byte (or 8-byte) pack loads/stores ---------------------------------------------------
template <class F, int VEC>
struct Pack {
  typename F::E v[VEC];
};

template <bool NT, class T>
__device__ __forceinline__ T ldg(const T* p) {
  if constexpr (NT)
    return __builtin_nontemporal_load(p);
  else
    return *p;
}
template <bool NT, class T>
__device__ __forceinline__ void stg(T* p, T v) {
  if constexpr (NT)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}

// p points at the first limb of element index e (already multiplied out by the caller)
template <class F, int VEC, bool NT>
__device__ __forceinline__ Pack<F, VEC> load_pack(const u64* p) {
  Pack<F, VEC> r;
  if constexpr (F::LIMBS == 1 && VEC == 2) {
    const u64x2 w = ldg<NT>(reinterpret_cast<const u64x2*>(p));
    r.v[0] = w.x;
    r.v[1] = w.y;
  } else if constexpr (F::LIMBS == 1 && VEC == 1) {
    r.v[0] = ldg<NT>(p);
  } else if constexpr (F::LIMBS == 2) {
    static_assert(VEC == 1, "128-bit fields use one element per lane");
    const u64x2 w = ldg<NT>(reinterpret_cast<const u64x2*>(p));
    r.v[0] = ((u128)w.y << 64) | w.x;
  } else {
    static_assert(VEC == 1 && F::LIMBS == 4, "256-bit fields use one element per lane");
    const u64x2 w0 = ldg<NT>(reinterpret_cast<const u64x2*>(p)), w1 = ldg<NT>(reinterpret_cast<const u64x2*>(p) + 1);
    r.v[0].w[0] = w0.x;
    r.v[0].w[1] = w0.y;
    r.v[0].w[2] = w1.x;
    r.v[0].w[3] = w1.y;
  }
  return r;
}

template <class F, int VEC, bool NT>
__device__ __forceinline__ void store_pack(u64* p, const Pack<F, VEC>& r) {
  if constexpr (F::LIMBS == 1 && VEC == 2) {
    u64x2 w;
    w.x = r.v[0];
    w.y = r.v[1];
    stg<NT>(reinterpret_cast<u64x2*>(p), w);
  } else if constexpr (F::LIMBS == 1 && VEC == 1) {
    stg<NT>(p, r.v[0]);
  } else if constexpr (F::LIMBS == 2) {
    u64x2 w;
    w.x = (u64)r.v[0];
    w.y = (u64)(r.v[0] >> 64);
    stg<NT>(reinterpret_cast<u64x2*>(p), w);
  } else {
    u64x2 w0, w1;
    w0.x = r.v[0].w[0];
    w0.y = r.v[0].w[1];
    w1.x = r.v[0].w[2];
    w1.y = r.v[0].w[3];
    stg<NT>(reinterpret_cast<u64x2*>(p), w0);
    stg<NT>(reinterpret_cast<u64x2*>(p) + 1, w1);
  }
}

template <class F>
struct Table {  // small kernel-argument table (wave-uniform, scalar loads)
  typename F::E v[FIXED_M_MAX];
};

// Large table: 2 KiB of kernel arguments, copied into LDS by the block.  Bounds the party count
// of the table-driven kernels: 256 for Mersenne61, 128 for the 128-bit fields, 64 for secp256k1_order.
template <class F>
struct BigTable {
  enum { CAP = 256 / F::LIMBS };
  typename F::E v[CAP];
};

#define SCL_GRID_STRIDE(q, npacks) \
  for (size_t q = (size_t)blockIdx.x * BLOCK + threadIdx.x; q < (npacks); q += (size_t)gridDim.x * BLOCK)

// ---- element-wise --------------------------------------------------------------------------
// what FF::invert / Z2k::invert refuse: zero in a field (small_ff.h:61-70), even values in a ring (z2k_ops.h:81-83)
template <class F>
__device__ __forceinline__ bool not_invertible(const typename F::E& a) {
  if constexpr (F::TAG == 5 || F::TAG == 6) return (F::low32(a) & 1u) == 0;  // the rings Z2k64 / Z2k128
  else return F::is_zero(a);
}

// the batch's "a zero was inverted" flag: set once -- a batch with many zeros (an error path, but 10^5 writes to one word cost a
// 0.3 ms kernel another 0.25 ms) reads the word first and leaves it alone when somebody has raised it already.  The word may
// be pinned HOST memory mapped into the device (scl_hip_ew) or device memory (scl_hip_ew_status): a plain system-scope STORE
// of 1 -- idempotent, so racing lanes need no read-modify-write, and a store reaches host memory on every platform where a
// device atomic would need PCIe atomic routing (absent under VFIO passthrough and on some root complexes).  The only other
// bit anyone sets in the word is this one, so storing 1 is the OR.
__device__ __forceinline__ void raise_flag(unsigned* flag) {
  if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0)
    __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Vector::add/subtract/multiplyEntryWise, FF::negate/invert/operator/ (vector.h:199-245, ff.h:203-246)
template <class F, int OP, int VEC, bool NT>
__global__ __launch_bounds__(BLOCK) void k_ew(typename F::Ctx ctx, u64* dst, const u64* a, const u64* b,
                                              size_t npacks, unsigned* zero_flag) {
  SCL_GRID_STRIDE(q, npacks) {
    const size_t off = q * VEC * F::LIMBS;
    Pack<F, VEC> x = load_pack<F, VEC, NT>(a + off), y, r;
    if constexpr (OP == 0 || OP == 1 || OP == 2 || OP == 5) y = load_pack<F, VEC, NT>(b + off);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if constexpr (OP == 0) r.v[v] = F::add(ctx, x.v[v], y.v[v]);
      if constexpr (OP == 1) r.v[v] = F::sub(ctx, x.v[v], y.v[v]);
      if constexpr (OP == 2) r.v[v] = F::mul(ctx, x.v[v], y.v[v]);
      if constexpr (OP == 3) r.v[v] = F::neg(ctx, x.v[v]);
      // a non-invertible operand raises the flag and leaves 0 in its slot: Fermat's chain maps 0 to 0 by itself; a ring's
      // Newton iteration on an even value would leave an arbitrary word, so the rings say so explicitly
      constexpr bool RING = F::TAG == 5 || F::TAG == 6;
      if constexpr (OP == 4) {
        const bool bad = not_invertible<F>(x.v[v]);
        if (bad) raise_flag(zero_flag);
        r.v[v] = F::inv(ctx, x.v[v]);
        if constexpr (RING)
          if (bad) r.v[v] = F::zero();
      }
      if constexpr (OP == 5) {
        const bool bad = not_invertible<F>(y.v[v]);
        if (bad) raise_flag(zero_flag);
        r.v[v] = F::mul(ctx, x.v[v], F::inv(ctx, y.v[v]));
        if constexpr (RING)
          if (bad) r.v[v] = F::zero();
      }
    }
    store_pack<F, VEC, NT>(dst + off, r);
  }
}

// ---- FF::invert / operator/ over a batch: Montgomery's simultaneous inversion, one chain per lane -------------------------
// The reference inverts element by element (extended Euclid, small_ff.h:61-92; Fermat for the mpn family,
// ff_ops_gmp.h:250-260; operator/ = multiply by the inverse, ff.h:203-205).  An inverse is unique, so any route to it gives the
// same bits (SURVEY 8a note C): a lane takes L = R * VEC elements, forms the prefix products c_i = x_0 ... x_i, inverts c_(L-1)
// ONCE (the field's Fermat chain) and walks back, x_i^-1 = c_(i-1) * (c_i)^-1, (c_(i-1))^-1 = (c_i)^-1 * x_i:
// 3 (L - 1) products and one inversion for L elements where k_ew<F, 4 | 5> spends 70 (Mersenne61) to 380 (secp256k1) products
// on each.  A zero sets the flag and goes through the chain as a one; its slot gets 0, which is what F::inv(0) returns.
// A workgroup owns BLOCK * R consecutive packs, lane l the packs l, l + BLOCK, ...: every access is a whole line.
// Register discipline (what took the kernel from 214 to 168 registers and from 2 to 3 waves per SIMD, +12 %): the loads go
// G packs at a time between scheduling barriers; a full tile is straight-line code; the zero mask is pinned after every group;
// the pack indices are recomputed where they are used (lane_now).  KEEP = false keeps only the prefix products and reads the
// operand a second time on the walk back (the tile is still in L2 / the memory-side cache): half the registers per slot, so L
// can double -- measured equal to KEEP at best (profiles/r5_ew_bench_grouped.txt), kept for the record.

// the lane's index through an empty asm: what is computed from the result cannot be hoisted out of the enclosing loop, so the 3 R
// pack indices of a tile are one add each where they are used, not 3 R registers (or 64-bit addresses) alive across the kernel
__device__ __forceinline__ unsigned lane_now(unsigned l) {
  asm volatile("" : "+v"(l));
  return l;
}

template <class F, bool DIV, int VEC, int R, int G, bool FULL, bool KEEP>
__device__ __forceinline__ void ew_inv_tile(const typename F::Ctx& ctx, u64* out, const u64* numer, const u64* src, unsigned rem,
                                                   unsigned* zero_flag) {
  typedef typename F::E E;
  constexpr int L = R * VEC;
  constexpr unsigned PW = VEC * F::LIMBS;  // words per pack
  const unsigned l = threadIdx.x;
  E c[L];
  E xk[KEEP ? L : 1];  // KEEP: the operand stays in registers too (no second read, twice the registers per slot)
  unsigned zm[2] = {0u, 0u};  // which slots held a zero
  // G packs in flight at a time: left alone the scheduler hoists every load of the tile to the top and the registers are gone.
  // (A full tile is straight-line code -- no branch around a load or a store -- or the barriers mean nothing: the products sink
  // into the blocks of their uses.)
#pragma unroll
  for (int r0 = 0; r0 < R; r0 += G) {
    Pack<F, VEC> pk[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const unsigned q = lane_now(l) + (unsigned)(r0 + g) * BLOCK;
      const unsigned qc = FULL ? q : (q < rem ? q : rem - 1);  // past the end: the tile's last pack, dropped below
      pk[g] = load_pack<F, VEC, KEEP>(src + qc * PW);  // streaming unless it is read again below
      if (!FULL && q >= rem) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) pk[g].v[v] = F::one(ctx);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int i = (r0 + g) * VEC + v;
        const bool z = F::is_zero(pk[g].v[v]);
        zm[i >> 5] |= z ? (1u << (i & 31)) : 0u;
        const E x = z ? F::one(ctx) : pk[g].v[v];
        if constexpr (KEEP) xk[i] = x;
        if (i == 0) c[0] = x;
        else c[i] = F::mul(ctx, c[i - 1], x);
      }
    }
    // the mask is final for this group here: without this the sixty-four tests are re-associated into one tree at the end of
    // the pass and every operand stays in its registers until then
    asm volatile("" : "+v"(zm[0]), "+v"(zm[1]));
    __builtin_amdgcn_sched_barrier(0);
  }
  E inv = F::inv(ctx, c[L - 1]);
#pragma unroll
  for (int r0 = R - G; r0 >= 0; r0 -= G) {
    Pack<F, VEC> pk[G], num[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const unsigned q = lane_now(l) + (unsigned)(r0 + g) * BLOCK;
      const unsigned qc = FULL ? q : (q < rem ? q : rem - 1);
      if constexpr (!KEEP) {
        pk[g] = load_pack<F, VEC, true>(src + qc * PW);
        if (!FULL && q >= rem) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) pk[g].v[v] = F::one(ctx);
        }
      }
      if constexpr (DIV) num[g] = load_pack<F, VEC, true>(numer + qc * PW);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = G - 1; g >= 0; --g) {
      const unsigned q = lane_now(l) + (unsigned)(r0 + g) * BLOCK;
      Pack<F, VEC> o;
#pragma unroll
      for (int v = VEC - 1; v >= 0; --v) {
        const int i = (r0 + g) * VEC + v;
        const bool z = (zm[i >> 5] >> (i & 31)) & 1u;
        E y = i > 0 ? F::mul(ctx, inv, c[i - 1]) : inv;
        if (i > 0) inv = F::mul(ctx, inv, KEEP ? xk[i] : (z ? F::one(ctx) : pk[g].v[v]));
        if (z) y = F::zero();
        if constexpr (DIV) y = F::mul(ctx, num[g].v[v], y);
        o.v[v] = y;
      }
      if (FULL || q < rem) store_pack<F, VEC, true>(out + q * PW, o);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (zm[0] | zm[1]) raise_flag(zero_flag);
}

template <class F, bool DIV, int VEC, int R, int G = (R < 4 ? R : 4), int WAVES = 2, bool KEEP = true>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_ew_inv(typename F::Ctx ctx, u64* dst, const u64* a, const u64* b, size_t npacks,
                                                         unsigned* zero_flag) {
  static_assert(R * VEC <= 64 && R % G == 0, "one bit of the zero mask per slot; whole groups of loads");
  constexpr unsigned PW = VEC * F::LIMBS;
  for (size_t tile = blockIdx.x; tile * ((size_t)BLOCK * R) < npacks; tile += gridDim.x) {
    // a uniform base per tile and 32-bit pack indices below it
    const size_t t0 = tile * ((size_t)BLOCK * R);
    const u64* src = (DIV ? b : a) + t0 * PW;
    if (npacks - t0 >= (size_t)BLOCK * R)
      ew_inv_tile<F, DIV, VEC, R, G, true, KEEP>(ctx, dst + t0 * PW, a + t0 * PW, src, BLOCK * R, zero_flag);
    else
      ew_inv_tile<F, DIV, VEC, R, G, false, KEEP>(ctx, dst + t0 * PW, a + t0 * PW, src, (unsigned)(npacks - t0), zero_flag);
  }
}

// ---- GF(2^128) element-wise products on a window table in LDS --------------------------------------------------------------
// gfx950 has no carry-less multiply.  Gf128::mul spends ~1.9 k vector instructions per product selecting among four shifted
// copies of a for every nibble of b; the comb product keeps the sixteen multiples u(x) a (deg u < 4) of THIS lane's a in LDS
// (256 bytes per lane, entry u of lane l at (u * BLK + l) * 16: the ds_read_b128 of a wave are 64 consecutive 16-byte slots
// whichever entries the lanes pick, so neither the writes nor the reads conflict) and adds, for nibble position k = 7..0, the
// entry picked by nibble k of each 32-bit word of b at that word's offset; the 256-bit sum moves up four bits between
// positions.  32 reads, 7 shifts, one fold: ~0.4 k vector instructions and 48 LDS accesses per product at 4-bit windows.  The
// kernels run 3-BIT windows: eight entries (128 bytes per lane), 44 reads, 10 shifts -- slightly more instructions, but twice the
// waves fit beside the tables and the vector ALU issues to five waves per SIMD instead of two and a half: 56-60 -> 76-80 G
// products/s at 10^7 elements (2-bit windows: 54; profiles/r5_ew_bench_windows.txt).
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

// W = window bits: 4 (sixteen entries, 256 bytes per lane: 32 lookups, 7 shifts) or 3 (eight entries, 128 bytes per lane: 44
// lookups, 10 shifts -- a few more instructions, but twice the waves fit beside the tables)
template <int BLK, int W = 4>
__device__ __forceinline__ void gf_table_store(u32x4* mine, u128 a) {  // mine = table base + this lane
  u128 t[16];
  Gf128::window_table(a, t);  // (entries 8.. are dead code at W = 3)
#pragma unroll
  for (int u = 0; u < (1 << W); ++u) {
    u32x4 w;
    w.x = (u32)t[u];
    w.y = (u32)(t[u] >> 32);
    w.z = (u32)(t[u] >> 64);
    w.w = (u32)(t[u] >> 96);
    mine[u * BLK] = w;
  }
}

template <int BLK, int W = 4>
__device__ __forceinline__ u128 gf_comb(const u32x4* mine, u128 b) {  // (the a of the table) * b
  const u32 bw[4] = {(u32)b, (u32)(b >> 32), (u32)(b >> 64), (u32)(b >> 96)};
  u32 c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int NPOS = (32 + W - 1) / W;  // window positions in a 32-bit word (the top one of W = 3 holds two bits)
#pragma unroll
  for (int k = NPOS - 1; k >= 0; --k) {
    u32x4 m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = mine[((bw[j] >> (W * k)) & ((1u << W) - 1)) * BLK];
    // word j + i of the sum takes word i of the entry picked for word j of b: sixteen terms as ten three-input xors
    c[0] ^= m[0].x;
    c[1] = xor3(c[1], m[0].y, m[1].x);
    c[2] = xor3(c[2], m[0].z, m[1].y) ^ m[2].x;
    c[3] = xor3(xor3(c[3], m[0].w, m[1].z), m[2].y, m[3].x);
    c[4] = xor3(c[4], m[1].w, m[2].z) ^ m[3].y;
    c[5] = xor3(c[5], m[2].w, m[3].z);
    c[6] ^= m[3].w;
    if (k) {
#pragma unroll
      for (int i = 7; i > 0; --i) c[i] = __builtin_amdgcn_alignbit(c[i], c[i - 1], 32 - W);
      c[0] <<= W;
    }
  }
  const u128 lo = (u128)c[0] | ((u128)c[1] << 32) | ((u128)c[2] << 64) | ((u128)c[3] << 96);
  const u128 hi = (u128)c[4] | ((u128)c[5] << 32) | ((u128)c[6] << 64) | ((u128)c[7] << 96);
  return Gf128::reduce256(lo, hi);
}

// multiplyEntryWise over GF(2^128) on the LDS table.  Dynamic LDS: BLK * 16 * 2^W bytes.
template <int BLK, int W = 4>
__global__ __launch_bounds__(BLK) void k_ew_gf128_mul(u64* dst, const u64* a, const u64* b, size_t n) {
  extern __shared__ u32x4 gf_tbl[];
  u32x4* mine = gf_tbl + threadIdx.x;
  typedef Gf128 F;
  for (size_t q = (size_t)blockIdx.x * BLK + threadIdx.x; q < n; q += (size_t)gridDim.x * BLK) {
    const Pack<F, 1> x = load_pack<F, 1, true>(a + q * 2), y = load_pack<F, 1, true>(b + q * 2);
    gf_table_store<BLK, W>(mine, x.v[0]);
    Pack<F, 1> r;
    r.v[0] = gf_comb<BLK, W>(mine, y.v[0]);
    store_pack<F, 1, true>(dst + q * 2, r);
  }
}

// Vector::scalarMultiply over GF(2^128): ONE window table, the scalar's, shared by the workgroup -- the sixteen entries fill the
// 64 banks exactly, so lanes that pick different entries read different banks and lanes that pick the same one share a broadcast.
template <int BLK>
__global__ __launch_bounds__(BLK) void k_scalar_mul_gf128(u64* dst, const u64* a, Table<Gf128> scalar, size_t n) {
  __shared__ u32x4 tbl[16];
  if (threadIdx.x == 0) gf_table_store<1>(tbl, scalar.v[0]);
  __syncthreads();
  for (size_t q = (size_t)blockIdx.x * BLK + threadIdx.x; q < n; q += (size_t)gridDim.x * BLK) {
    Pack<Gf128, 1> x = load_pack<Gf128, 1, true>(a + q * 2);
    x.v[0] = gf_comb<1>(tbl, x.v[0]);
    store_pack<Gf128, 1, true>(dst + q * 2, x);
  }
}

// Vector::dot / innerProd over GF(2^128) (vector.h:45-52, 252-255): the products on the per-lane LDS table, the sum an xor
// (wavefront shuffles, then one LDS slot per wave); per-workgroup partials like k_dot.  Dynamic LDS: BLK * 16 * 2^W bytes.
template <int BLK, int W = 3>
__global__ __launch_bounds__(BLK) void k_dot_gf128(u64* partial, const u64* a, const u64* b, size_t n) {
  extern __shared__ u32x4 gf_tbl[];
  __shared__ u128 red[BLK / 64];
  u32x4* mine = gf_tbl + threadIdx.x;
  u128 acc = 0;
  for (size_t q = (size_t)blockIdx.x * BLK + threadIdx.x; q < n; q += (size_t)gridDim.x * BLK) {
    const u128 x = load_pack<Gf128, 1, true>(a + q * 2).v[0], y = load_pack<Gf128, 1, true>(b + q * 2).v[0];
    gf_table_store<BLK, W>(mine, x);
    acc ^= gf_comb<BLK, W>(mine, y);
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    u32 w[4] = {(u32)acc, (u32)(acc >> 32), (u32)(acc >> 64), (u32)(acc >> 96)};
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] ^= (u32)__shfl_xor((int)w[i], m, 64);
    acc = (u128)w[0] | ((u128)w[1] << 32) | ((u128)w[2] << 64) | ((u128)w[3] << 96);
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    u128 tot = red[0];
#pragma unroll
    for (int w = 1; w < BLK / 64; ++w) tot ^= red[w];
    Gf128::st(partial + (size_t)blockIdx.x * 2, tot);
  }
}

// ---- simultaneous inversion with the chain in memory: the fields whose ONE inversion is dear ------------------------------------
// k_ew_inv keeps 2 L elements in registers, which stops at L = 16..32; for Mont128 (~250 products per Fermat inversion), secp256k1
// (~450) and GF(2^128) the inversion's share I / L still dominates there.  Here the loops over the chain stay rolled: the prefix
// products live in a per-lane array the compiler places in scratch memory (L1 / L2 traffic these compute-bound kernels have
// room for), the operand is read a second time on the walk back (an L2 hit: a workgroup's tile is BLK * L elements), the code is
// one copy of the product (the unrolled form of these fields runs to 60-130 KB, past the instruction cache) and L is free.
// ARITH supplies the products: the field's own (FieldArith) or GF(2^128)'s on the LDS window table (GfLdsArith).
template <class F>
struct FieldArith {
  typedef typename F::E E;
  enum { LDS_PER_LANE = 0 };
  typename F::Ctx ctx;
  __device__ __forceinline__ FieldArith(const typename F::Ctx& c, int) : ctx(c) {}
  __device__ __forceinline__ E one() const { return F::one(ctx); }
  __device__ __forceinline__ E product(const E& a, const E& b) const { return F::mul(ctx, a, b); }
  __device__ __forceinline__ void product2(const E& a, const E& b1, const E& b2, E& r1, E& r2) const {
    r1 = F::mul(ctx, a, b1);
    r2 = F::mul(ctx, a, b2);
  }
  __device__ __forceinline__ E inverse(const E& a) const { return F::inv(ctx, a); }
};

template <int BLK, int W = 4>
struct GfLdsArith {
  typedef u128 E;
  enum { LDS_PER_LANE = 16 << W };
  u32x4* mine;
  __device__ __forceinline__ GfLdsArith(const Gf128::Ctx&, int tid) {
    extern __shared__ u32x4 gf_tbl[];
    mine = gf_tbl + tid;
  }
  __device__ __forceinline__ E one() const { return 1; }
  __device__ __forceinline__ E product(E a, E b) const {
    gf_table_store<BLK, W>(mine, a);
    return gf_comb<BLK, W>(mine, b);
  }
  __device__ __forceinline__ void product2(E a, E b1, E b2, E& r1, E& r2) const {  // one table, two combs
    gf_table_store<BLK, W>(mine, a);
    r1 = gf_comb<BLK, W>(mine, b1);
    r2 = gf_comb<BLK, W>(mine, b2);
  }
  __device__ __forceinline__ E inverse(E a) const {
    return Gf128::inv_chain(Gf128::Ctx{}, a, [&](E x, E y) { return product(x, y); });
  }
};

template <class F, class ARITH, bool DIV, int L, int BLK>
__global__ __launch_bounds__(BLK) void k_ew_inv_rolled(typename F::Ctx ctx, u64* dst, const u64* a, const u64* b, size_t n,
                                                       unsigned* zero_flag) {
  typedef typename F::E E;
  // U elements per loop trip: their loads are issued together, ahead of the (sequentially dependent) products -- one memory
  // round trip per U elements on a wave's critical path instead of one per element; the code holds U copies of the product
  // (GF(2^128) on the LDS table stays at one: two interleaved comb products take 260 registers, and its waves are few already)
  constexpr int U = ARITH::LDS_PER_LANE ? 1 : (L % 4 == 0 && F::LIMBS <= 2) ? 4 : (L % 2 == 0 ? 2 : 1);
  const ARITH ar(ctx, (int)threadIdx.x);
  const u64* src = DIV ? b : a;
  for (size_t tile = blockIdx.x; tile * ((size_t)BLK * L) < n; tile += gridDim.x) {
    const size_t q0 = tile * ((size_t)BLK * L) + threadIdx.x;
    E c[L];
    E run = ar.one();
    bool any_zero = false;
#pragma unroll 1
    for (int i0 = 0; i0 < L; i0 += U) {
      E v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t q = q0 + (size_t)(i0 + u) * BLK;
        v[u] = ar.one();
        if (q < n) v[u] = load_pack<F, 1, false>(src + q * F::LIMBS).v[0];  // (read again on the walk back: keep it cached)
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (F::is_zero(v[u])) {
          any_zero = true;
          v[u] = ar.one();
        }
        run = (i0 + u) ? ar.product(run, v[u]) : v[u];
        c[i0 + u] = run;
      }
    }
    if (any_zero) raise_flag(zero_flag);
    E inv = ar.inverse(run);
#pragma unroll 1
    for (int i0 = L - U; i0 >= 0; i0 -= U) {
      E v[U], num[U], cp[U];
#pragma unroll
      for (int u = U - 1; u >= 0; --u) {
        const size_t q = q0 + (size_t)(i0 + u) * BLK;
        v[u] = ar.one();
        if (q < n) {
          v[u] = load_pack<F, 1, true>(src + q * F::LIMBS).v[0];
          if constexpr (DIV) num[u] = load_pack<F, 1, true>(a + q * F::LIMBS).v[0];
        }
        if (i0 + u) cp[u] = c[i0 + u - 1];
      }
#pragma unroll
      for (int u = U - 1; u >= 0; --u) {
        const size_t q = q0 + (size_t)(i0 + u) * BLK;
        const bool z = F::is_zero(v[u]);
        if (z) v[u] = ar.one();
        E o = inv;
        if (i0 + u) ar.product2(inv, cp[u], v[u], o, inv);
        if (q < n) {
          Pack<F, 1> out;
          out.v[0] = z ? F::zero() : o;
          if constexpr (DIV) out.v[0] = ar.product(num[u], out.v[0]);
          store_pack<F, 1, true>(dst + q * F::LIMBS, out);
        }
      }
    }
  }
}

// ---- the same with the chain in two levels: checkpoints, blocks recomputed ------------------------------------------------------------
// k_ew_inv_rolled moves 5 E bytes per element through HBM once the resident chains outgrow the caches (x read, every prefix
// product written to scratch and read back, x read again, the result written: 2.5 x the algorithmic 2 E, measured --
// profiles/r6_sq_counters.txt).  Here a lane's chain of L = L1 * L2 elements keeps only the prefix product at the END of each
// block of L2 (L1 checkpoints: a few scratch words per lane); the walk back takes the blocks last to first, recomputes the
// block's L2 prefix products from the checkpoint before it -- in registers, from the operands it has to read again anyway --
// and walks the block.  3 E bytes and 4 - 1/L2 + I/L products per element instead of 5 E and 3 + I/L: for a field whose
// rolled inversion is bound by that traffic (Mersenne127) a gain, for the ones that sit on vector issue a loss
// (ew_inverse_rolled chooses; "inv_two_level" pins).  Same unique inverses, same zero handling as k_ew_inv_rolled.
template <class F, class ARITH, bool DIV, int L1, int L2, int BLK>
__global__ __launch_bounds__(BLK) void k_ew_inv_blocked(typename F::Ctx ctx, u64* dst, const u64* a, const u64* b, size_t n,
                                                        unsigned* zero_flag) {
  typedef typename F::E E;
  constexpr int L = L1 * L2;
  const ARITH ar(ctx, (int)threadIdx.x);
  const u64* src = DIV ? b : a;
  for (size_t tile = blockIdx.x; tile * ((size_t)BLK * L) < n; tile += gridDim.x) {
    const size_t q0 = tile * ((size_t)BLK * L) + threadIdx.x;
    E ck[L1];  // ck[j] = x_0 ... x_(L2 (j + 1) - 1), zeros taken as ones
    E run = ar.one();
    bool any_zero = false;
#pragma unroll 1
    for (int j = 0; j < L1; ++j) {
      E v[L2];
#pragma unroll
      for (int k = 0; k < L2; ++k) {
        const size_t q = q0 + (size_t)(j * L2 + k) * BLK;
        v[k] = ar.one();
        if (q < n) v[k] = load_pack<F, 1, false>(src + q * F::LIMBS).v[0];
      }
#pragma unroll
      for (int k = 0; k < L2; ++k) {
        if (F::is_zero(v[k])) {
          any_zero = true;
          v[k] = ar.one();
        }
        run = (j | k) ? ar.product(run, v[k]) : v[k];
      }
      ck[j] = run;
    }
    if (any_zero) raise_flag(zero_flag);
    E inv = ar.inverse(run);
#pragma unroll 1
    for (int j = L1 - 1; j >= 0; --j) {
      E v[L2], pre[L2], num[L2];
#pragma unroll
      for (int k = 0; k < L2; ++k) {
        const size_t q = q0 + (size_t)(j * L2 + k) * BLK;
        v[k] = ar.one();
        if (q < n) {
          v[k] = load_pack<F, 1, true>(src + q * F::LIMBS).v[0];
          if constexpr (DIV) num[k] = load_pack<F, 1, true>(a + q * F::LIMBS).v[0];
        }
      }
      bool z[L2];
      pre[0] = j ? ck[j - 1] : ar.one();  // everything before the block
#pragma unroll
      for (int k = 0; k < L2; ++k) {
        z[k] = F::is_zero(v[k]);
        if (z[k]) v[k] = ar.one();
        if (k + 1 < L2) pre[k + 1] = (j | k) ? ar.product(pre[k], v[k]) : v[k];
      }
#pragma unroll
      for (int k = L2 - 1; k >= 0; --k) {
        const size_t q = q0 + (size_t)(j * L2 + k) * BLK;
        E o = inv;
        if (j | k) ar.product2(inv, pre[k], v[k], o, inv);
        if (q < n) {
          Pack<F, 1> out;
          out.v[0] = z[k] ? F::zero() : o;
          if constexpr (DIV) out.v[0] = ar.product(num[k], out.v[0]);
          store_pack<F, 1, true>(dst + q * F::LIMBS, out);
        }
      }
    }
  }
}

// Vector::scalarMultiply (vector.h:274-301)
template <class F, int VEC, bool NT>
__global__ __launch_bounds__(BLOCK) void k_scalar_mul(typename F::Ctx ctx, u64* dst, const u64* a,
                                                      Table<F> scalar, size_t npacks) {
  SCL_GRID_STRIDE(q, npacks) {
    const size_t off = q * VEC * F::LIMBS;
    Pack<F, VEC> x = load_pack<F, VEC, NT>(a + off);
#pragma unroll
    for (int v = 0; v < VEC; ++v) x.v[v] = F::mul(ctx, scalar.v[0], x.v[v]);
    store_pack<F, VEC, NT>(dst + off, x);
  }
}

// ---- block reduction of canonical elements: wavefront shuffles, then one LDS slot per wave ----------------
// (an 8-level LDS tree with a barrier per level before: 5.3 TB/s for sum / dot; the north star asks for wavefront
// shuffle reductions)
template <class F>
__device__ __forceinline__ typename F::E shfl_xor_elem(const typename F::E& v, int mask) {
  constexpr int W = (int)(sizeof(typename F::E) / 4);
  u32 w[W];
  __builtin_memcpy(w, &v, sizeof v);
#pragma unroll
  for (int i = 0; i < W; ++i) w[i] = (u32)__shfl_xor((int)w[i], mask, 64);
  typename F::E r;
  __builtin_memcpy(&r, w, sizeof r);
  return r;
}

template <class F>
__device__ __forceinline__ typename F::E block_reduce_add(const typename F::Ctx& ctx, typename F::E x) {
  __shared__ typename F::E red[BLOCK / 64];
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) x = F::add(ctx, x, shfl_xor_elem<F>(x, m));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  typename F::E tot = red[0];
#pragma unroll
  for (int w = 1; w < BLOCK / 64; ++w) tot = F::add(ctx, tot, red[w]);
  return tot;
}

constexpr int RED_UNROLL = 4;  // independent 16-byte loads in flight per lane and operand

// Vector::sum (vector.h:261-267): per-block partial sums -> partial[blockIdx].  Launched with ONE trip per thread
// (RED_UNROLL packs, a grid stride apart) wherever the batch allows: a read-only sweep by fresh workgroups reaches
// 7.0 TB/s where a few thousand resident workgroups that grid-stride reach 5.7-6.0 (profiles/r1_membench_hbm_ceilings.txt;
// issuing the next trip's loads ahead of this trip's adds did not help).  The partials -- up to 2^20 of them -- are
// reduced by a second launch of the same kernel.
template <class F, int VEC>
__global__ __launch_bounds__(BLOCK) void k_sum(typename F::Ctx ctx, u64* partial, const u64* a, size_t npacks) {
  typename F::Acc acc = F::acc_zero();
  int terms = 0;
  typename F::E run = F::zero();
  const size_t G = (size_t)gridDim.x * BLOCK;
  for (size_t q0 = (size_t)blockIdx.x * BLOCK + threadIdx.x; q0 < npacks; q0 += RED_UNROLL * G) {
    Pack<F, VEC> x[RED_UNROLL];
#pragma unroll
    for (int u = 0; u < RED_UNROLL; ++u)
      if (q0 + u * G < npacks) x[u] = load_pack<F, VEC, true>(a + (q0 + u * G) * VEC * F::LIMBS);
#pragma unroll
    for (int u = 0; u < RED_UNROLL; ++u) {
      if (q0 + u * G < npacks) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) F::acc_add(ctx, acc, x[u].v[v]);
        terms += VEC;
      }
    }
    if (terms + RED_UNROLL * VEC > F::ACC_TERMS) {
      run = F::add(ctx, run, F::acc_fold(ctx, acc));
      acc = F::acc_zero();
      terms = 0;
    }
  }
  run = F::add(ctx, run, F::acc_fold(ctx, acc));
  const typename F::E tot = block_reduce_add<F>(ctx, run);
  if (threadIdx.x == 0) F::st(partial + (size_t)blockIdx.x * F::LIMBS, tot);
}

// Vector::dot / innerProd (vector.h:45-52, 252-255)
template <class F, int VEC>
__global__ __launch_bounds__(BLOCK) void k_dot(typename F::Ctx ctx, u64* partial, const u64* a, const u64* b,
                                               size_t npacks) {
  typename F::Acc acc = F::acc_zero();
  int terms = 0;
  typename F::E run = F::zero();
  const size_t G = (size_t)gridDim.x * BLOCK;
  for (size_t q0 = (size_t)blockIdx.x * BLOCK + threadIdx.x; q0 < npacks; q0 += RED_UNROLL * G) {
    Pack<F, VEC> x[RED_UNROLL], y[RED_UNROLL];
#pragma unroll
    for (int u = 0; u < RED_UNROLL; ++u) {
      if (q0 + u * G < npacks) {
        const size_t off = (q0 + u * G) * VEC * F::LIMBS;
        x[u] = load_pack<F, VEC, true>(a + off);
        y[u] = load_pack<F, VEC, true>(b + off);
      }
    }
#pragma unroll
    for (int u = 0; u < RED_UNROLL; ++u) {
      if (q0 + u * G < npacks) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) F::mac(ctx, acc, x[u].v[v], y[u].v[v]);
        terms += VEC;
      }
    }
    if (terms + RED_UNROLL * VEC > F::ACC_TERMS) {
      run = F::add(ctx, run, F::acc_fold(ctx, acc));
      acc = F::acc_zero();
      terms = 0;
    }
  }
  run = F::add(ctx, run, F::acc_fold(ctx, acc));
  const typename F::E tot = block_reduce_add<F>(ctx, run);
  if (threadIdx.x == 0) F::st(partial + (size_t)blockIdx.x * F::LIMBS, tot);
}

// Vector::equals (vector.h:558-570): counts mismatching limbs
template <int = 0>  // (a template so that only the unit that launches it compiles it)
__global__ __launch_bounds__(BLOCK) void k_count_diff(unsigned long long* count, const u64* a, const u64* b,
                                                      size_t nwords) {
  unsigned long long local = 0;
  SCL_GRID_STRIDE(q, nwords) local += (a[q] != b[q]);
  if (local) atomicAdd(count, local);
}

// Z2k equality compares modulo 2^K (z2k_ops.h:97-103): mask = 2^K - 1 as (lo, hi) words, L limbs per element
template <int = 0>  // (a template so that only the unit that launches it compiles it)
__global__ __launch_bounds__(BLOCK) void k_count_diff_masked(unsigned long long* count, const u64* a, const u64* b,
                                                             size_t nwords, u64 mask_lo, u64 mask_hi, int L) {
  unsigned long long local = 0;
  SCL_GRID_STRIDE(q, nwords) {
    const u64 m = (L == 2 && (q & 1)) ? mask_hi : mask_lo;
    local += ((a[q] ^ b[q]) & m) != 0;
  }
  if (local) atomicAdd(count, local);
}

// ---- Shamir reconstruct ----------------------------------------------------------------------
// shamirRecoverP with a hoisted basis (shamir.h:81-104, lagrange.h:54-71, vector.h:45-52):
// out[s] = sum_i lambda[i] * shares[i][s].  M rows fully unrolled, lambda in scalar registers,
// all M loads issued before the first multiply.
//
// Launch geometry (tools/streambench.hip on plain allocations, profiles/r2_streambench_*.txt): SINGLE-WAVE workgroups
// (BLK = 64) and at most 8 of them resident per CU -- the host passes a dynamic LDS size that nothing reads, only to cap
// the residency.  Each resident wave keeps m + 1 DRAM streams open; with 32 waves per CU the streams evict each other's
// open rows, with 8 the CU still has 80 KiB of loads in flight (enough for HBM latency) and the kernel gains 6-7 % on
// every allocation tried (1.52 / 1.49 -> 1.41 / 1.40 ms at (10,3), 10^8 secrets); 4 waves per CU lose.  The same cap is
// available to k_share_small ("share_waves"), where it only pays with n and t compiled in.
template <class F, int VEC, int M, bool NT, int BLK = BLOCK>
__global__ __launch_bounds__(BLK) void k_recover_fixed(typename F::Ctx ctx, u64* out, const u64* shares,
                                                       size_t stride, Table<F> lam, size_t npacks) {
  for (size_t q = (size_t)blockIdx.x * BLK + threadIdx.x; q < npacks; q += (size_t)gridDim.x * BLK) {
    const size_t off = q * VEC * F::LIMBS;
    Pack<F, VEC> x[M];
#pragma unroll
    for (int i = 0; i < M; ++i) x[i] = load_pack<F, VEC, NT>(shares + (size_t)i * stride * F::LIMBS + off);
    Pack<F, VEC> r;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      typename F::Acc acc = F::acc_zero();
#pragma unroll
      for (int i = 0; i < M; ++i) F::mac(ctx, acc, lam.v[i], x[i].v[v]);
      r.v[v] = F::acc_fold(ctx, acc);
    }
    store_pack<F, VEC, NT>(out + off, r);
  }
}

// shamirRecoverP at the default nodes 1..n, x = 0, over a Montgomery field: there the Lagrange coefficients are the signed
// binomials lambda_i = (-1)^(i-1) C(n, i) -- prod_{j != i} j / (j - i) = (n! / i) / ((-1)^(i-1) (i-1)! (n-i)!) -- i.e. SMALL
// integers, and a residue x R times a plain integer v stays in Montgomery form ((x R) v = (x v) R): the sum needs no Montgomery
// product at all.  Positive and negative coefficients go to two lazy sums of limb x v products (the small-node accumulator of
// the share kernels, SAcc: one 32 x 32 multiply-add per 32-bit limb and term against the 64 + 64 of a full lazy product over
// secp256k1), each reduced once, then one modular subtraction -- the same residue shamirRecoverP reaches (shamir.h:81-104).
// The host takes this path whenever every coefficient, as an integer or as p minus one, is below 2^32 and both signed sums are
// (any nodes and evaluation point that happen to give such coefficients, not just the default ones).
struct SmallLam {
  u32 v[FIXED_M_MAX];  // |lambda_i|
  u32 neg;             // bit i: lambda_i = p - v[i]
};

template <class F, int M, bool NT, int BLK>
__global__ __launch_bounds__(BLK) void k_recover_small(typename F::Ctx ctx, u64* out, const u64* shares, size_t stride, SmallLam lam,
                                                       size_t n) {
  typedef typename F::E E;
  if constexpr (F::LIMBS == 4) {
    // 32-byte elements: a PAIR of lanes per secret, lane h of the pair holding limbs 2h, 2h + 1 -- a wave's loads and stores are
    // then 1 KiB of consecutive bytes each (a lane per element reads 16 bytes out of every 32 twice over), and the ten shares
    // cost a lane 40 registers instead of 80.  The limb sums are linear, so each lane accumulates its own four 32-bit columns;
    // the halves meet over DPP (quad_perm 1,0,3,2) before the two reductions, which both lanes run, each storing its half.
    for (size_t q = (size_t)blockIdx.x * BLK + threadIdx.x; q < 2 * n; q += (size_t)gridDim.x * BLK) {
      const int h = (int)(q & 1);
      const size_t off = (q >> 1) * 4 + 2 * h;
      u64x2 x[M];
#pragma unroll
      for (int i = 0; i < M; ++i) x[i] = ldg<NT>(reinterpret_cast<const u64x2*>(shares + (size_t)i * stride * 4 + off));
      u64 pos[4] = {0, 0, 0, 0}, neg[4] = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < M; ++i) {
        auto term = [&](u64 (&acc)[4]) {
          mad32(acc[0], (u32)x[i].x, lam.v[i]);
          mad32(acc[1], (u32)(x[i].x >> 32), lam.v[i]);
          mad32(acc[2], (u32)x[i].y, lam.v[i]);
          mad32(acc[3], (u32)(x[i].y >> 32), lam.v[i]);
        };
        if ((lam.neg >> i) & 1u) term(neg);  // (wave-uniform: a scalar branch)
        else term(pos);
      }
      auto partner = [](u64 v) {
        const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, 0xB1, 0xF, 0xF, false);
        const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), 0xB1, 0xF, 0xF, false);
        return ((u64)hi << 32) | lo;
      };
      typename F::SAcc sp, sn;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u64 pp = partner(pos[j]), pn = partner(neg[j]);
        sp.a[j] = h ? pp : pos[j];
        sp.a[4 + j] = h ? pos[j] : pp;
        sn.a[j] = h ? pn : neg[j];
        sn.a[4 + j] = h ? neg[j] : pn;
      }
      const E r = F::sub(ctx, F::sacc_fold(sp, F::zero()), F::sacc_fold(sn, F::zero()));
      u64x2 o;
      o.x = h ? r.w[2] : r.w[0];
      o.y = h ? r.w[3] : r.w[1];
      stg<NT>(reinterpret_cast<u64x2*>(out + off), o);
    }
  } else {
    for (size_t q = (size_t)blockIdx.x * BLK + threadIdx.x; q < n; q += (size_t)gridDim.x * BLK) {
      const size_t off = q * F::LIMBS;
      Pack<F, 1> x[M];
#pragma unroll
      for (int i = 0; i < M; ++i) x[i] = load_pack<F, 1, NT>(shares + (size_t)i * stride * F::LIMBS + off);
      typename F::SAcc pos, neg;
      F::sacc_zero(pos);
      F::sacc_zero(neg);
#pragma unroll
      for (int i = 0; i < M; ++i) {
        if ((lam.neg >> i) & 1u) F::sacc_mac(neg, x[i].v[0], lam.v[i]);  // (wave-uniform: a scalar branch)
        else F::sacc_mac(pos, x[i].v[0], lam.v[i]);
      }
      Pack<F, 1> r;
      r.v[0] = F::sub(ctx, F::sacc_fold(ctx, pos, F::zero()), F::sacc_fold(ctx, neg, F::zero()));
      store_pack<F, 1, NT>(out + off, r);
    }
  }
}

// Any m <= BigTable::CAP: lambda staged in LDS, rows consumed 8 at a time.
// prev != nullptr: out = prev + this block of parties' terms -- how more parties than one table holds are summed
// over several launches (canonical partial sums add exactly)
template <class F, int VEC, bool NT>
__global__ __launch_bounds__(BLOCK) void k_recover_table(typename F::Ctx ctx, u64* out, const u64* shares,
                                                         size_t stride, BigTable<F> tab, int m, size_t npacks,
                                                         const u64* prev) {
  __shared__ typename F::E lam[BigTable<F>::CAP];
  for (int i = threadIdx.x; i < m; i += BLOCK) lam[i] = tab.v[i];
  __syncthreads();
  SCL_GRID_STRIDE(q, npacks) {
    const size_t off = q * VEC * F::LIMBS;
    typename F::Acc acc[VEC];
    typename F::E run[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      acc[v] = F::acc_zero();
      run[v] = F::zero();
    }
    if (prev) {
      const Pack<F, VEC> p0 = load_pack<F, VEC, false>(prev + off);
#pragma unroll
      for (int v = 0; v < VEC; ++v) run[v] = p0.v[v];
    }
    int i = 0, terms = 0;
    for (; i + 8 <= m; i += 8) {
      Pack<F, VEC> x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = load_pack<F, VEC, NT>(shares + (size_t)(i + j) * stride * F::LIMBS + off);
      if (terms + 8 > F::ACC_TERMS) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          run[v] = F::add(ctx, run[v], F::acc_fold(ctx, acc[v]));
          acc[v] = F::acc_zero();
        }
        terms = 0;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const typename F::E l = lam[i + j];
#pragma unroll
        for (int v = 0; v < VEC; ++v) F::mac(ctx, acc[v], l, x[j].v[v]);
      }
      terms += 8;
    }
    for (; i < m; ++i) {
      const Pack<F, VEC> x = load_pack<F, VEC, NT>(shares + (size_t)i * stride * F::LIMBS + off);
      if (terms + 1 > F::ACC_TERMS) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          run[v] = F::add(ctx, run[v], F::acc_fold(ctx, acc[v]));
          acc[v] = F::acc_zero();
        }
        terms = 0;
      }
      const typename F::E l = lam[i];
#pragma unroll
      for (int v = 0; v < VEC; ++v) F::mac(ctx, acc[v], l, x.v[v]);
      terms += 1;
    }
    Pack<F, VEC> r;
#pragma unroll
    for (int v = 0; v < VEC; ++v) r.v[v] = F::add(ctx, run[v], F::acc_fold(ctx, acc[v]));
    store_pack<F, VEC, NT>(out + off, r);
  }
}

// GF(2^128) reconstruct.  Multiplying by the wave-uniform constant lambda_i is a 4-bit-window table
// walk: T_i[j] = j(x) * lambda_i for the 16 nibble values (256 B per party = one LDS bank row, so a
// ds_read_b128 with data-dependent nibbles is conflict-free).  The x^4 shifts are shared across
// parties: for nibble position k (high to low)  r = r * x^4  ^  XOR_i T_i[nib_k(s_i)].
template <class FieldG = Gf128>  // (a template so that only the unit that launches it compiles it)
__global__ __launch_bounds__(BLOCK) void k_recover_gf128(u64* out, const u64* shares, size_t stride,
                                                         BigTable<Gf128> tab, int m, size_t N, const u64* prev = nullptr) {
  __shared__ u128 T[BigTable<Gf128>::CAP * 16];
  for (int e = threadIdx.x; e < m * 16; e += BLOCK) {
    const u128 l0 = tab.v[e >> 4];
    const u128 l1 = Gf128::mulx(l0), l2 = Gf128::mulx(l1), l3 = Gf128::mulx(l2);
    const int j = e & 15;
    T[e] = (j & 1 ? l0 : (u128)0) ^ (j & 2 ? l1 : (u128)0) ^ (j & 4 ? l2 : (u128)0) ^ (j & 8 ? l3 : (u128)0);
  }
  __syncthreads();
  constexpr int G = 8;
  SCL_GRID_STRIDE(s, N) {
    u128 total = prev ? Gf128::ld(prev + s * 2) : (u128)0;
    for (int i0 = 0; i0 < m; i0 += G) {
      u32 w[G][4];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        if (i0 + j < m) {  // wave-uniform
          const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(shares + ((size_t)(i0 + j) * stride + s) * 2));
          w[j][0] = (u32)v.x;
          w[j][1] = (u32)(v.x >> 32);
          w[j][2] = (u32)v.y;
          w[j][3] = (u32)(v.y >> 32);
        }
      }
      u128 r = 0;
#pragma unroll
      for (int k = 31; k >= 0; --k) {
        r = Gf128::mulx4(r);
#pragma unroll
        for (int j = 0; j < G; ++j) {
          if (i0 + j < m) {
            const u32 nib = (w[j][k >> 3] >> (4 * (k & 7))) & 15u;
            r ^= T[(i0 + j) * 16 + nib];
          }
        }
      }
      total ^= r;
    }
    u64x2 o;
    o.x = (u64)total;
    o.y = (u64)(total >> 64);
    __builtin_nontemporal_store(o, reinterpret_cast<u64x2*>(out + s * 2));
  }
}

// GF(2^128) reconstruct, position tables.  T[i][p][j] = j(x) * lambda_i * x^(4p) for the 8 nibble positions p of a
// 32-bit word: 2 KiB per party (80 KiB at n = 40, two 512-thread workgroups per CU).  The 8 nibbles of word w of every share
// then add up with no shifting of the accumulator at all, and the four word sums are combined by three x^32 steps at the
// very end -- k_recover_gf128 pays 32 x^4 steps per group of 8 parties -- so a group can be small (few registers, many
// waves).  A lookup is one v_add_u32_sdwa (table base + a byte of the word masked to its high or low nibbles), one
// conflict-free ds_read_b128 and four xors.  The LDS reads of batch k + 1 (4 lookups) are in flight while batch k is folded
// (s_waitcnt lgkmcnt(4): LDS returns in order) -- left to itself the compiler waits for each batch right after issuing it;
// rounds 2-3 issued the reads from inline assembly for that, since round 4 they are plain loads between scheduling barriers
// (gfpos_pipe_cv) -- and the next group's shares are fetched while this group is worked on.
// Counters at (40,13), profiles/r2_pmc_gf128.txt: LDS array 68 % busy (4 cycles per ds_read_b128, no bank conflicts),
// vector ALU 60 % (5.8 instructions per lookup); the two overlap only partly, so neither saturates:
// 2.12 -> 2.44 TB/s.  At 100 % of the LDS array the form would reach 4.4 TB/s.
// capi.hip still checks this kernel against k_recover_gf128 once per process before it is used (gfpos_usable: the guard the
// hand-issued reads needed) and falls back to that kernel if the two ever disagree.
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

template <int B>
__device__ __forceinline__ u32 add_byte(u32 base, u32 word) {  // base + byte B of word, one instruction
  u32 d;
  if constexpr (B == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(d) : "v"(base), "v"(word));
  if constexpr (B == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(d) : "v"(base), "v"(word));
  if constexpr (B == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(d) : "v"(base), "v"(word));
  if constexpr (B == 3) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(d) : "v"(base), "v"(word));
  return d;
}
// The same batch with loads the COMPILER sees (plain LDS loads through an address-space-3 pointer): it then tracks the
// destination registers itself and places the s_waitcnt; gfpos_pipe_cv pins the order -- batch IDX + 1 issued, then batch IDX
// folded -- with scheduling barriers, which is all the hand-written form needed the assembly for.
typedef __attribute__((address_space(3))) const u32x4* lds_u32x4_ptr;
template <int G, int IDX>
__device__ __forceinline__ void gfpos_issue4_cv(u32x4 (&buf)[4], u32 gbase, const u32 (&w)[G][4]) {
  constexpr int wd = IDX / (2 * G), j = (IDX / 2) % G, half = IDX & 1;
  const u32 m = half ? ((w[j][wd] << 4) & 0xF0F0F0F0u) : (w[j][wd] & 0xF0F0F0F0u);
  buf[0] = *(lds_u32x4_ptr)(uintptr_t)(add_byte<0>(gbase, m) + (u32)(j * 2048 + (0 + (1 - half)) * 256));
  buf[1] = *(lds_u32x4_ptr)(uintptr_t)(add_byte<1>(gbase, m) + (u32)(j * 2048 + (2 + (1 - half)) * 256));
  buf[2] = *(lds_u32x4_ptr)(uintptr_t)(add_byte<2>(gbase, m) + (u32)(j * 2048 + (4 + (1 - half)) * 256));
  buf[3] = *(lds_u32x4_ptr)(uintptr_t)(add_byte<3>(gbase, m) + (u32)(j * 2048 + (6 + (1 - half)) * 256));
}
// acc ^= the four table entries of a batch: two three-way xors per word (v_bitop3_b32, 3.0 cycles each) where four two-way
// ones stood (2.7 each) -- the xors were two thirds of the kernel's vector instructions
__device__ __forceinline__ void gfpos_fold(u32x4& acc, const u32x4 (&cur)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int c = 0; c < 4; ++c)
    acc[c] = xor3(xor3(cur[0][c], cur[1][c], cur[2][c]), cur[3][c], acc[c]);
#else
  acc ^= (cur[0] ^ cur[1]) ^ (cur[2] ^ cur[3]);
#endif
}
// compiler-visible form of gfpos_pipe: nothing crosses a scheduling barrier, so the four reads of batch IDX + 1 are in
// flight while batch IDX is folded, and the compiler's own wait (lgkmcnt(4): LDS returns in order) covers exactly batch IDX
template <int G, int IDX>
__device__ __forceinline__ void gfpos_pipe_cv(u32x4 (&A)[4], u32x4 (&B)[4], u32x4 (&acc)[4], u32 gbase, const u32 (&w)[G][4]) {
  constexpr int NB = 8 * G;
  if constexpr (IDX < NB) {
    if constexpr (IDX + 1 < NB) gfpos_issue4_cv<G, IDX + 1>((IDX & 1) ? A : B, gbase, w);
    __builtin_amdgcn_sched_barrier(0);
    gfpos_fold(acc[IDX / (2 * G)], (IDX & 1) ? B : A);
    __builtin_amdgcn_sched_barrier(0);
    gfpos_pipe_cv<G, IDX + 1>(A, B, acc, gbase, w);
  }
}
__device__ __forceinline__ void gf_mulx32(u32 (&r)[4]) {  // r * x^32: the word that leaves the top times x^7 + x^2 + x + 1
  const u32 t = r[3];
  r[3] = r[2];
  r[2] = r[1];
  r[1] = r[0] ^ (t >> 31) ^ (t >> 30) ^ (t >> 25);
  r[0] = t ^ (t << 1) ^ (t << 2) ^ (t << 7);
}

// how a group's 8 G batches are read and folded: the policy of k_recover_gf128_pos.  The library's is the compiler-visible
// pipeline above; tools/gfpos_asm.hpp has the hand-issued forms of rounds 2-3 (inline-assembly ds_read_b128 one, two or
// three batches ahead of an s_waitcnt in a later statement) for A/B runs -- same speed within 1.5 %
// (profiles/r4_gfpos_bench.txt), and the compiler could not know their destination registers were still in flight.
struct GfposPipeCV {
  template <int G>
  static __device__ __forceinline__ void group(u32x4 (&acc)[4], u32 gbase, const u32 (&w)[G][4]) {
    u32x4 A[4], B[4];
    gfpos_issue4_cv<G, 0>(A, gbase, w);
    gfpos_pipe_cv<G, 0>(A, B, acc, gbase, w);
  }
};

constexpr int GFPOS_G = 5;  // parties per group: 8 G batches of 4 lookups, 20 share words + 20 prefetched
inline size_t gfpos_lds_bytes(size_t m) { return (m + GFPOS_G - 1) / GFPOS_G * GFPOS_G * 2048; }

template <int BLK, int WPS, class PIPE = GfposPipeCV>
__global__ __launch_bounds__(BLK, WPS) void k_recover_gf128_pos(u64* out, const u64* shares, size_t stride,
                                                                BigTable<Gf128> tab, int m, size_t N) {
  constexpr int G = GFPOS_G;
  extern __shared__ __align__(16) uint4 gfpos_T[];  // [mpad][8][16]; parties m .. mpad-1 are zero tables
  const int mpad = (m + G - 1) / G * G;
  for (int e = threadIdx.x; e < mpad * 128; e += BLK) {
    const int i = e >> 7, p = (e >> 4) & 7, j = e & 15;
    u128 l0 = i < m ? tab.v[i] : (u128)0;
    for (int k = 0; k < p; ++k) l0 = Gf128::mulx4(l0);
    const u128 l1 = Gf128::mulx(l0), l2 = Gf128::mulx(l1), l3 = Gf128::mulx(l2);
    const u128 v = (j & 1 ? l0 : (u128)0) ^ (j & 2 ? l1 : (u128)0) ^ (j & 4 ? l2 : (u128)0) ^ (j & 8 ? l3 : (u128)0);
    gfpos_T[e] = make_uint4((u32)v, (u32)(v >> 32), (u32)(v >> 64), (u32)(v >> 96));
  }
  __syncthreads();
  const u32 tbase = (u32)(uintptr_t)gfpos_T;  // low half of the flat address = the LDS byte address
  auto load_group = [&](u32(&w)[G][4], int i0, size_t s) {
#pragma unroll
    for (int j = 0; j < G; ++j) {
      u64x2 v;
      v.x = v.y = 0;
      if (i0 + j < m) v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(shares + ((size_t)(i0 + j) * stride + s) * 2));
      w[j][0] = (u32)v.x;
      w[j][1] = (u32)(v.x >> 32);
      w[j][2] = (u32)v.y;
      w[j][3] = (u32)(v.y >> 32);
    }
  };
  const size_t step = (size_t)gridDim.x * BLK;
  size_t s = (size_t)blockIdx.x * BLK + threadIdx.x;
  if (s >= N) return;
  u32 w[G][4], wn[G][4];
  load_group(w, 0, s);
  for (;; s += step) {
    u32x4 acc[4] = {0, 0, 0, 0};
    for (int i0 = 0; i0 < m; i0 += G) {
      // the next group's shares are fetched while this group is worked on: the next group of this secret, or -- round 4 --
      // the first group of this lane's NEXT secret (at m = 5, one rank's share of C4 on eight GPUs, every group is a
      // secret's only one and its load latency was exposed once per secret)
      if (i0 + G < m) load_group(wn, i0 + G, s);
      else if (s + step < N) load_group(wn, 0, s + step);
      const u32 gbase = tbase + (u32)i0 * 2048u;
      PIPE::template group<G>(acc, gbase, w);
#pragma unroll
      for (int j = 0; j < G; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) w[j][c] = wn[j][c];
    }
    u32 r[4] = {acc[3].x, acc[3].y, acc[3].z, acc[3].w};
#pragma unroll
    for (int wd = 2; wd >= 0; --wd) {
      gf_mulx32(r);
      r[0] ^= acc[wd].x;
      r[1] ^= acc[wd].y;
      r[2] ^= acc[wd].z;
      r[3] ^= acc[wd].w;
    }
    u64x2 o;
    o.x = (u64)r[0] | ((u64)r[1] << 32);
    o.y = (u64)r[2] | ((u64)r[3] << 32);
    __builtin_nontemporal_store(o, reinterpret_cast<u64x2*>(out + s * 2));
    if (s + step >= N) break;
  }
}

// ---- Shamir share ----------------------------------------------------------------------------
// Horner over TREG+1 register-resident coefficients (poly.h:56-64); only k <= t take part.
// SMALLX: every node is a small integer (< 2^F::SMALL_BITS; true for the default nodes 1..n), held as
// u32 in LDS, and y*x + c uses the field's cheap small-constant form.
template <class F, int VEC, int TREG, bool SMALLX>
__device__ __forceinline__ void horner_rows(const typename F::Ctx& ctx, const Pack<F, VEC> (&c)[TREG + 1], int t,
                                            const typename F::E* alpha_lds, const u32* alpha32_lds, int n,
                                            u64* shares, size_t stride, size_t off) {
  for (int i = 0; i < n; ++i) {
    Pack<F, VEC> y;
    if constexpr (SMALLX) {
      const u32 x = alpha32_lds[i];
#pragma unroll
      for (int k = TREG; k >= 0; --k) {
        if (k == t) {  // wave-uniform
          y = c[k];
        } else if (k < t) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) y.v[v] = F::muladd_small_lazy(ctx, y.v[v], x, c[k].v[v]);  // canonical once, below
        }
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) y.v[v] = F::canon(y.v[v]);
    } else {
      const typename F::E x = alpha_lds[i];
#pragma unroll
      for (int k = TREG; k >= 0; --k) {
        if (k == t) {
          y = c[k];
        } else if (k < t) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) y.v[v] = F::add(ctx, F::mul(ctx, y.v[v], x), c[k].v[v]);
        }
      }
    }
    store_pack<F, VEC, true>(shares + (size_t)i * stride * F::LIMBS + off, y);
  }
}

// node table of a Horner kernel: full elements, or u32 images when SMALLX
template <class F, bool SMALLX, int THREADS = BLOCK>
__device__ __forceinline__ void stage_nodes(const BigTable<F>& tab, int n, typename F::E* alpha, u32* alpha32) {
  for (int i = threadIdx.x; i < n; i += THREADS) {
    if constexpr (SMALLX) alpha32[i] = F::low32(tab.v[i]);
    else alpha[i] = tab.v[i];
  }
  __syncthreads();
}

// ---- GF(2^128) Horner at a node known at compile time ---------------------------------------------------------------------
// Gf128::muladd_small tests the bits of the (wave-uniform) node with a scalar branch each and reduces after every step:
// at (40,13) a Horner step costs 152 SIMD-cycles per wave for ~108 cycles of vector work (7 scalar test-and-branch pairs per
// step; profiles/r2_gf128_node_horner.txt).  With the node A a template parameter a step is straight-line -- one shifted copy
// of y per set bit, no tests -- and y lives in FIVE words: the bits shifted past x^127 collect in the fifth (deg A <= 5 per
// step) and come back down through x^128 = x^7 + x^2 + x + 1 once per 32 / deg A steps instead of every step.
__device__ __forceinline__ void gf_fold5(u32 (&y)[5]) {
  const u32 t = y[4];
  y[0] = xor3(xor3(y[0], t, t << 1), t << 2, t << 7);
  y[1] = xor3(y[1], t >> 31, t >> 30) ^ (t >> 25);
  y[4] = 0;
}
template <u32 A>
__device__ __forceinline__ void gf_node_step(u32 (&y)[5], u128 c) {
  u32 r0 = (u32)c, r1 = (u32)(c >> 32), r2 = (u32)(c >> 64), r3 = (u32)(c >> 96), r4 = 0;
  if constexpr (A & 1u) {
    r0 ^= y[0];
    r1 ^= y[1];
    r2 ^= y[2];
    r3 ^= y[3];
    r4 ^= y[4];
  }
#define SCL_GFN_BIT(B)                                        \
  if constexpr ((A >> B) & 1u) {                              \
    r0 ^= y[0] << B;                                          \
    r1 ^= __builtin_amdgcn_alignbit(y[1], y[0], 32 - B);      \
    r2 ^= __builtin_amdgcn_alignbit(y[2], y[1], 32 - B);      \
    r3 ^= __builtin_amdgcn_alignbit(y[3], y[2], 32 - B);      \
    r4 ^= __builtin_amdgcn_alignbit(y[4], y[3], 32 - B);      \
  }
  SCL_GFN_BIT(1) SCL_GFN_BIT(2) SCL_GFN_BIT(3) SCL_GFN_BIT(4) SCL_GFN_BIT(5) SCL_GFN_BIT(6)
#undef SCL_GFN_BIT
  y[0] = r0;
  y[1] = r1;
  y[2] = r2;
  y[3] = r3;
  y[4] = r4;
}
// The coefficients of one secret as two 32-word register vectors (word 4k + j of coefficient k: k < 8 in lo, 8 <= k < 16 in
// hi), so that the Horner loop can stay a LOOP: a wave-uniform k indexes them through the register-index mode
// (s_set_gpr_idx / v_movrel), four moves per step.  Why a loop: with the steps unrolled, the code of one (node, threshold)
// pair is 2.6 KB that a wave runs once per party -- 63 nodes of it is 1.4 MB per kernel, every wave streams 100 KB of
// instructions per secret and the kernel becomes instruction-fetch-bound (10.0 ms against 6.3 for the branchy form, at any
// occupancy; profiles/r2_gf128_node_horner.txt).  Rolled, a node's code is ~0.5 KB and all 63 stay in the instruction cache.
typedef u32 v32u __attribute__((ext_vector_type(32)));
struct GfCoeffs {
  v32u lo, hi;
};
// coefficient k of the pack at `off`: the secret for k = 0, row k - 1 of the coefficient matrix above that
__device__ __forceinline__ u128 gf_load_coeff(const u64* secrets, const u64* coeffs, size_t cstride, size_t off, int k) {
  const Pack<Gf128, 1> p = k == 0 ? load_pack<Gf128, 1, true>(secrets + off)
                                  : load_pack<Gf128, 1, true>(coeffs + (size_t)(k - 1) * cstride * Gf128::LIMBS + off);
  return p.v[0];
}
template <int T>
__device__ __forceinline__ GfCoeffs gf_load_coeffs(const u64* secrets, const u64* coeffs, size_t cstride, size_t off) {
  GfCoeffs cf;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const u128 v = k < T ? gf_load_coeff(secrets, coeffs, cstride, off, k) : (u128)0;
    if (k < 8) {
      cf.lo[4 * k] = (u32)v;
      cf.lo[4 * k + 1] = (u32)(v >> 32);
      cf.lo[4 * k + 2] = (u32)(v >> 64);
      cf.lo[4 * k + 3] = (u32)(v >> 96);
    } else {
      cf.hi[4 * (k - 8)] = (u32)v;
      cf.hi[4 * (k - 8) + 1] = (u32)(v >> 32);
      cf.hi[4 * (k - 8) + 2] = (u32)(v >> 64);
      cf.hi[4 * (k - 8) + 3] = (u32)(v >> 96);
    }
  }
  return cf;
}
template <int T, u32 A>
__device__ __forceinline__ u128 gf_horner_node(u128 top, const GfCoeffs& cf) {
  static_assert(A >= 1 && A < 64 && T >= 1 && T <= 16, "nodes of degree <= 5, at most 16 steps");
  constexpr int DEG = A >= 32 ? 5 : A >= 16 ? 4 : A >= 8 ? 3 : A >= 4 ? 2 : A >= 2 ? 1 : 0;
  constexpr int EVERY = DEG ? 32 / DEG : (1 << 30);  // steps the fifth word can take before it must come down
  u32 y[5] = {(u32)top, (u32)(top >> 32), (u32)(top >> 64), (u32)(top >> 96), 0};
  int since = 0;
  if constexpr (T > 8) {
#pragma unroll 1
    for (int b = 4 * (T - 9); b >= 0; b -= 4) {
      const u128 ck = (u128)cf.hi[b] | ((u128)cf.hi[b + 1] << 32) | ((u128)cf.hi[b + 2] << 64) | ((u128)cf.hi[b + 3] << 96);
      gf_node_step<A>(y, ck);
      if (++since == EVERY) {
        gf_fold5(y);
        since = 0;
      }
    }
  }
#pragma unroll 1
  for (int b = 4 * ((T < 8 ? T : 8) - 1); b >= 0; b -= 4) {
    const u128 ck = (u128)cf.lo[b] | ((u128)cf.lo[b + 1] << 32) | ((u128)cf.lo[b + 2] << 64) | ((u128)cf.lo[b + 3] << 96);
    gf_node_step<A>(y, ck);
    if (++since == EVERY) {
      gf_fold5(y);
      since = 0;
    }
  }
  gf_fold5(y);
  return (u128)y[0] | ((u128)y[1] << 32) | ((u128)y[2] << 64) | ((u128)y[3] << 96);
}
// the share at node a (wave-uniform): nodes 1..63 through their own code, anything else by the tested-bits form
template <int T>
__device__ __forceinline__ u128 gf_horner_at(u32 a, u128 top, const GfCoeffs& cf) {
#if defined(__HIP_DEVICE_COMPILE__)
  a = __builtin_amdgcn_readfirstlane(a);
#endif
  switch (a) {
#define SCL_GFN(A) \
  case A:          \
    return gf_horner_node<T, A>(top, cf);
    SCL_GFN(1) SCL_GFN(2) SCL_GFN(3) SCL_GFN(4) SCL_GFN(5) SCL_GFN(6) SCL_GFN(7) SCL_GFN(8) SCL_GFN(9) SCL_GFN(10) SCL_GFN(11) SCL_GFN(12) SCL_GFN(13) SCL_GFN(14) SCL_GFN(15) SCL_GFN(16) SCL_GFN(17) SCL_GFN(18) SCL_GFN(19) SCL_GFN(20) SCL_GFN(21) SCL_GFN(22) SCL_GFN(23) SCL_GFN(24) SCL_GFN(25) SCL_GFN(26) SCL_GFN(27) SCL_GFN(28) SCL_GFN(29) SCL_GFN(30) SCL_GFN(31) SCL_GFN(32) SCL_GFN(33) SCL_GFN(34) SCL_GFN(35) SCL_GFN(36) SCL_GFN(37) SCL_GFN(38) SCL_GFN(39) SCL_GFN(40) SCL_GFN(41) SCL_GFN(42) SCL_GFN(43) SCL_GFN(44) SCL_GFN(45) SCL_GFN(46) SCL_GFN(47) SCL_GFN(48) SCL_GFN(49) SCL_GFN(50) SCL_GFN(51) SCL_GFN(52) SCL_GFN(53) SCL_GFN(54) SCL_GFN(55) SCL_GFN(56) SCL_GFN(57) SCL_GFN(58) SCL_GFN(59) SCL_GFN(60) SCL_GFN(61) SCL_GFN(62) SCL_GFN(63)
#undef SCL_GFN
    default: break;
  }
  u128 y = top;
#pragma unroll
  for (int k = T - 1; k >= 0; --k) {
    const u128 ck = k < 8 ? ((u128)cf.lo[4 * k] | ((u128)cf.lo[4 * k + 1] << 32) | ((u128)cf.lo[4 * k + 2] << 64) | ((u128)cf.lo[4 * k + 3] << 96))
                          : ((u128)cf.hi[4 * (k - 8)] | ((u128)cf.hi[4 * (k - 8) + 1] << 32) | ((u128)cf.hi[4 * (k - 8) + 2] << 64) |
                             ((u128)cf.hi[4 * (k - 8) + 3] << 96));
    y = Gf128::muladd_small_lazy(y, a, ck);
  }
  return y;
}
// the GF(2^128) form of horner_pack_exact at small nodes
template <int T>
__device__ __forceinline__ void gf_horner_pack_exact(u64* shares, size_t stride, const u64* secrets, const u64* coeffs,
                                                     size_t cstride, const u32* alpha32_lds, int n, size_t off) {
  const GfCoeffs cf = gf_load_coeffs<T>(secrets, coeffs, cstride, off);
  const u128 top = gf_load_coeff(secrets, coeffs, cstride, off, T);
  for (int i = 0; i < n; ++i) {
    Pack<Gf128, 1> y;
    y.v[0] = gf_horner_at<T>(alpha32_lds[i], top, cf);
    store_pack<Gf128, 1, true>(shares + (size_t)i * stride * Gf128::LIMBS + off, y);
  }
}

// One pack with the threshold T known at compile time: the party loop carries no per-term control flow (the
// wave-uniform "k <= t" tests of horner_rows cost about as much as the arithmetic at t ~ 10).
template <class F, int VEC, int T, bool SMALLX>
__device__ __forceinline__ void horner_pack_exact(const typename F::Ctx& ctx, u64* shares, size_t stride,
                                                  const u64* secrets, const u64* coeffs, size_t cstride,
                                                  const typename F::E* alpha_lds, const u32* alpha32_lds, int n,
                                                  size_t off) {
  Pack<F, VEC> c[T + 1];
  c[0] = load_pack<F, VEC, true>(secrets + off);
#pragma unroll
  for (int k = 1; k <= T; ++k) c[k] = load_pack<F, VEC, true>(coeffs + (size_t)(k - 1) * cstride * F::LIMBS + off);
  for (int i = 0; i < n; ++i) {
    Pack<F, VEC> y = c[T];
    if constexpr (SMALLX) {
      const u32 x = alpha32_lds[i];
#pragma unroll
      for (int k = T - 1; k >= 0; --k) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) y.v[v] = F::muladd_small_lazy(ctx, y.v[v], x, c[k].v[v]);
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) y.v[v] = F::canon(y.v[v]);
    } else {
      const typename F::E x = alpha_lds[i];
#pragma unroll
      for (int k = T - 1; k >= 0; --k) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) y.v[v] = F::add(ctx, F::mul(ctx, y.v[v], x), c[k].v[v]);
      }
    }
    store_pack<F, VEC, true>(shares + (size_t)i * stride * F::LIMBS + off, y);
  }
}

// shamirSecretShare over GF(2^128) at small nodes (every node < 2^16 as a bit pattern), 5 <= t <= 16: per-node Horner code
// over register-indexed coefficients (gf_horner_at).  Four waves per SIMD asked of the register allocator: the two
// 32-word coefficient vectors and a dozen temporaries need no more.
template <class FieldG = Gf128>  // (a template so that only the unit that launches it compiles it)
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_share_gf_nodes(
    u64* shares, size_t stride, const u64* secrets, const u64* coeffs, size_t cstride, BigTable<Gf128> tab, int t, int n,
    size_t npacks) {
  __shared__ u32 alpha32[BigTable<Gf128>::CAP];
  for (int i = threadIdx.x; i < n; i += BLOCK) alpha32[i] = Gf128::low32(tab.v[i]);
  __syncthreads();
  SCL_GRID_STRIDE(q, npacks) {
    const size_t off = q * Gf128::LIMBS;
#define SCL_GFX_CASE(T) \
  case T: gf_horner_pack_exact<T>(shares, stride, secrets, coeffs, cstride, alpha32, n, off); break;
    switch (t) {  // wave-uniform
      SCL_GFX_CASE(5) SCL_GFX_CASE(6) SCL_GFX_CASE(7) SCL_GFX_CASE(8) SCL_GFX_CASE(9) SCL_GFX_CASE(10) SCL_GFX_CASE(11)
      SCL_GFX_CASE(12) SCL_GFX_CASE(13) SCL_GFX_CASE(14) SCL_GFX_CASE(15) SCL_GFX_CASE(16)
      default: break;
    }
#undef SCL_GFX_CASE
  }
}

// shamirSecretShare over GF(2^128) at the DEFAULT nodes (party i at the bit pattern of i + 1, n <= 64), 5 <= t <= 16.
//
// k_share_gf_nodes above runs one node's Horner steps, then the next node's: per step it pays the four register-indexed
// moves that fetch c_k, the loop's scalar bookkeeping and a switch per node -- 18.4 scalar + 4 branch instructions beside
// 31.8 vector ones per step, 121 SIMD-cycles against ~99 of vector work (profiles/r2_pmc_round_end.txt).  Three changes here:
//
// 1. EIGHT nodes advance together: the nodes 8H .. 8H+7 of a tile are compile-time constants, a loop iteration fetches
//    its coefficients once and applies them to eight accumulators -- eight independent straight-line steps, the fetch and
//    the loop control amortised over them.  A tile's code is a few KB; all tiles of a threshold stay in the instruction cache.
// 2. Horner in a^2 over coefficient PAIRS:  y <- y * a^2 + (c_2B + a * c_2B+1).  Squaring is linear in characteristic 2,
//    a^2 = sum_b a_b x^(2b) has as few terms as a, so the outer product costs what a plain Horner step costs (one shifted
//    copy of y per set bit: 1 v_lshlrev + 4 v_alignbit at 4.5 cycles each + 5 v_xor at 2.6) -- but there are half as
//    many of them: at (40,13) 6 outer products per node instead of 13.
// 3. The inner term g(a) = c_2B + a * c_2B+1 is LINEAR in the bits of a, and the tile's nodes are 8H + l, l = 0 .. 7: with
//    base = c_2B + (H x^3) * c_2B+1 (built once per tile and pair) the eight values are base + {0, d0, d1, d0+d1, ..}
//    over d_e = c_2B+1 << e, e = 0, 1, 2 -- walked in Gray-code order, ONE five-word xor per node, no shifts.
// The fifth accumulator word takes the bits pushed past x^127 (2 deg a per outer step) and comes down through
// x^128 = x^7 + x^2 + x + 1 once per (32 - deg) / (2 deg) steps, for the tile as a whole.
template <u32 A>
__device__ __forceinline__ void gf_outer_step(u32 (&y)[5], const u32 (&g)[5]) {  // y <- y * A^2 + g  (A^2: bit b of A at 2b)
  // g and one copy of y per set bit of A, added up with THREE-way xors (v_bitop3_b32): the shifted copies (bits 1 .. 6) first,
  // copy number idx among them waits in p when idx is even and goes in together with the next one; the unshifted copy (bit 0)
  // comes last and pairs with a waiting shifted one or goes in alone -- y itself is never copied
  u32 r[5] = {g[0], g[1], g[2], g[3], g[4]}, p[5] = {0, 0, 0, 0, 0};
#define SCL_GFN_OUT(B)                                                       \
  if constexpr ((A >> B) & 1u) {                                             \
    constexpr int idx = __builtin_popcount(A & ((1u << B) - 2u));            \
    const u32 e0 = y[0] << (2 * B);                                          \
    const u32 e1 = __builtin_amdgcn_alignbit(y[1], y[0], 32 - 2 * B);        \
    const u32 e2 = __builtin_amdgcn_alignbit(y[2], y[1], 32 - 2 * B);        \
    const u32 e3 = __builtin_amdgcn_alignbit(y[3], y[2], 32 - 2 * B);        \
    const u32 e4 = __builtin_amdgcn_alignbit(y[4], y[3], 32 - 2 * B);        \
    if constexpr (idx % 2 == 0) {                                            \
      p[0] = e0, p[1] = e1, p[2] = e2, p[3] = e3, p[4] = e4;                 \
    } else {                                                                 \
      r[0] = xor3(r[0], p[0], e0);                                           \
      r[1] = xor3(r[1], p[1], e1);                                           \
      r[2] = xor3(r[2], p[2], e2);                                           \
      r[3] = xor3(r[3], p[3], e3);                                           \
      r[4] = xor3(r[4], p[4], e4);                                           \
    }                                                                        \
  }
  SCL_GFN_OUT(1) SCL_GFN_OUT(2) SCL_GFN_OUT(3) SCL_GFN_OUT(4) SCL_GFN_OUT(5) SCL_GFN_OUT(6)
#undef SCL_GFN_OUT
  constexpr bool waiting = __builtin_popcount(A & 0x7Eu) % 2 == 1;  // a shifted copy still sits in p
#pragma unroll
  for (int w = 0; w < 5; ++w) {
    if constexpr ((A & 1u) && waiting) r[w] = xor3(r[w], p[w], y[w]);
    else if constexpr ((A & 1u) != 0) r[w] ^= y[w];
    else if constexpr (waiting) r[w] ^= p[w];
  }
#pragma unroll
  for (int w = 0; w < 5; ++w) y[w] = r[w];
}
__device__ __forceinline__ void gf_words(u32 (&w)[4], u128 c) {
  w[0] = (u32)c;
  w[1] = (u32)(c >> 32);
  w[2] = (u32)(c >> 64);
  w[3] = (u32)(c >> 96);
}
// g <- c0 + (H x^3) * c1: the part of the inner term every node of tile H shares
template <int H>
__device__ __forceinline__ void gf_tile_base(u32 (&g)[5], const u32 (&c0)[4], const u32 (&c1)[4]) {
  g[0] = c0[0];
  g[1] = c0[1];
  g[2] = c0[2];
  g[3] = c0[3];
  g[4] = 0;
#define SCL_GFN_HB(B)                                               \
  if constexpr ((H >> B) & 1) {                                      \
    g[0] ^= c1[0] << (B + 3);                                        \
    g[1] ^= __builtin_amdgcn_alignbit(c1[1], c1[0], 32 - (B + 3));   \
    g[2] ^= __builtin_amdgcn_alignbit(c1[2], c1[1], 32 - (B + 3));   \
    g[3] ^= __builtin_amdgcn_alignbit(c1[3], c1[2], 32 - (B + 3));   \
    g[4] ^= c1[3] >> (32 - (B + 3));                                 \
  }
  SCL_GFN_HB(0) SCL_GFN_HB(1) SCL_GFN_HB(2) SCL_GFN_HB(3)
#undef SCL_GFN_HB
}
// one pair of coefficients into the eight accumulators of tile H.  OUTER = false: the start value y = g(a) (top pair).
// `nv` (RAGGED tiles only, wave-uniform): the nodes 8H + l with l < nv exist; the others are skipped.
template <int H, bool OUTER, bool RAGGED>
__device__ __forceinline__ void gf_tile_pair(u32 (&y)[8][5], u128 c0v, u128 c1v, int nv) {
  u32 c0[4], c1[4], g[5], d1[5], d2[5];
  gf_words(c0, c0v);
  gf_words(c1, c1v);
  gf_tile_base<H>(g, c0, c1);
  d1[0] = c1[0] << 1;
  d1[1] = __builtin_amdgcn_alignbit(c1[1], c1[0], 31);
  d1[2] = __builtin_amdgcn_alignbit(c1[2], c1[1], 31);
  d1[3] = __builtin_amdgcn_alignbit(c1[3], c1[2], 31);
  d1[4] = c1[3] >> 31;
  d2[0] = c1[0] << 2;
  d2[1] = __builtin_amdgcn_alignbit(c1[1], c1[0], 30);
  d2[2] = __builtin_amdgcn_alignbit(c1[2], c1[1], 30);
  d2[3] = __builtin_amdgcn_alignbit(c1[3], c1[2], 30);
  d2[4] = c1[3] >> 30;
  auto node = [&](auto L) {  // g holds g(8H + l)
    constexpr int l = decltype(L)::value;
    if constexpr (H == 0 && l == 0) {
      return;  // node 0 is nobody's
    } else {
      if (RAGGED && l >= nv) return;
      if constexpr (OUTER) {
        gf_outer_step<(u32)(8 * H + l)>(y[l], g);
      } else {
#pragma unroll
        for (int w = 0; w < 5; ++w) y[l][w] = g[w];
      }
    }
  };
  auto add0 = [&]() {
    g[0] ^= c1[0];
    g[1] ^= c1[1];
    g[2] ^= c1[2];
    g[3] ^= c1[3];
  };
  auto add = [&](const u32(&d)[5]) {
#pragma unroll
    for (int w = 0; w < 5; ++w) g[w] ^= d[w];
  };
  // Gray-code walk over l: 0, 1, 3, 2, 6, 7, 5, 4
  node(std::integral_constant<int, 0>{});
  add0();
  node(std::integral_constant<int, 1>{});
  add(d1);
  node(std::integral_constant<int, 3>{});
  add0();
  node(std::integral_constant<int, 2>{});
  add(d2);
  node(std::integral_constant<int, 6>{});
  add0();
  node(std::integral_constant<int, 7>{});
  add(d1);
  node(std::integral_constant<int, 5>{});
  add0();
  node(std::integral_constant<int, 4>{});
}
__device__ __forceinline__ void gf_tile_fold(u32 (&y)[8][5]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) gf_fold5(y[j]);
}
// Coefficients 0 .. 7 of a lane's secret wait in LDS (a 16-byte slot per lane and coefficient that only the lane itself
// touches: 32 KiB per 256-thread workgroup, four workgroups = four waves per SIMD per CU), those above in a 32-word
// register vector read through the register-index mode: with all sixteen in registers beside 40 accumulator words the
// kernel spills (98 registers at a budget of 128).
template <int T, int H, bool RAGGED>
__device__ __forceinline__ void gf_tile(u64* shares, size_t stride, size_t off, int n, u128 top, const v32u& hi, const uint4* lo) {
  constexpr int TOPNODE = 8 * H + 7;
  constexpr int DEG = TOPNODE >= 64 ? 6 : TOPNODE >= 32 ? 5 : TOPNODE >= 16 ? 4 : TOPNODE >= 8 ? 3 : 2;
  // outer products the fifth word can take before it must come down: from the start value (degree <= 127 + DEG) and from a
  // folded one (degree <= 127); each adds 2 DEG bits and the word holds 32
  constexpr int EVERY0 = (32 - DEG) / (2 * DEG), EVERY1 = 32 / (2 * DEG);
  // coefficients c_0 .. c_T; pairs (c_2B, c_2B+1); with T even the top coefficient starts the recurrence alone
  constexpr int BTOP = (T % 2 == 0 ? T / 2 : (T - 1) / 2) - 1;  // the highest pair below the start value
  const int nv = n - 8 * H + 1;                                 // nodes 8H + l <= n  <=>  l < nv
  u32 y[8][5];
  auto hi_coeff = [&](int k) {  // 8 <= k < 16, wave-uniform
    const int b = 4 * (k - 8);
    return (u128)hi[b] | ((u128)hi[b + 1] << 32) | ((u128)hi[b + 2] << 64) | ((u128)hi[b + 3] << 96);
  };
  auto lo_coeff = [&](int k) {  // k < 8
    const uint4 w = lo[k * BLOCK];
    return (u128)w.x | ((u128)w.y << 32) | ((u128)w.z << 64) | ((u128)w.w << 96);
  };
  if constexpr (T % 2 == 0) {  // y = c_T
    u32 c[4];
    gf_words(c, top);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      y[j][0] = c[0];
      y[j][1] = c[1];
      y[j][2] = c[2];
      y[j][3] = c[3];
      y[j][4] = 0;
    }
  } else {  // y = c_(T-1) + a * c_T: the top pair, no outer product
    constexpr int k = T - 1;
    u128 ck;
    if constexpr (k >= 8) ck = hi_coeff(k);
    else ck = lo_coeff(k);
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j][0] = y[j][1] = y[j][2] = y[j][3] = y[j][4] = 0;
    gf_tile_pair<H, false, RAGGED>(y, ck, top, nv);
  }
  int left = EVERY0;  // outer products until the next fold
  if constexpr (BTOP >= 4) {
#pragma unroll 1
    for (int B = BTOP; B >= 4; --B) {
      gf_tile_pair<H, true, RAGGED>(y, hi_coeff(2 * B), hi_coeff(2 * B + 1), nv);
      if (--left == 0) {
        gf_tile_fold(y);
        left = EVERY1;
      }
    }
  }
#pragma unroll 1
  for (int B = (BTOP < 3 ? BTOP : 3); B >= 0; --B) {
    gf_tile_pair<H, true, RAGGED>(y, lo_coeff(2 * B), lo_coeff(2 * B + 1), nv);
    if (--left == 0 && B > 0) {  // (the last pair is followed by the final fold)
      gf_tile_fold(y);
      left = EVERY1;
    }
  }
  gf_tile_fold(y);
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    if (8 * H + l >= 1 && (!RAGGED || l < nv)) {  // wave-uniform
      Pack<Gf128, 1> r;
      r.v[0] = (u128)y[l][0] | ((u128)y[l][1] << 32) | ((u128)y[l][2] << 64) | ((u128)y[l][3] << 96);
      store_pack<Gf128, 1, true>(shares + (size_t)(8 * H + l - 1) * stride * Gf128::LIMBS + off, r);
    }
  }
}
template <int T, int H>
__device__ __forceinline__ void gf_tile_any(u64* shares, size_t stride, size_t off, int n, u128 top, const v32u& hi, const uint4* lo) {
  if (n >= 8 * H + 7) gf_tile<T, H, false>(shares, stride, off, n, top, hi, lo);
  else if (n >= 8 * H && (H > 0 || n >= 1)) gf_tile<T, H, true>(shares, stride, off, n, top, hi, lo);
}
template <int T>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_share_gf_tiles(
    u64* shares, size_t stride, const u64* secrets, const u64* coeffs, size_t cstride, int n, size_t npacks) {
  __shared__ uint4 gft_lo[8 * BLOCK];
  uint4* const lo = gft_lo + threadIdx.x;
  SCL_GRID_STRIDE(q, npacks) {
    const size_t off = q * Gf128::LIMBS;
    // the row offsets i * stride of the 40+ stores are loop invariants; hoisted out of this loop they take two scalar
    // registers each and spill (278 scalar spills, ~600 v_readlane / v_writelane per secret).  An opaque copy of the
    // stride per iteration keeps them where they are used: a scalar multiply per row.
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(stride), "+s"(cstride), "+s"(n));  // (n: the "node exists" tests of the ragged tiles, likewise)
#endif
#pragma unroll
    for (int k = 0; k < (T < 8 ? T : 8); ++k) {
      const u128 v = gf_load_coeff(secrets, coeffs, cstride, off, k);
      lo[k * BLOCK] = make_uint4((u32)v, (u32)(v >> 32), (u32)(v >> 64), (u32)(v >> 96));
    }
    v32u hi = {};
#pragma unroll
    for (int k = 8; k < T; ++k) {
      const u128 v = gf_load_coeff(secrets, coeffs, cstride, off, k);
      hi[4 * (k - 8)] = (u32)v;
      hi[4 * (k - 8) + 1] = (u32)(v >> 32);
      hi[4 * (k - 8) + 2] = (u32)(v >> 64);
      hi[4 * (k - 8) + 3] = (u32)(v >> 96);
    }
    const u128 top = gf_load_coeff(secrets, coeffs, cstride, off, T);
    gf_tile_any<T, 0>(shares, stride, off, n, top, hi, lo);
    gf_tile_any<T, 1>(shares, stride, off, n, top, hi, lo);
    gf_tile_any<T, 2>(shares, stride, off, n, top, hi, lo);
    gf_tile_any<T, 3>(shares, stride, off, n, top, hi, lo);
    gf_tile_any<T, 4>(shares, stride, off, n, top, hi, lo);
    gf_tile_any<T, 5>(shares, stride, off, n, top, hi, lo);
    gf_tile_any<T, 6>(shares, stride, off, n, top, hi, lo);
    gf_tile_any<T, 7>(shares, stride, off, n, top, hi, lo);
    gf_tile_any<T, 8>(shares, stride, off, n, top, hi, lo);
  }
}

// shamirSecretShare with explicit coefficients (shamir.h:51-68): c_0 = secret, c_k = coeffs[k-1]
template <class F, int VEC, int TREG, bool SMALLX>
__global__ __launch_bounds__(BLOCK) void k_share(typename F::Ctx ctx, u64* shares, size_t stride,
                                                 const u64* secrets, const u64* coeffs, size_t cstride,
                                                 BigTable<F> tab, int t, int n, size_t npacks) {
  __shared__ typename F::E alpha[SMALLX ? 1 : BigTable<F>::CAP];
  __shared__ u32 alpha32[SMALLX ? BigTable<F>::CAP : 1];
  stage_nodes<F, SMALLX>(tab, n, alpha, alpha32);
  if constexpr (TREG == 16 && F::LIMBS <= 2) {  // 5 <= t <= 16: one specialised body per threshold
    SCL_GRID_STRIDE(q, npacks) {
      const size_t off = q * VEC * F::LIMBS;
#define SCL_HX_CASE(T) \
  case T: horner_pack_exact<F, VEC, T, SMALLX>(ctx, shares, stride, secrets, coeffs, cstride, alpha, alpha32, n, off); break;
      switch (t) {  // wave-uniform
        SCL_HX_CASE(5) SCL_HX_CASE(6) SCL_HX_CASE(7) SCL_HX_CASE(8) SCL_HX_CASE(9) SCL_HX_CASE(10) SCL_HX_CASE(11)
        SCL_HX_CASE(12) SCL_HX_CASE(13) SCL_HX_CASE(14) SCL_HX_CASE(15) SCL_HX_CASE(16)
        default: break;
      }
#undef SCL_HX_CASE
    }
    return;
  }
  SCL_GRID_STRIDE(q, npacks) {
    const size_t off = q * VEC * F::LIMBS;
    Pack<F, VEC> c[TREG + 1];
    c[0] = load_pack<F, VEC, true>(secrets + off);
#pragma unroll
    for (int k = 1; k <= TREG; ++k) {
      if (k <= t) c[k] = load_pack<F, VEC, true>(coeffs + (size_t)(k - 1) * cstride * F::LIMBS + off);
    }
    horner_rows<F, VEC, TREG, SMALLX>(ctx, c, t, alpha, alpha32, n, shares, stride, off);
  }
}

// Vandermonde-row formulation for the Montgomery fields, whose nodes are full-width residues: share_i =
// c_0 + sum_{k=1..t} c_k * alpha_i^k with the powers alpha_i^k (host table in device memory, residues scaled
// by F::table_scale) staged in LDS, the t products of a share summed unreduced in the field's lazy
// accumulator and reduced once -- half the multiplier work of Horner's t Montgomery products.
struct VdmLds {
  enum { WORDS = 4096 };  // 32 KiB: n * t * LIMBS <= WORDS
};

template <class F, int TREG>
__device__ __forceinline__ void vdm_rows(const typename F::Ctx& ctx, const Pack<F, 1> (&c)[TREG + 1], int t,
                                         const u64* vdm_lds, int n, u64* shares, size_t stride, size_t off) {
  for (int i = 0; i < n; ++i) {
    typename F::Acc acc = F::acc_zero();
    F::acc_add(ctx, acc, c[0].v[0]);
    const u64* row = vdm_lds + (size_t)i * t * F::LIMBS;
#pragma unroll
    for (int k = 1; k <= TREG; ++k) {
      if (k <= t) F::mac(ctx, acc, c[k].v[0], F::ld(row + (k - 1) * F::LIMBS));  // wave-uniform
    }
    Pack<F, 1> y;
    y.v[0] = F::acc_fold_scaled(ctx, acc);
    store_pack<F, 1, true>(shares + (size_t)i * stride * F::LIMBS + off, y);
  }
}

template <class F, int TREG>
__global__ __launch_bounds__(BLOCK) void k_share_vdm(typename F::Ctx ctx, u64* shares, size_t stride,
                                                     const u64* secrets, const u64* coeffs, size_t cstride,
                                                     const u64* vdm, int t, int n, size_t N) {
  __shared__ u64 V[VdmLds::WORDS];
  for (int i = threadIdx.x; i < n * t * F::LIMBS; i += BLOCK) V[i] = vdm[i];
  __syncthreads();
  SCL_GRID_STRIDE(q, N) {
    const size_t off = q * F::LIMBS;
    Pack<F, 1> c[TREG + 1];
    c[0] = load_pack<F, 1, true>(secrets + off);
#pragma unroll
    for (int k = 1; k <= TREG; ++k) {
      if (k <= t) c[k] = load_pack<F, 1, true>(coeffs + (size_t)(k - 1) * cstride * F::LIMBS + off);
    }
    vdm_rows<F, TREG>(ctx, c, t, V, n, shares, stride, off);
  }
}

// Vandermonde-row formulation for SMALL nodes (the default nodes 1..n): share_i = sum_k c_k * V[i][k]
// with V[i][k] = alpha_i^k < 2^29 held as u32 in LDS.  A 61/127-bit coefficient times a 29-bit
// constant is one v_mad_u64_u32 per 32-bit limb, accumulated lazily per limb (<= 7 terms cannot
// overflow 64 bits) and folded once per share -- about half the multiplier work of Horner with
// full-width nodes, which is what lets the (10,3) share kernel run at the HBM rate.
struct SmallVdm {
  enum { CAP = 512, TMAX = 7 };
  u32 v[CAP];  // row-major [n][t+1]
};

template <class F>
struct SmallAcc;

template <>
struct SmallAcc<M61> {
  u64 a0, a1;
  __device__ __forceinline__ void init() { a0 = a1 = 0; }
  __device__ __forceinline__ void mac(u64 c, u32 v) {
    a0 += (u64)(u32)c * v;          // c_lo * v < 2^61
    a1 += (u64)(u32)(c >> 32) * v;  // c_hi < 2^29, * v < 2^58
  }
  // c0 + a0 + a1*2^32 mod p, with 2^61 = 1: a1*2^32 = (a1 >> 29) + ((a1 & (2^29-1)) << 32)
  __device__ __forceinline__ u64 fold(const M61::Ctx&, u64 c0) const {
    const u64 P = M61::P;
    const u64 s = (a0 & P) + (a0 >> 61) + c0 + (a1 >> 29) + ((a1 & 0x1FFFFFFFull) << 32);  // < 2^63
    const u64 r = (s & P) + (s >> 61);
    return r >= P ? r - P : r;
  }
};

template <>
struct SmallAcc<M127> {
  u64 a[4];
  __device__ __forceinline__ void init() { a[0] = a[1] = a[2] = a[3] = 0; }
  __device__ __forceinline__ void mac(u128 c, u32 v) {
    a[0] += (u64)(u32)c * v;
    a[1] += (u64)(u32)(c >> 32) * v;
    a[2] += (u64)(u32)(c >> 64) * v;
    a[3] += (u64)(u32)(c >> 96) * v;
  }
  // c0 + sum_j a_j 2^(32j) mod p: the four accumulators overlap by 32 bits, so sum them limb-wise (each limb
  // sum < 2^34), ripple the carries once, then 2^128 = 2 (mod p)
  __device__ __forceinline__ u128 fold(const M127::Ctx&, u128 c0) const {
    u64 l0 = (u64)(u32)a[0] + (u32)c0;
    u64 l1 = (a[0] >> 32) + (u64)(u32)a[1] + (u32)(c0 >> 32);
    u64 l2 = (a[1] >> 32) + (u64)(u32)a[2] + (u32)(c0 >> 64);
    u64 l3 = (a[2] >> 32) + (u64)(u32)a[3] + (u32)(c0 >> 96);
    u64 l4 = (a[3] >> 32);
    l1 += l0 >> 32;
    l2 += l1 >> 32;
    l3 += l2 >> 32;
    l4 += l3 >> 32;  // < 2^33
    const u128 x = (u128)((u64)(u32)l0 | (l1 << 32)) | ((u128)((u64)(u32)l2 | (l3 << 32)) << 64);
    const u128 v = (x & M127::P()) + (x >> 127) + ((u128)l4 << 1);  // < 2^127 + 2^35
    const u128 r = (v & M127::P()) + (v >> 127);
    return r >= M127::P() ? r - M127::P() : r;
  }
};

// The 256-bit Montgomery primes: (x R) * v = (x v) R needs no Montgomery reduction (field.hpp, Mont256::SAcc)
template <class PRM>
struct SmallAcc<Mont256<PRM>> {
  typename Mont256<PRM>::SAcc s;
  __device__ __forceinline__ void init() { Mont256<PRM>::sacc_zero(s); }
  __device__ __forceinline__ void mac(const U256& c, u32 v) { Mont256<PRM>::sacc_mac(s, c, v); }
  __device__ __forceinline__ U256 fold(const typename Mont256<PRM>::Ctx&, const U256& c0) const { return Mont256<PRM>::sacc_fold(s, c0); }
};

// The 128-bit Montgomery prime of full width: the same, with a Barrett step for the one reduction (field.hpp, Mont128::SAcc)
template <>
struct SmallAcc<Mont128> {
  Mont128::SAcc s;
  __device__ __forceinline__ void init() { Mont128::sacc_zero(s); }
  __device__ __forceinline__ void mac(u128 c, u32 v) { Mont128::sacc_mac(s, c, v); }
  __device__ __forceinline__ u128 fold(const Mont128::Ctx& ctx, u128 c0) const { return Mont128::sacc_fold(ctx, s, c0); }
};

template <class F, int VEC>
__device__ __forceinline__ void small_rows(const typename F::Ctx& ctx, const Pack<F, VEC> (&c)[SmallVdm::TMAX + 1],
                                           const u32* V, int t, int n, u64* shares, size_t stride, size_t off) {
  for (int i = 0; i < n; ++i) {
    const u32* row = V + i * (t + 1);
    SmallAcc<F> acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v].init();
#pragma unroll
    for (int k = 1; k <= SmallVdm::TMAX; ++k) {
      if (k <= t) {
        const u32 w = row[k];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v].mac(c[k].v[v], w);
      }
    }
    Pack<F, VEC> y;
#pragma unroll
    for (int v = 0; v < VEC; ++v) y.v[v] = acc[v].fold(ctx, c[0].v[v]);  // V[i][0] = 1
    store_pack<F, VEC, true>(shares + (size_t)i * stride * F::LIMBS + off, y);
  }
}

template <class F, int VEC, int BLK = BLOCK>
__global__ __launch_bounds__(BLK) void k_share_small(typename F::Ctx ctx, u64* shares, size_t stride, const u64* secrets,
                                                     const u64* coeffs, size_t cstride, SmallVdm tab, int t, int n,
                                                     size_t npacks) {
  __shared__ u32 V[SmallVdm::CAP];
  for (int i = threadIdx.x; i < n * (t + 1); i += BLK) V[i] = tab.v[i];
  __syncthreads();
  for (size_t q = (size_t)blockIdx.x * BLK + threadIdx.x; q < npacks; q += (size_t)gridDim.x * BLK) {
    const size_t off = q * VEC * F::LIMBS;
    Pack<F, VEC> c[SmallVdm::TMAX + 1];
    c[0] = load_pack<F, VEC, true>(secrets + off);
#pragma unroll
    for (int k = 1; k <= SmallVdm::TMAX; ++k) {
      if (k <= t) c[k] = load_pack<F, VEC, true>(coeffs + (size_t)(k - 1) * cstride * F::LIMBS + off);
    }
    small_rows<F, VEC>(ctx, c, V, t, n, shares, stride, off);
  }
}

// The same with the threshold T compiled in and single-wave workgroups under a residency cap (see "Launch geometry"
// at k_recover_fixed): with at most 8 waves per CU nothing hides per-term control flow, so the party loop must be
// straight-line -- then the cap gains 2-3 % on allocations the kernel writes fast and 9 % on those it writes slowly
// (2.00 -> 1.83 ms at (10,3), 10^8 secrets; profiles/r2_streambench_pitch_and_regions.txt).  Mersenne61 only: Mersenne127's
// heavier fold wants the occupancy.
template <class F, int VEC, int T, int BLK>
__global__ __launch_bounds__(BLK) void k_share_small_t(typename F::Ctx ctx, u64* shares, size_t stride, const u64* secrets,
                                                       const u64* coeffs, size_t cstride, SmallVdm tab, int n,
                                                       size_t npacks) {
  __shared__ u32 V[SmallVdm::CAP];
  for (int i = threadIdx.x; i < n * (T + 1); i += BLK) V[i] = tab.v[i];
  __syncthreads();
  for (size_t q = (size_t)blockIdx.x * BLK + threadIdx.x; q < npacks; q += (size_t)gridDim.x * BLK) {
    const size_t off = q * VEC * F::LIMBS;
    Pack<F, VEC> c[T + 1];
    c[0] = load_pack<F, VEC, true>(secrets + off);
#pragma unroll
    for (int k = 1; k <= T; ++k) c[k] = load_pack<F, VEC, true>(coeffs + (size_t)(k - 1) * cstride * F::LIMBS + off);
    for (int i = 0; i < n; ++i) {
      const u32* row = V + i * (T + 1);
      SmallAcc<F> acc[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v].init();
#pragma unroll
      for (int k = 1; k <= T; ++k) {
        const u32 w = row[k];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v].mac(c[k].v[v], w);
      }
      Pack<F, VEC> y;
#pragma unroll
      for (int v = 0; v < VEC; ++v) y.v[v] = acc[v].fold(ctx, c[0].v[v]);  // V[i][0] = 1
      store_pack<F, VEC, true>(shares + (size_t)i * stride * F::LIMBS + off, y);
    }
  }
}

// Small-node sharing over the 256-bit fields with a PAIR of lanes per secret (cf. k_recover_small): lane h of the pair holds
// limbs 2h, 2h + 1 of the secret and of every coefficient, so each load and each store of a wave is 1 KiB of consecutive bytes.
// (One lane per 32-byte element writes 16 bytes out of every 32 per instruction: 7 % more bytes written than stored and 4 % more
// read, by the counters -- profiles/pmc_traffic.json, F3 before this kernel -- and 0.64 of peak.)  The parties go two at a time:
// each lane sums its own four 32-bit columns for both, the pair swaps columns over DPP so that lane 0 holds party i's eight and
// lane 1 party i + 1's, each reduces ONE share (as many reductions as a lane per element does; a first form that had both lanes
// reduce every share lost: 0.94 against 0.88 ms, profiles/r5_probe_secp_share_pair_no_gain.txt), and the halves swap back for the stores.
template <class F, int T, int BLK>
__global__ __launch_bounds__(BLK) void k_share_small_pair(typename F::Ctx ctx, u64* shares, size_t stride, const u64* secrets,
                                                          const u64* coeffs, size_t cstride, SmallVdm tab, int n, size_t N) {
  static_assert(F::LIMBS == 4, "lane pairs: 32-byte elements");
  typedef typename F::E E;
  __shared__ u32 V[SmallVdm::CAP];
  for (int i = threadIdx.x; i < n * (T + 1); i += BLK) V[i] = tab.v[i];
  __syncthreads();
  auto partner = [](u64 v) {
    const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, 0xB1, 0xF, 0xF, false);
    const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), 0xB1, 0xF, 0xF, false);
    return ((u64)hi << 32) | lo;
  };
  for (size_t q = (size_t)blockIdx.x * BLK + threadIdx.x; q < 2 * N; q += (size_t)gridDim.x * BLK) {
    const int h = (int)(q & 1);
    const size_t off = (q >> 1) * 4 + 2 * h;
    const u64x2 s = ldg<true>(reinterpret_cast<const u64x2*>(secrets + off));
    u64x2 c[T];
#pragma unroll
    for (int k = 0; k < T; ++k) c[k] = ldg<true>(reinterpret_cast<const u64x2*>(coeffs + (size_t)k * cstride * 4 + off));
    const u64 ox = partner(s.x), oy = partner(s.y);
    const E c0 = h ? F::make(ox, oy, s.x, s.y) : F::make(s.x, s.y, ox, oy);  // the whole secret: V[i][0] = 1
    for (int i = 0; i < n; i += 2) {
      const bool two = i + 1 < n;
      const u32* row0 = V + i * (T + 1);
      const u32* row1 = two ? row0 + (T + 1) : row0;
      u64 col0[4] = {0, 0, 0, 0}, col1[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < T; ++k) {
        const u32 w0 = row0[k + 1], w1 = row1[k + 1];
        const u32 l0 = (u32)c[k].x, l1 = (u32)(c[k].x >> 32), l2 = (u32)c[k].y, l3 = (u32)(c[k].y >> 32);
        mad32(col0[0], l0, w0);
        mad32(col0[1], l1, w0);
        mad32(col0[2], l2, w0);
        mad32(col0[3], l3, w0);
        mad32(col1[0], l0, w1);
        mad32(col1[1], l1, w1);
        mad32(col1[2], l2, w1);
        mad32(col1[3], l3, w1);
      }
      // lane 0 reduces party i, lane 1 party i + 1: each keeps its own columns of "its" party and receives the partner's
      typename F::SAcc sa;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u64 got = partner(h ? col0[j] : col1[j]);  // what the partner needs from me / what I need from it
        sa.a[j] = h ? got : col0[j];
        sa.a[4 + j] = h ? col1[j] : got;
      }
      const E y = F::sacc_fold(sa, c0);  // lane 0: share of party i; lane 1: of party i + 1
      // lane h stores half h of both shares: it keeps half h of its own and receives half h of the partner's
      const u64 gx = partner(h ? y.w[0] : y.w[2]), gy = partner(h ? y.w[1] : y.w[3]);
      u64x2 o0, o1;  // my half of party i's share, of party i + 1's
      o0.x = h ? gx : y.w[0];
      o0.y = h ? gy : y.w[1];
      o1.x = h ? y.w[2] : gx;
      o1.y = h ? y.w[3] : gy;
      stg<true>(reinterpret_cast<u64x2*>(shares + (size_t)i * stride * 4 + off), o0);
      if (two) stg<true>(reinterpret_cast<u64x2*>(shares + (size_t)(i + 1) * stride * 4 + off), o1);
    }
  }
}

// Blocked form of the small-node evaluation for larger t: the polynomial is cut into groups of G coefficients,
//   f(x) = sum_j x^(G j) * g_j(x),   g_j(x) = sum_{r<G} c_{Gj+r} x^r,
// each g_j(x_i) summed lazily against the small powers x_i^r < 2^29 (one v_mad_u64_u32 per 32-bit limb and
// term, as in k_share_small) and folded once, the groups then combined by Horner in x_i^G < 2^32 with the
// field's small-constant step.  Per term that is the bare multiply-accumulate plus 1/G of a fold and a Horner
// step, against a full lazy Horner step per term in k_share<SMALLX>.
struct BlockVdm {
  enum { CAP = 768, TMAX = 16 };
  u32 v[CAP];  // [n][G] powers x_i^0 .. x_i^(G-1), then x_i^G at v[n*G + i]
};

// One pack, exact threshold T known at compile time: no per-term control flow in the party loop.
template <class F, int VEC, int G, int T>
__device__ __forceinline__ void blocked_pack(const typename F::Ctx& ctx, u64* shares, size_t stride, const u64* secrets,
                                             const u64* coeffs, size_t cstride, const u32* V, int n, size_t off) {
  Pack<F, VEC> c[T + 1];
  c[0] = load_pack<F, VEC, true>(secrets + off);
#pragma unroll
  for (int k = 1; k <= T; ++k) c[k] = load_pack<F, VEC, true>(coeffs + (size_t)(k - 1) * cstride * F::LIMBS + off);
  for (int i = 0; i < n; ++i) {
    const u32* row = V + i * G;
    u32 w[G];
#pragma unroll
    for (int r = 1; r < G; ++r) w[r] = row[r];
    const u32 xg = V[n * G + i];
    SmallAcc<F> acc[VEC];
    Pack<F, VEC> y;
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v].init();
#pragma unroll
    for (int k = T; k >= 0; --k) {
      if (k % G != 0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v].mac(c[k].v[v], w[k % G]);
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const typename F::E gv = acc[v].fold(ctx, c[k].v[v]);  // g_j(x_i), canonical
          acc[v].init();
          if (k / G == T / G) y.v[v] = gv;
          else y.v[v] = F::muladd_small_lazy(ctx, y.v[v], xg, gv);
        }
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) y.v[v] = F::canon(y.v[v]);
    store_pack<F, VEC, true>(shares + (size_t)i * stride * F::LIMBS + off, y);
  }
}

template <class F, int VEC, int G>
__global__ __launch_bounds__(BLOCK) void k_share_blocked(typename F::Ctx ctx, u64* shares, size_t stride,
                                                         const u64* secrets, const u64* coeffs, size_t cstride,
                                                         BlockVdm tab, int t, int n, size_t npacks) {
  static_assert(G >= 2 && G <= 8, "at most 7 lazy terms per group (SmallAcc bound)");
  __shared__ u32 V[BlockVdm::CAP];
  for (int i = threadIdx.x; i < n * (G + 1); i += BLOCK) V[i] = tab.v[i];
  __syncthreads();
  SCL_GRID_STRIDE(q, npacks) {
    const size_t off = q * VEC * F::LIMBS;
#define SCL_BLK_CASE(T) \
  case T: blocked_pack<F, VEC, G, T>(ctx, shares, stride, secrets, coeffs, cstride, V, n, off); break;
    switch (t) {  // wave-uniform; BlockVdm::TMAX cases
      SCL_BLK_CASE(1) SCL_BLK_CASE(2) SCL_BLK_CASE(3) SCL_BLK_CASE(4) SCL_BLK_CASE(5) SCL_BLK_CASE(6) SCL_BLK_CASE(7)
      SCL_BLK_CASE(8) SCL_BLK_CASE(9) SCL_BLK_CASE(10) SCL_BLK_CASE(11) SCL_BLK_CASE(12) SCL_BLK_CASE(13)
      SCL_BLK_CASE(14) SCL_BLK_CASE(15) SCL_BLK_CASE(16)
      default: break;
    }
#undef SCL_BLK_CASE
  }
}

// ---- AES-128 for util::PRG (src/scl/util/prg.cc) ------------------------------------------------
// One 256-entry T-table (SubBytes+MixColumns of one byte), the other three by rotation.  In LDS the
// table is replicated 32 times, entry x of copy c at word x*32 + c, and lane l reads copy l & 31:
// every ds_read_b32 of a wave then touches 32 distinct banks per half-wave -- conflict-free for
// arbitrary (data-dependent) indices, where a single shared copy costs ~3.5 LDS cycles per read.
struct AesKey {
  u32 rk[44];    // 11 round keys, little-endian column words (wave-uniform: scalar loads)
  u32 te0[256];  // te0[x] = (2S, S, S, 3S) as bytes 0..3, S = sbox[x]
  u32 r1[4];     // the key-only half of round 1 (aes_key_round1 below)
  u32 hi_same;   // every counter of the launch has the same upper word (aes_key_range below): round 1 reads r1h and
  u32 r1h[4];    // only the lower word's four table entries
};

// The PRG's input block is LE64(counter) || LE64(0x0123456789ABCDEF) (prg.h:34-43): its upper two columns do not depend on the
// counter, so the eight table terms of round 1 that read them, and round key 1, are one constant per key and column:
//   r1[0] = T2[s2.b2] ^ T3[s3.b3] ^ rk[4]     r1[1] = T1[s2.b1] ^ T2[s3.b2] ^ rk[5]
//   r1[2] = T0[s2.b0] ^ T1[s3.b1] ^ rk[6]     r1[3] = T0[s3.b0] ^ T3[s2.b3] ^ rk[7]
// with s2 = 0x89ABCDEF ^ rk[2], s3 = 0x01234567 ^ rk[3], T_r[x] = te0[x] rotated left by 8 r.  Round 1 of a block is then
// eight lookups instead of sixteen (152 per block instead of 160).  Call after rk and te0 are filled.
inline void aes_key_round1(AesKey& k) {
  auto T = [&](int r, u32 word, int byte) {
    const u32 v = k.te0[(word >> (8 * byte)) & 255u];
    return r ? (v << (8 * r)) | (v >> (32 - 8 * r)) : v;
  };
  const u32 s2 = 0x89ABCDEFu ^ k.rk[2], s3 = 0x01234567u ^ k.rk[3];
  k.r1[0] = T(2, s2, 2) ^ T(3, s3, 3) ^ k.rk[4];
  k.r1[1] = T(1, s2, 1) ^ T(2, s3, 2) ^ k.rk[5];
  k.r1[2] = T(0, s2, 0) ^ T(1, s3, 1) ^ k.rk[6];
  k.r1[3] = T(0, s3, 0) ^ T(3, s2, 3) ^ k.rk[7];
  k.hi_same = 0;
  for (int c = 0; c < 4; ++c) k.r1h[c] = k.r1[c];
}

// A launch that draws blocks first .. first + count - 1: when all of them (and the few a kernel rounds its work up by) share
// the upper 32 bits of the counter -- every launch that does not straddle a multiple of 2^32 blocks = 64 GiB of stream -- the
// four table terms of round 1 that read the upper word are constants too:
//   r1h[0] = r1[0] ^ T1[s1.b1]   r1h[1] = r1[1] ^ T0[s1.b0]   r1h[2] = r1[2] ^ T3[s1.b3]   r1h[3] = r1[3] ^ T2[s1.b2]
// with s1 = (first >> 32) ^ rk[1]; round 1 is then four lookups (146 per block).  Otherwise hi_same = 0 and the kernels take
// the eight-lookup round.  Call after aes_key_round1.
inline void aes_key_range(AesKey& k, u64 first, u64 count) {
  const u64 last = first + count + 64;  // slack for rounded-up work; on wrap-around of the 64-bit counter: no folding
  k.hi_same = last >= first && (first >> 32) == (last >> 32);
  if (!k.hi_same) return;
  auto T = [&](int r, u32 word, int byte) {
    const u32 v = k.te0[(word >> (8 * byte)) & 255u];
    return r ? (v << (8 * r)) | (v >> (32 - 8 * r)) : v;
  };
  const u32 s1 = (u32)(first >> 32) ^ k.rk[1];
  k.r1h[0] = k.r1[0] ^ T(1, s1, 1);
  k.r1h[1] = k.r1[1] ^ T(0, s1, 0);
  k.r1h[2] = k.r1[2] ^ T(3, s1, 3);
  k.r1h[3] = k.r1[3] ^ T(2, s1, 2);
}

constexpr int AES_LDS_WORDS = 256 * 32;
constexpr int AES_GRID_CAP = 256 * 4;  // 4 x 34 KiB blocks per CU; the PRG kernels grid-stride

__device__ __forceinline__ u32 rotl32(u32 x, int r) { return (x << r) | (x >> (32 - r)); }

// block = AES( LE64(counter) || LE64(0x0123456789ABCDEF) ) (prg.h:34-43, prg.cc:82-84)
// tl = te0 + (lane & 31): the lane's private copy, entries 32 words apart
__device__ __forceinline__ void aes_ctr_block(const u32* tl, const AesKey& key, u64 counter, u64& out_lo, u64& out_hi) {
#define SCL_T(x) tl[(x) << 5]
  u32 s0 = (u32)counter ^ key.rk[0], s1 = (u32)(counter >> 32) ^ key.rk[1], s2, s3;
  if (key.hi_same) {  // round 1 from the counter's lower word only (aes_key_range; wave-uniform)
    const u32 t0 = SCL_T(s0 & 255) ^ key.r1h[0];
    const u32 t1 = rotl32(SCL_T(s0 >> 24), 24) ^ key.r1h[1];
    const u32 t2 = rotl32(SCL_T((s0 >> 16) & 255), 16) ^ key.r1h[2];
    const u32 t3 = rotl32(SCL_T((s0 >> 8) & 255), 8) ^ key.r1h[3];
    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
  } else {  // round 1: the counter's two columns only (aes_key_round1)
    const u32 t0 = xor3(SCL_T(s0 & 255), rotl32(SCL_T((s1 >> 8) & 255), 8), key.r1[0]);
    const u32 t1 = xor3(SCL_T(s1 & 255), rotl32(SCL_T(s0 >> 24), 24), key.r1[1]);
    const u32 t2 = xor3(rotl32(SCL_T((s0 >> 16) & 255), 16), rotl32(SCL_T(s1 >> 24), 24), key.r1[2]);
    const u32 t3 = xor3(rotl32(SCL_T((s0 >> 8) & 255), 8), rotl32(SCL_T((s1 >> 16) & 255), 16), key.r1[3]);
    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
  }
#pragma unroll
  for (int r = 2; r < 10; ++r) {
    const u32 t0 = xor3(xor3(SCL_T(s0 & 255), rotl32(SCL_T((s1 >> 8) & 255), 8), rotl32(SCL_T((s2 >> 16) & 255), 16)),
                        rotl32(SCL_T(s3 >> 24), 24), key.rk[4 * r + 0]);
    const u32 t1 = xor3(xor3(SCL_T(s1 & 255), rotl32(SCL_T((s2 >> 8) & 255), 8), rotl32(SCL_T((s3 >> 16) & 255), 16)),
                        rotl32(SCL_T(s0 >> 24), 24), key.rk[4 * r + 1]);
    const u32 t2 = xor3(xor3(SCL_T(s2 & 255), rotl32(SCL_T((s3 >> 8) & 255), 8), rotl32(SCL_T((s0 >> 16) & 255), 16)),
                        rotl32(SCL_T(s1 >> 24), 24), key.rk[4 * r + 2]);
    const u32 t3 = xor3(xor3(SCL_T(s3 & 255), rotl32(SCL_T((s0 >> 8) & 255), 8), rotl32(SCL_T((s1 >> 16) & 255), 16)),
                        rotl32(SCL_T(s2 >> 24), 24), key.rk[4 * r + 3]);
    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
  }
  // last round: SubBytes + ShiftRows only; S = byte 1 of te0
#define SCL_SB(x) ((SCL_T(x) >> 8) & 255u)
  const u32 o0 = (SCL_SB(s0 & 255) | (SCL_SB((s1 >> 8) & 255) << 8) | (SCL_SB((s2 >> 16) & 255) << 16) |
                  (SCL_SB(s3 >> 24) << 24)) ^ key.rk[40];
  const u32 o1 = (SCL_SB(s1 & 255) | (SCL_SB((s2 >> 8) & 255) << 8) | (SCL_SB((s3 >> 16) & 255) << 16) |
                  (SCL_SB(s0 >> 24) << 24)) ^ key.rk[41];
  const u32 o2 = (SCL_SB(s2 & 255) | (SCL_SB((s3 >> 8) & 255) << 8) | (SCL_SB((s0 >> 16) & 255) << 16) |
                  (SCL_SB(s1 >> 24) << 24)) ^ key.rk[42];
  const u32 o3 = (SCL_SB(s3 & 255) | (SCL_SB((s0 >> 8) & 255) << 8) | (SCL_SB((s1 >> 16) & 255) << 16) |
                  (SCL_SB(s2 >> 24) << 24)) ^ key.rk[43];
#undef SCL_SB
#undef SCL_T
  out_lo = (u64)o0 | ((u64)o1 << 32);
  out_hi = (u64)o2 | ((u64)o3 << 32);
}

// NB independent blocks in lockstep, one round at a time: 16*NB table reads are in flight per round,
// which is what hides the LDS latency at the 4 waves/SIMD the 32 KiB table allows.  The round loop
// is kept rolled so the body stays small in the instruction cache.
template <int NB>
__device__ __forceinline__ void aes_ctr_multi(const u32* tl, const AesKey& key, const u64 (&ctr)[NB], u64 (&lo)[NB],
                                              u64 (&hi)[NB]) {
#define SCL_T(x) tl[(x) << 5]
  u32 s[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    s[b][0] = (u32)ctr[b] ^ key.rk[0];
    s[b][1] = (u32)(ctr[b] >> 32) ^ key.rk[1];
    s[b][2] = 0x89ABCDEFu ^ key.rk[2];
    s[b][3] = 0x01234567u ^ key.rk[3];
  }
#pragma unroll 1
  for (int r = 1; r < 10; ++r) {
    const u32 k0 = key.rk[4 * r], k1 = key.rk[4 * r + 1], k2 = key.rk[4 * r + 2], k3 = key.rk[4 * r + 3];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const u32 s0 = s[b][0], s1 = s[b][1], s2 = s[b][2], s3 = s[b][3];
      s[b][0] = SCL_T(s0 & 255) ^ rotl32(SCL_T((s1 >> 8) & 255), 8) ^ rotl32(SCL_T((s2 >> 16) & 255), 16) ^
                rotl32(SCL_T(s3 >> 24), 24) ^ k0;
      s[b][1] = SCL_T(s1 & 255) ^ rotl32(SCL_T((s2 >> 8) & 255), 8) ^ rotl32(SCL_T((s3 >> 16) & 255), 16) ^
                rotl32(SCL_T(s0 >> 24), 24) ^ k1;
      s[b][2] = SCL_T(s2 & 255) ^ rotl32(SCL_T((s3 >> 8) & 255), 8) ^ rotl32(SCL_T((s0 >> 16) & 255), 16) ^
                rotl32(SCL_T(s1 >> 24), 24) ^ k2;
      s[b][3] = SCL_T(s3 & 255) ^ rotl32(SCL_T((s0 >> 8) & 255), 8) ^ rotl32(SCL_T((s1 >> 16) & 255), 16) ^
                rotl32(SCL_T(s2 >> 24), 24) ^ k3;
    }
  }
#define SCL_SB(x) ((SCL_T(x) >> 8) & 255u)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const u32 s0 = s[b][0], s1 = s[b][1], s2 = s[b][2], s3 = s[b][3];
    const u32 o0 = (SCL_SB(s0 & 255) | (SCL_SB((s1 >> 8) & 255) << 8) | (SCL_SB((s2 >> 16) & 255) << 16) |
                    (SCL_SB(s3 >> 24) << 24)) ^ key.rk[40];
    const u32 o1 = (SCL_SB(s1 & 255) | (SCL_SB((s2 >> 8) & 255) << 8) | (SCL_SB((s3 >> 16) & 255) << 16) |
                    (SCL_SB(s0 >> 24) << 24)) ^ key.rk[41];
    const u32 o2 = (SCL_SB(s2 & 255) | (SCL_SB((s3 >> 8) & 255) << 8) | (SCL_SB((s0 >> 16) & 255) << 16) |
                    (SCL_SB(s1 >> 24) << 24)) ^ key.rk[42];
    const u32 o3 = (SCL_SB(s3 & 255) | (SCL_SB((s0 >> 8) & 255) << 8) | (SCL_SB((s1 >> 16) & 255) << 16) |
                    (SCL_SB(s2 >> 24) << 24)) ^ key.rk[43];
    lo[b] = (u64)o0 | ((u64)o1 << 32);
    hi[b] = (u64)o2 | ((u64)o3 << 32);
  }
#undef SCL_SB
#undef SCL_T
}

// ---- four tables, no rotations, one instruction per table address -------------------------------------------------
// The three rotations per output word of the single-table form are 20 % of its VALU work (tools/aes_bench.hip: 51.8 ->
// 64.9 G blocks/s), so four tables te_r[x] = rotl(te0[x], 8 r) are kept, each replicated 32 times (a lane reads copy
// lane % 32: no bank conflicts whatever the 64 indices are).  128 KiB of LDS: one workgroup of ABLOCK = 1024 threads per CU
// shares them (dynamic LDS; the kernels that also keep a 32 KiB Vandermonde table in LDS stay on the single-table form).
//
// Layout: a 256-byte row per byte value x = [table 0, copies 0..31 | table 1, copies 0..31]; tables 2 / 3 the same 64 KiB
// higher.  A lane's address register is then (region << 16) | (x << 8) | 4 * copy, and a lookup's whole address
// computation is ONE v_mov_b32_sdwa that drops byte n of the state word into byte 1 of that register
// (dst_unused:UNUSED_PRESERVE keeps the other three); the table select (0 / 128) rides in the ds_read offset field.  The
// compiler's form of te[(s >> 8n) & 255] is v_bfe_u32 + v_lshl_add_u32, two slow-class instructions per lookup
// (tools/oprate.hip): 526 -> 342 vector instructions per block, 66 -> 81 G blocks/s (profiles/r2_aes_sdwa.txt).
// For byte 1 of the register to be the row, the tables must start at LDS address 0: these kernels declare no static
// __shared__ and carve whatever else they stage from the dynamic block above the tables (AES4_EXTRA_BYTES).
constexpr int ABLOCK = 1024;
constexpr int AES4_TABLE_WORDS = 4 * 256 * 32;
constexpr int AES4_EXTRA_BYTES = 4096;
constexpr int AES4_LDS_BYTES = AES4_TABLE_WORDS * 4 + AES4_EXTRA_BYTES;
constexpr int AES4_GRID_CAP = 256;  // one workgroup per CU; the kernels grid-stride

struct Aes4 {
  u32 lane;  // 4 * (the lane's copy)
  __device__ __forceinline__ void block(const AesKey& key, u64 counter, u64& out_lo, u64& out_hi) const;
  // table word at LDS address a + OFF
  template <int OFF>
  static __device__ __forceinline__ u32 ld(u32 a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const __attribute__((address_space(3))) u32*>((uintptr_t)(a + OFF));
#else
    return a + OFF;
#endif
  }
  // byte BYTE of s becomes the row of address register a; the word at table select OFF (0 or 128) of that row
  template <int BYTE, int OFF>
  static __device__ __forceinline__ u32 look(u32& a, u32 s) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (BYTE == 0) asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(a) : "v"(s));
    if constexpr (BYTE == 1) asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1" : "+v"(a) : "v"(s));
    if constexpr (BYTE == 2) asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(a) : "v"(s));
    if constexpr (BYTE == 3) asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(a) : "v"(s));
#else
    a = (a & ~0xFF00u) | (((s >> (8 * BYTE)) & 255u) << 8);
#endif
    return ld<OFF>(a);
  }
};
struct Aes1 {  // the single-table form above behind the same interface
  const u32* tl;
  __device__ __forceinline__ void block(const AesKey& key, u64 counter, u64& out_lo, u64& out_hi) const {
    aes_ctr_block(tl, key, counter, out_lo, out_hi);
  }
};

// one block, rounds unrolled (as aes_ctr_block): consecutive independent blocks of a lane can be interleaved by the
// scheduler.  Four address registers (one per table), each a chain of byte-1 replacements.
__device__ __forceinline__ void Aes4::block(const AesKey& key, u64 counter, u64& out_lo, u64& out_hi) const {
  u32 a0 = lane, a1 = lane, a2 = lane + 65536u, a3 = lane + 65536u;
  u32 s0 = (u32)counter ^ key.rk[0], s1 = (u32)(counter >> 32) ^ key.rk[1], s2, s3;
  if (key.hi_same) {  // round 1 from the counter's lower word only (aes_key_range; wave-uniform)
    const u32 u0 = look<0, 0>(a0, s0) ^ key.r1h[0];
    const u32 u1 = look<3, 128>(a3, s0) ^ key.r1h[1];
    const u32 u2 = look<2, 0>(a2, s0) ^ key.r1h[2];
    const u32 u3 = look<1, 128>(a1, s0) ^ key.r1h[3];
    s0 = u0; s1 = u1; s2 = u2; s3 = u3;
  } else {  // round 1: the counter's two columns only (aes_key_round1)
    const u32 u0 = xor3(look<0, 0>(a0, s0), look<1, 128>(a1, s1), key.r1[0]);
    const u32 u1 = xor3(look<0, 0>(a0, s1), look<3, 128>(a3, s0), key.r1[1]);
    const u32 u2 = xor3(look<2, 0>(a2, s0), look<3, 128>(a3, s1), key.r1[2]);
    const u32 u3 = xor3(look<1, 128>(a1, s0), look<2, 0>(a2, s1), key.r1[3]);
    s0 = u0; s1 = u1; s2 = u2; s3 = u3;
  }
#pragma unroll
  for (int r = 2; r < 10; ++r) {
    // two three-way xors per column (v_bitop3_b32) where four two-way ones stood
    const u32 u0 = xor3(xor3(look<0, 0>(a0, s0), look<1, 128>(a1, s1), look<2, 0>(a2, s2)), look<3, 128>(a3, s3), key.rk[4 * r + 0]);
    const u32 u1 = xor3(xor3(look<0, 0>(a0, s1), look<1, 128>(a1, s2), look<2, 0>(a2, s3)), look<3, 128>(a3, s0), key.rk[4 * r + 1]);
    const u32 u2 = xor3(xor3(look<0, 0>(a0, s2), look<1, 128>(a1, s3), look<2, 0>(a2, s0)), look<3, 128>(a3, s1), key.rk[4 * r + 2]);
    const u32 u3 = xor3(xor3(look<0, 0>(a0, s3), look<1, 128>(a1, s0), look<2, 0>(a2, s1)), look<3, 128>(a3, s2), key.rk[4 * r + 3]);
    s0 = u0; s1 = u1; s2 = u2; s3 = u3;
  }
  // last round (SubBytes + ShiftRows only): te0[x] = (2s, s, s, 3s) from the top byte down, so the S-box byte already sits in
  // byte 0 of table 3, byte 1 of table 0, byte 2 of table 1 and byte 3 of table 2
  // (three bit selects per word -- byte 0 of the first, byte 1 of the second, .. -- where four masks and three ors stood)
  const u32 o0 = bitsel(bitsel(bitsel(look<0, 128>(a3, s0), look<1, 0>(a0, s1), 0xFFu), look<2, 128>(a1, s2), 0xFFFFu),
                        look<3, 0>(a2, s3), 0xFFFFFFu) ^ key.rk[40];
  const u32 o1 = bitsel(bitsel(bitsel(look<0, 128>(a3, s1), look<1, 0>(a0, s2), 0xFFu), look<2, 128>(a1, s3), 0xFFFFu),
                        look<3, 0>(a2, s0), 0xFFFFFFu) ^ key.rk[41];
  const u32 o2 = bitsel(bitsel(bitsel(look<0, 128>(a3, s2), look<1, 0>(a0, s3), 0xFFu), look<2, 128>(a1, s0), 0xFFFFu),
                        look<3, 0>(a2, s1), 0xFFFFFFu) ^ key.rk[42];
  const u32 o3 = bitsel(bitsel(bitsel(look<0, 128>(a3, s3), look<1, 0>(a0, s0), 0xFFu), look<2, 128>(a1, s1), 0xFFFFu),
                        look<3, 0>(a2, s2), 0xFFFFFFu) ^ key.rk[43];
  out_lo = (u64)o0 | ((u64)o1 << 32);
  out_hi = (u64)o2 | ((u64)o3 << 32);
}

// NB independent blocks, each through the unrolled single-block form: straight-line code in which the scheduler runs
// the blocks' table reads ahead of one another (tools/aes_bench.hip: 67 G blocks/s against 62 for the lock-step loop)
template <int NB>
__device__ __forceinline__ void aes4_blocks(const Aes4& a, const AesKey& key, const u64 (&ctr)[NB], u64 (&lo)[NB], u64 (&hi)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b) a.block(key, ctr[b], lo[b], hi[b]);
}

// the workgroup (ABLOCK threads) builds the four replicated tables at the bottom of its dynamic LDS: word e belongs to
// region e >> 14, row (e >> 6) & 255, table 2 * region + ((e >> 5) & 1).  aes4_extra = the AES4_EXTRA_BYTES above them.
#define SCL_AES4_PROLOGUE(key)                                                                     \
  extern __shared__ __align__(16) u32 aes4_lds[];                                                  \
  if ((u32)(uintptr_t)aes4_lds != 0) __builtin_trap(); /* a static __shared__ crept into the kernel */ \
  for (int e_ = threadIdx.x; e_ < AES4_TABLE_WORDS; e_ += ABLOCK) {                                \
    const int r_ = 2 * (e_ >> 14) + ((e_ >> 5) & 1);                                               \
    const u32 v_ = (key).te0[(e_ >> 6) & 255];                                                     \
    aes4_lds[e_] = r_ == 0 ? v_ : (v_ << (8 * r_)) | (v_ >> (32 - 8 * r_));                        \
  }                                                                                                \
  __syncthreads();                                                                                 \
  u32* const aes4_extra = aes4_lds + AES4_TABLE_WORDS;                                             \
  (void)aes4_extra;                                                                                \
  const Aes4 aes{4u * (threadIdx.x & 31)};

#define SCL_AES4_GRID_STRIDE(q, npacks) \
  for (size_t q = (size_t)blockIdx.x * ABLOCK + threadIdx.x; q < (npacks); q += (size_t)gridDim.x * ABLOCK)

// FF::read over the AES stream: element bytes = F::LIMBS/2 consecutive blocks (128- and 256-bit fields)
// RAW: the integer the bytes spell, not yet a residue (for kernels whose constant tables absorb the conversion)
template <class F, bool RAW = false>
__device__ __forceinline__ typename F::E elem_from_blocks(const typename F::Ctx& ctx, const u64* lo, const u64* hi) {
  if constexpr (F::LIMBS == 2) {
    const u128 raw = ((u128)hi[0] << 64) | lo[0];
    if constexpr (RAW) return F::raw_from_le_word(raw);
    else return F::from_le_word(ctx, raw);
  } else {
    typename F::E raw;
    raw.w[0] = lo[0];
    raw.w[1] = hi[0];
    raw.w[2] = lo[1];
    raw.w[3] = hi[1];
    if constexpr (RAW) return F::raw_from_le_word(raw);
    else return F::from_le_word(ctx, raw);
  }
}

#define SCL_AES_PROLOGUE(key)                                                              \
  __shared__ u32 te0_lds[AES_LDS_WORDS];                                                   \
  for (int e_ = threadIdx.x; e_ < AES_LDS_WORDS; e_ += BLOCK) te0_lds[e_] = (key).te0[e_ >> 5]; \
  __syncthreads();                                                                         \
  const u32* te0 = te0_lds + (threadIdx.x & 31);

// PRG::next as raw counter-addressed blocks (prg.cc:124-146); each lane computes 4 blocks a grid
// stride apart so that every store instruction is a contiguous 1 KiB per wave.
template <int = 0>  // (a template so that only the unit that launches it compiles it)
__global__ __launch_bounds__(ABLOCK) void k_prg_blocks(u64* dst, AesKey key, u64 counter0, size_t nblocks) {
  SCL_AES4_PROLOGUE(key)
  const size_t G = (size_t)gridDim.x * ABLOCK;
  for (size_t q = (size_t)blockIdx.x * ABLOCK + threadIdx.x; q < nblocks; q += 4 * G) {
    u64 ctr[4], lo[4], hi[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) ctr[b] = counter0 + q + b * G;
    aes4_blocks<4>(aes, key, ctr, lo, hi);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (q + b * G < nblocks) {
        u64x2 w;
        w.x = lo[b];
        w.y = hi[b];
        *reinterpret_cast<u64x2*>(dst + 2 * (q + b * G)) = w;
      }
    }
  }
}

// FF::read over a byte buffer (ff.h:63-67)
template <class F>
__global__ __launch_bounds__(BLOCK) void k_from_bytes(typename F::Ctx ctx, u64* dst, const unsigned char* src,
                                                      size_t n) {
  SCL_GRID_STRIDE(q, n) {
    if constexpr (F::LIMBS == 1) {
      u64 w;
      __builtin_memcpy(&w, src + 8 * q, 8);
      dst[q] = F::from_le_word(ctx, w);
    } else if constexpr (F::LIMBS == 2) {
      u64 w[2];
      __builtin_memcpy(w, src + 16 * q, 16);
      F::st(dst + 2 * q, F::from_le_word(ctx, ((u128)w[1] << 64) | w[0]));
    } else {
      typename F::E raw;
      __builtin_memcpy(raw.w, src + 32 * q, 32);
      F::st(dst + 4 * q, F::from_le_word(ctx, raw));
    }
  }
}

// Z2k::read at a stride of byteSize = (K-1)/8 + 1 bytes (z2k.h:50-52,71-75, z2k_ops.h:107-112): the bytes
// are gathered one at a time (the stride is odd in general), the mask drops what lies above bit K
template <class F>
__global__ __launch_bounds__(BLOCK) void k_ring_from_bytes(typename F::Ctx ctx, u64* dst, const unsigned char* src,
                                                           size_t n, int bs) {
  SCL_GRID_STRIDE(q, n) {
    const unsigned char* p = src + q * (size_t)bs;
    typename F::E v = 0;
    for (int b = 0; b < bs; ++b) v |= (typename F::E)p[b] << (8 * b);
    F::st(dst + q * F::LIMBS, F::from_le_word(ctx, v));
  }
}

// Vector::random(n, prg) for a PRG at counter0 (vector.h:507-519): element e = bytes [e*bs,(e+1)*bs)
// PAIRS: a Mersenne61 destination that is 16-byte aligned takes both elements of a block in one 16-byte store
template <class F, bool PAIRS = true>
__global__ __launch_bounds__(ABLOCK) void k_vector_random(typename F::Ctx ctx, u64* dst, AesKey key, u64 counter0,
                                                          size_t n) {
  SCL_AES4_PROLOGUE(key)
  const size_t G = (size_t)gridDim.x * ABLOCK;
  constexpr int BPE = F::LIMBS >= 2 ? F::LIMBS / 2 : 1;        // AES blocks per element (wide fields)
  const size_t nb = F::LIMBS == 1 ? (n + 1) / 2 : n * BPE;    // AES blocks needed
  // 256-bit fields: q enumerates elements (two blocks each), two elements per trip
  const size_t units = F::LIMBS == 4 ? n : nb;
  for (size_t q = (size_t)blockIdx.x * ABLOCK + threadIdx.x; q < units; q += (F::LIMBS == 4 ? 2 : 4) * G) {
    u64 ctr[4], lo[4], hi[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) ctr[b] = counter0 + (F::LIMBS == 4 ? 2 * (q + (b >> 1) * G) + (b & 1) : q + b * G);
    aes4_blocks<4>(aes, key, ctr, lo, hi);
    if constexpr (F::LIMBS == 4) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const size_t e = q + u * G;
        if (e < n) F::st(dst + 4 * e, elem_from_blocks<F>(ctx, lo + 2 * u, hi + 2 * u));
      }
      continue;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const size_t blk = q + b * G;
      if (blk < nb) {
        if constexpr (F::LIMBS == 1) {
          if (2 * blk + 1 < n) {
            if constexpr (PAIRS) {
              u64x2 w;
              w.x = F::from_le_word(ctx, lo[b]);
              w.y = F::from_le_word(ctx, hi[b]);
              *reinterpret_cast<u64x2*>(dst + 2 * blk) = w;
            } else {
              dst[2 * blk] = F::from_le_word(ctx, lo[b]);
              dst[2 * blk + 1] = F::from_le_word(ctx, hi[b]);
            }
          } else {
            dst[2 * blk] = F::from_le_word(ctx, lo[b]);
          }
        } else if constexpr (F::LIMBS == 2) {
          F::st(dst + 2 * blk, F::from_le_word(ctx, ((u128)hi[b] << 64) | lo[b]));
        }
      }
    }
  }
}

// Coefficients of secret q*VEC + v of a batch whose first draw starts at block counter0, under the reference PRG discipline (SURVEY.md
// section 8a note P): Vector::random(t+1) from counters [s*B, (s+1)*B), B = ceil((t+1)*byteSize/16);
// c_0's draw is discarded and replaced by the secret (shamir.h:56-57).
//
// Sharing over math::Array<FF, W> (array.h:69-415; pedersenSecretShare uses W = 2, pedersen.h:138): the draw is
// Vector<Array>::random(t+1) = (t+1)*W consecutive elements and component `lane` of coefficient k is element
// k*W + lane.  W = 1, lane = 0 is the plain case.
struct ArrayLane {
  int W, lane;
};

template <class F, int VEC, int TREG, bool RAW = false, class AES>
__device__ __forceinline__ void prg_coeffs(const typename F::Ctx& ctx, Pack<F, VEC> (&c)[TREG + 1], const AES& aes,
                                           const AesKey& key, u64 counter0, size_t q, int t, ArrayLane al = {1, 0}) {
  constexpr int BPE = F::LIMBS >= 2 ? F::LIMBS / 2 : 1;  // AES blocks per coefficient (wide fields)
  const u64 B = F::LIMBS == 1 ? ((u64)(t + 1) * al.W + 1) / 2 : (u64)(t + 1) * BPE * al.W;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const u64 ctr0 = counter0 + (q * VEC + v) * B;
    if constexpr (F::LIMBS == 1) {
      if (al.W != 1) {  // wave-uniform: element e = k*W + lane is half (e & 1) of block e >> 1
#pragma unroll
        for (int k = 1; k <= TREG; ++k) {
          if (k <= t) {
            const u64 e = (u64)k * al.W + al.lane;
            u64 lo, hi;
            aes.block(key, ctr0 + (e >> 1), lo, hi);
            c[k].v[v] = F::from_le_word(ctx, (e & 1) ? hi : lo);
          }
        }
        continue;
      }
      // block j holds coefficients 2j (low 8 bytes) and 2j+1 (high 8 bytes)
#pragma unroll
      for (int j = 0; j <= TREG / 2; ++j) {
        if (2 * j <= t) {
          u64 lo, hi;
          aes.block(key, ctr0 + j, lo, hi);
          if (j > 0) c[2 * j].v[v] = F::from_le_word(ctx, lo);
          if (2 * j + 1 <= TREG && 2 * j + 1 <= t) c[2 * j + 1].v[v] = F::from_le_word(ctx, hi);
        }
      }
    } else {
      // blocks k*BPE .. hold coefficient k; the blocks of c_0's draw are never needed
#pragma unroll
      for (int k = 1; k <= TREG; ++k) {
        if (k <= t) {
          u64 lo[BPE], hi[BPE];
#pragma unroll
          for (int b = 0; b < BPE; ++b)
            aes.block(key, ctr0 + ((u64)k * al.W + al.lane) * BPE + b, lo[b], hi[b]);
          c[k].v[v] = elem_from_blocks<F, RAW>(ctx, lo, hi);
        }
      }
    }
  }
}

// shamirSecretShare(secret_s, t, n, prg) for a whole batch, bit-identical to the per-secret calls on
// ONE PRG (shamir.h:51-68); Horner evaluation at the default nodes 1..n.
// Thresholds up to 16 (register-resident coefficients fit the 128 registers of a 1024-thread workgroup) run on the
// four-table AES; the 48-coefficient form and the 256-bit field keep 256-thread workgroups and the single table.
template <class F, int TREG>
constexpr bool share_prg_four_tables() { return TREG <= 16 && F::LIMBS < 4; }

template <class F, int VEC, int TREG, bool SMALLX>
__global__ __launch_bounds__((share_prg_four_tables<F, TREG>() ? ABLOCK : BLOCK)) void k_share_prg(
    typename F::Ctx ctx, u64* shares, size_t stride, const u64* secrets, AesKey key, u64 counter0, BigTable<F> tab, int t,
    int n, size_t npacks, ArrayLane al) {
  constexpr bool FOUR = share_prg_four_tables<F, TREG>();
  constexpr int THREADS = FOUR ? ABLOCK : BLOCK;
  constexpr int NALPHA = SMALLX ? 1 : BigTable<F>::CAP, NALPHA32 = SMALLX ? BigTable<F>::CAP : 1;
  auto body = [&](const auto& aes, typename F::E* alpha, u32* alpha32) {
    stage_nodes<F, SMALLX, THREADS>(tab, n, alpha, alpha32);
    for (size_t q = (size_t)blockIdx.x * THREADS + threadIdx.x; q < npacks; q += (size_t)gridDim.x * THREADS) {
      const size_t off = q * VEC * F::LIMBS;
      Pack<F, VEC> c[TREG + 1];
      c[0] = load_pack<F, VEC, true>(secrets + off);
      prg_coeffs<F, VEC, TREG, false>(ctx, c, aes, key, counter0, q, t, al);
      horner_rows<F, VEC, TREG, SMALLX>(ctx, c, t, alpha, alpha32, n, shares, stride, off);
    }
  };
  if constexpr (FOUR) {  // no static __shared__ here: the node table goes above the AES tables
    static_assert(NALPHA * sizeof(typename F::E) + NALPHA32 * 4 <= AES4_EXTRA_BYTES, "node table outgrew the dynamic block");
    SCL_AES4_PROLOGUE(key)
    typename F::E* alpha = reinterpret_cast<typename F::E*>(aes4_extra);
    body(aes, alpha, reinterpret_cast<u32*>(alpha + NALPHA));
  } else {
    __shared__ typename F::E alpha[NALPHA];
    __shared__ u32 alpha32[NALPHA32];
    SCL_AES_PROLOGUE(key)
    body(Aes1{te0}, alpha, alpha32);
  }
}

// Same for the Montgomery fields: Vandermonde rows in LDS, lazy products (see k_share_vdm).  The drawn
// coefficients stay the plain integers their bytes spell (FF::read's montyIn would cost a Montgomery product
// each); the table rows carry the extra factor R instead, so each share still comes out as the residue
// c_0 + sum_k montyIn(x_k) * alpha_i^k.
template <class F, int TREG>
__global__ __launch_bounds__(BLOCK) void k_share_prg_vdm(typename F::Ctx ctx, u64* shares, size_t stride,
                                                         const u64* secrets, AesKey key, u64 counter0,
                                                         const u64* vdm, int t, int n, size_t N, ArrayLane al) {
  SCL_AES_PROLOGUE(key)
  __shared__ u64 V[VdmLds::WORDS];
  for (int i = threadIdx.x; i < n * t * F::LIMBS; i += BLOCK) V[i] = vdm[i];
  __syncthreads();
  SCL_GRID_STRIDE(q, N) {
    const size_t off = q * F::LIMBS;
    Pack<F, 1> c[TREG + 1];
    c[0] = load_pack<F, 1, true>(secrets + off);
    prg_coeffs<F, 1, TREG, true>(ctx, c, Aes1{te0}, key, counter0, q, t, al);  // raw integers: the table holds alpha^k * R
    vdm_rows<F, TREG>(ctx, c, t, V, n, shares, stride, off);
  }
}

// Same, small-node Vandermonde evaluation (see k_share_small) for t <= SmallVdm::TMAX.  NBLK = AES
// blocks per secret that carry a used coefficient: t/2+1 for M61 (block j = c_2j, c_2j+1), t for the
// 128-bit fields (block k = c_k, block 0 skipped).  All VEC*NBLK blocks of a lane run in lockstep.
template <class F, int VEC, int NBLK>
__global__ __launch_bounds__(ABLOCK) void k_share_prg_small(u64* shares, size_t stride, const u64* secrets, AesKey key,
                                                            u64 counter0, SmallVdm tab, int t, int n,
                                                            size_t npacks) {
  static_assert(SmallVdm::CAP * 4 <= AES4_EXTRA_BYTES, "Vandermonde table outgrew the dynamic block");
  SCL_AES4_PROLOGUE(key)
  u32* const V = aes4_extra;  // (no static __shared__: the AES tables must start at LDS address 0)
  for (int i = threadIdx.x; i < n * (t + 1); i += ABLOCK) V[i] = tab.v[i];
  __syncthreads();
  const typename F::Ctx ctx{};
  const u64 B = F::LIMBS == 1 ? (u64)(t + 2) / 2 : (u64)(t + 1);
  SCL_AES4_GRID_STRIDE(q, npacks) {
    const size_t off = q * VEC * F::LIMBS;
    Pack<F, VEC> c[SmallVdm::TMAX + 1];
#pragma unroll
    for (int k = 1; k <= SmallVdm::TMAX; ++k)
#pragma unroll
      for (int v = 0; v < VEC; ++v) c[k].v[v] = F::zero();
    c[0] = load_pack<F, VEC, true>(secrets + off);
    u64 ctr[VEC * NBLK], lo[VEC * NBLK], hi[VEC * NBLK];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
      for (int j = 0; j < NBLK; ++j)
        ctr[v * NBLK + j] = counter0 + (q * VEC + v) * B + j + (F::LIMBS == 1 ? 0 : 1);
    aes4_blocks<VEC * NBLK>(aes, key, ctr, lo, hi);
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
      for (int j = 0; j < NBLK; ++j) {
        if constexpr (F::LIMBS == 1) {
          if (j > 0) c[2 * j].v[v] = F::from_le_word(ctx, lo[v * NBLK + j]);
          if (2 * j + 1 <= SmallVdm::TMAX) c[2 * j + 1].v[v] = F::from_le_word(ctx, hi[v * NBLK + j]);
        } else {
          c[j + 1].v[v] = F::from_le_word(ctx, ((u128)hi[v * NBLK + j] << 64) | lo[v * NBLK + j]);
        }
      }
    small_rows<F, VEC>(ctx, c, V, t, n, shares, stride, off);
  }
}

// The same with the threshold T compiled in (the shape BASELINE quotes: T = 3).  k_share_prg_small walks a party's row of
// the Vandermonde table in LDS under a run-time "k <= t" test per term: seven scalar compare-and-branch pairs and up to
// seven LDS reads per share, in a kernel whose LDS array is already the AES's bottleneck.  Here the term loop is straight
// line and the small powers come from the kernel argument by scalar loads (wave-uniform, no LDS traffic).
template <class F, int VEC, int NBLK, int T>
__global__ __launch_bounds__(ABLOCK) void k_share_prg_small_t(u64* shares, size_t stride, const u64* secrets, AesKey key,
                                                              u64 counter0, SmallVdm tab, int n, size_t npacks) {
  static_assert(T >= 1 && T <= SmallVdm::TMAX && NBLK == (F::LIMBS == 1 ? T / 2 + 1 : T), "blocks that carry a used coefficient");
  SCL_AES4_PROLOGUE(key)
  const typename F::Ctx ctx{};
  constexpr u64 B = F::LIMBS == 1 ? (u64)(T + 2) / 2 : (u64)(T + 1);
  SCL_AES4_GRID_STRIDE(q, npacks) {
    const size_t off = q * VEC * F::LIMBS;
    Pack<F, VEC> c[T + 2];
    c[0] = load_pack<F, VEC, true>(secrets + off);
    u64 ctr[VEC * NBLK], lo[VEC * NBLK], hi[VEC * NBLK];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
      for (int j = 0; j < NBLK; ++j)
        ctr[v * NBLK + j] = counter0 + (q * VEC + v) * B + j + (F::LIMBS == 1 ? 0 : 1);
    aes4_blocks<VEC * NBLK>(aes, key, ctr, lo, hi);
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
      for (int j = 0; j < NBLK; ++j) {
        if constexpr (F::LIMBS == 1) {
          if (j > 0) c[2 * j].v[v] = F::from_le_word(ctx, lo[v * NBLK + j]);
          if (2 * j + 1 <= T) c[2 * j + 1].v[v] = F::from_le_word(ctx, hi[v * NBLK + j]);
        } else {
          c[j + 1].v[v] = F::from_le_word(ctx, ((u128)hi[v * NBLK + j] << 64) | lo[v * NBLK + j]);
        }
      }
    for (int i = 0; i < n; ++i) {
      const u32* row = tab.v + i * (T + 1);  // kernel argument: scalar loads
      SmallAcc<F> acc[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v].init();
#pragma unroll
      for (int k = 1; k <= T; ++k) {
        const u32 w = row[k];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v].mac(c[k].v[v], w);
      }
      Pack<F, VEC> y;
#pragma unroll
      for (int v = 0; v < VEC; ++v) y.v[v] = acc[v].fold(ctx, c[0].v[v]);  // V[i][0] = 1
      store_pack<F, VEC, true>(shares + (size_t)i * stride * F::LIMBS + off, y);
    }
  }
}

// ---- any threshold: the polynomial in chunks of at most SHARE_CHUNK_T + 1 coefficients, one launch per chunk -------
// shamirSecretShare has no bound on t (shamir.h:51-68); the register-resident kernels above hold 49 coefficients.  A longer
// polynomial f(x) = sum_j x^(C j) g_j(x), C = SHARE_CHUNK_T + 1, is evaluated by Horner over its chunks from the top: the
// first launch writes g_top(alpha_i), every later one y_i <- y_i * alpha_i^C + g_j(alpha_i) (its chunk has exactly C
// coefficients).  tables = [nodes | nodes^C] in device memory.  Generic full-width arithmetic: not a tuned path.
template <class F>
constexpr int share_chunk_t() { return F::LIMBS >= 4 ? 23 : 47; }

// SMALLX: tables[0 .. n) hold the nodes as plain integers < 2^32 (the fields' small-constant Horner step)
template <class F, bool SMALLX = false>
__global__ __launch_bounds__(BLOCK) void k_share_chunk(typename F::Ctx ctx, u64* shares, size_t stride, const u64* c0,
                                                       const u64* crest, size_t cstride, const u64* tables, int tc, int n,
                                                       size_t N, int accumulate) {
  constexpr int TC = share_chunk_t<F>();
  __shared__ typename F::E alpha[BigTable<F>::CAP], xpow[BigTable<F>::CAP];
  for (int i = threadIdx.x; i < n; i += BLOCK) {
    alpha[i] = F::ld(tables + (size_t)i * F::LIMBS);
    xpow[i] = F::ld(tables + (size_t)(n + i) * F::LIMBS);
  }
  __syncthreads();
  SCL_GRID_STRIDE(q, N) {
    const size_t off = q * F::LIMBS;
    Pack<F, 1> c[TC + 1];
    c[0] = load_pack<F, 1, true>(c0 + off);
#pragma unroll
    for (int k = 1; k <= TC; ++k) {
      if (k <= tc) c[k] = load_pack<F, 1, true>(crest + (size_t)(k - 1) * cstride * F::LIMBS + off);
    }
    for (int i = 0; i < n; ++i) {
      const typename F::E x = alpha[i];
      typename F::E y = F::zero();
#pragma unroll
      for (int k = TC; k >= 0; --k) {
        if (k == tc) y = c[k].v[0];                                     // wave-uniform
        else if (k < tc) {
          if constexpr (SMALLX) y = F::muladd_small(ctx, y, F::low32(x), c[k].v[0]);
          else y = F::add(ctx, F::mul(ctx, y, x), c[k].v[0]);
        }
      }
      u64* dst = shares + (size_t)i * stride * F::LIMBS + off;
      if (accumulate) y = F::add(ctx, F::mul(ctx, F::ld(dst), xpow[i]), y);
      F::st(dst, y);
    }
  }
}

// Coefficient rows of a PRG-driven sharing (rows[k-1][s] = c_k of secret s, k = 1..t) under the reference's counter
// discipline (prg_coeffs above; al = component `lane` of an Array<FF, W> draw).  The first pass of the two-pass form
// of shamirSecretShare(secret, t, n, prg): AES at the rate of the four-table kernel with a rolled block loop (a fused
// kernel that keeps 17 or 49 coefficients in registers unrolls 9 .. 25 AES blocks per lane -- 100 to 260 KB of code per
// launch, beyond the instruction cache), the explicit-coefficient kernel of the shape (small-node, blocked, matrix cores,
// per-node GF(2^128)) as the second (profiles/r2_probe_prg_share.txt).
template <class F>
__global__ __launch_bounds__(ABLOCK) void k_prg_coeff_rows(typename F::Ctx ctx, u64* rows, size_t rstride, AesKey key,
                                                           u64 counter0, int t, size_t N, ArrayLane al) {
  SCL_AES4_PROLOGUE(key)
  constexpr int BPE = F::LIMBS >= 2 ? F::LIMBS / 2 : 1;
  const u64 B = F::LIMBS == 1 ? ((u64)(t + 1) * al.W + 1) / 2 : (u64)(t + 1) * BPE * al.W;
  SCL_AES4_GRID_STRIDE(s, N) {
    const u64 ctr0 = counter0 + s * B;
    if constexpr (F::LIMBS == 1) {
      if (al.W == 1) {
#pragma unroll 1
        for (int j = 0; 2 * j <= t; ++j) {  // block j = c_2j (low 8 bytes), c_2j+1 (high 8 bytes)
          u64 lo, hi;
          aes.block(key, ctr0 + j, lo, hi);
          if (j > 0) rows[(size_t)(2 * j - 1) * rstride + s] = F::from_le_word(ctx, lo);
          if (2 * j + 1 <= t) rows[(size_t)(2 * j) * rstride + s] = F::from_le_word(ctx, hi);
        }
      } else {
#pragma unroll 1
        for (int k = 1; k <= t; ++k) {  // element e = k*W + lane is half (e & 1) of block e >> 1
          const u64 e = (u64)k * al.W + al.lane;
          u64 lo, hi;
          aes.block(key, ctr0 + (e >> 1), lo, hi);
          rows[(size_t)(k - 1) * rstride + s] = F::from_le_word(ctx, (e & 1) ? hi : lo);
        }
      }
    } else {
#pragma unroll 1
      for (int k = 1; k <= t; ++k) {
        u64 lo[BPE], hi[BPE];
#pragma unroll
        for (int b = 0; b < BPE; ++b) aes.block(key, ctr0 + ((u64)k * al.W + al.lane) * BPE + b, lo[b], hi[b]);
        F::st(rows + ((size_t)(k - 1) * rstride + s) * F::LIMBS, elem_from_blocks<F>(ctx, lo, hi));
      }
    }
  }
}

// ---- additive ----------------------------------------------------------------------------------
// additiveShare with explicit randomness (additive.h:41-53)
template <class F, int VEC>
__global__ __launch_bounds__(BLOCK) void k_additive_share(typename F::Ctx ctx, u64* shares, size_t stride,
                                                          const u64* secrets, const u64* rnd, size_t rstride, int n,
                                                          size_t npacks) {
  SCL_GRID_STRIDE(q, npacks) {
    const size_t off = q * VEC * F::LIMBS;
    Pack<F, VEC> last = load_pack<F, VEC, true>(secrets + off);
    for (int i = 0; i + 1 < n; ++i) {
      const Pack<F, VEC> r = load_pack<F, VEC, true>(rnd + (size_t)i * rstride * F::LIMBS + off);
#pragma unroll
      for (int v = 0; v < VEC; ++v) last.v[v] = F::sub(ctx, last.v[v], r.v[v]);
      store_pack<F, VEC, true>(shares + (size_t)i * stride * F::LIMBS + off, r);
    }
    store_pack<F, VEC, true>(shares + (size_t)(n - 1) * stride * F::LIMBS + off, last);
  }
}

// PRG-driven: share i < n-1 of secret s = FF::random on counter counter0 + s*(n-1)+i (ff.h:72-76: one block each)
template <class F, int VEC>
__global__ __launch_bounds__(ABLOCK) void k_additive_share_prg(typename F::Ctx ctx, u64* shares, size_t stride,
                                                               const u64* secrets, AesKey key, u64 counter0, int n,
                                                               size_t npacks) {
  SCL_AES4_PROLOGUE(key)
  SCL_AES4_GRID_STRIDE(q, npacks) {
    const size_t off = q * VEC * F::LIMBS;
    Pack<F, VEC> last = load_pack<F, VEC, true>(secrets + off);
    constexpr int BPE = F::LIMBS >= 2 ? F::LIMBS / 2 : 1;  // FF::random burns ceil(byteSize/16) blocks
    for (int i = 0; i + 1 < n; ++i) {
      Pack<F, VEC> r;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        u64 lo[BPE], hi[BPE];
        const u64 c0 = counter0 + ((q * VEC + v) * (u64)(n - 1) + i) * BPE;
#pragma unroll
        for (int b = 0; b < BPE; ++b) aes.block(key, c0 + b, lo[b], hi[b]);
        if constexpr (F::LIMBS == 1)
          r.v[v] = F::from_le_word(ctx, lo[0]);
        else
          r.v[v] = elem_from_blocks<F>(ctx, lo, hi);
        last.v[v] = F::sub(ctx, last.v[v], r.v[v]);
      }
      store_pack<F, VEC, true>(shares + (size_t)i * stride * F::LIMBS + off, r);
    }
    store_pack<F, VEC, true>(shares + (size_t)(n - 1) * stride * F::LIMBS + off, last);
  }
}

// reconstruct = Vector::sum per secret (vector.h:261-267)
template <class F, int VEC, bool NT>
__global__ __launch_bounds__(BLOCK) void k_additive_recover(typename F::Ctx ctx, u64* out, const u64* shares,
                                                            size_t stride, int n, size_t npacks) {
  SCL_GRID_STRIDE(q, npacks) {
    const size_t off = q * VEC * F::LIMBS;
    typename F::Acc acc[VEC];
    typename F::E run[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      acc[v] = F::acc_zero();
      run[v] = F::zero();
    }
    int terms = 0;
    int i = 0;
    for (; i + 4 <= n; i += 4) {
      Pack<F, VEC> x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = load_pack<F, VEC, NT>(shares + (size_t)(i + j) * stride * F::LIMBS + off);
      if (terms + 4 > F::ACC_TERMS) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          run[v] = F::add(ctx, run[v], F::acc_fold(ctx, acc[v]));
          acc[v] = F::acc_zero();
        }
        terms = 0;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int v = 0; v < VEC; ++v) F::acc_add(ctx, acc[v], x[j].v[v]);
      terms += 4;
    }
    for (; i < n; ++i) {
      const Pack<F, VEC> x = load_pack<F, VEC, NT>(shares + (size_t)i * stride * F::LIMBS + off);
      if (terms + 1 > F::ACC_TERMS) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          run[v] = F::add(ctx, run[v], F::acc_fold(ctx, acc[v]));
          acc[v] = F::acc_zero();
        }
        terms = 0;
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) F::acc_add(ctx, acc[v], x.v[v]);
      terms += 1;
    }
    Pack<F, VEC> r;
#pragma unroll
    for (int v = 0; v < VEC; ++v) r.v[v] = F::add(ctx, run[v], F::acc_fold(ctx, acc[v]));
    store_pack<F, VEC, NT>(out + off, r);
  }
}

// ---- a block of table rows times the share columns of one pack ---------------------------------------
// y[j] = sum_k L[k][j] * shares[k][pack], j < RB: each share is loaded once and feeds RB lazy accumulators through
// F::kmac (prepared constants: six carry-free 32-bit multiply-adds per term for Mersenne61, 24 for Mersenne127).
// L is one row block of a table laid out [row block][k][RB] (zero rows pad the last block), already in LDS and
// read wave-uniformly; the next row's constant is on its way from LDS while this row's multiplies issue.
template <class F, int VEC, int RB>
__device__ __forceinline__ void rows_times_shares(const typename F::Ctx& ctx, const typename F::KC* L, const u64* shares,
                                                  size_t stride, size_t off, int d1, Pack<F, VEC> (&y)[RB]) {
  typedef typename F::KC KC;
  typename F::KAcc acc[RB][VEC];
  typename F::E run[RB][VEC];
#pragma unroll
  for (int j = 0; j < RB; ++j)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      acc[j][v] = F::kacc_zero();
      run[j][v] = F::zero();
    }
  int terms = 0;
  Pack<F, VEC> xn = load_pack<F, VEC, false>(shares + off);
  for (int k = 0; k < d1; ++k) {
    const Pack<F, VEC> x = xn;
    if (k + 1 < d1) xn = load_pack<F, VEC, false>(shares + (size_t)(k + 1) * stride * F::LIMBS + off);
    if (terms + 1 > F::K_TERMS) {
#pragma unroll
      for (int j = 0; j < RB; ++j)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          run[j][v] = F::add(ctx, run[j][v], F::kacc_fold(ctx, acc[j][v]));
          acc[j][v] = F::kacc_zero();
        }
      terms = 0;
    }
    KC l = L[k * RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      KC ln = l;
      if (j + 1 < RB) ln = L[k * RB + j + 1];
#pragma unroll
      for (int v = 0; v < VEC; ++v) F::kmac(ctx, acc[j][v], l, x.v[v]);
      l = ln;
    }
    ++terms;
  }
#pragma unroll
  for (int j = 0; j < RB; ++j)
#pragma unroll
    for (int v = 0; v < VEC; ++v) y[j].v[v] = F::add(ctx, run[j][v], F::kacc_fold(ctx, acc[j][v]));
}

// the workgroup copies row block b of the table (RB * d1 prepared constants) into LDS; barriers on both sides
template <class KC, int RB>
__device__ __forceinline__ void stage_row_block(unsigned char* smem, const KC* Lk, int b, int d1) {
  static_assert((sizeof(KC) * RB) % 16 == 0, "row block image is copied 16 bytes at a time");
  if (b) __syncthreads();
  const uint4* src = reinterpret_cast<const uint4*>(Lk + (size_t)b * d1 * RB);
  uint4* dst = reinterpret_cast<uint4*>(smem);
  const int n16 = (int)((size_t)d1 * RB * sizeof(KC) / 16);
  for (int i = threadIdx.x; i < n16; i += BLOCK) dst[i] = src[i];
  __syncthreads();
}

// ---- error-detecting recovery ---------------------------------------------------------------------
// shamirRecoverD (shamir.h:116-139): rows 0..nchk-1 of L re-derive share d+1+r from the first d+1
// shares, row nchk evaluates at x.  The nchk+1 inner products of one secret are taken RB rows at a time
// (rows_times_shares): the shares are read once per row block (once in all when nchk < RB; the later passes
// hit L2), so the kernel streams at HBM rate for small t and is bound by multiply issue for large t.
// One pack per thread: the host launches ceil(npacks / BLOCK) workgroups (every thread reaches the barriers).
template <class F, int VEC, int RB>
__global__ __launch_bounds__(BLOCK) void k_recover_detect(typename F::Ctx ctx, u64* out, unsigned char* status,
                                                          const u64* shares, size_t stride,
                                                          const typename F::KC* Lk, int d1, int nchk, size_t npacks,
                                                          unsigned long long* bad_count) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  typedef typename F::KC KC;
  const int nblk = (nchk + RB) / RB;  // ceil((nchk + 1) / RB)
  const size_t q = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const bool live = q < npacks;
  const size_t off = q * VEC * F::LIMBS;
  bool bad[VEC];
  Pack<F, VEC> result;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    bad[v] = false;
    result.v[v] = F::zero();
  }
  for (int b = 0; b < nblk; ++b) {
    stage_row_block<KC, RB>(smem_raw, Lk, b, d1);
    if (!live) continue;
    Pack<F, VEC> y[RB];
    rows_times_shares<F, VEC, RB>(ctx, reinterpret_cast<const KC*>(smem_raw), shares, stride, off, d1, y);
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int r = b * RB + j;
      if (r < nchk) {
        const Pack<F, VEC> got = load_pack<F, VEC, false>(shares + (size_t)(d1 + r) * stride * F::LIMBS + off);
#pragma unroll
        for (int v = 0; v < VEC; ++v) bad[v] |= !F::eq(y[j].v[v], got.v[v]);
      } else if (r == nchk) {
        result = y[j];
      }
    }
  }
  if (live) {
    unsigned nbad = 0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      status[q * VEC + v] = bad[v] ? 1 : 0;
      if (bad[v]) result.v[v] = F::zero();
      nbad += bad[v];
    }
    store_pack<F, VEC, false>(out + off, result);
    if (nbad) atomicAdd(bad_count, (unsigned long long)nbad);
  }
}

// shamirRecoverD at large (t, d) over Mersenne61 as a contraction (shamir.h:129-136 is nchk + 1 Lagrange rows applied to the
// same d + 1 shares): Y = L * S runs on the matrix cores (share_mfma.hpp, the kernel that evaluates V * C when sharing);
// this pass compares rows 0 .. nchk-1 of Y with the shares they re-derive and hands out row nchk, the value at x.
// Same outputs as k_recover_detect: status 1 and a zero result where a check fails, the failures counted.
template <class F>
__global__ __launch_bounds__(BLOCK) void k_detect_compare(u64* out, unsigned char* status, const u64* Y, size_t ldy,
                                                          const u64* checked, size_t stride, int nchk, size_t N,
                                                          unsigned long long* bad_count) {
  static_assert(F::LIMBS == 1, "one-word elements");
  SCL_GRID_STRIDE(s, N) {
    bool bad = false;
    for (int r = 0; r < nchk; ++r) bad |= Y[(size_t)r * ldy + s] != checked[(size_t)r * stride + s];
    out[s] = bad ? 0 : Y[(size_t)nchk * ldy + s];
    status[s] = bad ? 1 : 0;
    const unsigned long long nb = __popcll(__ballot(bad));
    if (nb && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)__ballot(bad)) - 1)) atomicAdd(bad_count, nb);
  }
}

// ---- error-correcting recovery: shamirRecoverC (Berlekamp-Welch, shamir.h:202-259) ----------------------------
// What the reference computes per secret is fixed by algebra, not by its elimination order: for e = t, t-1, .. 0
// it builds the n x n system  s_i E(a_i) = Q(a_i)  (E monic of degree e, deg Q <= n-1-e, n = 3t+1) and takes the
// first e whose matrix is NONSINGULAR (solveLinearSystem accepts unique solutions only, matrix.h:811-828); then
// f = Q / E, and a non-zero remainder is "could not correct shares".  e = 0 is plain interpolation and always
// succeeds.  Two kernels:
//   k_bw_consistent  one thread per secret: if the n shares lie on one polynomial of degree <= t (the common
//                    case; then every e > 0 system is singular and e = 0 returns that polynomial) write its
//                    coefficients and E = 1, else queue the secret;
//   k_bw_solve       one wavefront per queued secret: the systems, in LDS, by division-free Gauss-Jordan with lanes
//                    over rows; the only inversions are the n diagonal ones of the accepted system, done in parallel.
// Lk is [row block][k][RB] prepared constants over nchk + d1 rows: nchk rows that re-derive share d1+r from the
// first d1 = t+1 shares, then d1 rows that give coefficient k of the interpolant (rows_times_shares, as in
// k_recover_detect; one secret per thread, the host launches ceil(N / BLOCK) workgroups).
template <class F, int RB>
__global__ __launch_bounds__(BLOCK) void k_bw_consistent(typename F::Ctx ctx, u64* f_out, size_t f_stride, u64* e_out,
                                                         size_t e_stride, unsigned char* status, unsigned* nerr,
                                                         const u64* shares, size_t stride, const typename F::KC* Lk,
                                                         int d1, int nchk, size_t N, unsigned* queue, unsigned* queued) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  typedef typename F::KC KC;
  const int n = d1 + nchk;
  const int nblk = (n + RB - 1) / RB;
  const size_t s = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const bool live = s < N;
  const size_t off = s * F::LIMBS;
  bool bad = false;
  for (int b = 0; b < nblk; ++b) {
    stage_row_block<KC, RB>(smem_raw, Lk, b, d1);
    if (!live || (bad && b * RB >= nchk)) continue;  // inconsistent: the coefficient rows are not needed
    Pack<F, 1> y[RB];
    rows_times_shares<F, 1, RB>(ctx, reinterpret_cast<const KC*>(smem_raw), shares, stride, off, d1, y);
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int r = b * RB + j;
      if (r < nchk) bad |= !F::eq(y[j].v[0], F::ld(shares + (size_t)(d1 + r) * stride * F::LIMBS + off));
      else if (r < n && !bad) F::st(f_out + (size_t)(r - nchk) * f_stride * F::LIMBS + off, y[j].v[0]);
    }
  }
  if (!live) return;
  if (bad) {
    queue[atomicAdd(queued, 1u)] = (unsigned)s;
  } else {
    for (int k = d1; k < n; ++k) F::st(f_out + (size_t)k * f_stride * F::LIMBS + off, F::zero());
    F::st(e_out + off, F::one(ctx));
    for (int k = 1; k < d1; ++k) F::st(e_out + (size_t)k * e_stride * F::LIMBS + off, F::zero());
    status[s] = 0;
    nerr[s] = 0;
  }
}

constexpr int BW_WAVE = 64;

// elements of k_bw_solve's work area for n shares (t = (n-1)/3): the augmented matrix, the nodes, the shares, the
// quotient and the locator
__host__ __device__ inline size_t bw_lds_elems(size_t n) { return n * (n + 1) + 3 * n + (n - 1) / 3 + 1; }

// One workgroup per queued secret, threads over rows (a thread owns rows tid, tid + blockDim.x, ..: one wavefront while
// n <= 64).  The work area is the workgroup's dynamic LDS when it fits and a slice of `work` (global memory, one slice
// per workgroup) when it does not -- shamirRecoverC has no bound on the number of shares.
template <class F>
__global__ __launch_bounds__(1024) void k_bw_solve(typename F::Ctx ctx, u64* f_out, size_t f_stride, u64* e_out,
                                                   size_t e_stride, unsigned char* status, unsigned* nerr,
                                                   const u64* shares, size_t stride, const u64* nodes, int n,
                                                   const unsigned* queue, unsigned nqueued, unsigned* failed, u64* work) {
  typedef typename F::E E;
  extern __shared__ unsigned char smem_raw[];
  __shared__ int pivot_row;
  __shared__ int all_ok;
  const int t = (n - 1) / 3, m = n + 1, tid = threadIdx.x, T = blockDim.x;
  E* M = work ? reinterpret_cast<E*>(work) + (size_t)blockIdx.x * bw_lds_elems(n) : reinterpret_cast<E*>(smem_raw);  // [n][m]
  E* al = M + (size_t)n * m;  // [n]
  E* sh = al + n;             // [n]
  E* fq = sh + n;             // [n] quotient
  E* ec = fq + n;             // [t+1] locator
  for (int r = tid; r < n; r += T) al[r] = F::ld(nodes + (size_t)r * F::LIMBS);
  for (unsigned item = blockIdx.x; item < nqueued; item += gridDim.x) {
    const size_t s = queue[item];
    __syncthreads();
    for (int r = tid; r < n; r += T) sh[r] = F::ld(shares + ((size_t)r * stride + s) * F::LIMBS);
    __syncthreads();
    int e = t;
    for (; e >= 0; --e) {
      // row i of [A | b] (shamir.h:217-231): s a^j for j < e, -1 at e, -a^(j-e) above, b = -s a^e
      for (int r = tid; r < n; r += T) {
        E* row = M + (size_t)r * m;
        const E a = al[r], si = sh[r];
        E v = si;
        for (int j = 0; j < e; ++j) {
          row[j] = v;
          v = F::mul(ctx, v, a);
        }
        row[n] = F::neg(ctx, v);
        v = F::neg(ctx, F::one(ctx));
        for (int j = e; j < n; ++j) {
          row[j] = v;
          v = F::mul(ctx, v, a);
        }
      }
      // Division-free Gauss-Jordan: rows k != c become row_k * pivot - row_c * M[k][c], which keeps the solution
      // set (the pivot is non-zero) and needs no inversion to decide the rank; the matrix ends up diagonal.
      bool singular = false;
      for (int c = 0; c < n; ++c) {
        if (tid == 0) pivot_row = n;
        __syncthreads();  // (also: the rows written above / by the previous column are visible)
        for (int r = c + tid; r < n; r += T)
          if (!F::is_zero(M[(size_t)r * m + c])) {
            atomicMin(&pivot_row, r);
            break;  // a thread's rows ascend
          }
        __syncthreads();
        const int p = pivot_row;
        if (p >= n) {  // workgroup-uniform: no pivot in this column, the matrix is singular
          singular = true;
          break;
        }
        if (p != c) {
          for (int j = tid; j < m; j += T) {
            const E tmp = M[(size_t)p * m + j];
            M[(size_t)p * m + j] = M[(size_t)c * m + j];
            M[(size_t)c * m + j] = tmp;
          }
          __syncthreads();
        }
        const E pv = M[(size_t)c * m + c];
        for (int r = tid; r < n; r += T) {
          if (r == c) continue;
          E* row = M + (size_t)r * m;
          const E nt = F::neg(ctx, row[c]);
          row[c] = F::zero();
          // columns left of c: zero in row c, so only the scaling by the pivot remains (rows above c carry their
          // own diagonal entry there)
          if (r < c) row[r] = F::mul(ctx, row[r], pv);
          for (int j = c + 1; j < m; ++j) row[j] = F::add(ctx, F::mul(ctx, row[j], pv), F::mul(ctx, M[(size_t)c * m + j], nt));
        }
        __syncthreads();  // every thread has read pivot_row and finished its rows before the next column resets it
      }
      __syncthreads();
      if (!singular) break;
    }
    // the e = 0 system is a Vandermonde system, nonsingular for distinct nodes (the host checks them)
    const bool solved = e >= 0;
    if (!solved) e = 0;
    // the system is diagonal now: x_j = M[j][n] / M[j][j], every thread inverting its own diagonal entries
    if (solved)
      for (int r = tid; r < n; r += T) M[(size_t)r * m + n] = F::mul(ctx, M[(size_t)r * m + n], F::inv(ctx, M[(size_t)r * m + r]));
    __syncthreads();
    // E = x[0..e) then 1, Q = x[e..n)
    if (tid == 0) {
      bool ok = solved;
      if (solved) {
        for (int j = 0; j <= t; ++j) ec[j] = j < e ? M[(size_t)j * m + n] : (j == e ? F::one(ctx) : F::zero());
        const int qn = n - e;  // Q has qn coefficients; reuse column n of M as the running remainder
        for (int j = 0; j < n; ++j) fq[j] = F::zero();
        for (int d = qn - 1 - e; d >= 0; --d) {  // synthetic division by the monic E
          const E lead = M[(size_t)(e + d + e) * m + n];
          fq[d] = lead;
          for (int j = 0; j < e; ++j) {
            E& r = M[(size_t)(e + d + j) * m + n];
            r = F::sub(ctx, r, F::mul(ctx, lead, ec[j]));
          }
        }
        const int rem = qn - 1 - e >= 0 ? e : qn;  // remainder coefficients left in Q[0..rem)
        for (int j = 0; j < rem; ++j) ok = ok && F::is_zero(M[(size_t)(e + j) * m + n]);
      }
      all_ok = ok ? 1 : 0;
    }
    __syncthreads();
    const bool ok = all_ok != 0;
    for (int k = tid; k < n; k += T) F::st(f_out + ((size_t)k * f_stride + s) * F::LIMBS, ok ? fq[k] : F::zero());
    for (int k = tid; k <= t; k += T) F::st(e_out + ((size_t)k * e_stride + s) * F::LIMBS, ok ? ec[k] : F::zero());
    if (tid == 0) {
      status[s] = ok ? 0 : 1;
      nerr[s] = ok ? (unsigned)e : 0u;
      if (!ok) atomicAdd(failed, 1u);
    }
  }
}

// The accumulator of the matrix kernels.  For most fields it is the field's own lazy accumulator (F::Acc / F::mac).  Mersenne127's
// folds every 256-bit product to 128 bits as it goes (~70 instructions a term): right for the streaming reconstruct kernels, which
// are HBM-bound and want few registers (the column-sum form below cost k_recover_fixed<M127> 8 % in an A/B on one box,
// profiles/r5_probe_ab_c3_m127_acc.txt), wrong for a product's inner loop -- there the products are summed unreduced as seven
// column sums of 32 x 32 partial products (LazyCols: 32 instructions a term) and reduced once: 2^128 = 2, 2^256 = 4 (mod p).
template <class F>
struct MatAcc {
  typename F::Acc a;
  enum { TERMS = F::ACC_TERMS };
  __device__ __forceinline__ void zero() { a = F::acc_zero(); }
  __device__ __forceinline__ void mac(const typename F::Ctx& c, const typename F::E& x, const typename F::E& y) { F::mac(c, a, x, y); }
  __device__ __forceinline__ void add(const typename F::Ctx& c, const typename F::E& x) { F::acc_add(c, a, x); }
  __device__ __forceinline__ typename F::E fold(const typename F::Ctx& c) const { return F::acc_fold(c, a); }
};
template <>
struct MatAcc<M127> {
  LazyCols<4> cols;
  u128 plain;  // elements added as they are (the restart value after a fold): < 2^127 each, few of them
  enum { TERMS = 1 << 24 };
  __device__ __forceinline__ void zero() {
    lazy_zero(cols);
    plain = 0;
  }
  __device__ __forceinline__ void mac(const M127::Ctx&, u128 x, u128 y) {
    const u32 xl[4] = {(u32)x, (u32)(x >> 32), (u32)(x >> 64), (u32)(x >> 96)};
    const u32 yl[4] = {(u32)y, (u32)(y >> 32), (u32)(y >> 64), (u32)(y >> 96)};
    lazy_mac<4>(cols, xl, yl);
  }
  __device__ __forceinline__ void add(const M127::Ctx& c, u128 x) { plain = M127::add(c, plain, x); }
  __device__ __forceinline__ u128 fold(const M127::Ctx& c) const {
    u32 t[9];
    lazy_limbs<4, 9>(cols, t);  // the products' sum, < 2^288
    const u128 lo = (u128)t[0] | ((u128)t[1] << 32) | ((u128)t[2] << 64) | ((u128)t[3] << 96);
    const u128 hi = (u128)t[4] | ((u128)t[5] << 32) | ((u128)t[6] << 64) | ((u128)t[7] << 96);
    M127::Acc s = M127::acc_zero();
    M127::acc_add_raw(s, lo);
    M127::acc_add_raw(s, hi);  // 2^128 = 2
    M127::acc_add_raw(s, hi);
    M127::acc_add_raw(s, (u128)t[8] << 2);  // 2^256 = 4
    M127::acc_add_raw(s, plain);
    return M127::acc_fold(c, s);
  }
};

// ---- matrices ---------------------------------------------------------------------------------------
// Matrix::multiply (matrix.h:477-495): C[M x N] = A[M x K] * B[K x N].  One thread per column j,
// RM rows of A at a time (A tile in LDS, wave-uniform reads), B streamed coalesced.
template <class F, int RM>
__global__ __launch_bounds__(BLOCK) void k_matmul(typename F::Ctx ctx, u64* C, size_t ldc, const u64* A, size_t lda,
                                                  const u64* B, size_t ldb, int M, int K, size_t N, int rows_per_tile) {
  extern __shared__ unsigned char smem_raw[];
  typename F::E* As = reinterpret_cast<typename F::E*>(smem_raw);
  const int row0 = blockIdx.y * rows_per_tile;
  const int rows = min(rows_per_tile, M - row0);
  for (int i = threadIdx.x; i < rows * K; i += BLOCK)
    As[i] = F::ld(A + ((size_t)(row0 + i / K) * lda + (i % K)) * F::LIMBS);
  __syncthreads();
  SCL_GRID_STRIDE(j, N) {
    for (int r0 = 0; r0 < rows; r0 += RM) {
      MatAcc<F> acc[RM];
      typename F::E run[RM];
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        acc[r].zero();
        run[r] = F::zero();
      }
      int terms = 0;
      for (int k = 0; k < K; ++k) {
        const typename F::E b = F::ld(B + ((size_t)k * ldb + j) * F::LIMBS);
        if (terms + 1 > (int)MatAcc<F>::TERMS) {
#pragma unroll
          for (int r = 0; r < RM; ++r) {
            run[r] = F::add(ctx, run[r], acc[r].fold(ctx));
            acc[r].zero();
          }
          terms = 0;
        }
#pragma unroll
        for (int r = 0; r < RM; ++r) {
          const int rr = (r0 + r < rows) ? r0 + r : rows - 1;  // clamp: duplicates are discarded below
          acc[r].mac(ctx, As[rr * K + k], b);
        }
        ++terms;
      }
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        if (r0 + r < rows)
          F::st(C + ((size_t)(row0 + r0 + r) * ldc + j) * F::LIMBS, F::add(ctx, run[r], acc[r].fold(ctx)));
      }
    }
  }
}

// A thin inner dimension against a long right factor -- Matrix::vandermonde(n, t + 1) times the coefficient matrix, the
// reference's own way of writing a sharing (test/scl/math/test_matrix.cc:342-365): the K <= KMAX rows of B a lane needs stay in
// registers (16-byte packs: two columns of a one-limb field per lane), A sits in LDS and is read as broadcasts, every row of C
// is one sum of K products folded once.  B is read once and C written once, whole lines both: HBM-bound like the share kernels
// (k_matmul re-reads B for every four rows of C and moves 8 bytes per lane).
template <class F, int VEC, int KMAX>
__global__ __launch_bounds__(BLOCK) void k_matmul_thin(typename F::Ctx ctx, u64* C, size_t ldc, const u64* A, size_t lda, const u64* B,
                                                       size_t ldb, int M, int K, size_t npacks) {
  extern __shared__ unsigned char smem_raw[];
  typename F::E* As = reinterpret_cast<typename F::E*>(smem_raw);  // [M][K]
  for (int i = threadIdx.x; i < M * K; i += BLOCK) As[i] = F::ld(A + ((size_t)(i / K) * lda + (i % K)) * F::LIMBS);
  __syncthreads();
  SCL_GRID_STRIDE(q, npacks) {
    const size_t off = q * VEC * F::LIMBS;
    Pack<F, VEC> b[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) b[k] = load_pack<F, VEC, true>(B + (size_t)k * ldb * F::LIMBS + off);
    for (int r = 0; r < M; ++r) {
      MatAcc<F> acc[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v].zero();
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
          const typename F::E a = As[r * K + k];
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v].mac(ctx, a, b[k].v[v]);
        }
      }
      Pack<F, VEC> y;
#pragma unroll
      for (int v = 0; v < VEC; ++v) y.v[v] = acc[v].fold(ctx);
      store_pack<F, VEC, true>(C + (size_t)r * ldc * F::LIMBS + off, y);
    }
  }
}

// Matrix::multiply(Matrix) for any shape (matrix.h:477-495 is an unbounded i-k-j loop): C tile TM x TN per workgroup, K walked in
// steps of TK through LDS (A tile row-major, B tile row-major), a thread owns RM x RN outputs -- rows ty*RM + r, columns
// tx + 16*c, so that a wave's B reads and C stores run along consecutive columns -- each a lazy accumulator folded every
// F::ACC_TERMS terms.  Edges are zero-filled on the way in and masked on the way out; K has no bound.  The micro-tile follows
// the accumulator's size (u128 for Mersenne61, column sums for the Montgomery fields): MatmulShape<F>.
template <class F>
struct MatmulShape {
  enum {
    RM = F::LIMBS == 4 ? 2 : (F::TAG == 2 ? 2 : 4),
    RN = F::LIMBS == 4 ? 1 : (F::TAG == 2 ? 2 : 4),
    TM = 16 * RM,
    TN = 16 * RN,
    TK = F::LIMBS == 4 ? 8 : 16
  };
};

// ksplit > 1: gridDim.y workgroups share a tile, each taking a slice of K of `kslice` columns (a multiple of TK) and writing its
// canonical partial product to C + blockIdx.y * cslice: short-and-wide-K shapes ((200 x 7000)(7000 x 300) is 20 tiles) then fill
// the chip, and one pass of Vector::sum over the slices finishes the product (canonical partial sums add exactly).
template <class F>
__global__ __launch_bounds__(BLOCK) void k_matmul_tiled(typename F::Ctx ctx, u64* C, size_t ldc, const u64* A, size_t lda,
                                                        const u64* B, size_t ldb, size_t M, size_t K, size_t N, size_t kslice,
                                                        size_t cslice) {
  typedef typename F::E E;
  typedef MatmulShape<F> S;
  constexpr int RM = S::RM, RN = S::RN, TM = S::TM, TN = S::TN, TK = S::TK;
  __shared__ E As[TM * TK];  // [row][k]
  __shared__ E Bs[TK * TN];  // [k][col]
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const size_t tiles_n = (N + TN - 1) / TN, tiles_m = (M + TM - 1) / TM;
  for (size_t tile = blockIdx.x; tile < tiles_m * tiles_n; tile += gridDim.x) {
    const size_t row0 = (tile / tiles_n) * TM, col0 = (tile % tiles_n) * TN;
    MatAcc<F> acc[RM][RN];
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
      for (int c = 0; c < RN; ++c) acc[r][c].zero();
    int terms = 0;
    const size_t kbeg = blockIdx.y * kslice, kend = kbeg + kslice < K ? kbeg + kslice : K;
    for (size_t k0 = kbeg; k0 < kend; k0 += TK) {
      __syncthreads();  // the previous step's reads are done
      for (int i = threadIdx.x; i < TM * TK; i += BLOCK) {
        const size_t r = row0 + i / TK, k = k0 + i % TK;
        As[i] = (r < M && k < kend) ? F::ld(A + (r * lda + k) * F::LIMBS) : F::zero();
      }
      for (int i = threadIdx.x; i < TK * TN; i += BLOCK) {
        const size_t k = k0 + i / TN, c = col0 + i % TN;
        Bs[i] = (k < kend && c < N) ? F::ld(B + (k * ldb + c) * F::LIMBS) : F::zero();
      }
      __syncthreads();
      if (terms + TK > (int)MatAcc<F>::TERMS) {  // fold: the accumulator restarts from its own canonical value
#pragma unroll
        for (int r = 0; r < RM; ++r)
#pragma unroll
          for (int c = 0; c < RN; ++c) {
            const E f = acc[r][c].fold(ctx);
            acc[r][c].zero();
            acc[r][c].add(ctx, f);
          }
        terms = 1;
      }
#pragma unroll 4
      for (int k = 0; k < TK; ++k) {
        E a[RM], b[RN];
#pragma unroll
        for (int r = 0; r < RM; ++r) a[r] = As[(ty * RM + r) * TK + k];
#pragma unroll
        for (int c = 0; c < RN; ++c) b[c] = Bs[k * TN + tx + 16 * c];
#pragma unroll
        for (int r = 0; r < RM; ++r)
#pragma unroll
          for (int c = 0; c < RN; ++c) acc[r][c].mac(ctx, a[r], b[c]);
      }
      terms += TK;
    }
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
      for (int c = 0; c < RN; ++c) {
        const size_t row = row0 + ty * RM + r, col = col0 + tx + 16 * c;
        if (row < M && col < N) F::st(C + blockIdx.y * cslice + (row * ldc + col) * F::LIMBS, acc[r][c].fold(ctx));
      }
  }
}

// Matrix::multiply(Vector) (matrix.h:497-513: one innerProd per row) and any product with ONE column: a wavefront per row, its
// lanes stride over the row (coalesced), lazy accumulation, a shuffle reduction -- rows spread over the whole chip instead of
// the single workgroup a one-column launch of the column-per-thread kernels gets.  x is read with stride ldb (a column of B).
template <class F>
__global__ __launch_bounds__(BLOCK) void k_matvec(typename F::Ctx ctx, u64* y, size_t ldc, const u64* A, size_t lda, const u64* x,
                                                  size_t ldb, size_t M, size_t K) {
  typedef typename F::E E;
  const int lane = threadIdx.x & 63;
  for (size_t row = (size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); row < M; row += (size_t)gridDim.x * (BLOCK / 64)) {
    MatAcc<F> acc;
    acc.zero();
    E run = F::zero();
    int terms = 0;
    for (size_t k = lane; k < K; k += 64) {
      if (terms + 1 > (int)MatAcc<F>::TERMS) {
        run = F::add(ctx, run, acc.fold(ctx));
        acc.zero();
        terms = 0;
      }
      acc.mac(ctx, F::ld(A + (row * lda + k) * F::LIMBS), F::ld(x + k * ldb * F::LIMBS));
      ++terms;
    }
    run = F::add(ctx, run, acc.fold(ctx));
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) run = F::add(ctx, run, shfl_xor_elem<F>(run, m));
    if (lane == 0) F::st(y + row * ldc * F::LIMBS, run);
  }
}

// ---- layout ----------------------------------------------------------------------------------------------
// AoS [N][n] (reference Vector per secret) <-> SoA [n][stride], staged through LDS so that both the
// global reads and the global writes are contiguous.  Tile = TS secrets x n parties.
template <int LIMBS, bool TO_SOA>
__global__ __launch_bounds__(BLOCK) void k_transpose(u64* dst, const u64* src, size_t stride, size_t N, int n,
                                                     int tile_secrets) {
  extern __shared__ unsigned char smem_raw[];
  u64* tile = reinterpret_cast<u64*>(smem_raw);  // [tile_secrets][n] elements, AoS order
  const size_t ntiles = (N + tile_secrets - 1) / tile_secrets;
  for (size_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
    const size_t s0 = tix * tile_secrets;
    const int ts = (int)min((size_t)tile_secrets, N - s0);
    const int words = ts * n * LIMBS;
    if constexpr (TO_SOA) {
      const u64* a = src + s0 * n * LIMBS;
      for (int w = threadIdx.x; w < words; w += BLOCK) tile[w] = a[w];
      __syncthreads();
      for (int e = threadIdx.x; e < ts * n; e += BLOCK) {
        const int i = e / ts, s = e % ts;
#pragma unroll
        for (int l = 0; l < LIMBS; ++l) dst[((size_t)i * stride + s0 + s) * LIMBS + l] = tile[(s * n + i) * LIMBS + l];
      }
    } else {
      for (int e = threadIdx.x; e < ts * n; e += BLOCK) {
        const int i = e / ts, s = e % ts;
#pragma unroll
        for (int l = 0; l < LIMBS; ++l) tile[(s * n + i) * LIMBS + l] = src[((size_t)i * stride + s0 + s) * LIMBS + l];
      }
      __syncthreads();
      u64* a = dst + s0 * n * LIMBS;
      for (int w = threadIdx.x; w < words; w += BLOCK) a[w] = tile[w];
    }
    __syncthreads();
  }
}

// The same bridge with 16-byte global accesses on BOTH sides (k_transpose moves 8 bytes per lane on its strided side).  The
// LDS tile is the AoS image of TS secrets, copied linearly 16 bytes per lane; on the SoA side a lane owns 16 consecutive bytes
// of ONE party's row -- two consecutive secrets of a one-limb field, one element of a two-limb field, half an element of a
// four-limb one -- so every global instruction of a wave is 1 KiB of whole lines, and the strided walk happens in LDS
// (n x LIMBS x 8 bytes between a lane's neighbours: a few-way bank conflict the array has room for at these rates).
// Needs 16-byte aligned bases and, for one-limb fields, an even row stride; ragged last tiles take the scalar tail below.
template <int LIMBS, bool TO_SOA>
__global__ __launch_bounds__(BLOCK) void k_transpose16(u64* dst, const u64* src, size_t stride, size_t N, int n, int tile_secrets) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u64* tile = reinterpret_cast<u64*>(smem_raw);  // [tile_secrets][n][LIMBS] words, AoS order
  u64x2* tile16 = reinterpret_cast<u64x2*>(smem_raw);
  constexpr int PER = LIMBS == 1 ? 2 : 1;             // secrets per 16-byte piece (one-limb fields)
  constexpr int PIECES = LIMBS == 1 ? 1 : LIMBS / 2;  // 16-byte pieces per element
  const size_t ntiles = (N + tile_secrets - 1) / tile_secrets;
  for (size_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
    const size_t s0 = tix * tile_secrets;
    const int ts = (int)min((size_t)tile_secrets, N - s0);
    const int words = ts * n * LIMBS;  // (s0 * n * LIMBS is even: tile_secrets is)
    if constexpr (TO_SOA) {
      const u64* aos = src + s0 * n * LIMBS;
      for (int c = threadIdx.x; c < words / 2; c += BLOCK) tile16[c] = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(aos) + c);
      if ((words & 1) && threadIdx.x == 0) tile[words - 1] = aos[words - 1];
      __syncthreads();
    }
    // SoA side: one flat index over (party, 16-byte piece of its row segment [s0, s0 + ts)): every thread has work whatever the
    // tile size, and all of a tile's strided-side accesses are in flight together.  The piece count is a power of two on full
    // tiles (tile_secrets is): a shift splits the index.
    const int npieces = LIMBS == 1 ? ts / 2 : ts * PIECES;
    const int total = n * npieces;
    const bool pow2 = npieces > 0 && (npieces & (npieces - 1)) == 0;
    const int sh = pow2 ? __builtin_ctz((unsigned)npieces) : 0;
    for (int idx = threadIdx.x; idx < total; idx += BLOCK) {
      const int i = pow2 ? idx >> sh : idx / npieces;
      const int p = pow2 ? idx & (npieces - 1) : idx % npieces;
      const size_t row_off = ((size_t)i * stride + s0) * LIMBS;  // party i's row segment
      if constexpr (LIMBS == 1) {
        const int s = p * PER;
        if constexpr (TO_SOA) {
          u64x2 v;
          v.x = tile[s * n + i];
          v.y = tile[(s + 1) * n + i];
          __builtin_nontemporal_store(v, reinterpret_cast<u64x2*>(dst + row_off) + p);
        } else {
          const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(src + row_off) + p);
          tile[s * n + i] = v.x;
          tile[(s + 1) * n + i] = v.y;
        }
      } else {
        const int s = p / PIECES, c = p % PIECES;
        if constexpr (TO_SOA) __builtin_nontemporal_store(tile16[(s * n + i) * PIECES + c], reinterpret_cast<u64x2*>(dst + row_off) + p);
        else tile16[(s * n + i) * PIECES + c] = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(src + row_off) + p);
      }
    }
    if constexpr (LIMBS == 1) {
      if (ts & 1) {  // the odd secret out of a ragged last tile
        for (int i = threadIdx.x; i < n; i += BLOCK) {
          const size_t at = (size_t)i * stride + s0 + ts - 1;
          if constexpr (TO_SOA) dst[at] = tile[(ts - 1) * n + i];
          else tile[(ts - 1) * n + i] = src[at];
        }
      }
    }
    if constexpr (!TO_SOA) {
      __syncthreads();
      u64* aos = dst + s0 * n * LIMBS;
      for (int c = threadIdx.x; c < words / 2; c += BLOCK) __builtin_nontemporal_store(tile16[c], reinterpret_cast<u64x2*>(aos) + c);
      if ((words & 1) && threadIdx.x == 0) aos[words - 1] = tile[words - 1];
    }
    __syncthreads();  // the tile is free for the next trip
  }
}

// ---- wire image --------------------------------------------------------------------------------------
// seri::Serializer<Vector<FF>> (include/scl/serialization/serializer.h:157-190, ff.h:355-391): u32 count
// then count elements as FF::write emits them.  The payload starts 4 bytes into the buffer, so it is moved
// as 32-bit words (buffers must be 4-byte aligned).
// Serializer<Matrix> (matrix.h:910-963) puts u32 rows, u32 cols in front of the vector image of the row-major
// values; WireGeom carries that header and the row pitch of the device matrix (ld >= cols elements).
struct WireGeom {
  u32 hdr[4];
  int nhdr;         // 1: Vector (count)   3: Matrix (rows, cols, count)   +1 in front for a frame (packet size)
  size_t cols, ld;  // element e of the image lives at element (e / cols) * ld + e % cols of the device buffer
  __device__ __forceinline__ size_t at(size_t e) const { return ld == cols ? e : (e / cols) * ld + e % cols; }
};

template <class F>
__global__ __launch_bounds__(BLOCK) void k_wire_pack(typename F::Ctx ctx, u32* dst, const u64* src, size_t n,
                                                     WireGeom g) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int i = 0; i < g.nhdr; ++i) dst[i] = g.hdr[i];
  }
  SCL_GRID_STRIDE(e, n) {
    u32* o = dst + g.nhdr + e * (2 * F::LIMBS);
    const size_t se = g.at(e);
    if constexpr (F::LIMBS == 1) {
      const u64 v = src[se];
      o[0] = (u32)v;
      o[1] = (u32)(v >> 32);
    } else if constexpr (F::LIMBS == 2) {
      u128 v = F::ld(src + 2 * se);
      if constexpr (F::TAG == 2) v = bswap128(F::from_mont(ctx, v));  // Montgomery family: value, big-endian
      o[0] = (u32)v;
      o[1] = (u32)(v >> 32);
      o[2] = (u32)(v >> 64);
      o[3] = (u32)(v >> 96);
    } else {
      const typename F::E img = F::to_be_image(ctx, F::ld(src + 4 * se));  // montyToBytes
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[2 * j] = (u32)img.w[j];
        o[2 * j + 1] = (u32)(img.w[j] >> 32);
      }
    }
  }
}

template <class F>
__global__ __launch_bounds__(BLOCK) void k_wire_unpack(typename F::Ctx ctx, u64* dst, const u32* src, size_t n,
                                                       WireGeom g) {
  SCL_GRID_STRIDE(e, n) {
    const u32* in = src + g.nhdr + e * (2 * F::LIMBS);
    const size_t de = g.at(e);
    if constexpr (F::LIMBS == 1) {
      dst[de] = F::from_le_word(ctx, (u64)in[0] | ((u64)in[1] << 32));
    } else if constexpr (F::LIMBS == 2) {
      const u128 raw = (u128)in[0] | ((u128)in[1] << 32) | ((u128)in[2] << 64) | ((u128)in[3] << 96);
      F::st(dst + 2 * de, F::from_le_word(ctx, raw));
    } else {
      typename F::E raw;
#pragma unroll
      for (int j = 0; j < 4; ++j) raw.w[j] = (u64)in[2 * j] | ((u64)in[2 * j + 1] << 32);
      F::st(dst + 4 * de, F::from_le_word(ctx, raw));
    }
  }
}

// ---- roofline probe ------------------------------------------------------------------------------------
template <int = 0>  // (a template so that only the unit that launches it compiles it)
__global__ __launch_bounds__(BLOCK) void k_copy16(u64x2* dst, const u64x2* src, size_t n16) {
  SCL_GRID_STRIDE(q, n16) __builtin_nontemporal_store(__builtin_nontemporal_load(src + q), dst + q);
}

}  // namespace sclhip
