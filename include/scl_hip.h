/* include/scl_hip.h -- the drop-in boundary: a C ABI over the MI355X (gfx950)
 * finite-field / secret-sharing engine in libscl_hip.so.
 *
 * The reference (anderspkd/secure-computation-library 0.1.0) has no FFI of its
 * own: its batch-shaped call sites are C++ header templates that own their
 * storage in std::vector.  Each entry point below names the
 * reference interface it replaces (paths relative to the reference tree);
 * INTEGRATION.md shows the binding a maintainer would add on the reference
 * side.
 *
 * Conventions
 *  - field tags: SCL_M61 = scl::math::ff::Mersenne61 (1 uint64 limb),
 *    SCL_M127 = Mersenne127 (2 limbs, little-endian, 16-byte aligned),
 *    SCL_SECP256K1_SCALAR = scl::math::ff::Secp256k1Scalar, the secp256k1 group
 *    order, and SCL_SECP256K1_FIELD = scl::math::ff::Secp256k1Field, the prime the
 *    curve is defined over (src/scl/math/fields/secp256k1_field.cc:43-135) -- the
 *    two N = 4 instances of the reference's Montgomery family (4 limbs, 16-byte aligned); SCL_MONT128 is the N = 2 instance of
 *    the same family (the templates of include/scl/math/fields/ff_ops_gmp.h:44-392, which the reference itself instantiates at
 *    N = 4 only; pinned by tests/golden/golden_mont128.json, emitted by those templates compiled at two limbs) over a run-time
 *    128-bit modulus, SCL_GF2_128 a plug-in field the reference does not have (GHASH's field; 2 limbs each).  An element's limbs are the
 *    in-memory image of FF::m_value (what std::vector<FF>::data() holds): the
 *    canonical integer for the Mersenne fields, the Montgomery residue
 *    x*2^256 mod p for secp256k1_order exactly as the reference keeps it
 *    (src/scl/math/fields/secp256k1_scalar.cc:47-135), x*2^128 mod p for MONT128.
 *  - values are canonical (in [0,p)) on entry and on exit.
 *  - "dev" pointers are device (HBM) pointers, "host" pointers are host
 *    memory.  The caller owns every buffer.
 *  - share matrices are SoA: row i is party i's share vector, element s of
 *    row i lives at base + (i*stride + s)*limbs uint64 words; stride >= N lets
 *    a caller pass a window of a larger matrix (sharding, chunking).  An AoS
 *    [secret][party] image (the reference's Vector per secret) converts with
 *    scl_hip_aos_to_soa / scl_hip_soa_to_aos.
 *  - stream is a hipStream_t passed as void* (NULL = the default stream);
 *    calls are asynchronous w.r.t. the host unless they return a host value or
 *    must report a data-dependent error (documented per function).
 *  - every function returns an scl_status; scl_hip_last_error() gives the
 *    thread's last diagnostic and scl_hip_status_message() the text of the
 *    exception the reference throws for that condition.
 *  - thread-safe: callable concurrently from several host threads on different streams / devices.  The only
 *    mutable state is per host thread (a scratch buffer, two temporary arenas -- kept and grown: at most 1 GiB for the
 *    coefficient rows of PRG-driven sharing in two passes, otherwise what one call's tables and queues need --, the
 *    tuning knobs: each thread sets its own and starts from the defaults), the Mont128 modulus (a process-wide default
 *    behind a mutex that a thread latches at its first use or replaces with scl_hip_mont128_set_prime; see there) or immutable
 *    once built (device tables of Vandermonde rows, behind a mutex; a caller holds a reference to the table it uses
 *    until its kernel is enqueued, and past 16 entries the least recently used one is freed once nobody holds it).
 *    A thread that exits calls scl_hip_thread_cleanup() to release its device buffers.
 */
#ifndef SCL_HIP_H
#define SCL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { SCL_M61 = 0, SCL_M127 = 1, SCL_MONT128 = 2, SCL_GF2_128 = 3, SCL_SECP256K1_SCALAR = 4,
               SCL_SECP256K1_FIELD = 5 } scl_field;
/* Rings scl::math::Z2k<K> (include/scl/math/z2k.h:39-320, z2k/z2k_ops.h:32-150), 1 <= K <= 128: tag
 * SCL_Z2K(K).  One limb for K <= 64, two above (Z2k::ValueType); values are taken modulo 2^K on entry and are
 * returned masked.  Accepted by the entry points that make sense in a ring: scl_hip_ew (INV / DIV need odd
 * operands, else SCL_ERR_NOT_INVERTIBLE_2K), scalar_mul, sum, dot, equals (compares modulo 2^K), from_bytes and
 * vector_random (stride Z2k::byteSize() = (K-1)/8 + 1 bytes), additive_share(_prg), additive_recover, matmul,
 * aos_to_soa / soa_to_aos.  The Shamir / Lagrange / Vandermonde entry points refuse ring tags, and so do the
 * wire functions: the reference has no Serializer for Z2k. */
#define SCL_Z2K(K) (0x100 + (K))

typedef enum { SCL_OP_ADD = 0, SCL_OP_SUB = 1, SCL_OP_MUL = 2, SCL_OP_NEG = 3, SCL_OP_INV = 4,
               SCL_OP_DIV = 5 } scl_op;

typedef enum {
  SCL_OK = 0,
  SCL_ERR_SIZE_MISMATCH = 1,  /* std::invalid_argument("Vec sizes mismatch")             vector.h:481-485 */
  SCL_ERR_ZERO_INVERSE = 2,   /* std::logic_error("0 not invertible modulo prime")       small_ff.h:70    */
  SCL_ERR_BAD_ARG = 3,
  SCL_ERR_HIP = 4,            /* a HIP runtime call failed; see scl_hip_last_error()                       */
  SCL_ERR_NO_DEVICE = 5,
  SCL_ERR_ERROR_DETECTED = 6, /* std::logic_error("error detected during recovery")      shamir.h:135     */
  SCL_ERR_NOT_ENOUGH_SHARES = 7, /* std::logic_error("not enough shares provided to detect errors") shamir.h:123 */
  SCL_ERR_MATMUL_DIMS = 8,    /* std::invalid_argument("matmul: this->cols() != that->rows()") matrix.h:480 */
  SCL_ERR_VANDERMONDE_XS = 9, /* std::invalid_argument("|xs| != number of rows")         matrix.h:449     */
  SCL_ERR_INVALID_RANGE = 10, /* std::invalid_argument("invalid range")                  vector.h:493     */
  SCL_ERR_NOT_INVERTIBLE_2K = 11 /* std::invalid_argument("value not invertible modulo 2^K") z2k/z2k_ops.h:82 */
} scl_status;

/* Version of this boundary.  2: scl_hip_ew_status (round 6), scl_hip_open_reduce_scatter, scl_hip_mont128_relatch and the
 * export map (round 5: nothing but these prototypes is a dynamic symbol).  A binding checks it BEFORE it looks up symbols:
 * a stale library then fails with a version message, not with an undefined symbol. */
#define SCL_HIP_ABI_VERSION 2
int scl_hip_abi_version(void);
const char* scl_hip_last_error(void);
const char* scl_hip_status_message(int status);
int scl_hip_limbs(int field);              /* FF::byteSize()/8, ff.h:43-45 */
const char* scl_hip_field_name(int field); /* FF::name(), ff.h:57-59 */

/* ---- device plumbing (for callers without their own HIP runtime) ---------- */
int scl_hip_device_count(int* count);
int scl_hip_set_device(int device);
int scl_hip_malloc(void** dev, size_t bytes);
int scl_hip_free(void* dev);
int scl_hip_memcpy_h2d(void* dev, const void* host, size_t bytes, void* stream);
int scl_hip_memcpy_d2h(void* host, const void* dev, size_t bytes, void* stream);
int scl_hip_memset(void* dev, int value, size_t bytes, void* stream);
int scl_hip_stream_create(void** stream);
int scl_hip_stream_destroy(void* stream);
int scl_hip_stream_sync(void* stream);
/* HIP-event timing on `stream`: start, stop -> elapsed milliseconds */
int scl_hip_timer_create(void** timer);
int scl_hip_timer_destroy(void* timer);
int scl_hip_timer_start(void* timer, void* stream);
int scl_hip_timer_stop(void* timer, void* stream);
int scl_hip_timer_elapsed_ms(void* timer, float* ms); /* synchronises on the stop event */
/* knobs for tuning and for the tests that pin a kernel path (0 = built-in default unless noted); they belong to the
 * calling host thread:
 *   max_blocks, aes_blocks   grid caps;  nontemporal (default 1);  force_scalar (no 16-byte packs)
 *   force_table  1: no small-node / blocked / Vandermonde-table share kernels, 2: also no small-node Horner,
 *                3: GF(2^128) reconstruct on the shared-shift nibble tables instead of the position tables
 *   stream_block (default 64) workgroup size of the (m <= 16) reconstruct kernel: 64 or 256
 *   stream_waves (default -1) resident waves per CU that kernel is capped at; 0 = no cap, -1 = 10 for one-word
 *                elements and 12 for wider ones
 *   share_waves  (default 9) the same cap for the Mersenne61 small-node share kernel (threshold compiled in, stream_block
 *                workgroups); 0 = the 256-thread kernel with the threshold at run time, no cap
 *   share_waves128 (default 12) the same for the 16-byte fields' small-node share kernel (Mersenne127, Mont128); 0 = the
 *                256-thread kernel
 *   mfma         1: force the matrix-core share / matmul / recover_detect path, -1: never use it, 2: as 1 with matmul's
 *                (row block, k-chunk) form on the sharing kernels instead of the general kernel
 *   open_gather_always (default 0) scl_hip_open_all_gather on a ONE-rank communicator: 1 = through the collective path all the
 *                same (tests); 0 = reconstruct straight from the slab
 *   prg_t3       (default 1) PRG-driven sharing at t = 3 over the Mersenne fields: the fused kernel with the threshold
 *                compiled in; 0 = the fused kernel that takes any t <= 7
 *   gf_tiles     (default 1) GF(2^128) sharing at the default nodes, 5 <= t <= 16: eight nodes per Horner loop; 0 = one node
 *                at a time (the kernel that serves any other small nodes)
 *   prg_two_pass PRG-driven sharing: 1 = always draw the coefficient rows into a temporary and share from there, -1 = always the
 *                fused kernels (where one exists), 0 (default) = by shape and field
 *   inv_batch    element-wise inverse / divide: 0 (default) simultaneous inversion with the chain length chosen by the batch,
 *                8 | 16 | 32 | 64 | 128 that chain length, -1 one Fermat chain per element and GF(2^128)'s register-only product
 *   inv_two_level  the chained inversion of the 16-byte prime fields in two levels (checkpoints + recomputed blocks: 3 E instead of
 *                5 E bytes per element, one more product): 0 (default) by field and chain length (Mersenne127 from chains of 64),
 *                4 | 8 that block length for chains of 32 and more (also Mont128), -1 never; with inv_batch 256 a chain of 256
 *   matmul_lds_min  columns from which the thread-per-column matrix kernels are taken instead of the tiled one (0 = by shape);
 *   transpose_tile  secrets per LDS tile of the 16-byte layout bridge (0 = up to 512);  gemm_slab_mib  digit planes per factor
 *                and launch of the general matrix-core product in MiB (0 = 1024)
 *   mfma_areg    (default 1) matrix-core kernel keeps V's digit fragments in registers for 97..128 parties
 *   mfma_pipe    (default 2) matrix-core kernel for 97..128 parties: 2 = two software-pipelined waves per SIMD on
 *                16x16x64 tiles (thresholds 32..63; smaller ones as 1), 1 = one pipelined wave per SIMD, 0 = word bursts */
int scl_hip_set_tuning(const char* key, long value);
/* frees the calling thread's device scratch and temporary arena (for host threads that exit; see Conventions) */
int scl_hip_thread_cleanup(void);

/* ---- element-wise: scl::math::Vector<FF> members ------------------------- */
/* Vector::add / subtract / multiplyEntryWise (+InPlace) (vector.h:199-245,521-556),
 * FF::negate / invert / operator/ per element (ff.h:203-246).  dst may alias a or b.
 * b is ignored for NEG and INV.  INV and DIV are synchronous: a zero operand yields
 * SCL_ERR_ZERO_INVERSE after the launch completes (the reference throws at the
 * first zero; here every other element is still computed, the slot of a zero -- of an even element, in a ring -- gets 0). */
int scl_hip_ew(int field, int op, uint64_t* dst_dev, const uint64_t* a_dev, const uint64_t* b_dev,
               size_t n, void* stream);
/* The same call, asynchronous for every op: the "an operand was not invertible" report of INV / DIV (the reference's throw,
 * small_ff.h:61-70, ff_ops_gmp.h:250-260, z2k_ops.h:81-83) stays on the device.  status_dev points to ONE 32-bit word in
 * device memory: the call ORs 1 into it when an operand of INV / DIV is zero (even, in a ring) -- every other element is still
 * computed, the offending slot gets 0 -- and never clears it, so one word can collect a sequence of calls; clear it with
 * scl_hip_memset / hipMemsetAsync on the same stream, read it whenever the caller synchronises anyway (cf. the status vector of
 * scl_hip_shamir_recover_detect).  No host synchronisation, no host read: small batches do not pay a stream round trip and the
 * call can be captured into a hipGraph.  status_dev may be NULL for ops that cannot fail (ADD, SUB, MUL, NEG). */
int scl_hip_ew_status(int field, int op, uint64_t* dst_dev, const uint64_t* a_dev, const uint64_t* b_dev,
                      size_t n, unsigned* status_dev, void* stream);
/* Vector::scalarMultiply(InPlace) (vector.h:274-301): dst[i] = scalar * a[i] */
int scl_hip_scalar_mul(int field, uint64_t* dst_dev, const uint64_t* a_dev,
                       const uint64_t* scalar_host, size_t n, void* stream);
/* Vector::sum (vector.h:261-267) and Vector::dot / innerProd (vector.h:45-52,252-255);
 * synchronous, result to host. */
int scl_hip_sum(int field, uint64_t* out_host, const uint64_t* a_dev, size_t n, void* stream);
int scl_hip_dot(int field, uint64_t* out_host, const uint64_t* a_dev, const uint64_t* b_dev,
                size_t n, void* stream);
/* Vector::equals (vector.h:558-570): *equal_host = 1 iff all n elements match; synchronous */
int scl_hip_equals(int field, int* equal_host, const uint64_t* a_dev, const uint64_t* b_dev,
                   size_t n, void* stream);

/* ---- randomness: scl::util::PRG and FF::read ------------------------------ */
/* PRG::next as a counter-addressed stream (src/scl/util/prg.cc:124-146): writes
 * nblocks*16 bytes, block i = AES128_key(LE64(counter0+i) || LE64(PRG_NONCE)), key = seed
 * zero-padded / truncated to 16 bytes (prg.cc:88-101). */
int scl_hip_prg_blocks(unsigned char* dst_dev, size_t nblocks, const unsigned char* seed_host,
                       size_t seed_len, uint64_t counter0, void* stream);
/* FF::read / ff::fromBytes (mersenne61.cc:86-90, mersenne127.cc:114-118): n elements from
 * n*byteSize raw bytes, reduced mod p. */
int scl_hip_from_bytes(int field, uint64_t* dst_dev, const unsigned char* src_dev, size_t n,
                       void* stream);
/* Vector::random(n, prg) for a PRG whose counter stands at counter0 (vector.h:507-519):
 * consumes ceil(n*byteSize/16) blocks. */
int scl_hip_vector_random(int field, uint64_t* dst_dev, size_t n, const unsigned char* seed_host,
                          size_t seed_len, uint64_t counter0, void* stream);

/* ---- Shamir: scl::ss::shamirSecretShare / shamirRecoverP ------------------ */
/* computeLagrangeBasis(nodes, x) (include/scl/math/lagrange.h:54-71) on the host:
 * lambda[i] = prod_{j!=i} (x - a_j)/(a_i - a_j).  alphas_host == NULL means
 * Vector::range(1, m+1) (vector.h:490-505); x_host == NULL means 0.  Duplicate nodes give
 * SCL_ERR_ZERO_INVERSE like the reference's exception. */
int scl_hip_lagrange_basis(int field, uint64_t* lambda_host, const uint64_t* alphas_host, size_t m,
                           const uint64_t* x_host);
/* Batched shamirSecretShare (include/scl/ss/shamir.h:51-68) with explicit coefficients:
 * shares[i][s] = secrets[s] + sum_{k=1..t} coeffs[k-1][s] * alpha_i^k for i < n.
 * coeffs is SoA [t][N] with row stride coeff_stride; alphas_host NULL = 1..n. */
int scl_hip_shamir_share(int field, uint64_t* shares_dev, size_t share_stride,
                         const uint64_t* secrets_dev, const uint64_t* coeffs_dev,
                         size_t coeff_stride, size_t N, size_t t, size_t n,
                         const uint64_t* alphas_host, void* stream);
/* The same driven by the reference's PRG discipline: bit-identical to calling
 * shamirSecretShare(secret_s, t, n, prg) for s = 0..N-1 on a PRG seeded with `seed` whose block
 * counter stands at counter0 when the batch begins (secret s draws Vector::random(t+1) from blocks
 * [counter0 + s*B, counter0 + (s+1)*B), B = ceil((t+1)*byteSize/16); the draw for c_0 is made and
 * discarded, shamir.h:56-57).  A shard that starts at secret index f of a longer run passes
 * counter0 = f*B.  The PRG has consumed N*B blocks afterwards. */
int scl_hip_shamir_share_prg(int field, uint64_t* shares_dev, size_t share_stride,
                             const uint64_t* secrets_dev, size_t N, size_t t, size_t n,
                             const unsigned char* seed_host, size_t seed_len,
                             uint64_t counter0, void* stream);
/* shamirSecretShare over math::Array<FF, W> (include/scl/math/array.h:69-415) -- what pedersenSecretShare runs with
 * W = 2: {secret, blinding} (include/scl/ss/pedersen.h:127-140).  Vector<Array>::random(t+1) is one draw of
 * (t+1)*W elements and component j of coefficient k is element k*W + j; arithmetic is component-wise, nodes 1..n.
 * secrets_dev: component j at secrets_dev + j * secret_stride elements; shares_dev: component j, party i at
 * shares_dev + (j * n + i) * share_stride elements.  Secret s starts at block counter0 + s * ceil((t+1)*W*byteSize/16).
 * The EC commitments of Feldman / Pedersen stay with the caller. */
int scl_hip_shamir_share_prg_packed(int field, uint64_t* shares_dev, size_t share_stride, const uint64_t* secrets_dev,
                                    size_t secret_stride, size_t N, size_t t, size_t n, size_t width,
                                    const unsigned char* seed_host, size_t seed_len, uint64_t counter0, void* stream);
/* Batched shamirRecoverP (shamir.h:81-104) with the basis hoisted out of the per-secret
 * call: out[s] = sum_{i<m} lambda[i] * shares[i][s]. */
int scl_hip_shamir_recover(int field, uint64_t* out_dev, const uint64_t* shares_dev,
                           size_t share_stride, const uint64_t* lambda_host, size_t m, size_t N,
                           void* stream);
/* Batched shamirRecoverD(shares, alphas, t, d, x) (shamir.h:116-139): the first d+1 shares
 * define the polynomial, shares d+1..d+t-1 are checked.  status_dev[s] = 1 where the
 * reference would throw "error detected during recovery" (out[s] = 0 there).
 * Synchronous; returns SCL_ERR_ERROR_DETECTED if any status is set, with
 * *num_bad_host = how many.  alphas_host NULL = 1..m, x_host NULL = 0. */
int scl_hip_shamir_recover_detect(int field, uint64_t* out_dev, unsigned char* status_dev,
                                  const uint64_t* shares_dev, size_t share_stride, size_t m,
                                  size_t N, size_t t, size_t d, const uint64_t* alphas_host,
                                  const uint64_t* x_host, size_t* num_bad_host, void* stream);

/* Batched shamirRecoverC(shares, alphas) -- Berlekamp-Welch error correction (shamir.h:202-259, with
 * solveLinearSystem matrix.h:811-828 and Polynomial::divide poly.h:261-278).  t = (m - 1) / 3 and only the first
 * n = 3t + 1 shares are used, as in the reference; n is unbounded, as there.  Per secret s:
 *   f_dev[k * f_stride + s], k < n      coefficients of the corrected polynomial f (zero padded); the secret is
 *                                       f(0) = row 0
 *   e_dev[k * e_stride + s], k <= t     the monic error locator E (zero padded); nerr_dev[s] = its degree
 *   status_dev[s]                       1 where the reference throws std::logic_error("could not correct shares")
 *                                       (f, E and nerr zeroed there)
 * Secrets whose n shares already lie on one polynomial of degree <= t take a streaming kernel (E = 1); the rest are
 * queued and solved one workgroup each (one wavefront up to 64 shares; the systems in LDS while they fit, in device
 * memory beyond that).  Synchronous.  *num_queued_host = secrets that needed the solver,
 * *num_failed_host = how many of them could not be corrected (either may be NULL).  alphas_host NULL = 1..n. */
int scl_hip_shamir_recover_correct(int field, uint64_t* f_dev, size_t f_stride, uint64_t* e_dev, size_t e_stride,
                                   unsigned char* status_dev, unsigned* nerr_dev, const uint64_t* shares_dev,
                                   size_t share_stride, size_t m, size_t N, const uint64_t* alphas_host,
                                   size_t* num_queued_host, size_t* num_failed_host, void* stream);

/* ---- additive: scl::ss::additiveShare, Vector::sum ------------------------ */
/* additiveShare (include/scl/ss/additive.h:41-53) with explicit randomness rnd [n-1][N]:
 * shares[i] = rnd[i] for i < n-1, shares[n-1] = secret - sum. */
int scl_hip_additive_share(int field, uint64_t* shares_dev, size_t share_stride,
                           const uint64_t* secrets_dev, const uint64_t* rnd_dev, size_t rnd_stride,
                           size_t N, size_t n, void* stream);
/* PRG-driven: share i < n-1 of secret s is FF::random on block counter0 + s*(n-1) + i (one whole
 * AES block per element, ff.h:72-76); N*(n-1) blocks are consumed. */
int scl_hip_additive_share_prg(int field, uint64_t* shares_dev, size_t share_stride,
                               const uint64_t* secrets_dev, size_t N, size_t n,
                               const unsigned char* seed_host, size_t seed_len,
                               uint64_t counter0, void* stream);
/* reconstruct = Vector::sum per secret (vector.h:261-267): out[s] = sum_i shares[i][s] */
int scl_hip_additive_recover(int field, uint64_t* out_dev, const uint64_t* shares_dev,
                             size_t share_stride, size_t n, size_t N, void* stream);

/* ---- matrices: scl::math::Matrix ------------------------------------------ */
/* Matrix::vandermonde(n, m, xs) (matrix.h:444-460), row-major into V_dev; xs NULL = 1..n */
int scl_hip_vandermonde(int field, uint64_t* V_dev, size_t n, size_t m, const uint64_t* xs_host,
                        void* stream);
/* Matrix::multiply (matrix.h:477-495): C[M x N] = A[M x K] * B[K x N], all row-major, with
 * leading dimensions lda/ldb/ldc in elements. */
int scl_hip_matmul(int field, uint64_t* C_dev, size_t ldc, const uint64_t* A_dev, size_t lda,
                   const uint64_t* B_dev, size_t ldb, size_t M, size_t K, size_t N, void* stream);

/* ---- layout ---------------------------------------------------------------- */
/* AoS [N][n] (one reference Vector per secret) <-> SoA [n][stride] */
int scl_hip_aos_to_soa(int field, uint64_t* soa_dev, size_t stride, const uint64_t* aos_dev,
                       size_t N, size_t n, void* stream);
int scl_hip_soa_to_aos(int field, uint64_t* aos_dev, const uint64_t* soa_dev, size_t stride,
                       size_t N, size_t n, void* stream);

/* ---- wire image: seri::Serializer<Vector<FF>> ------------------------------------------------ */
/* What `packet << vector` puts on the wire (include/scl/serialization/serializer.h:157-190,
 * include/scl/math/ff.h:355-391, vector.h:595-629): u32 count (little-endian) followed by count
 * elements as FF::write emits them.  Buffers must be 4-byte aligned. */
size_t scl_hip_wire_size(int field, size_t n); /* 4 + n * byteSize */
int scl_hip_wire_pack(int field, unsigned char* dst_dev, const uint64_t* src_dev, size_t n, void* stream);
/* Reads the count (synchronous), checks it against nbytes and capacity, then FF::read per element.
 * *n_host receives the element count. */
int scl_hip_wire_unpack(int field, uint64_t* dst_dev, size_t capacity, const unsigned char* src_dev,
                        size_t nbytes, size_t* n_host, void* stream);

/* seri::Serializer<Matrix<FF>> (include/scl/math/matrix.h:910-963): u32 rows, u32 cols, then the vector image
 * (u32 count = rows*cols, elements row-major).  src/dst are row-major device matrices with a pitch of ld >= cols
 * elements.  unpack reads the 12-byte header synchronously; an image whose count differs from rows*cols is
 * refused (the reference does not check, matrix.h:420). */
size_t scl_hip_wire_size_matrix(int field, size_t rows, size_t cols); /* 12 + rows * cols * byteSize */
int scl_hip_wire_pack_matrix(int field, unsigned char* dst_dev, const uint64_t* src_dev, size_t ld, size_t rows,
                             size_t cols, void* stream);
int scl_hip_wire_unpack_matrix(int field, uint64_t* dst_dev, size_t ld, size_t capacity_rows,
                               const unsigned char* src_dev, size_t nbytes, size_t* rows_host, size_t* cols_host,
                               void* stream);

/* What TcpChannel::send writes for a Packet holding one Vector / Matrix (include/scl/net/tcp_channel.h:125-160,
 * include/scl/net/packet.h:65-313): u32 packet size, then the packet bytes = the wire image above.
 * scl_hip_frame_size(image_bytes) = 4 + image_bytes.  frame_unpack reads a Vector frame the way TcpChannel::recv +
 * `packet.read<Vector>()` would: it looks at exactly packet-size bytes.  (A Matrix frame is its 4-byte size in
 * front of scl_hip_wire_unpack_matrix's input.) */
size_t scl_hip_frame_size(size_t image_bytes);
int scl_hip_frame_pack(int field, unsigned char* dst_dev, const uint64_t* src_dev, size_t n, void* stream);
int scl_hip_frame_pack_matrix(int field, unsigned char* dst_dev, const uint64_t* src_dev, size_t ld, size_t rows,
                              size_t cols, void* stream);
int scl_hip_frame_unpack(int field, uint64_t* dst_dev, size_t capacity, const unsigned char* src_dev, size_t nbytes,
                         size_t* n_host, void* stream);

/* ---- roofline probe -------------------------------------------------------- */
/* plain device copy kernel (16 B per lane) used to measure achievable HBM bandwidth */
int scl_hip_stream_copy(void* dst_dev, const void* src_dev, size_t bytes, void* stream);

/* ---- the open step: every party sends its shares to every party, then reconstructs ----------------------------
 * Replaces Network::send to each party + Network::recv from each party (include/scl/net/network.h:148-152,178-185; the
 * pattern of test/scl/protocol/beaver.h:43-55) followed by shamirRecoverP per secret (shamir.h:81-104), for a whole
 * batch: the n parties are dealt to the ranks of an RCCL communicator in contiguous blocks of ceil(n / world), one
 * all-gather per chunk of secrets over xGMI brings the share rows together, the reconstruct kernel of chunk k runs
 * beside the gather of chunk k + 1.  RCCL is opened with dlopen at the first of these calls (librccl.so.1; an instance
 * the process already holds is reused), so the library has no link-time dependency on it.
 *
 * A communicator handle wraps an ncclComm_t with a stream for the collectives, events and two gather buffers:
 *   scl_hip_comm_unique_id   ncclGetUniqueId on one rank; the caller distributes the 128 bytes (MPI, a file, gloo ..)
 *   scl_hip_comm_init_rank   ncclCommInitRank on every rank (the current HIP device is the rank's device)
 *   scl_hip_comm_adopt       wrap an ncclComm_t the caller created itself (not destroyed by scl_hip_comm_destroy)
 *   scl_hip_comm_info / scl_hip_comm_destroy
 * scl_hip_open_row_order (host only): row q = j * world + r of a gathered chunk holds party r * per + j (each of a
 * rank's `per` rows is all-gathered on its own: no packing copy); order[q] = that party or -1 for padding.
 *
 * One host thread at a time per handle (different handles -- the ranks of an in-process world -- run concurrently); calls on
 * one handle may come from different streams: a call waits for the previous call's readers of the handle's buffers.  The
 * padding rows of a slab (party >= n) are never read: zeros are sent in their place.  The calling thread's current device
 * must be the one the handle was made on (SCL_ERR_BAD_ARG otherwise).  SCL_HIP_RCCL_LIBRARY (environment, read at the first
 * of these calls) names the library to bind instead of librccl.so.1. */
int scl_hip_comm_unique_id(unsigned char id[128]);
int scl_hip_comm_init_rank(void** comm, int world, int rank, const unsigned char id[128]);
int scl_hip_comm_adopt(void** comm, void* nccl_comm);
int scl_hip_comm_info(void* comm, int* world, int* rank);
int scl_hip_comm_destroy(void* comm);
int scl_hip_open_row_order(size_t n, size_t world, long* order);
/* out_dev[s] = sum_p lambda[p] * share_p[s] for all N secrets on every rank.  local_dev: this rank's
 * [ceil(n / world)][stride] share rows (device; rows past the rank's party count are padding); lambda_host: the n
 * coefficients in party order (scl_hip_lagrange_basis).  chunk = secrets per all-gather (0: 2^24).  Asynchronous on
 * `stream` like every batch call. */
int scl_hip_open_all_gather(void* comm, int field, uint64_t* out_dev, const uint64_t* local_dev, size_t stride, size_t n,
                            const uint64_t* lambda_host, size_t N, size_t chunk, void* stream);
/* The same result from 1 / ceil(n / world) of the xGMI volume: each rank first reduces its own parties to one partial
 * sum per secret, the ranks all-gather the partials and add them (Vector::sum per secret, vector.h:261-267).
 * local_dev: the rank's OWN [parties_mine][stride] rows, lambda_local_host: their coefficients; parties_mine may be 0. */
int scl_hip_open_partial_gather(void* comm, int field, uint64_t* out_dev, const uint64_t* local_dev, size_t stride,
                                size_t parties_mine, const uint64_t* lambda_local_host, size_t N, size_t chunk,
                                void* stream);

/* Mersenne61 only: each rank's canonical partial sums go through ncclReduceScatter(ncclSum, ncclUint64)
 * -- at most 8 ranks, so that the 64-bit sum of values <= 2^61 - 2 cannot wrap -- rank r folds the r-th slice of every chunk
 * mod p, and (all_ranks != 0) an all-gather of the folded slices gives every rank every secret, bit-identical to
 * scl_hip_open_all_gather.  all_ranks == 0 writes out_dev[s] only on the rank that owns s: within chunk c (of the even chunk
 * size in force, cnt_c secrets) the slice index (s mod chunk) / ceil(cnt_c / world).  Arguments as scl_hip_open_partial_gather. */
int scl_hip_open_reduce_scatter(void* comm, int field, uint64_t* out_dev, const uint64_t* local_dev, size_t stride,
                                size_t parties_mine, const uint64_t* lambda_local_host, size_t N, size_t chunk, int all_ranks,
                                void* stream);

/* MONT128: choose the modulus (odd, < 2^128).  The call sets it for the CALLING host thread and as the process-wide
 * default: a thread that has called this keeps its own modulus whatever other threads choose later; a thread that never
 * did (a pool worker started after the main thread chose the prime) LATCHES the default at its first Mont128 call -- the
 * modulus set last by any thread before that, 2^128 - 159 if none was -- and keeps it: a worker between a share and its
 * recover does not change field because another thread picked a different prime.  It does not go on silently either: once
 * the default has changed after a thread latched it, that thread's next Mont128 call returns SCL_ERR_BAD_ARG (its buffers may
 * hold residues of either modulus) until the thread calls scl_hip_mont128_set_prime (a modulus of its own) or
 * scl_hip_mont128_relatch (take the current default).  scl_hip_mont128_get_prime reports the calling thread's modulus
 * (latching it if need be) and never fails. */
int scl_hip_mont128_set_prime(const uint64_t p[2]);
int scl_hip_mont128_get_prime(uint64_t p[2]);
int scl_hip_mont128_relatch(void);

#ifdef __cplusplus
}
#endif
#endif /* SCL_HIP_H */
