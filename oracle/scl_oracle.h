/* oracle/scl_oracle.h -- TEST INFRASTRUCTURE, not product code.
 *
 * Plain-C CPU restatement of the reference's finite-field / secret-sharing hot
 * path.  It is the checker the HIP path is compared
 * against; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.  Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 *
 * Parity status: PINNED for Mersenne61 / Mersenne127 / secp256k1_order / PRG / Shamir / additive
 * / Lagrange / Vandermonde / matmul -- tests/test_oracle_golden.py checks every
 * function here against golden vectors emitted by the real reference
 * (oracle/_ref, tests/golden/make_golden.py) and, where /root/reference is
 * present, against the live reference library.
 * PARITY UNPINNED for field tags SCLO_MONT128 and SCLO_GF2_128: the reference
 * has no such fields; they are checked against Python
 * big-integer arithmetic and algebraic identities only.
 *
 * Encoding: an element is 1 (M61), 2 (M127, MONT128, GF2_128) or 4 (SECP256K1_SCALAR)
 * little-endian uint64 limbs -- the in-memory image of FF::m_value: the canonical integer
 * for the Mersenne fields, the Montgomery residue for MONT128 / SECP256K1_SCALAR (as the
 * reference keeps its 256-bit fields, secp256k1_scalar.cc:47-135).
 * Values are canonical (in [0,p)) on entry and exit.
 * Share matrices on this face are AoS [secret][party] like the reference's
 * per-secret Vector (include/scl/ss/shamir.h:52-68).
 */
#ifndef SCL_ORACLE_H
#define SCL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SCLO_M61 = 0, SCLO_M127 = 1, SCLO_MONT128 = 2, SCLO_GF2_128 = 3, SCLO_SECP256K1_SCALAR = 4,
       SCLO_SECP256K1_FIELD = 5 };
/* rings Z2k<K> (include/scl/math/z2k.h): element-wise ops, from_bytes, vector_random, additive sharing, sum, dot,
 * scalar_mul and matmul take these tags; one limb for K <= 64, two above */
#define SCLO_Z2K(K) (0x100 + (K))
enum { SCLO_ADD = 0, SCLO_SUB = 1, SCLO_MUL = 2, SCLO_NEG = 3, SCLO_INV = 4, SCLO_DIV = 5 };
/* status codes */
enum { SCLO_OK = 0, SCLO_ZERO_INVERSE = 1, SCLO_BAD_ARG = 2, SCLO_BAD_HEX_LEN = 3,
       SCLO_BAD_HEX_CHAR = 4, SCLO_ERROR_DETECTED = 5, SCLO_NOT_INVERTIBLE_2K = 6 };

int sclo_limbs(int field);
const char* sclo_field_name(int field);
/* message the reference would put in its exception for a status code */
const char* sclo_status_message(int status);

/* element-wise over n elements; b may be NULL for unary ops */
int sclo_ew(int field, int op, uint64_t* dst, const uint64_t* a, const uint64_t* b, size_t n);
int sclo_from_int(int field, int v, uint64_t* dst);
int sclo_from_bytes(int field, const unsigned char* src, size_t n, uint64_t* dst);
int sclo_from_hex(int field, const char* hex, uint64_t* dst);
int sclo_to_hex(int field, const uint64_t* a, char* out, size_t outlen);
int sclo_exp(int field, const uint64_t* base, size_t e, uint64_t* dst);

/* PRG: AES-128-CTR as src/scl/util/prg.cc.  use_aesni: 0 portable C, 1 AES-NI
 * (if the CPU has it), -1 auto. */
void sclo_aes_force(int use_aesni);
int sclo_prg(const unsigned char* seed, size_t seed_len, const size_t* sizes, size_t ncalls,
             unsigned char* out);
/* raw counter-addressed blocks: out[16*i..] = AES_key(LE64(counter0+i) || LE64(NONCE)) */
int sclo_prg_blocks(const unsigned char* seed, size_t seed_len, uint64_t counter0,
                    size_t nblocks, unsigned char* out);

int sclo_vector_random(int field, const unsigned char* seed, size_t seed_len, size_t n,
                       uint64_t* out);

/* faithful per-secret calls driven by ONE PRG in secret order */
int sclo_shamir_share(int field, const unsigned char* seed, size_t seed_len,
                      const uint64_t* secrets, size_t N, size_t t, size_t n, uint64_t* shares);
/* same polynomial evaluation but with explicit coefficients coeffs[N][t] (c_1..c_t) */
/* shamirSecretShare over math::Array<FF, W> (pedersen.h:138 uses W = 2): secrets [N][W], shares [N][n][W] */
int sclo_shamir_share_packed(int field, const unsigned char* seed, size_t seed_len, const uint64_t* secrets, size_t N,
                             size_t t, size_t n, size_t W, uint64_t* shares);
int sclo_shamir_share_coeffs(int field, const uint64_t* secrets, const uint64_t* coeffs, size_t N,
                             size_t t, size_t n, uint64_t* shares);
/* faithful: recompute the basis for alphas 1..n at x=0 on every secret */
int sclo_shamir_recover(int field, const uint64_t* shares, size_t n, size_t N, uint64_t* out);
/* hoisted basis: out[s] = sum_i lambda[i]*shares[s][i] */
int sclo_shamir_recover_lambda(int field, const uint64_t* shares, const uint64_t* lambda,
                               size_t n, size_t N, uint64_t* out);
int sclo_shamir_recover_at(int field, const uint64_t* shares, const uint64_t* alphas,
                           const uint64_t* x, size_t m, size_t N, uint64_t* out);
/* short overload shamirRecoverD(shares, t): status[s]=1 where the reference throws */
int sclo_shamir_recover_d(int field, const uint64_t* shares, size_t n, size_t t, size_t N,
                          uint64_t* out, unsigned char* status);
/* shamirRecoverC (Berlekamp-Welch, shamir.h:202-259) per secret: t = (count-1)/3, the first n = 3t+1 shares are
 * used.  f_out [N][n] = coefficients of the corrected polynomial (zero padded), e_out [N][t+1] = the monic error
 * locator (zero padded), nerr[N] its degree; status[s] = 1 where the reference throws "could not correct shares"
 * (outputs zeroed). */
int sclo_shamir_recover_c(int field, const uint64_t* shares, const uint64_t* alphas, size_t count, size_t N,
                          uint64_t* f_out, uint64_t* e_out, unsigned char* status, unsigned* nerr);
int sclo_lagrange_basis(int field, const uint64_t* nodes, size_t m, const uint64_t* x,
                        uint64_t* out);

int sclo_additive_share(int field, const unsigned char* seed, size_t seed_len,
                        const uint64_t* secrets, size_t N, size_t n, uint64_t* shares);
int sclo_additive_recover(int field, const uint64_t* shares, size_t n, size_t N, uint64_t* out);

int sclo_dot(int field, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);
int sclo_sum(int field, const uint64_t* a, size_t n, uint64_t* out);
int sclo_scalar_mul(int field, const uint64_t* a, const uint64_t* scalar, size_t n, uint64_t* out);
int sclo_poly_eval(int field, const uint64_t* coeffs, size_t ncoeff, const uint64_t* xs,
                   size_t nx, uint64_t* out);
int sclo_vandermonde(int field, size_t n, size_t m, const uint64_t* xs, uint64_t* out);
int sclo_matmul(int field, const uint64_t* A, const uint64_t* B, size_t n, size_t k, size_t m,
                uint64_t* C);

/* wire image of a vector: seri::Serializer<std::vector<T>> = u32 count (little-endian) followed by
 * count elements as FF::write emits them (include/scl/serialization/serializer.h:157-190, ff.h:355-391).
 * wire: returns bytes written (out may be NULL to size).  unwire: FF::read per element (reduces mod p). */
size_t sclo_wire_vector(int field, const uint64_t* elems, size_t n, unsigned char* out);
int sclo_unwire_vector(int field, const unsigned char* in, size_t nbytes, uint64_t* elems, size_t capacity,
                       size_t* n);
/* wire image of a matrix: seri::Serializer<math::Matrix<T>> (matrix.h:910-963) = u32 rows, u32 cols, then the
 * vector image of the rows*cols row-major values */
size_t sclo_wire_matrix(int field, const uint64_t* elems, size_t rows, size_t cols, unsigned char* out);
int sclo_unwire_matrix(int field, const unsigned char* in, size_t nbytes, uint64_t* elems, size_t capacity,
                       size_t* rows, size_t* cols);

/* MONT128 plugin field: choose the modulus (odd, 2^127 < p < 2^128 not required;
 * any odd p >= 3 below 2^128).  Default: p = 2^128 - 159.  Not thread safe. */
int sclo_mont128_set_prime(const uint64_t p[2]);
void sclo_mont128_get_prime(uint64_t p[2]);

/* The CPU baseline when oracle/_ref is unavailable ("kind":"port"): per secret
 * share then recover with the reference's algorithmic shape, timed inside. */
int sclo_time_shamir(int field, size_t N, size_t t, size_t n, const unsigned char* seed,
                     size_t seed_len, double* share_s, double* recover_s, uint64_t* mismatches,
                     uint64_t* checksum);

#ifdef __cplusplus
}
#endif
#endif
