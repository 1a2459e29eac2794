// include/scl_hip/math/fields/ff_ops.h -- the field plug-in boundary, as the reference has it.
//
// A finite field is a traits struct { using ValueType; NAME; BYTE_SIZE; BIT_SIZE; } plus explicit specialisations
// of the eleven function templates below on that struct (reference: include/scl/math/fields/ff_ops.h:35-118; a
// complete worked example is test/scl/gf7.h:26-31 + test/scl/gf7.cc:26-103).  scl::math::FF<FIELD> (../ff.h)
// is written against these functions only, so a field defined the reference's way compiles here unchanged.
//
// Where such a field runs: a traits struct with nothing more is a HOST field -- Vector, Matrix, Polynomial,
// computeLagrangeBasis and scl::ss::* take their generic per-element paths for it, exactly the reference's
// algorithms.  The fields that have kernels (Mersenne61, Mersenne127, Secp256k1Scalar and the two plug-ins Mont128,
// GF2_128, all in ../ff.h) additionally carry `TAG` (the C ABI's scl_field) and `Impl` (the per-lane arithmetic of
// ../../detail/field.hpp, shared with the HIP kernels); the batch-shaped members go to the GPU for them
// (`if constexpr (requires { FIELD::TAG; })`).
//
// All functions work in place on their first argument, are synchronous and re-entrant, and keep no state.  Errors
// are C++ exceptions: invert(0) throws std::logic_error("0 not invertible modulo prime") (test_ff.cc:168-171).
#ifndef SCL_HIP_MATH_FIELDS_FF_OPS_H
#define SCL_HIP_MATH_FIELDS_FF_OPS_H

#include <string>

namespace scl::math::ff {

/// out = the field element for the integer `value` (negative values wrap to p - |value|)
template <typename FIELD>
void convertTo(typename FIELD::ValueType& out, int value);

/// out = the field element a hex string spells
template <typename FIELD>
void convertTo(typename FIELD::ValueType& out, const std::string& src);

template <typename FIELD>
void add(typename FIELD::ValueType& out, const typename FIELD::ValueType& op);

template <typename FIELD>
void subtract(typename FIELD::ValueType& out, const typename FIELD::ValueType& op);

template <typename FIELD>
void multiply(typename FIELD::ValueType& out, const typename FIELD::ValueType& op);

template <typename FIELD>
void negate(typename FIELD::ValueType& out);

/// throws std::logic_error("0 not invertible modulo prime") for zero
template <typename FIELD>
void invert(typename FIELD::ValueType& out);

template <typename FIELD>
bool equal(const typename FIELD::ValueType& in1, const typename FIELD::ValueType& in2);

/// writes FIELD::BYTE_SIZE bytes
template <typename FIELD>
void toBytes(unsigned char* dest, const typename FIELD::ValueType& src);

/// reads FIELD::BYTE_SIZE bytes and reduces them into the field
template <typename FIELD>
void fromBytes(typename FIELD::ValueType& dest, const unsigned char* src);

template <typename FIELD>
std::string toString(const typename FIELD::ValueType& in);

}  // namespace scl::math::ff

#endif
