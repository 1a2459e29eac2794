// csrc/tu_config.hpp -- which part of the library a translation unit builds.
//
// libscl_hip.so is capi.hip compiled once per field family, in parallel (Makefile): a unit instantiates the kernels of the
// fields in its SCL_TU_FIELDS mask only, its field-taking entry points carry the unit's suffix (capi_names.inc) and
// capi_route.cc forwards each public entry point to the unit that holds the field.  Bits: 0 Mersenne61, 1 Mersenne127,
// 2 Mont128, 3 GF(2^128), 4 secp256k1 scalars, 5 secp256k1 field, 6 the rings Z2k, 7 = the common unit (entry points without
// a field, the shared state, the open step).  Default: everything in one unit (what the tools under tools/ build).
#pragma once
#ifndef SCL_TU_FIELDS
#define SCL_TU_FIELDS 0xff
#endif
#define SCL_TU_HAS(bit) (((SCL_TU_FIELDS) >> (bit)) & 1)
#define SCL_TU_COMMON SCL_TU_HAS(7)
