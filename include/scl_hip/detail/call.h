// include/scl_hip/detail/call.h -- status code -> the exception the reference throws.
//
// The C ABI reports conditions as scl_status codes; this header turns them back into the C++
// exception types and messages a user of the reference sees:
//   "Vec sizes mismatch"                      std::invalid_argument   include/scl/math/vector.h:481-485
//   "0 not invertible modulo prime"           std::logic_error        src/scl/math/fields/small_ff.h:70
//   "error detected during recovery"          std::logic_error        include/scl/ss/shamir.h:135
//   "not enough shares provided to detect errors"  std::logic_error   include/scl/ss/shamir.h:123
//   "matmul: this->cols() != that->rows()"    std::invalid_argument   include/scl/math/matrix.h:480
//   "|xs| != number of rows"                  std::invalid_argument   include/scl/math/matrix.h:449
//   "invalid range"                           std::invalid_argument   include/scl/math/vector.h:493
#ifndef SCL_HIP_DETAIL_CALL_H
#define SCL_HIP_DETAIL_CALL_H

#include <stdexcept>
#include <string>

#include "../../scl_hip.h"

namespace scl::hip::detail {

[[noreturn]] inline void raise(int status) {
  const std::string ref = scl_hip_status_message(status);
  switch (status) {
    case SCL_ERR_ZERO_INVERSE:
    case SCL_ERR_ERROR_DETECTED:
    case SCL_ERR_NOT_ENOUGH_SHARES:
      throw std::logic_error(ref);
    case SCL_ERR_SIZE_MISMATCH:
    case SCL_ERR_MATMUL_DIMS:
    case SCL_ERR_VANDERMONDE_XS:
    case SCL_ERR_INVALID_RANGE:
    case SCL_ERR_NOT_INVERTIBLE_2K:
      throw std::invalid_argument(ref);
    default:
      throw std::runtime_error(ref + ": " + scl_hip_last_error());
  }
}

inline void check(int status) {
  if (status != SCL_OK) raise(status);
}

}  // namespace scl::hip::detail

#endif
