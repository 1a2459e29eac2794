// include/scl_hip/scl.h -- umbrella header: the hot-path surface of <scl/scl.h> on the MI355X engine.
#ifndef SCL_HIP_SCL_H
#define SCL_HIP_SCL_H

#include "hip/device.h"
#include "hip/elementwise.h"
#include "hip/open.h"
#include "math/fields/ff_ops.h"
#include "math/ff.h"
#include "math/array.h"
#include "math/lagrange.h"
#include "math/matrix.h"
#include "math/poly.h"
#include "math/vector.h"
#include "math/z2k.h"
#include "serialization/serializer.h"
#include "ss/additive.h"
#include "ss/shamir.h"
#include "util/prg.h"
#include "names.h"

#endif
