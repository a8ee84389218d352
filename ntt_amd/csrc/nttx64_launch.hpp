// Internal interface between the translation units of the 64-bit-lane path.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "nttx_kernels64.hpp"

namespace nttx {

// One column pass (nttx64_col.hip); safe / plo1 select the reduction class and the REDC form of the modulus, fp the
// floating-point class (tables of (w, w/p) pairs).
// A.nvalid > 0 selects the zero-padded-half forward variant, A.psi the twisted one.  Returns 0 or an NTTX_E*
// code with *msg set to a static text.
int launch_col64_kernels(bool inv, bool safe, bool plo1, bool fp, int logT, const ColArgs64 &A, int nvec, hipStream_t st, const char **msg);

}  // namespace nttx
