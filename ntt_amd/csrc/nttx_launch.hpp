// Internal interface between the translation units of the 32-bit-lane path.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "nttx_kernels.hpp"

namespace nttx {

// One column pass (nttx_col.hip).  A.nvalid > 0 selects the zero-padded-half forward variant, A.fsa the
// split-table variant, A.psi the twisted one.  Returns 0 or an NTTX_E* code with *msg set to a static text.
int launch_col_kernels(bool inv, int h, int logT, const ColArgs &A, int nvec, hipStream_t st, const char **msg);

}  // namespace nttx
