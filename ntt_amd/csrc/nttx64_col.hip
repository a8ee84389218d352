// Column-pass launches of the 64-bit-lane path: every col64_kernel instantiation lives in this translation
// unit so that it compiles in parallel with the row kernels (nttx64.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#include "../../include/nttx.h"
#include "nttx_arith.hpp"
#include "nttx_kernels64.hpp"
#include "nttx64_launch.hpp"

namespace nttx {

static thread_local const char *g_msg = nullptr;
static int colfail(int code, const char *msg) { g_msg = msg; return code; }
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
static int grid_for(long long tiles, int cap) { return (int)std::max<long long>(1, std::min<long long>(tiles, cap)); }

template <bool INV, bool SAFE, bool PLO1, bool ZHI>
static int launch_col64_t(int logT, const ColArgs64 &A, int nvec, hipStream_t st) {
  const long long tiles = A.nblocks << (A.logS - 4);
  static const int cap_col = env_int("NTTX_GRID_COL64", 1 << 20);   // one tile per workgroup, as on the 32-bit path
  const int grid = grid_for(tiles, std::max(1, cap_col / nvec));
#define NTTX_COL64_CASE(K) case K: \
    if constexpr (!ZHI) { if (A.psi) { hipLaunchKernelGGL((col64_kernel<K, INV, SAFE, PLO1, false, true>), dim3(grid, 1, nvec), dim3(512), 0, st, A); break; } } \
    hipLaunchKernelGGL((col64_kernel<K, INV, SAFE, PLO1, ZHI>), dim3(grid, 1, nvec), dim3(512), 0, st, A); break;
  switch (logT) {
    NTTX_COL64_CASE(6) NTTX_COL64_CASE(7) NTTX_COL64_CASE(8) NTTX_COL64_CASE(9)
    default: return colfail(NTTX_ERANGE, "column pass of this size unsupported");
  }
#undef NTTX_COL64_CASE
  if (hipGetLastError() != hipSuccess) return colfail(NTTX_EDEVICE, "column kernel launch failed");
  return 0;
}
template <bool INV, bool ZHI>
static int launch_col64_c(bool safe, bool plo1, int logT, const ColArgs64 &A, int nvec, hipStream_t st) {
  if (safe) return plo1 ? launch_col64_t<INV, true, true, ZHI>(logT, A, nvec, st) : launch_col64_t<INV, true, false, ZHI>(logT, A, nvec, st);
  return plo1 ? launch_col64_t<INV, false, true, ZHI>(logT, A, nvec, st) : launch_col64_t<INV, false, false, ZHI>(logT, A, nvec, st);
}

template <bool INV, bool ZHI>
static int launch_col64_fp(int logT, const ColArgs64 &A, int nvec, hipStream_t st) {
  const long long tiles = A.nblocks << (A.logS - 4);
  static const int cap_col = env_int("NTTX_GRID_COL64", 1 << 20);
  const int grid = grid_for(tiles, std::max(1, cap_col / nvec));
#define NTTX_COL64FP_CASE(K) case K: \
    if constexpr (!ZHI) { if (A.psi) { hipLaunchKernelGGL((col64_kernel<K, INV, false, false, false, true, true>), dim3(grid, 1, nvec), dim3(512), 0, st, A); break; } } \
    hipLaunchKernelGGL((col64_kernel<K, INV, false, false, ZHI, false, true>), dim3(grid, 1, nvec), dim3(512), 0, st, A); break;
  // (a plain forward pass without the zero half is the inner column pass of a three-pass plan on this class: 2^7 points
  // and more; 2^6-point passes exist with the zero half, twisted, or inverse)
  switch (logT) {
    case 6:
      if constexpr (!ZHI) { if (A.psi) { hipLaunchKernelGGL((col64_kernel<6, INV, false, false, false, true, true>), dim3(grid, 1, nvec), dim3(512), 0, st, A); break; } }
      if constexpr (INV || ZHI) { hipLaunchKernelGGL((col64_kernel<6, INV, false, false, ZHI, false, true>), dim3(grid, 1, nvec), dim3(512), 0, st, A); break; }
      else return colfail(NTTX_ERANGE, "column pass of this size unsupported");
    NTTX_COL64FP_CASE(7) NTTX_COL64FP_CASE(8) NTTX_COL64FP_CASE(9)
    default: return colfail(NTTX_ERANGE, "column pass of this size unsupported");
  }
#undef NTTX_COL64FP_CASE
  if (hipGetLastError() != hipSuccess) return colfail(NTTX_EDEVICE, "column kernel launch failed");
  return 0;
}

int launch_col64_kernels(bool inv, bool safe, bool plo1, bool fp, int logT, const ColArgs64 &A, int nvec, hipStream_t st, const char **msg) {
  g_msg = nullptr;
  if (fp) {
    const int rc = inv ? launch_col64_fp<true, false>(logT, A, nvec, st)
                       : (A.nvalid > 0 ? launch_col64_fp<false, true>(logT, A, nvec, st) : launch_col64_fp<false, false>(logT, A, nvec, st));
    if (msg) *msg = g_msg;
    return rc;
  }
  const int rc = inv ? launch_col64_c<true, false>(safe, plo1, logT, A, nvec, st)
                     : (A.nvalid > 0 ? launch_col64_c<false, true>(safe, plo1, logT, A, nvec, st)
                                     : launch_col64_c<false, false>(safe, plo1, logT, A, nvec, st));
  if (msg) *msg = g_msg;
  return rc;
}

}  // namespace nttx
