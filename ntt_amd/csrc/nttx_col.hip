// Column-pass launches of the 32-bit-lane path: every col_kernel instantiation lives in this translation
// unit so that it compiles in parallel with the row kernels and the C ABI (nttx.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#include "../../include/nttx.h"
#include "nttx_arith.hpp"
#include "nttx_kernels.hpp"
#include "nttx_launch.hpp"

namespace nttx {

static thread_local const char *g_msg = nullptr;
static int colfail(int code, const char *msg) { g_msg = msg; return code; }
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
static int grid_for(long long tiles, int cap) { return (int)std::max<long long>(1, std::min<long long>(tiles, cap)); }
constexpr int kMinColLog = 6;

// Column tile shape: 32 columns (full 128-byte lines) wherever the slab has that many, 16 columns for slabs of 16.
// S64: the input is the caller's u64 coefficients, reduced modulo the pass's prime as the tile is read (ColArgs::in64)
template <bool INV, int H, bool ZHI, bool FS2, bool S64 = false>
static int launch_col_t(int logT, const ColArgs &A, int nvec, hipStream_t st) {
  static_assert(!S64 || (!INV && ZHI && !FS2), "u64 source: plain zero-half forward passes");
  const int wide = A.logS >= 5 ? 1 : 0;
  // Few tiles (one product, or a handful): full-line tiles would leave most of the 256 CUs idle while each of the busy
  // ones works through its whole tile, so passes of 2^6 .. 2^9 points switch to tiles of 8 or 4 columns with one radix-8
  // unit per thread and round, until about 256 workgroups exist (one 2^18-point product: 7.8 -> about 2.5 us per pass).
  // NTTX_COL_NARROW=0 keeps full-line tiles (A/B measurements), 2 or 3 forces a width.
  int lc = wide ? 5 : 4;
  if (wide && logT >= 6 && !FS2 && !A.psi && !A.piece_stride) {
    static const int narrow = env_int("NTTX_COL_NARROW", -1);
    const int lcmin = logT == 6 ? 3 : 2;       // (a tile of 2^6 rows x 4 columns would be half a wavefront of units)
    long long w = (A.nblocks << (A.logS - 5)) * nvec;
    if (narrow == 2 || narrow == 3) lc = std::max(narrow, lcmin);
    else if (narrow != 0) while (lc > lcmin && w < 192) { --lc; w <<= 1; }
    if (lc == 4) lc = 3;                       // (the 16-column shape keeps four units per thread: no gain there)
  }
  const long long tiles = A.nblocks << (A.logS - lc);
  // workgroups per launch (tuning knobs; any value is correct, workgroups stride over the tiles)
  // Measured on MI355X: one tile per workgroup (the hardware dispatcher refills CUs as workgroups retire,
  // which staggers their load / butterfly / store phases) beats a resident grid striding over the tiles
  // by 12 % on the forward and 8 % on the inverse pass.
  static const int capf = env_int("NTTX_GRID_COL", 1 << 20), capi = env_int("NTTX_GRID_COLI", 1 << 20);
  const int cap = std::max(1, (INV ? capi : capf) / nvec);
  const int g16 = (grid_for(tiles, cap) + 15) / 16 * 16;
  const int grid = g16 <= tiles ? g16 : grid_for(tiles, cap);
#define NTTX_COL_CASE(K, LC, NT) \
    hipLaunchKernelGGL((col_kernel<K, INV, H, LC, NT, ZHI, FS2, false, 0, S64>), dim3(grid, 1, nvec), dim3(NT), 0, st, A); break;
  if constexpr (!FS2 && !ZHI) {
    if (A.psi) {   // outermost pass of a negacyclic product (full 128-byte tiles only: the caller checks)
#define NTTX_COLT_CASE(K, NT) \
      hipLaunchKernelGGL((col_kernel<K, INV, H, 5, NT, false, false, true>), dim3(grid, 1, nvec), dim3(NT), 0, st, A); break;
      switch (wide ? logT : 0) {
        case 6: NTTX_COLT_CASE(6, 256)  case 7: NTTX_COLT_CASE(7, 256)  case 8: NTTX_COLT_CASE(8, 256)  case 9: NTTX_COLT_CASE(9, 1024)
        default: return colfail(NTTX_ERANGE, "twisted column pass of this size unsupported");
      }
#undef NTTX_COLT_CASE
      if (hipGetLastError() != hipSuccess) return colfail(NTTX_EDEVICE, "column kernel launch failed");
      return 0;
    }
  }
  if constexpr (FS2) {
    switch (logT * 2 + wide) {
      case 15: NTTX_COL_CASE(7, 5, 256)   case 17: NTTX_COL_CASE(8, 5, 256)
      case 19: NTTX_COL_CASE(9, 5, 512)   // 512 threads: the 1024-thread shape has no registers left for the second factor
      default: return colfail(NTTX_ERANGE, "split-table column pass of this size unsupported");
    }
    if (hipGetLastError() != hipSuccess) return colfail(NTTX_EDEVICE, "column kernel launch failed");
    return 0;
  } else {
  if (A.piece_stride) {   // rows in the pieces layout of an all-to-all (full-line tiles of 2^6 .. 2^8-point passes)
    if constexpr (!FS2 && !ZHI) {
#define NTTX_COLP_CASE(K) \
      hipLaunchKernelGGL((col_kernel<K, INV, H, 5, 256, false, false, false, (INV ? 2 : 1)>), dim3(grid, 1, nvec), dim3(256), 0, st, A); break;
      switch ((wide && !A.psi) ? logT : 0) {
        case 6: NTTX_COLP_CASE(6)  case 7: NTTX_COLP_CASE(7)  case 8: NTTX_COLP_CASE(8)
        case 9:   // (inverse only: the sliced residues of 2^18-point CRT products)
          if constexpr (INV) { hipLaunchKernelGGL((col_kernel<9, true, H, 5, 1024, false, false, false, 2>), dim3(grid, 1, nvec), dim3(1024), 0, st, A); break; }
          return colfail(NTTX_ERANGE, "column pass of this size has no pieces-layout variant");
        default: return colfail(NTTX_ERANGE, "column pass of this size has no pieces-layout variant");
      }
#undef NTTX_COLP_CASE
      if (hipGetLastError() != hipSuccess) return colfail(NTTX_EDEVICE, "column kernel launch failed");
      return 0;
    } else {
      return colfail(NTTX_ERANGE, "pieces layout needs a plain column pass");
    }
  }
  // 2^6-point passes (products of 2^15 points) run on tiles of 64 columns (256-byte row segments) instead of 32: these
  // passes run at the rate of their access pattern, and a bare copy of the wider shape moves 10 % more (forward pass
  // 0.197 -> 0.170 ms, inverse 0.114 -> 0.103 ms per 2^26 points, the step 4.6 % faster).  At 2^7 and 2^8 points the
  // same change makes the column kernels 1 - 7 % faster and the step no faster (the row kernel between them then runs
  // that much slower: the part holds a power budget, not a clock).  NTTX_COL_SEG256=0 keeps 128-byte segments.
  if constexpr (!FS2) {
    static const int seg256 = env_int("NTTX_COL_SEG256", 1);
    if (seg256 && wide && logT == 6 && A.logS >= 6 && !A.psi && tiles >= 1024) {
      const long long t64 = A.nblocks << (A.logS - 6);
      hipLaunchKernelGGL((col_kernel<6, INV, H, 6, 256, ZHI, false, false, 0, S64>), dim3(grid_for(t64, cap), 1, nvec), dim3(256), 0, st, A);
      if (hipGetLastError() != hipSuccess) return colfail(NTTX_EDEVICE, "column kernel launch failed");
      return 0;
    }
  }
  if (lc < 4) {
    if constexpr (!FS2) {
#define NTTX_COLN_CASE(K, LC, NT) \
      hipLaunchKernelGGL((col_kernel<K, INV, H, LC, NT, ZHI, false, false, 0, S64>), dim3(grid, 1, nvec), dim3(NT), 0, st, A); break;
      switch (logT * 4 + lc) {                   // one radix-8 unit per thread: 2^(logT + lc - 3) threads
        case 6 * 4 + 3: NTTX_COLN_CASE(6, 3, 64)
        case 7 * 4 + 3: NTTX_COLN_CASE(7, 3, 128)  case 7 * 4 + 2: NTTX_COLN_CASE(7, 2, 64)
        case 8 * 4 + 3: NTTX_COLN_CASE(8, 3, 256)  case 8 * 4 + 2: NTTX_COLN_CASE(8, 2, 128)
        case 9 * 4 + 3: NTTX_COLN_CASE(9, 3, 512)  case 9 * 4 + 2: NTTX_COLN_CASE(9, 2, 256)
        default: return colfail(NTTX_ERANGE, "column pass of this size has no narrow-tile variant");
      }
#undef NTTX_COLN_CASE
    }
    if (hipGetLastError() != hipSuccess) return colfail(NTTX_EDEVICE, "column kernel launch failed");
    return 0;
  }
  if constexpr (S64) {   // (the plans that take this route have at least 2^9 columns: full-line tiles only)
    switch (wide ? logT : 0) {
      case 6: NTTX_COL_CASE(6, 5, 256)  case 7: NTTX_COL_CASE(7, 5, 256)  case 8: NTTX_COL_CASE(8, 5, 256)  case 9: NTTX_COL_CASE(9, 5, 1024)
      default: return colfail(NTTX_ERANGE, "column pass of this size has no u64-source variant");
    }
    if (hipGetLastError() != hipSuccess) return colfail(NTTX_EDEVICE, "column kernel launch failed");
    return 0;
  } else {
  switch (logT * 2 + wide) {
    case 12: NTTX_COL_CASE(6, 4, 256)   case 13: NTTX_COL_CASE(6, 5, 256)
    case 14: NTTX_COL_CASE(7, 4, 256)   case 15: NTTX_COL_CASE(7, 5, 256)
    case 16: NTTX_COL_CASE(8, 4, 256)   case 17: NTTX_COL_CASE(8, 5, 256)
    case 18: NTTX_COL_CASE(9, 4, 256)   case 19: NTTX_COL_CASE(9, 5, 1024)
    default: return colfail(NTTX_ERANGE, "column pass of this size unsupported");
  }
  if (hipGetLastError() != hipSuccess) return colfail(NTTX_EDEVICE, "column kernel launch failed");
  return 0;
  }   // !S64
  }   // !FS2
#undef NTTX_COL_CASE
}
template <bool INV, bool ZHI>
static int launch_col_i(int h, int logT, const ColArgs &A, int nvec, hipStream_t st) {
  if (A.fsa)
    return h == 9 ? launch_col_t<INV, 9, ZHI, true>(logT, A, nvec, st) : h == 4 ? launch_col_t<INV, 4, ZHI, true>(logT, A, nvec, st)
                                                                                : launch_col_t<INV, 2, ZHI, true>(logT, A, nvec, st);
  return h == 9 ? launch_col_t<INV, 9, ZHI, false>(logT, A, nvec, st) : h == 4 ? launch_col_t<INV, 4, ZHI, false>(logT, A, nvec, st)
                                                                               : launch_col_t<INV, 2, ZHI, false>(logT, A, nvec, st);
}
// A.nvalid > 0 selects the zero-padded-half forward variant (needs nvalid <= N/2)
static int launch_col_sel(bool inv, int h, int logT, const ColArgs &A, int nvec, hipStream_t st) {
  if (logT < kMinColLog || A.logS < 4) return colfail(NTTX_ERANGE, "column pass shape unsupported");
  if (A.in64[0]) {
    if (inv || A.nvalid <= 0 || A.fsa || A.psi || A.piece_stride || A.logS < 5 || logT > 9)
      return colfail(NTTX_ERANGE, "u64 source needs a plain zero-half forward pass of 2^6 .. 2^9 points");
    return h == 9 ? launch_col_t<false, 9, true, false, true>(logT, A, nvec, st) : h == 4 ? launch_col_t<false, 4, true, false, true>(logT, A, nvec, st)
                                                                                   : launch_col_t<false, 2, true, false, true>(logT, A, nvec, st);
  }
  if (inv) return launch_col_i<true, false>(h, logT, A, nvec, st);
  if (A.nvalid > 0) return launch_col_i<false, true>(h, logT, A, nvec, st);
  return launch_col_i<false, false>(h, logT, A, nvec, st);
}

int launch_col_kernels(bool inv, int h, int logT, const ColArgs &A, int nvec, hipStream_t st, const char **msg) {
  g_msg = nullptr;
  const int rc = launch_col_sel(inv, h, logT, A, nvec, st);
  if (msg) *msg = g_msg;
  return rc;
}

}  // namespace nttx
