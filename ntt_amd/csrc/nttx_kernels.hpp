// gfx950 kernels for the NTT polynomial-multiplication path.
//
// Replaces the CPU hot loop of the reference: bit-reversal + log2(L) radix-2
// butterfly stages (ntt_iter_barrett main.cc:143-182, ntt_worker
// monttry2.cpp:259-280), the pointwise product and n^{-1} scaling
// (main.cc:196-206) and their orchestration (ntt_compute_omp main.cc:185-209).
//
// Structure (MI355X-first, not a translation):
//  * A length-L transform is factored L = T1*T2(*T3) ("four-step"): strided
//    column passes of length T1 (T2) followed by contiguous row passes, every
//    pass doing all of its log2(T) butterfly stages inside LDS.
//  * Forward passes are decimation-in-frequency (natural in, bit-reversed out),
//    inverse passes decimation-in-time (bit-reversed in, natural out), so the
//    fused multiply never runs a bit-reversal permutation.
//  * Inside a pass each thread owns a radix-8 (or 4/2) unit: 8 points in
//    registers, 3 stages, then one exchange through LDS.
//  * Residues are 32-bit, Montgomery radix 2^32; twiddles are precomputed in
//    Montgomery form. Between stages values are kept lazily in [0,2p) when
//    p < 2^30 and canonically in [0,p) otherwise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nttx_arith.hpp"

namespace nttx {

constexpr int WG = 256;          // threads per workgroup of the row kernels

// ------------------------------------------------------------------------------------
// modular butterflies
// ------------------------------------------------------------------------------------
template <bool LAZY>
struct Mod {
  uint32_t p, npinv, m;  // m = representation bound: 2p (lazy) or p (canonical)
  __device__ __forceinline__ Mod(const PrimeCtx &c) : p(c.p), npinv(c.npinv), m(LAZY ? 2 * c.p : c.p) {}
  __device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) const { return csub(a + b, m); }
  __device__ __forceinline__ uint32_t sub(uint32_t a, uint32_t b) const {
    uint32_t d = a - b, e = d + m;   // d wraps when a < b; then e = d + m is the small value
    return d < e ? d : e;
  }
  // x*w with w canonical Montgomery-form constant; x any value below 2^32
  __device__ __forceinline__ uint32_t mul(uint32_t x, uint32_t w) const {
    uint32_t t = mont_mul_lazy(x, w, p, npinv);
    return LAZY ? t : csub(t, p);
  }
  // decimation-in-frequency (Gentleman-Sande): (a, b) -> (a + b, (a - b) * w)
  __device__ __forceinline__ void dif(uint32_t &a, uint32_t &b, uint32_t w) const {
    uint32_t s = add(a, b);
    b = mul(a - b + m, w);
    a = s;
  }
  __device__ __forceinline__ void dif1(uint32_t &a, uint32_t &b) const {  // twiddle 1
    uint32_t s = add(a, b);
    b = sub(a, b);
    a = s;
  }
  // decimation-in-time (Cooley-Tukey): (a, b) -> (a + b*w, a - b*w)
  __device__ __forceinline__ void dit(uint32_t &a, uint32_t &b, uint32_t w) const {
    uint32_t v = mul(b, w);
    b = sub(a, v);
    a = add(a, v);
  }
  __device__ __forceinline__ void dit1(uint32_t &a, uint32_t &b) const {
    uint32_t v = b;
    b = sub(a, v);
    a = add(a, v);
  }
  __device__ __forceinline__ uint32_t canon(uint32_t x) const { return LAZY ? csub(x, p) : x; }
};


// ------------------------------------------------------------------------------------
// Radix-8 units on 8 registers: three butterfly stages at point distances 4D, 2D, D.
// w[m] = tw[lo + m*D] (m = 1..7); UNIT_D1 marks D == 1, where w[1] == 1 needs no product.
// ------------------------------------------------------------------------------------
template <bool LAZY, bool UNIT_D1>
__device__ __forceinline__ void r8_dif(uint32_t (&x)[8], const uint32_t (&w)[8], const Mod<LAZY> &M) {
#pragma unroll
  for (int q = 0; q < 4; ++q) M.dif(x[q], x[q + 4], w[4 + q]);
#pragma unroll
  for (int q = 0; q < 2; ++q) { M.dif(x[q], x[q + 2], w[2 + q]); M.dif(x[4 + q], x[6 + q], w[2 + q]); }
#pragma unroll
  for (int q = 0; q < 8; q += 2) { if (!UNIT_D1) M.dif(x[q], x[q + 1], w[1]); else M.dif1(x[q], x[q + 1]); }
}
template <bool LAZY, bool UNIT_D1>
__device__ __forceinline__ void r8_dit(uint32_t (&x)[8], const uint32_t (&w)[8], const Mod<LAZY> &M) {
#pragma unroll
  for (int q = 0; q < 8; q += 2) { if (!UNIT_D1) M.dit(x[q], x[q + 1], w[1]); else M.dit1(x[q], x[q + 1]); }
#pragma unroll
  for (int q = 0; q < 2; ++q) { M.dit(x[q], x[q + 2], w[2 + q]); M.dit(x[4 + q], x[6 + q], w[2 + q]); }
#pragma unroll
  for (int q = 0; q < 4; ++q) M.dit(x[q], x[q + 4], w[4 + q]);
}

// ------------------------------------------------------------------------------------
// LDS tile layouts.  A tile holds NT transforms of length T.
//  RowLayout: transform-contiguous, 8 pad words after every 64 (conflict-free strides 8, 64)
//  ColLayout: point-major [T][COLS], one pad row after every 8 rows
// Both paddings satisfy pad(base + q*D) == pad(base) + pad(q*D) for every unit
// (base = hi*2^R*D + lo, lo < D), so per-point offsets are compile-time constants.
// ------------------------------------------------------------------------------------
template <int LOGT_>
struct RowLayout {
  static constexpr int LOGT = LOGT_;
  static constexpr int T = 1 << LOGT;
  static constexpr int TP = T + (T >> 6) * 8;              // padded transform length (words)
  static constexpr int NT = (T >= 2048) ? 1 : (2048 >> LOGT);  // transforms per tile
  static constexpr int WORDS = NT * TP;
  static constexpr int NTHR = WG;
  __device__ static constexpr int padj(int j) { return j + ((j >> 6) << 3); }
  __device__ static __forceinline__ int addr(int c, int j) { return c * TP + padj(j); }
  // unit index g -> (transform c, unit u): consecutive lanes walk one transform
  template <int LOGU>
  __device__ static __forceinline__ void split(int g, int &c, int &u) { c = g >> LOGU; u = g & ((1 << LOGU) - 1); }
};

// LOGC = log2(contiguous columns per tile): 4 -> 64-byte row segments, 5 -> full 128-byte lines
template <int LOGT_, int LOGC_, int NTHR_>
struct ColLayout {
  static constexpr int LOGT = LOGT_, LOGC = LOGC_;
  static constexpr int T = 1 << LOGT;
  static constexpr int COLS = 1 << LOGC;
  static constexpr int NT = COLS;
  static constexpr int NTHR = NTHR_;
  static constexpr int ROWS = T + (T >> 3);
  static constexpr int WORDS = ROWS * COLS;
  __device__ static constexpr int padj(int j) { return (j + (j >> 3)) * COLS; }
  __device__ static __forceinline__ int addr(int c, int j) { return padj(j) + c; }
  // consecutive lanes walk the columns of one unit
  template <int LOGU>
  __device__ static __forceinline__ void split(int g, int &c, int &u) { c = g & (COLS - 1); u = g >> LOGC; }
};

// ------------------------------------------------------------------------------------
// One round: every radix-2^R unit at point distance D = 2^LOGD of every transform in
// the tile, for NV vectors that share twiddles (a and b of a product).
// tw is the stage-major table: tw[h + j] = g^(j*T/(2h)), j < h, Montgomery form.
// ------------------------------------------------------------------------------------
template <class L, int R, int LOGD, bool INV, bool LAZY, int NV>
__device__ __forceinline__ void ntt_round(uint32_t *const (&s)[NV], const uint32_t *tw, const Mod<LAZY> &M) {
  constexpr int D = 1 << LOGD;
  constexpr int LOG_UNITS = L::LOGT - R;  // log2(units per transform)
  constexpr int TOTAL = L::NT << LOG_UNITS;
  constexpr int NP = 1 << R;
  for (int g = threadIdx.x; g < TOTAL; g += L::NTHR) {
    int c, u;
    L::template split<LOG_UNITS>(g, c, u);
    const int lo = u & (D - 1), hi = u >> LOGD;
    const int base = (hi << (LOGD + R)) + lo;
    const int a0 = L::addr(c, base);
    uint32_t w[NP];
    if (D > 1) {
#pragma unroll
      for (int m = 1; m < NP; ++m) w[m] = tw[lo + m * D];
    } else {
#pragma unroll
      for (int m = 2; m < NP; ++m) w[m] = tw[m];     // tw[1] == 1: handled by dif1/dit1
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      uint32_t x[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q) x[q] = s[v][a0 + L::padj(q * D)];
      if constexpr (!INV) {
        if constexpr (R == 3) {
          r8_dif<LAZY, D == 1>(x, w, M);
        } else if constexpr (R == 2) {
          M.dif(x[0], x[2], w[2]); M.dif(x[1], x[3], w[3]);
          if (D > 1) { M.dif(x[0], x[1], w[1]); M.dif(x[2], x[3], w[1]); }
          else { M.dif1(x[0], x[1]); M.dif1(x[2], x[3]); }
        } else {
          if (D > 1) M.dif(x[0], x[1], w[1]); else M.dif1(x[0], x[1]);
        }
      } else {
        if constexpr (R == 3) {
          r8_dit<LAZY, D == 1>(x, w, M);
        } else if constexpr (R == 2) {
          if (D > 1) { M.dit(x[0], x[1], w[1]); M.dit(x[2], x[3], w[1]); }
          else { M.dit1(x[0], x[1]); M.dit1(x[2], x[3]); }
          M.dit(x[0], x[2], w[2]); M.dit(x[1], x[3], w[3]);
        } else {
          if (D > 1) M.dit(x[0], x[1], w[1]); else M.dit1(x[0], x[1]);
        }
      }
#pragma unroll
      for (int q = 0; q < NP; ++q) s[v][a0 + L::padj(q * D)] = x[q];
    }
  }
}

// Forward (DIF) rounds: largest distance first.  REM = log2 of the sub-transform still to do.
template <class L, int REM, bool LAZY, int NV>
__device__ __forceinline__ void dif_rounds(uint32_t *const (&s)[NV], const uint32_t *tw, const Mod<LAZY> &M) {
  if constexpr (REM >= 3) {
    ntt_round<L, 3, REM - 3, false, LAZY, NV>(s, tw, M);
    if constexpr (REM > 3) { __syncthreads(); dif_rounds<L, REM - 3, LAZY, NV>(s, tw, M); }
  } else if constexpr (REM == 2) {
    ntt_round<L, 2, 0, false, LAZY, NV>(s, tw, M);
  } else if constexpr (REM == 1) {
    ntt_round<L, 1, 0, false, LAZY, NV>(s, tw, M);
  }
}

// Inverse (DIT) rounds: the same partition walked from the smallest distance up.
template <class L, int REM, bool LAZY, int NV>
__device__ __forceinline__ void dit_rounds(uint32_t *const (&s)[NV], const uint32_t *tw, const Mod<LAZY> &M) {
  if constexpr (REM >= 3) {
    if constexpr (REM > 3) { dit_rounds<L, REM - 3, LAZY, NV>(s, tw, M); __syncthreads(); }
    ntt_round<L, 3, REM - 3, true, LAZY, NV>(s, tw, M);
  } else if constexpr (REM == 2) {
    ntt_round<L, 2, 0, true, LAZY, NV>(s, tw, M);
  } else if constexpr (REM == 1) {
    ntt_round<L, 1, 0, true, LAZY, NV>(s, tw, M);
  }
}

// ------------------------------------------------------------------------------------
// Row kernels: transforms are contiguous runs of T words.
// ------------------------------------------------------------------------------------
enum RowMode { ROW_FWD = 0, ROW_INV = 1, ROW_MUL = 2 };

struct RowArgs {
  const uint32_t *in0;   // a (or x)
  const uint32_t *in1;   // b (ROW_MUL only)
  uint32_t *out;
  const uint32_t *tw_fwd;  // stage-major table, T words (ROW_FWD, ROW_MUL)
  const uint32_t *tw_inv;  // inverse-root table (ROW_INV, ROW_MUL)
  PrimeCtx ctx;
  uint32_t scale;        // ROW_MUL: extra Montgomery factor applied to the pointwise product (0 = none)
  int canon_out;         // fully reduce outputs to [0,p)
  long long nrows;       // total transforms (batch * L / T)
};

template <class L>
__device__ __forceinline__ void row_load(uint32_t *s, const uint32_t *in, long long row0, long long nrows) {
  constexpr int T = L::T;
  if constexpr (T >= 4) {
    constexpr int V = L::NT * T / 4;
    for (int v = threadIdx.x; v < V; v += WG) {
      const int c = (v * 4) >> L::LOGT, j = (v * 4) & (T - 1);
      uint4 x = make_uint4(0, 0, 0, 0);
      if (row0 + c < nrows) x = *reinterpret_cast<const uint4 *>(in + (row0 + c) * T + j);
      *reinterpret_cast<uint4 *>(s + L::addr(c, j)) = x;
    }
  } else {
    for (int v = threadIdx.x; v < L::NT * T; v += WG) {
      const int c = v / T, j = v % T;
      s[L::addr(c, j)] = (row0 + c < nrows) ? in[(row0 + c) * T + j] : 0u;
    }
  }
}

template <class L, bool LAZY>
__device__ __forceinline__ void row_store(const uint32_t *s, uint32_t *out, long long row0, long long nrows,
                                          const Mod<LAZY> &M, bool canon) {
  constexpr int T = L::T;
  if constexpr (T >= 4) {
    constexpr int V = L::NT * T / 4;
    for (int v = threadIdx.x; v < V; v += WG) {
      const int c = (v * 4) >> L::LOGT, j = (v * 4) & (T - 1);
      if (row0 + c < nrows) {
        uint4 x = *reinterpret_cast<const uint4 *>(s + L::addr(c, j));
        if (canon) { x.x = M.canon(x.x); x.y = M.canon(x.y); x.z = M.canon(x.z); x.w = M.canon(x.w); }
        *reinterpret_cast<uint4 *>(out + (row0 + c) * T + j) = x;
      }
    }
  } else {
    for (int v = threadIdx.x; v < L::NT * T; v += WG) {
      const int c = v / T, j = v % T;
      if (row0 + c < nrows) { uint32_t x = s[L::addr(c, j)]; out[(row0 + c) * T + j] = canon ? M.canon(x) : x; }
    }
  }
}

template <int LOGT, int MODE, bool LAZY>
__global__ void __launch_bounds__(WG) row_kernel(RowArgs A) {
  using L = RowLayout<LOGT>;
  constexpr int T = L::T;
  constexpr int NVEC = (MODE == ROW_MUL) ? 2 : 1;
  __shared__ __attribute__((aligned(16))) uint32_t lds[NVEC * L::WORDS + (MODE == ROW_MUL ? 2 : 1) * T];
  uint32_t *sa = lds, *sb = lds + (NVEC - 1) * L::WORDS;
  uint32_t *twf = lds + NVEC * L::WORDS;
  uint32_t *twi = twf + (MODE == ROW_MUL ? T : 0);
  const Mod<LAZY> M(A.ctx);

  if (MODE != ROW_INV) for (int i = threadIdx.x; i < T; i += WG) twf[i] = A.tw_fwd[i];
  if (MODE != ROW_FWD) for (int i = threadIdx.x; i < T; i += WG) twi[i] = A.tw_inv[i];

  for (long long row0 = (long long)blockIdx.x * L::NT; row0 < A.nrows; row0 += (long long)gridDim.x * L::NT) {
    __syncthreads();  // previous tile fully stored before LDS is overwritten
    row_load<L>(sa, A.in0, row0, A.nrows);
    if (MODE == ROW_MUL) row_load<L>(sb, A.in1, row0, A.nrows);
    __syncthreads();
    if (MODE == ROW_FWD) {
      uint32_t *const v[1] = {sa};
      dif_rounds<L, LOGT, LAZY, 1>(v, twf, M);
    } else if (MODE == ROW_INV) {
      uint32_t *const v[1] = {sa};
      dit_rounds<L, LOGT, LAZY, 1>(v, twi, M);
    } else {
      uint32_t *const v2[2] = {sa, sb};
      dif_rounds<L, LOGT, LAZY, 2>(v2, twf, M);
      __syncthreads();
      // pointwise product (plus the optional n^{-1} R^2 factor), into sa
      for (int i = threadIdx.x; i < L::NT * T; i += WG) {
        const int c = i >> LOGT, j = i & (T - 1);
        const int ad = L::addr(c, j);
        uint32_t xa = sa[ad], xb = sb[ad];
        uint32_t pr = M.mul(xa, xb);  // operands < 2p < 2^31 (lazy) or < p: product < 2^32 p
        if (A.scale) pr = M.mul(pr, A.scale);
        sa[ad] = pr;
      }
      __syncthreads();
      uint32_t *const v[1] = {sa};
      dit_rounds<L, LOGT, LAZY, 1>(v, twi, M);
    }
    __syncthreads();
    row_store<L, LAZY>(sa, A.out, row0, A.nrows, M, A.canon_out != 0);
  }
}


// ------------------------------------------------------------------------------------
// Fused product on 512-point rows, one wavefront per row, no workgroup barrier.
// Lane l owns 8 points per vector; the three radix-8 rounds use the register layouts
//   R1: i = l + 64 q         R2: i = 64 (l>>3) + (l&7) + 8 j         R3: i = 8 l + j
// and move between them through a wave-private LDS buffer (DS operations of one wave
// execute in order, so a read issued after a write by other lanes sees it).  The last
// forward round, the pointwise product and the first inverse round all act on the same
// 8 contiguous points of a lane, so they stay in registers.  Twiddles of a lane do not
// depend on the row: they are loaded once and stay in registers across the row loop.
// ------------------------------------------------------------------------------------
constexpr int ROW9_PAD = 576;  // 512 + 8 pad words per 64

template <bool LAZY>
__global__ void __launch_bounds__(WG) row_mul9_kernel(RowArgs A) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[(WG / 64) * 2 * ROW9_PAD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *sa = lds + wave * 2 * ROW9_PAD, *sb = sa + ROW9_PAD;
  const Mod<LAZY> M(A.ctx);
  const int k8 = lane >> 3, m8 = lane & 7;
  const int o1 = lane;                       // R1: + 72 q
  const int o2 = 72 * k8 + m8;               // R2: + 8 j
  const int o3 = 8 * lane + 8 * k8;          // R3: + j (contiguous)

  uint32_t wf1[8], wf2[8], wf3[8], wi1[8], wi2[8], wi3[8];
#pragma unroll
  for (int m = 1; m < 8; ++m) {
    wf1[m] = A.tw_fwd[lane + 64 * m]; wf2[m] = A.tw_fwd[m8 + 8 * m]; wf3[m] = A.tw_fwd[m];
    wi1[m] = A.tw_inv[lane + 64 * m]; wi2[m] = A.tw_inv[m8 + 8 * m]; wi3[m] = A.tw_inv[m];
  }
  wf1[0] = wf2[0] = wf3[0] = wi1[0] = wi2[0] = wi3[0] = 0;

  const long long nwaves = (long long)gridDim.x * (WG / 64);
  long long row = (long long)blockIdx.x * (WG / 64) + wave;
  uint32_t na[8], nb[8];
  if (row < A.nrows) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { na[q] = A.in0[row * 512 + lane + 64 * q]; nb[q] = A.in1[row * 512 + lane + 64 * q]; }
  }
  for (; row < A.nrows; row += nwaves) {
    uint32_t a[8], b[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { a[q] = na[q]; b[q] = nb[q]; }
    const long long nrow = row + nwaves;
    if (nrow < A.nrows) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { na[q] = A.in0[nrow * 512 + lane + 64 * q]; nb[q] = A.in1[nrow * 512 + lane + 64 * q]; }
    }
    // forward round 1 (distances 256, 128, 64)
    r8_dif<LAZY, false>(a, wf1, M);
    r8_dif<LAZY, false>(b, wf1, M);
#pragma unroll
    for (int q = 0; q < 8; ++q) { sa[o1 + 72 * q] = a[q]; sb[o1 + 72 * q] = b[q]; }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = sa[o2 + 8 * j]; b[j] = sb[o2 + 8 * j]; }
    // forward round 2 (32, 16, 8)
    r8_dif<LAZY, false>(a, wf2, M);
    r8_dif<LAZY, false>(b, wf2, M);
#pragma unroll
    for (int j = 0; j < 8; ++j) { sa[o2 + 8 * j] = a[j]; sb[o2 + 8 * j] = b[j]; }
    __builtin_amdgcn_wave_barrier();
    {
      const uint4 a0 = *reinterpret_cast<const uint4 *>(sa + o3), a1 = *reinterpret_cast<const uint4 *>(sa + o3 + 4);
      const uint4 b0 = *reinterpret_cast<const uint4 *>(sb + o3), b1 = *reinterpret_cast<const uint4 *>(sb + o3 + 4);
      a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
      b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
    }
    // forward round 3 (4, 2, 1), pointwise product, inverse round 1 (1, 2, 4): all on a lane's own 8 points
    r8_dif<LAZY, true>(a, wf3, M);
    r8_dif<LAZY, true>(b, wf3, M);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t pr = M.mul(a[j], b[j]);
      if (A.scale) pr = M.mul(pr, A.scale);
      a[j] = pr;
    }
    r8_dit<LAZY, true>(a, wi3, M);
    *reinterpret_cast<uint4 *>(sa + o3) = make_uint4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<uint4 *>(sa + o3 + 4) = make_uint4(a[4], a[5], a[6], a[7]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = sa[o2 + 8 * j];
    // inverse round 2 (8, 16, 32)
    r8_dit<LAZY, false>(a, wi2, M);
#pragma unroll
    for (int j = 0; j < 8; ++j) sa[o2 + 8 * j] = a[j];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = sa[o1 + 72 * q];
    // inverse round 3 (64, 128, 256)
    r8_dit<LAZY, false>(a, wi1, M);
#pragma unroll
    for (int q = 0; q < 8; ++q) A.out[row * 512 + lane + 64 * q] = A.canon_out ? M.canon(a[q]) : a[q];
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------
// Column kernels: T points at stride S inside blocks of N = T*S words; a tile is
// T rows x COLS contiguous columns.  The forward pass multiplies point (j', s) by
// fs[j'*S + s] = w_N^(s * bitrev_T(j')) on the way out (four-step twiddle); the inverse
// pass multiplies by the inverse table on the way in.
// ------------------------------------------------------------------------------------
struct ColArgs {
  const uint32_t *in[2];
  uint32_t *out[2];
  const uint32_t *tw;     // stage-major table for T (forward or inverse root)
  const uint32_t *fs;     // four-step table, N words
  PrimeCtx ctx;
  int logS;               // S = 2^logS >= columns per tile
  int canon_out;
  long long nblocks;      // batch * L / N
#ifdef NTTX_DBG
  int dbg;                // ablation: 1 = skip butterflies, 2 = skip global loads, 3 = skip global stores
#endif
};

// Each thread moves PER = T*COLS/(4*NTHR) 16-byte pieces of a tile; LPR = COLS/4 lanes cover one
// row segment.  The next tile's pieces are fetched into registers before the butterflies of the
// current tile start, so HBM latency overlaps with compute inside one workgroup.
template <int LOGT, bool INV, bool LAZY, int LOGC, int NTHR>
__global__ void __launch_bounds__(NTHR) col_kernel(ColArgs A) {
  using L = ColLayout<LOGT, LOGC, NTHR>;
  constexpr int T = L::T, COLS = L::COLS;
  constexpr int LPR = COLS / 4, LOG_LPR = LOGC - 2;
  constexpr int PER = (T * LPR) / NTHR;
  constexpr int RSTEP = NTHR / LPR;              // rows covered by one piece index
  static_assert(PER >= 1 && PER * NTHR == T * LPR, "tile must be a whole number of pieces per thread");
  __shared__ __attribute__((aligned(16))) uint32_t lds[L::WORDS + T];
  uint32_t *s = lds, *tw = lds + L::WORDS;
  const Mod<LAZY> M(A.ctx);
  const uint32_t *in = A.in[blockIdx.z];
  uint32_t *out = A.out[blockIdx.z];
  const int logS = A.logS;
  const int log_tpb = logS - LOGC;               // log2(tiles per block)
  const long long ntiles = A.nblocks << log_tpb;
  const int j0 = threadIdx.x >> LOG_LPR, c4 = (threadIdx.x & (LPR - 1)) * 4;

  for (int i = threadIdx.x; i < T; i += NTHR) tw[i] = A.tw[i];

  auto tile_base = [&](long long t) -> long long {
    const long long blk = t >> log_tpb;
    const long long ct = t & ((1ll << log_tpb) - 1);
    return (blk << (logS + LOGT)) + (ct << LOGC);
  };
  auto fs_base = [&](long long t) -> long long { return (t & ((1ll << log_tpb) - 1)) << LOGC; };

  uint4 nx[PER];
  long long t = blockIdx.x;
  if (t < ntiles) {
    const uint32_t *src = in + tile_base(t) + c4;
#pragma unroll
    for (int k = 0; k < PER; ++k) nx[k] = *reinterpret_cast<const uint4 *>(src + ((long long)(j0 + k * RSTEP) << logS));
  }
  for (; t < ntiles; t += gridDim.x) {
    const long long gbase = tile_base(t), fbase = fs_base(t);
    __syncthreads();   // previous tile's LDS reads are done
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      uint4 x = nx[k];
      const int j = j0 + k * RSTEP;
      if (INV) {
        const uint4 f = *reinterpret_cast<const uint4 *>(A.fs + fbase + ((long long)j << logS) + c4);
        x.x = M.mul(x.x, f.x); x.y = M.mul(x.y, f.y); x.z = M.mul(x.z, f.z); x.w = M.mul(x.w, f.w);
      }
      *reinterpret_cast<uint4 *>(s + L::addr(c4, j)) = x;
    }
    const long long tn = t + gridDim.x;
    if (tn < ntiles) {
      const uint32_t *src = in + tile_base(tn) + c4;
#pragma unroll
      for (int k = 0; k < PER; ++k) nx[k] = *reinterpret_cast<const uint4 *>(src + ((long long)(j0 + k * RSTEP) << logS));
    }
    __syncthreads();
    uint32_t *const vec[1] = {s};
    if (!INV) dif_rounds<L, LOGT, LAZY, 1>(vec, tw, M);
    else dit_rounds<L, LOGT, LAZY, 1>(vec, tw, M);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int j = j0 + k * RSTEP;
      uint4 x = *reinterpret_cast<const uint4 *>(s + L::addr(c4, j));
      if (!INV) {
        const uint4 f = *reinterpret_cast<const uint4 *>(A.fs + fbase + ((long long)j << logS) + c4);
        x.x = M.mul(x.x, f.x); x.y = M.mul(x.y, f.y); x.z = M.mul(x.z, f.z); x.w = M.mul(x.w, f.w);
      }
      if (A.canon_out) { x.x = M.canon(x.x); x.y = M.canon(x.y); x.z = M.canon(x.z); x.w = M.canon(x.w); }
      *reinterpret_cast<uint4 *>(out + gbase + ((long long)j << logS) + c4) = x;
    }
  }
}

// ------------------------------------------------------------------------------------
// Element-wise helpers around the transforms
// ------------------------------------------------------------------------------------

// out[i] = in[bitrev_logn(i)] within each length-2^logn vector (stand-alone transform API:
// the reference's transform is natural-order in and out, main.cc:146-152).
__global__ void bitrev_kernel(const uint32_t *in, uint32_t *out, int logn, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long vec = i >> logn;
    const uint32_t k = (uint32_t)(i & ((1ll << logn) - 1));
    const uint32_t r = logn ? (__brev(k) >> (32 - logn)) : 0u;
    out[i] = in[(vec << logn) + r];
  }
}

// u64 coefficients -> zero-padded u32 residues mod up to 4 primes (residue split,
// main.cc:292-303).  src is [batch][n]; dst[k] is [batch][len].
struct SplitArgs {
  const uint64_t *src;
  uint32_t *dst[4];
  uint32_t mod[4];
  int nmod;
  int n, len;
  long long total;  // batch * len
};
__global__ void split_kernel(SplitArgs A) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < A.total; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / A.len;
    const int j = (int)(i - b * A.len);
    const uint64_t x = (j < A.n) ? A.src[b * A.n + j] : 0ull;
    for (int k = 0; k < A.nmod; ++k) A.dst[k][i] = (uint32_t)(x % A.mod[k]);
  }
}

// i32 / u32 coefficients -> zero-padded canonical residues
__global__ void pad_kernel(const uint32_t *src, uint32_t *dst, uint32_t p, int n, int len, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / len;
    const int j = (int)(i - b * len);
    const uint32_t x = (j < n) ? src[b * n + j] : 0u;
    dst[i] = x >= p ? x % p : x;
  }
}

// gather the first nout words of every length-len vector (u32 -> u32 or u64)
template <class OUT>
__global__ void unpad_kernel(const uint32_t *src, OUT *dst, int nout, int len, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / nout;
    const int j = (int)(i - b * nout);
    dst[i] = (OUT)src[b * len + j];
  }
}

// Multi-prime CRT recombination (replaces the 128-bit sum/% M loop of main.cc:323-337):
// Garner mixed-radix digits v_k, then Horner evaluation of
//   x = v0 + m0 (v1 + m1 (v2 + m2 v3))  modulo the caller's p on 64-bit Montgomery lanes.
struct CrtArgs {
  const uint32_t *res[4];   // [batch][len] canonical residues per prime
  uint64_t *out;            // [batch][nout]
  PrimeCtx ctx[4];
  uint32_t inv[4][4];       // inv[i][j] = m_i^{-1} mod m_j in Montgomery form (i < j)
  PrimeCtx64 pctx;          // caller modulus
  uint64_t mR[4];           // m_k * 2^64 mod p
  int nmod, nout, len;
  long long total;          // batch * nout
};
__global__ void crt_kernel(CrtArgs A) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < A.total; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / A.nout;
    const int j = (int)(i - b * A.nout);
    const long long src = b * A.len + j;
    uint32_t v[4];
    for (int k = 0; k < A.nmod; ++k) {
      const uint32_t mk = A.ctx[k].p;
      uint32_t t = A.res[k][src];
      for (int q = 0; q < k; ++q) {
        uint32_t vq = v[q] % mk;
        t = t >= vq ? t - vq : t + mk - vq;
        t = mont_mul(t, A.inv[q][k], mk, A.ctx[k].npinv);
      }
      v[k] = t;
    }
    const uint64_t p = A.pctx.p;
    uint64_t r = v[A.nmod - 1] % p;
    for (int k = A.nmod - 2; k >= 0; --k) {
      r = mont_mul64(r, A.mR[k], p, A.pctx.npinv);
      const uint64_t add = v[k] % p;
      r += add;
      if (r >= p) r -= p;
    }
    A.out[i] = r;
  }
}

}  // namespace nttx
