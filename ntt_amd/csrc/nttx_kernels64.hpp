// gfx950 kernels of the 64-bit-lane direct path: one transform modulo p itself for
// NTT-friendly moduli beyond 32 bits (2^31 <= p < 2^62, 2^k | p-1), instead of four
// 32-bit transforms plus CRT recombination (main.cc:286-338 always takes the latter; the
// single-prime structure is get_result of monttry2.cpp:319-343 on wider lanes).
//
// Same pass structure as the 32-bit kernels (column passes with a four-step multiply,
// contiguous row passes, forward decimation-in-frequency / inverse decimation-in-time so no
// permutation is run), radix-8 units on registers exchanged through LDS.  Residues are u64 in
// ordinary form, constants in Montgomery form with R = 2^64.
//
// Reduction comes in two classes chosen per modulus:
//   LAZY (p < 2^51): no conditional subtraction anywhere.  Forward sums double their bound per
//        stage (2^(s+1) p after s stages, at most 2^12 p < 2^63), inverse sums add 2p per stage,
//        and every multiplication by a constant brings a value back below 2p whatever it held.
//   SAFE (p < 2^62): every sum and difference is brought back below 2p (Harvey butterflies).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "nttx_arith.hpp"

namespace nttx {

// x*w*2^-64 mod p in [0, 2p) for any 64-bit x and canonical w < p < 2^62.
// PLO1: p = 1 mod 2^32 (true for every modulus whose p-1 is divisible by 2^32, such as
// 19*2^46+1): word-serial REDC with radix 2^32, where -p^-1 mod 2^32 = -1 and p's low word
// is 1, so each of the two reduction steps is one negation and one 32x32 multiply:
// six v_mad_u64_u32 in all.  Otherwise the two-multiply-high form is used.
// 1 if x != 0 else 0, as one v_min_u32 (the compiler's compare + select pair issues far slower)
__device__ __forceinline__ uint32_t nonzero32(uint32_t x) {
  uint32_t r;
  asm("v_min_u32 %0, 1, %1" : "=v"(r) : "v"(x));
  return r;
}

template <bool PLO1>
__device__ __forceinline__ uint64_t mont64_lazy(uint64_t x, uint64_t w, uint64_t p, uint64_t npinv) {
  if constexpr (PLO1) {
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint32_t ph = (uint32_t)(p >> 32);
    const uint64_t u = (uint64_t)x0 * w0;
    const uint32_t a0 = (uint32_t)u;
    const uint64_t v = (uint64_t)x0 * w1 + (u >> 32);                         // words 1..2 of x0*w
    const uint32_t m = 0u - a0;                                               // a0 + m = 2^32 * (a0 != 0)
    const uint64_t v2 = (uint64_t)m * ph + ((uint64_t)(uint32_t)v + nonzero32(a0));
    const uint32_t b1 = (uint32_t)(v >> 32) + (uint32_t)(v2 >> 32);           // (x0*w + m*p) / 2^32 < 2^63
    const uint32_t b0 = (uint32_t)v2;
    const uint64_t u2 = (uint64_t)x1 * w0 + b0;
    const uint32_t c0 = (uint32_t)u2;
    const uint64_t v3 = (uint64_t)x1 * w1 + ((uint64_t)b1 + (u2 >> 32));
    const uint32_t m2 = 0u - c0;
    const uint64_t v4 = (uint64_t)m2 * ph + ((uint64_t)(uint32_t)v3 + nonzero32(c0));
    const uint32_t d1 = (uint32_t)(v3 >> 32) + (uint32_t)(v4 >> 32);
    return ((uint64_t)d1 << 32) | (uint32_t)v4;
  } else {
    const uint64_t lo = x * w, hi = __umul64hi(x, w);
    const uint64_t m = lo * npinv;
    return hi + __umul64hi(m, p) + (lo != 0 ? 1u : 0u);
  }
}

__device__ __forceinline__ uint64_t csub64(uint64_t x, uint64_t m) { return x >= m ? x - m : x; }

struct PrimeArgs64 {
  uint64_t p, npinv;
  uint64_t one;    // R mod p: multiplying by it reduces a value without changing it
  uint64_t r2;     // R^2 mod p
};

template <bool SAFE, bool PLO1>
struct Arith64 {
  using Tw = uint64_t;                 // a twiddle: w R mod p
  static constexpr bool kFP = false;
  uint64_t p, npinv, p2;
  __device__ __forceinline__ Arith64(const PrimeArgs64 &c) : p(c.p), npinv(c.npinv), p2(2 * c.p) {}
  __device__ __forceinline__ uint64_t mul(uint64_t x, uint64_t w) const { return mont64_lazy<PLO1>(x, w, p, npinv); }
  // bound (in multiples of p, as a shift) of forward values before stage s of a pass: 2^(s+1)
  // bound of inverse values before stage s: 2, 4, 8, 16, then +2 per stage (twiddle-1 stages double)
  __device__ __forceinline__ static constexpr int inv_bound(int s) { return s <= 3 ? (2 << s) : 16 + 2 * (s - 3); }
  // Gentleman-Sande: (a, b) -> (a + b, (a - b) w); unit_w: w = 1; zhi: b is known to be zero
  __device__ __forceinline__ void dif(uint64_t &a, uint64_t &b, uint64_t w, bool unit_w, bool zhi, int s) const {
    if (zhi) { b = unit_w ? a : mul(a, w); return; }
    if constexpr (SAFE) {
      const uint64_t sum = csub64(a + b, p2), d = a - b + p2;
      b = unit_w ? csub64(d, p2) : mul(d, w);
      a = sum;
    } else {
      const uint64_t sum = a + b, d = a - b + (p << (s + 1));
      b = unit_w ? d : mul(d, w);
      a = sum;
    }
  }
  // Cooley-Tukey: (a, b) -> (a + b w, a - b w)
  __device__ __forceinline__ void dit(uint64_t &a, uint64_t &b, uint64_t w, bool unit_w, int s, int /*stage in unit*/ = 0,
                                      int /*stages of the unit*/ = 0) const {
    const uint64_t t = unit_w ? b : mul(b, w);
    if constexpr (SAFE) {
      b = csub64(a - t + p2, p2);
      a = csub64(a + t, p2);
    } else {
      const uint64_t off = unit_w ? p * (uint64_t)inv_bound(s) : p2;
      b = a - t + off;
      a = a + t;
    }
  }
  // fully reduced value of x (any history in the LAZY class, below 2p in the SAFE class)
  __device__ __forceinline__ uint64_t canon(uint64_t x, uint64_t one) const {
    if constexpr (SAFE) return csub64(x, p);
    else return csub64(mul(x, one), p);
  }
};

// ------------------------------------------------------------------------------------
// Floating-point class for p < 1.35e15 (2^50.26): residues travel as doubles holding integers of magnitude
// at most 1.5 p (the bit pattern in the same 64-bit words), twiddles as pairs (w, w/p rounded).
//   x w mod p:  h = fl(x w), l = fma(x, w, -h)  (x w = h + l exactly);  q = rndne(x * (w/p));  r = fma(-q, p, h) + l.
// h - q p is an integer of magnitude below 2^53, so the fused multiply-add returns it exactly and r = x w - q p with
// |q - x w / p| <= 0.5 + |x| 2^-52, i.e. |r| <= (0.5 + |x| / 2^52) p.  Every operand of a product or of a reduction
// x - rndne(x / p) p is a sum of two stored values, at most 3.25 p < 2^52, so every result is at most 1.5 p again
// (products of two stored values: q from h / p, |r| <= 1.63 p).  Six FP64 instructions replace the fourteen integer
// ones of a 64-bit Montgomery product; FP64 issues at the rate of v_mad_u64_u32 on gfx950.
// ------------------------------------------------------------------------------------
struct TwF { double w, wq; };
struct ArithFP {
  using Tw = TwF;
  static constexpr bool kFP = true;
  double p, pinv;
  uint64_t pi, npinv, r2;
  __device__ __forceinline__ ArithFP(const PrimeArgs64 &c) : p((double)c.p), pinv(1.0 / (double)c.p), pi(c.p), npinv(c.npinv), r2(c.r2) {}
  __device__ __forceinline__ static double D(uint64_t x) { return __longlong_as_double((long long)x); }
  __device__ __forceinline__ static uint64_t U(double x) { return (uint64_t)__double_as_longlong(x); }
  __device__ __forceinline__ double fpmod(double x, const TwF &w) const {
    const double h = x * w.w;
    const double l = __builtin_fma(x, w.w, -h);
    const double q = __builtin_rint(x * w.wq);
    return __builtin_fma(-q, p, h) + l;
  }
  __device__ __forceinline__ double fpred(double x) const { return __builtin_fma(-__builtin_rint(x * pinv), p, x); }
  __device__ __forceinline__ uint64_t mul(uint64_t x, const TwF &w) const { return U(fpmod(D(x), w)); }
  // product of two stored values
  __device__ __forceinline__ uint64_t mul2(uint64_t x, uint64_t y) const {
    const double a = D(x), b = D(y);
    const double h = a * b;
    const double l = __builtin_fma(a, b, -h);
    const double q = __builtin_rint(h * pinv);
    return U(__builtin_fma(-q, p, h) + l);
  }
  __device__ __forceinline__ void dif(uint64_t &a, uint64_t &b, const TwF &w, bool unit_w, bool zhi, int) const {
    const double x = D(a), y = D(b);
    if (zhi) { b = unit_w ? a : U(fpmod(x, w)); return; }
    const double d = x - y;
    a = U(fpred(x + y));
    b = U(unit_w ? fpred(d) : fpmod(d, w));
  }
  // Cooley-Tukey butterfly.  The sums a +- b w are reduced after the second stage of a unit only (and in units of one
  // stage, and where w = 1 leaves b unreduced): with unit inputs at most 1.63 p the first stage leaves at most
  // 1.63 + (0.5 + 1.63 rho) p <= 2.62 p (rho = p / 2^52 < 0.3), the second reduces to 0.5 p, the third leaves at most
  // 0.5 + (0.5 + 0.5 rho) p <= 1.15 p: every operand stays below 3.25 p and every stored value below 1.5 p.
  __device__ __forceinline__ void dit(uint64_t &a, uint64_t &b, const TwF &w, bool unit_w, int, int st = 1, int nst = 1) const {
    const double x = D(a), t = unit_w ? D(b) : fpmod(D(b), w);
    if (unit_w || st == 1 || nst == 1) { a = U(fpred(x + t)); b = U(fpred(x - t)); }
    else { a = U(x + t); b = U(x - t); }
  }
  // integer -> stored form; a word that is not below p (rare: callers pass residues) is reduced first, by two
  // Montgomery reductions as in pad64_kernel
  __device__ __forceinline__ uint64_t from_int(uint64_t v) const {
    if (v >= pi) v = csub64(mont64_lazy<false>(mont64_lazy<false>(v, r2, pi, npinv), 1ull, pi, npinv), pi);
    return U((double)v);
  }
  // stored form -> canonical integer residue
  __device__ __forceinline__ uint64_t canon(uint64_t x, uint64_t) const {
    double r = fpred(D(x));
    if (r < 0.0) r += p;
    if (r >= p) r -= p;
    return (uint64_t)r;
  }
};

// radix-2^R unit on registers, stages at point distances 2^(R-1) D ... D (forward) or D ... (inverse);
// w[m] = tw[lo + m*D]; D1: D == 1, where the twiddles of local index 0 equal 1; sbase = index of
// the unit's first stage inside its pass (for the LAZY bounds)
template <int R, bool INV, bool D1, bool ZHI, class AR>
__device__ __forceinline__ void unit64(uint64_t (&x)[1 << R], const typename AR::Tw (&w)[1 << R], const AR &M, int sbase) {
  constexpr int NP = 1 << R;
  if constexpr (!INV) {
#pragma unroll
    for (int st = 0; st < R; ++st) {
      const int half = NP >> (st + 1);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        if (i & half) continue;
        const int j = i & (half - 1);
        M.dif(x[i], x[i + half], w[half + j], D1 && j == 0, ZHI && st == 0, sbase + st);
      }
    }
  } else {
#pragma unroll
    for (int st = 0; st < R; ++st) {
      const int half = 1 << st;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        if (i & half) continue;
        const int j = i & (half - 1);
        M.dit(x[i], x[i + half], w[half + j], D1 && j == 0, sbase + st, st, R);
      }
    }
  }
}

// LDS layouts in 64-bit words (the padding keeps per-point offsets compile-time constants, as in
// the 32-bit layouts): rows get 4 pad words after every 32, column tiles one pad row after every 8.
template <int LOGT_>
struct RowLayout64 {
  static constexpr int LOGT = LOGT_;
  static constexpr int T = 1 << LOGT;
  static constexpr int TP = T + (T >> 5) * 4;
  static constexpr int NT = (T >= 2048) ? 1 : (2048 >> LOGT);
  static constexpr int WORDS = NT * TP;
  static constexpr int NTHR = (LOGT <= 11) ? 256 : (T / 8);   // longer rows: one row per workgroup, eight points per thread
  __device__ static constexpr int padj(int j) { return j + ((j >> 5) << 2); }
  __device__ static __forceinline__ int addr(int c, int j) { return c * TP + padj(j); }
  template <int LOGU>
  __device__ static __forceinline__ void split(int g, int &c, int &u) { c = g >> LOGU; u = g & ((1 << LOGU) - 1); }
};

template <int LOGT_, int LOGC_, int NTHR_>
struct ColLayout64 {
  static constexpr int LOGT = LOGT_, LOGC = LOGC_;
  static constexpr int T = 1 << LOGT;
  static constexpr int COLS = 1 << LOGC;
  static constexpr int NT = COLS;
  static constexpr int NTHR = NTHR_;
  static constexpr int ROWS = T + (T >> 3);
  static constexpr int WORDS = ROWS * COLS;
  __device__ static constexpr int padj(int j) { return (j + (j >> 3)) * COLS; }
  __device__ static __forceinline__ int addr(int c, int j) { return padj(j) + c; }
  template <int LOGU>
  __device__ static __forceinline__ void split(int g, int &c, int &u) { c = g & (COLS - 1); u = g >> LOGC; }
};

// one round: every radix-2^R unit at distance 2^LOGD of every transform of the tile, NV vectors
template <class L, int R, int LOGD, bool INV, class AR, int NV, bool ZHI>
__device__ __forceinline__ void ntt_round64(uint64_t *const (&s)[NV], const typename AR::Tw *tw, const AR &M) {
  constexpr int D = 1 << LOGD;
  constexpr int LOG_UNITS = L::LOGT - R;
  constexpr int TOTAL = L::NT << LOG_UNITS;
  constexpr int NP = 1 << R;
  constexpr int SBASE = INV ? LOGD : L::LOGT - (LOGD + R);
  for (int g = threadIdx.x; g < TOTAL; g += L::NTHR) {
    int c, u;
    L::template split<LOG_UNITS>(g, c, u);
    const int lo = u & (D - 1), hi = u >> LOGD;
    const int base = (hi << (LOGD + R)) + lo;
    const int a0 = L::addr(c, base);
    typename AR::Tw w[NP];
    w[0] = typename AR::Tw{};
#pragma unroll
    for (int m = 1; m < NP; ++m) w[m] = tw[lo + m * D];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      uint64_t x[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q) x[q] = (ZHI && q >= NP / 2) ? 0ull : s[v][a0 + L::padj(q * D)];
      unit64<R, INV, D == 1, ZHI, AR>(x, w, M, SBASE);
#pragma unroll
      for (int q = 0; q < NP; ++q) s[v][a0 + L::padj(q * D)] = x[q];
    }
  }
}

template <class L, int REM, class AR, int NV, bool ZTOP>
__device__ __forceinline__ void dif_rounds64(uint64_t *const (&s)[NV], const typename AR::Tw *tw, const AR &M) {
  constexpr bool zt = ZTOP && REM == L::LOGT;
  if constexpr (REM > 3) {
    ntt_round64<L, 3, REM - 3, false, AR, NV, zt>(s, tw, M);
    __syncthreads();
    dif_rounds64<L, REM - 3, AR, NV, ZTOP>(s, tw, M);
  } else {
    ntt_round64<L, REM, 0, false, AR, NV, zt>(s, tw, M);
  }
}

template <class L, int REM, class AR, int NV>
__device__ __forceinline__ void dit_rounds64(uint64_t *const (&s)[NV], const typename AR::Tw *tw, const AR &M) {
  if constexpr (REM > 3) {
    dit_rounds64<L, REM - 3, AR, NV>(s, tw, M);
    __syncthreads();
    ntt_round64<L, 3, REM - 3, true, AR, NV, false>(s, tw, M);
  } else {
    ntt_round64<L, REM, 0, true, AR, NV, false>(s, tw, M);
  }
}

// ------------------------------------------------------------------------------------
// Row kernel: transforms are contiguous runs of T words.
//   MODE 0: forward (natural in, bit-reversed out), 1: inverse, 2: forward a, forward b,
//   pointwise product, inverse.
// ------------------------------------------------------------------------------------
struct RowArgs64 {
  const uint64_t *in0, *in1;
  uint64_t *out;
  const uint64_t *tw_fwd, *tw_inv;   // stage-major tables, T words
  PrimeArgs64 ctx;
  uint64_t pre;          // MODE 2: Montgomery constant applied to a before a*b: R^2 (times n^-1 on single-pass lengths)
  int canon_out;
  long long nrows;
  // negacyclic products (TWIST kernels, rows = whole vectors): inputs times psi_in[i] on the way in, outputs times
  // psi_out[i] on the way out (Montgomery form, T words each)
  const uint64_t *psi_in, *psi_out;
  // floating-point class: tables hold (w, w/p) pairs of doubles; pre / pre_q = n^-1 and n^-1 / p as doubles on
  // single-pass lengths (0: none); int_in: the inputs are integers (first pass of a product), else stored doubles
  uint64_t pre_q;
  int int_in;
};

template <class L, bool TW = false, class AR = void, bool CVT = false>
__device__ __forceinline__ void row_load64(uint64_t *s, const uint64_t *in, long long row0, long long nrows,
                                           const uint64_t *psi = nullptr, const AR *M = nullptr, bool cvt = false) {
  constexpr int T = L::T;
  if constexpr (T >= 2) {
    constexpr int NV2 = L::NT * T / 2;
    for (int v = threadIdx.x; v < NV2; v += L::NTHR) {
      const int c = (v * 2) >> L::LOGT, j = (v * 2) & (T - 1);
      ulonglong2 x = make_ulonglong2(0, 0);
      if (row0 + c < nrows) {
        x = *reinterpret_cast<const ulonglong2 *>(in + (row0 + c) * T + j);
        if constexpr (CVT) { if (cvt) { x.x = M->from_int(x.x); x.y = M->from_int(x.y); } }
        if constexpr (TW) {
          if constexpr (CVT) {      // floating-point class: the table holds (psi^i, psi^i / p) pairs
            const typename AR::Tw *f = reinterpret_cast<const typename AR::Tw *>(psi) + j;
            x.x = M->mul(x.x, f[0]); x.y = M->mul(x.y, f[1]);
          } else {
            const ulonglong2 f = *reinterpret_cast<const ulonglong2 *>(psi + j);
            x.x = M->mul(x.x, f.x); x.y = M->mul(x.y, f.y);
          }
        }
      }
      *reinterpret_cast<ulonglong2 *>(s + L::addr(c, j)) = x;
    }
  }
}

template <int LOGT, int MODE, bool SAFE, bool PLO1>
__global__ void __launch_bounds__(RowLayout64<LOGT>::NTHR) row64_kernel(RowArgs64 A) {
  using L = RowLayout64<LOGT>;
  using AR = Arith64<SAFE, PLO1>;
  constexpr int T = L::T;
  constexpr int NVEC = (MODE == 2) ? 2 : 1;
  __shared__ __attribute__((aligned(16))) uint64_t lds[NVEC * L::WORDS + (MODE == 2 ? 2 : 1) * T];
  uint64_t *sa = lds, *sb = lds + (NVEC - 1) * L::WORDS;
  uint64_t *twf = lds + NVEC * L::WORDS;
  uint64_t *twi = twf + (MODE == 2 ? T : 0);
  const AR M(A.ctx);
  if (MODE != 1) for (int i = threadIdx.x; i < T; i += L::NTHR) twf[i] = A.tw_fwd[i];
  if (MODE != 0) for (int i = threadIdx.x; i < T; i += L::NTHR) twi[i] = A.tw_inv[i];

  for (long long row0 = (long long)blockIdx.x * L::NT; row0 < A.nrows; row0 += (long long)gridDim.x * L::NT) {
    __syncthreads();
    row_load64<L>(sa, A.in0, row0, A.nrows);
    if (MODE == 2) row_load64<L>(sb, A.in1, row0, A.nrows);
    __syncthreads();
    if (MODE == 0) {
      uint64_t *const v[1] = {sa};
      dif_rounds64<L, LOGT, AR, 1, false>(v, twf, M);
    } else if (MODE == 1) {
      uint64_t *const v[1] = {sa};
      dit_rounds64<L, LOGT, AR, 1>(v, twi, M);
    } else {
      uint64_t *const v2[2] = {sa, sb};
      dif_rounds64<L, LOGT, AR, 2, false>(v2, twf, M);
      __syncthreads();
      // pointwise: (a * pre) is below 2p whatever a held, so its product with b (< 2^12 p) fits REDC
      for (int i = threadIdx.x; i < L::NT * T; i += L::NTHR) {
        const int ad = L::addr(i >> LOGT, i & (T - 1));
        sa[ad] = M.mul(M.mul(sa[ad], A.pre), sb[ad]);
      }
      __syncthreads();
      uint64_t *const v[1] = {sa};
      dit_rounds64<L, LOGT, AR, 1>(v, twi, M);
    }
    __syncthreads();
    constexpr int NV2 = L::NT * T / 2;
    for (int v = threadIdx.x; v < NV2; v += L::NTHR) {
      const int c = (v * 2) >> LOGT, j = (v * 2) & (T - 1);
      if (row0 + c < A.nrows) {
        ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(sa + L::addr(c, j));
        if (A.canon_out) { x.x = M.canon(x.x, A.ctx.one); x.y = M.canon(x.y, A.ctx.one); }
        *reinterpret_cast<ulonglong2 *>(A.out + (row0 + c) * T + j) = x;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Fused product on rows: forward a, forward b, pointwise, inverse, through ONE LDS buffer.
// The last forward round, the pointwise product and the first inverse round all act on the
// same 8 points of a thread (units at distance 1), so a's and b's transformed values wait in
// registers while the other vector uses the buffer: half the LDS of keeping both vectors
// resident (twice the workgroups per CU) and two LDS round trips fewer.
// ------------------------------------------------------------------------------------
// all rounds of a forward pass but the last (which has R = RL stages at distance 1); rounds whose
// twiddles lie within the first TWL table words read the LDS copy twl, the others the table itself
template <class L, int REM, class AR, int TWL>
__device__ __forceinline__ void dif_head64(uint64_t *s, const typename AR::Tw *twg, const typename AR::Tw *twl, const AR &M) {
  if constexpr (REM > 3) {
    uint64_t *const v[1] = {s};
    ntt_round64<L, 3, REM - 3, false, AR, 1, false>(v, (8 << (REM - 3)) <= TWL ? twl : twg, M);
    __syncthreads();
    dif_head64<L, REM - 3, AR, TWL>(s, twg, twl, M);
  }
}
// all rounds of an inverse pass after the first
template <class L, int REM, class AR, int TWL>
__device__ __forceinline__ void dit_tail64(uint64_t *s, const typename AR::Tw *twg, const typename AR::Tw *twl, const AR &M) {
  if constexpr (REM > 3) {
    dit_tail64<L, REM - 3, AR, TWL>(s, twg, twl, M);
    __syncthreads();
    uint64_t *const v[1] = {s};
    ntt_round64<L, 3, REM - 3, true, AR, 1, false>(v, (8 << (REM - 3)) <= TWL ? twl : twg, M);
  }
}
// the distance-1 round on registers: r[u*NP + q] = point q of the thread's u-th unit
template <class L, int RL, bool INV, bool LOAD, bool STORE, class AR>
__device__ __forceinline__ void edge_round64(uint64_t *s, const typename AR::Tw *tw, const AR &M, uint64_t (&r)[8]) {
  constexpr int NP = 1 << RL, NU = 8 / NP;
  constexpr int LOG_UNITS = L::LOGT - RL;
  static_assert((L::NT << LOG_UNITS) == NU * L::NTHR, "eight points per thread");
  typename AR::Tw w[NP];
  w[0] = typename AR::Tw{};
#pragma unroll
  for (int m = 1; m < NP; ++m) w[m] = tw[m];     // distance 1: lo = 0
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int g = threadIdx.x + u * L::NTHR;
    int c, un;
    L::template split<LOG_UNITS>(g, c, un);
    const int a0 = L::addr(c, un << RL);
    uint64_t x[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) x[q] = LOAD ? s[a0 + L::padj(q)] : r[u * NP + q];
    unit64<RL, INV, true, false, AR>(x, w, M, INV ? 0 : L::LOGT - RL);
#pragma unroll
    for (int q = 0; q < NP; ++q) { if (STORE) s[a0 + L::padj(q)] = x[q]; else r[u * NP + q] = x[q]; }
  }
}

// TWIST: negacyclic product of whole vectors (single pass, canonical output): psi factors on load and store.
// FP: the floating-point class (ArithFP) instead of Arith64<SAFE, PLO1>; tables are (w, w/p) pairs.
template <int LOGT, bool SAFE, bool PLO1, bool TWIST = false, bool FP = false>
__global__ void __launch_bounds__(RowLayout64<LOGT>::NTHR, (LOGT >= 12 ? 4 : 1)) rowmul64_kernel(RowArgs64 A) {
  using L = RowLayout64<LOGT>;
  using AR = typename std::conditional<FP, ArithFP, Arith64<SAFE, PLO1>>::type;
  using Tw = typename AR::Tw;
  constexpr int T = L::T;
  constexpr int RL = (LOGT - 1) % 3 + 1;          // stages of the distance-1 round
  // stage tables in LDS: whole up to 2^11 points (2^9 for the 16-byte twiddles of the floating-point class); of
  // longer rows only the first 512 entries (the rounds at distance <= 64), the outer rounds read the tables
  // through the caches
  constexpr int TWL = (LOGT <= (FP ? 9 : 11)) ? T : 512;
  constexpr int TWW = sizeof(Tw) / 8;
  __shared__ __attribute__((aligned(16))) uint64_t lds[L::WORDS + 2 * TWL * TWW];
  uint64_t *sa = lds;
  Tw *twf = reinterpret_cast<Tw *>(lds + L::WORDS), *twi = twf + TWL;
  const Tw *gtwf = reinterpret_cast<const Tw *>(A.tw_fwd), *gtwi = reinterpret_cast<const Tw *>(A.tw_inv);
  const AR M(A.ctx);
  for (int i = threadIdx.x; i < TWL; i += L::NTHR) { twf[i] = gtwf[i]; twi[i] = gtwi[i]; }

  for (long long row0 = (long long)blockIdx.x * L::NT; row0 < A.nrows; row0 += (long long)gridDim.x * L::NT) {
    uint64_t ra[8], rb[8];
    __syncthreads();
    row_load64<L, TWIST, AR, FP>(sa, A.in0, row0, A.nrows, A.psi_in, &M, A.int_in != 0);
    __syncthreads();
    dif_head64<L, LOGT, AR, TWL>(sa, gtwf, twf, M);
    edge_round64<L, RL, false, true, false, AR>(sa, twf, M, ra);
    __syncthreads();
    row_load64<L, TWIST, AR, FP>(sa, A.in1, row0, A.nrows, A.psi_in, &M, A.int_in != 0);
    __syncthreads();
    dif_head64<L, LOGT, AR, TWL>(sa, gtwf, twf, M);
    edge_round64<L, RL, false, true, false, AR>(sa, twf, M, rb);
    // pointwise: (a * pre) is below 2p whatever a held, so its product with b (< 2^12 p after up to 11 lazy
    // stages, p < 2^51) fits REDC; longer lazy rows leave b up to 2^14 p, so b is reduced first
    if constexpr (FP) {     // a b, then n^-1 where this pass is the whole transform
      const TwF pre{ArithFP::D(A.pre), ArithFP::D(A.pre_q)};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint64_t ab = M.mul2(ra[k], rb[k]);
        ra[k] = A.pre ? M.mul(ab, pre) : ab;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint64_t bk = (!SAFE && LOGT >= 12) ? M.mul(rb[k], A.ctx.one) : rb[k];
        ra[k] = M.mul(M.mul(ra[k], A.pre), bk);
      }
    }
    edge_round64<L, RL, true, false, true, AR>(sa, twi, M, ra);
    __syncthreads();
    dit_tail64<L, LOGT, AR, TWL>(sa, gtwi, twi, M);
    __syncthreads();
    constexpr int NV2 = L::NT * T / 2;
    for (int v = threadIdx.x; v < NV2; v += L::NTHR) {
      const int c = (v * 2) >> LOGT, j = (v * 2) & (T - 1);
      if (row0 + c < A.nrows) {
        ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(sa + L::addr(c, j));
        if constexpr (TWIST && FP) {
          const Tw *f = reinterpret_cast<const Tw *>(A.psi_out) + j;
          x.x = M.canon(M.mul(x.x, f[0]), 0); x.y = M.canon(M.mul(x.y, f[1]), 0);
        } else if constexpr (TWIST) {     // the multiplication is also the reduction: below 2p, then canonical
          const ulonglong2 f = *reinterpret_cast<const ulonglong2 *>(A.psi_out + j);
          x.x = csub64(M.mul(x.x, f.x), M.p); x.y = csub64(M.mul(x.y, f.y), M.p);
        } else if (A.canon_out) { x.x = M.canon(x.x, A.ctx.one); x.y = M.canon(x.y, A.ctx.one); }
        *reinterpret_cast<ulonglong2 *>(A.out + (row0 + c) * T + j) = x;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Column kernel: T points at stride S inside blocks of N = T*S words; tile = T rows x 16
// contiguous columns (128-byte row segments).  Forward multiplies point (j', s) by
// fs[j'*S + s] = w_N^(s * bitrev_T(j')) on the way out, inverse by the inverse table on the way in.
// ZHI (forward): rows from T/2 on hold the zero padding of a linear product and are not read;
// nvalid = words of each block that hold data.
// ------------------------------------------------------------------------------------
struct ColArgs64 {
  const uint64_t *in[2];
  uint64_t *out[2];
  const uint64_t *tw, *fs;
  PrimeArgs64 ctx;
  int logS;
  int canon_out;
  long long nblocks;
  long long nvalid;
  const uint64_t *psi;   // TWIST kernels (outermost pass of a negacyclic product): psi^i forward, psi^-i inverse, N words
  int int_in;            // floating-point class, forward: the inputs are integers (first pass of a product)
};

// TWIST: outermost pass of a negacyclic product: forward inputs times psi^i as they are read, inverse outputs times
// psi^-i before the canonical store.
// FP: the floating-point class; tw and fs hold (w, w/p) pairs, int_in marks integer inputs (first forward pass).
// NTHR: threads per workgroup.  One tile is 2^LOGT rows x 16 columns; with 2^(LOGT + 1) threads every thread moves four
// 16-byte pieces and runs one radix-8 unit per round, with 2^LOGT threads eight pieces and two units.
template <int LOGT, bool INV, bool SAFE, bool PLO1, bool ZHI, bool TWIST = false, bool FP = false, int NTHR = 512>
__global__ void __launch_bounds__(NTHR, (NTHR == 512 ? 4 : 1)) col64_kernel(ColArgs64 A) {
  static_assert(!TWIST || !ZHI, "twisted passes read whole vectors");
  constexpr int LOGC = 4;
  using L = ColLayout64<LOGT, LOGC, NTHR>;
  using AR = typename std::conditional<FP, ArithFP, Arith64<SAFE, PLO1>>::type;
  using Tw = typename AR::Tw;
  constexpr int TWW = sizeof(Tw) / 8;
  constexpr int T = L::T, COLS = L::COLS;
  constexpr int LPR = COLS / 2, LOG_LPR = LOGC - 1;     // lanes per row segment, 16 bytes each
  constexpr int PER = (T * LPR) / NTHR;
  constexpr int RSTEP = NTHR / LPR;
  constexpr int PERL = ZHI ? (PER + 1) / 2 : PER;
  static_assert(PER >= 1 && PER * NTHR == T * LPR, "tile must be a whole number of pieces per thread");
  static_assert(!ZHI || (!INV && (PER % 2 == 0 || PER == 1)), "zero-half variant is forward only");
  __shared__ __attribute__((aligned(16))) uint64_t lds[L::WORDS + T * TWW];
  uint64_t *s = lds;
  Tw *tw = reinterpret_cast<Tw *>(lds + L::WORDS);
  const Tw *gtw = reinterpret_cast<const Tw *>(A.tw), *gfs = reinterpret_cast<const Tw *>(A.fs);
  const AR M(A.ctx);
  const uint64_t *in = A.in[blockIdx.z];
  uint64_t *out = A.out[blockIdx.z];
  const int logS = A.logS;
  const int log_tpb = logS - LOGC;
  const long long ntiles = A.nblocks << log_tpb;
  const int j0 = threadIdx.x >> LOG_LPR, c2 = (threadIdx.x & (LPR - 1)) * 2;

  for (int i = threadIdx.x; i < T; i += NTHR) tw[i] = gtw[i];

  auto tile_base = [&](long long t) -> long long {
    const long long blk = t >> log_tpb;
    const long long ct = t & ((1ll << log_tpb) - 1);
    return (blk << (logS + LOGT)) + (ct << LOGC);
  };
  auto fs_base = [&](long long t) -> long long { return (t & ((1ll << log_tpb) - 1)) << LOGC; };
  auto load_piece = [&](long long base, long long t_, int j) -> ulonglong2 {
    if constexpr (!ZHI) {
      return *reinterpret_cast<const ulonglong2 *>(in + base + ((long long)j << logS) + c2);
    } else {
      const long long ct = t_ & ((1ll << log_tpb) - 1);
      const long long w0 = ((long long)j << logS) + (ct << LOGC) + c2;
      ulonglong2 x = make_ulonglong2(0, 0);
      if (j < T / 2 && w0 < A.nvalid) {
        x = *reinterpret_cast<const ulonglong2 *>(in + base + ((long long)j << logS) + c2);
        if (w0 + 1 >= A.nvalid) x.y = 0;
      }
      return x;
    }
  };
  ulonglong2 nx[PERL];
  long long t0 = blockIdx.x;
  if (t0 < ntiles) {
    const long long base = tile_base(t0);
#pragma unroll
    for (int k = 0; k < PERL; ++k) nx[k] = load_piece(base, t0, j0 + k * RSTEP);
  }
  for (; t0 < ntiles; t0 += gridDim.x) {
    const long long gbase = tile_base(t0), fbase = fs_base(t0);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PERL; ++k) {
      ulonglong2 x = nx[k];
      const int j = j0 + k * RSTEP;
      if (INV) {
        if constexpr (FP) {
          const Tw *f = gfs + fbase + ((long long)j << logS) + c2;
          x.x = M.mul(x.x, f[0]); x.y = M.mul(x.y, f[1]);
        } else {
          const ulonglong2 f = *reinterpret_cast<const ulonglong2 *>(A.fs + fbase + ((long long)j << logS) + c2);
          x.x = M.mul(x.x, f.x); x.y = M.mul(x.y, f.y);
        }
      } else if constexpr (FP) {
        if (A.int_in) { x.x = M.from_int(x.x); x.y = M.from_int(x.y); }
        if constexpr (TWIST) {
          const Tw *f = reinterpret_cast<const Tw *>(A.psi) + fbase + ((long long)j << logS) + c2;
          x.x = M.mul(x.x, f[0]); x.y = M.mul(x.y, f[1]);
        }
      } else if constexpr (TWIST) {
        const ulonglong2 f = *reinterpret_cast<const ulonglong2 *>(A.psi + fbase + ((long long)j << logS) + c2);
        x.x = M.mul(x.x, f.x); x.y = M.mul(x.y, f.y);
      }
      *reinterpret_cast<ulonglong2 *>(s + L::addr(c2, j)) = x;
    }
    if (t0 + gridDim.x < ntiles) {
      const long long tn = t0 + gridDim.x, base = tile_base(tn);
#pragma unroll
      for (int k = 0; k < PERL; ++k) nx[k] = load_piece(base, tn, j0 + k * RSTEP);
    }
    __syncthreads();
    uint64_t *const vec[1] = {s};
    if constexpr (!INV) dif_rounds64<L, LOGT, AR, 1, ZHI>(vec, tw, M);
    else dit_rounds64<L, LOGT, AR, 1>(vec, tw, M);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int j = j0 + k * RSTEP;
      ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(s + L::addr(c2, j));
      if (!INV) {
        if constexpr (FP) {
          const Tw *f = gfs + fbase + ((long long)j << logS) + c2;
          x.x = M.mul(x.x, f[0]); x.y = M.mul(x.y, f[1]);
          if (A.canon_out) { x.x = M.canon(x.x, 0); x.y = M.canon(x.y, 0); }
        } else {
          const ulonglong2 f = *reinterpret_cast<const ulonglong2 *>(A.fs + fbase + ((long long)j << logS) + c2);
          x.x = M.mul(x.x, f.x); x.y = M.mul(x.y, f.y);
          if (A.canon_out) { x.x = csub64(x.x, M.p); x.y = csub64(x.y, M.p); }
        }
      } else if constexpr (TWIST && FP) {
        const Tw *f = reinterpret_cast<const Tw *>(A.psi) + fbase + ((long long)j << logS) + c2;
        x.x = M.canon(M.mul(x.x, f[0]), 0); x.y = M.canon(M.mul(x.y, f[1]), 0);
      } else if constexpr (TWIST) {
        const ulonglong2 f = *reinterpret_cast<const ulonglong2 *>(A.psi + fbase + ((long long)j << logS) + c2);
        x.x = csub64(M.mul(x.x, f.x), M.p); x.y = csub64(M.mul(x.y, f.y), M.p);
      } else if (A.canon_out) {
        x.x = M.canon(x.x, A.ctx.one); x.y = M.canon(x.y, A.ctx.one);
      }
      *reinterpret_cast<ulonglong2 *>(out + gbase + ((long long)j << logS) + c2) = x;
    }
  }
}

// compact [batch][n] -> zero-padded [batch][len] (words n..zfill-1 zeroed), values >= p reduced
struct PadArgs64 {
  const uint64_t *src[2];
  uint64_t *dst[2];
  PrimeArgs64 ctx;
  int n, len, zfill;
  int batch;
};
template <bool PLO1>
__global__ void pad64_kernel(PadArgs64 A) {
  for (long long b = blockIdx.y; b < A.batch; b += gridDim.y) {
    const uint64_t *src = A.src[blockIdx.z] + b * A.n;
    uint64_t *dst = A.dst[blockIdx.z] + b * A.len;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < A.zfill; j += gridDim.x * blockDim.x) {
      uint64_t x = (j < A.n) ? src[j] : 0ull;
      if (x >= A.ctx.p) {   // x -> x R -> x, each step a REDC that leaves a value below 2p
        const uint64_t xr = mont64_lazy<PLO1>(x, A.ctx.r2, A.ctx.p, A.ctx.npinv);
        x = csub64(mont64_lazy<PLO1>(xr, 1ull, A.ctx.p, A.ctx.npinv), A.ctx.p);
      }
      dst[j] = x;
    }
  }
}

// dst[i] = src[i] * tab[i] (tab in Montgomery form): the psi^i / psi^-i factors of a negacyclic product.
// Results are below 2p, or canonical when asked; any 64-bit input word is admissible.
struct Twist64Args {
  const uint64_t *src[2];
  uint64_t *dst[2];
  const uint64_t *tab;
  PrimeArgs64 ctx;
  int len, canon;
  int batch;
};
template <bool PLO1>
__global__ void twist64_kernel(Twist64Args A) {
  for (long long b = blockIdx.y; b < A.batch; b += gridDim.y) {
    const uint64_t *src = A.src[blockIdx.z] + b * A.len;
    uint64_t *dst = A.dst[blockIdx.z] + b * A.len;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2; i < A.len; i += gridDim.x * blockDim.x * 2) {
      const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(src + i), f = *reinterpret_cast<const ulonglong2 *>(A.tab + i);
      ulonglong2 y;
      y.x = mont64_lazy<PLO1>(x.x, f.x, A.ctx.p, A.ctx.npinv);
      y.y = mont64_lazy<PLO1>(x.y, f.y, A.ctx.p, A.ctx.npinv);
      if (A.canon) { y.x = csub64(y.x, A.ctx.p); y.y = csub64(y.y, A.ctx.p); }
      *reinterpret_cast<ulonglong2 *>(dst + i) = y;
    }
  }
}

static __global__ void unpad64_kernel(const uint64_t *src, uint64_t *dst, int nout, int len, int batch) {
  for (long long b = blockIdx.y; b < batch; b += gridDim.y)
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nout; j += gridDim.x * blockDim.x)
      dst[b * nout + j] = src[b * len + j];
}

}  // namespace nttx
