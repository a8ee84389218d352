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
//    pass doing all of its log2(T) butterfly stages on chip.
//  * Forward passes are decimation-in-frequency (natural in, bit-reversed out),
//    inverse passes decimation-in-time (bit-reversed in, natural out), so the
//    fused multiply never runs a bit-reversal permutation.
//  * Inside a pass each thread owns a radix-8 (or 4/2) unit: 8 points in
//    registers, 3 stages, then one exchange through LDS.
//  * Residues are 32-bit, Montgomery radix 2^32; twiddles are precomputed in
//    Montgomery form.  Values are reduced lazily: every register value carries a
//    compile-time bound (a multiple of p) and a conditional subtraction is emitted
//    only where a sum would otherwise leave the 32-bit lane.
//
// Measured gfx950 issue costs behind these choices (cycles per wave64 instruction per
// SIMD with >= 2 waves): v_add/v_sub 2.5, v_min_u32 4.2, v_mul_lo_u32 4.2,
// v_mad_u64_u32 4.5.  The path is integer-VALU bound, not MFMA material.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "nttx_arith.hpp"

namespace nttx {

constexpr int WG = 256;          // threads per workgroup of the row kernels

// ------------------------------------------------------------------------------------
// Bounded lazy arithmetic.  V<K> holds a value v < K*p.  H = how many multiples of p are
// known to fit below 2^32 for the modulus class: 2 (p < 2^31), 4 (p < 2^30), 9 (p < 2^32/9).
// ------------------------------------------------------------------------------------
template <int K>
struct V { uint32_t v; static constexpr int bound = K; };
template <class A, class B>
struct Pair { A a; B b; };
template <class A, class B>
__device__ __forceinline__ Pair<A, B> mkpair(A a, B b) { return Pair<A, B>{a, b}; }

template <int H>
struct Arith {
  // Bounds kept by values that cross an LDS exchange between rounds, per direction.  Chosen by
  // counting the conditional subtractions a radix-8 unit needs under each contract (forward units
  // want small inputs because sums add bounds; inverse units only ever add 2p per stage).
  static constexpr int KF = 2;                                   // forward (decimation-in-frequency) rounds
  static constexpr int KI = (H >= 9) ? 8 : (H >= 4) ? 4 : 2;     // inverse (decimation-in-time) rounds
  // bound left on the two factors of a pointwise product (KP*KP <= H where possible)
  static constexpr int KP = (H >= 9) ? 3 : 2;
  // a value about to be multiplied by a Montgomery constant needs no reduction at all
  static constexpr int KANY = H;
  // bound of a value fresh from a multiplication by a constant
  static constexpr int KM = 2;
  uint32_t p, npinv;
  // The multiples of p also live in vector registers: on gfx950 a 32-bit add/subtract whose
  // operands are all VGPRs issues in about 2.5 cycles per wave, one with a scalar or literal
  // operand in about 4.2.  Copies that no butterfly uses are dropped by the compiler.
  uint32_t vp[H + 1];
  __device__ __forceinline__ Arith(const PrimeCtx &c) : p(c.p), npinv(c.npinv) {
#pragma unroll
    for (int k = 1; k <= H; ++k) asm("v_mov_b32 %0, %1" : "=v"(vp[k]) : "s"((uint32_t)k * c.p));
  }
  template <int C>
  __device__ __forceinline__ uint32_t kp() const {
    static_assert(C >= 1 && C <= H, "multiple of p outside the modulus class");
    return vp[C];
  }
  // one conditional subtraction: [0, K p) -> [0, ceil(K/2) p)
  template <int K>
  __device__ __forceinline__ V<(K + 1) / 2> half(V<K> x) const {
    return V<(K + 1) / 2>{csub(x.v, kp<(K + 1) / 2>())};
  }
  template <int T, int K>
  __device__ __forceinline__ auto reduce(V<K> x) const {
    if constexpr (K <= T) return x; else return reduce<T>(half(x));
  }
  // x * w, w a canonical Montgomery-form constant: any 32-bit x is admissible
  template <int K>
  __device__ __forceinline__ V<2> mul(V<K> x, uint32_t w) const { return V<2>{mont_mul_lazy(x.v, w, p, npinv)}; }
  // data * data: needs a*b < 2^32 p, i.e. KA*KB <= H
  template <int KA, int KB>
  __device__ __forceinline__ V<2> mul2(V<KA> a, V<KB> b) const {
    if constexpr (KA * KB <= H) return V<2>{mont_mul_lazy(a.v, b.v, p, npinv)};
    else if constexpr (KA >= KB) return mul2(half(a), b);
    else return mul2(a, half(b));
  }
  // Gentleman-Sande butterfly (a, b) -> (a + b, (a - b) w)
  template <int KA, int KB>
  __device__ __forceinline__ auto dif(V<KA> a, V<KB> b, uint32_t w) const {
    if constexpr (KA + KB <= H) return mkpair(V<KA + KB>{a.v + b.v}, mul(V<KA + KB>{a.v - b.v + kp<KB>()}, w));
    else if constexpr (KA >= KB) return dif(half(a), b, w);
    else return dif(a, half(b), w);
  }
  // twiddle 1: (a + b, a - b), used by both directions
  template <int KA, int KB>
  __device__ __forceinline__ auto bf1(V<KA> a, V<KB> b) const {
    if constexpr (KA + KB <= H) return mkpair(V<KA + KB>{a.v + b.v}, V<KA + KB>{a.v - b.v + kp<KB>()});
    else if constexpr (KA >= KB) return bf1(half(a), b);
    else return bf1(a, half(b));
  }
  // Cooley-Tukey butterfly (a, b) -> (a + b w, a - b w); b may carry any bound
  template <int KA, int KB>
  __device__ __forceinline__ auto dit(V<KA> a, V<KB> b, uint32_t w) const { return bf1(a, mul(b, w)); }
  template <int K>
  __device__ __forceinline__ uint32_t canon(V<K> x) const { return reduce<1>(x).v; }
  template <int KOUT, int K>
  __device__ __forceinline__ uint32_t out(V<K> x) const { return reduce<KOUT>(x).v; }
  // V<1> -> canonical residue (already is one here)
  __device__ __forceinline__ uint32_t fix1(uint32_t v) const { return v; }
  // bound of the values a fused row kernel finds in memory, and what it may leave there for an
  // inverse column pass (which multiplies first): anything
  static constexpr int KROW = KM;
  __device__ __forceinline__ uint32_t raw_out(uint32_t v) const { return v; }
};

// Signed classes (H = 1 and H = 0; H = 17 and H = 16 below), used by the fused row kernels for p < 2^29, where 4p fits a signed 32-bit lane: V<K> holds an int32 of
// magnitude below K*p (K <= 4).  Sums and differences need no offset, a signed REDC returns a value in
// (-p, p), and any value is brought back into that range by one multiplication by 1 (three
// instructions, the price of one and a half conditional subtractions) instead of a chain of them.
// Counted over a radix-8 unit this is 12-18 % fewer instructions than the unsigned class with 9p < 2^32.
// ASM: the two multiply-adds of the REDC as inline assembly (the 512-point kernel, where the fixed
// pairing is 1 % faster) or left to the compiler's scheduler (the long-row kernels, where inline
// assembly costs hazard no-ops the scheduler cannot fill: 4-11 % slower).
// HS = 16 (Arith<16>, p < 2^27): the same lanes with sixteen multiples of headroom.  A forward radix-8 unit then reduces
// two of its outputs instead of six values, an inverse unit only the one input every output's additive chain starts from
// (DIT_PRE), and the pointwise product one factor: 20 % fewer REDCs per 512-point product row than with four multiples
// (18 % fewer VALU instructions, 10 % less time).  The column kernels gain nothing from it (forward 9 % fewer
// instructions in the same time, inverse 5 % slower: its reductions become REDCs where the unsigned class spends
// conditional subtractions), so they stay on the unsigned class for these moduli too.
template <bool ASM, int HS_ = 4>
struct SignedArith {
  static constexpr int HS = HS_;      // largest admissible bound
  static constexpr bool WIDE = HS >= 16;
  static constexpr int KM = 1, KF = WIDE ? 2 : 1, KI = WIDE ? 8 : 4, KP = WIDE ? HS : 2, KANY = HS;
  static constexpr bool DIT_PRE = WIDE;
  uint32_t p, pinv, one;
  int32_t negp;
  __device__ __forceinline__ SignedArith(const PrimeCtx &c) : p(c.p), pinv(0u - c.npinv), one(c.one), negp(-(int32_t)c.p) {}
  // x*w*2^-32 mod p in (-p, p) for |x| < 2^31, |w| < p: t - m p has a zero low word and |t - m p| < 2^32 p
  // (three instructions: v_mad_i64_i32, v_mul_lo_u32, v_mad_i64_i32; the subtraction is folded into the
  // second multiply-add by multiplying with -p)
  __device__ __forceinline__ uint32_t sredc(uint32_t x, uint32_t w) const {
    if constexpr (ASM) {
      int64_t t, r;
      uint32_t m;
      asm("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"(x), "v"(w) : "vcc");
      m = (uint32_t)t * pinv;
      asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(m), "s"(negp), "v"(t) : "vcc");
      return (uint32_t)((uint64_t)r >> 32);
    } else {
      const int64_t t = (int64_t)(int32_t)x * (int64_t)(int32_t)w;
      const uint32_t m = (uint32_t)t * pinv;
      const int64_t r = (int64_t)(int32_t)m * (int64_t)negp + t;
      return (uint32_t)((uint64_t)r >> 32);
    }
  }
  template <int K>
  __device__ __forceinline__ V<1> mul(V<K> x, uint32_t w) const { return V<1>{sredc(x.v, w)}; }
  template <int K>
  __device__ __forceinline__ V<1> half(V<K> x) const { return mul(x, one); }
  template <int T, int K>
  __device__ __forceinline__ auto reduce(V<K> x) const {
    if constexpr (K <= T) return x; else return half(x);
  }
  template <int KA, int KB>
  __device__ __forceinline__ V<1> mul2(V<KA> a, V<KB> b) const {      // |a b| < 2^31 p  <=>  KA*KB*p < 2^31
    if constexpr (KA * KB <= HS) return V<1>{sredc(a.v, b.v)};
    else if constexpr (KA >= KB) return mul2(half(a), b);
    else return mul2(a, half(b));
  }
  template <int KA, int KB>
  __device__ __forceinline__ auto dif(V<KA> a, V<KB> b, uint32_t w) const {
    if constexpr (KA + KB <= HS) return mkpair(V<KA + KB>{a.v + b.v}, mul(V<KA + KB>{a.v - b.v}, w));
    else if constexpr (KA >= KB) return dif(half(a), b, w);
    else return dif(a, half(b), w);
  }
  template <int KA, int KB>
  __device__ __forceinline__ auto bf1(V<KA> a, V<KB> b) const {
    if constexpr (KA + KB <= HS) return mkpair(V<KA + KB>{a.v + b.v}, V<KA + KB>{a.v - b.v});
    else if constexpr (KA >= KB) return bf1(half(a), b);
    else return bf1(a, half(b));
  }
  template <int KA, int KB>
  __device__ __forceinline__ auto dit(V<KA> a, V<KB> b, uint32_t w) const { return bf1(a, mul(b, w)); }
  __device__ __forceinline__ uint32_t fix1(uint32_t v) const {       // (-p, p) -> [0, p)
    const uint32_t y = v + p;
    return v < y ? v : y;
  }
  template <int K>
  __device__ __forceinline__ uint32_t canon(V<K> x) const { return fix1(reduce<1>(x).v); }
  template <int KOUT, int K>
  __device__ __forceinline__ uint32_t out(V<K> x) const { return reduce<KOUT>(x).v; }
  // The column passes around a fused row kernel run on the unsigned class: rows arrive below 2p, and a
  // value left for the inverse column pass (any unsigned 32-bit word) gets HS*p added to be non-negative
  // (2*HS*p < 2^32 in both classes).
  static constexpr int KROW = 2;
  __device__ __forceinline__ uint32_t raw_out(uint32_t v) const { return v + (uint32_t)HS * p; }
};
template <>
struct Arith<1> : SignedArith<true> { __device__ __forceinline__ Arith(const PrimeCtx &c) : SignedArith<true>(c) {} };
template <>
struct Arith<0> : SignedArith<false> { __device__ __forceinline__ Arith(const PrimeCtx &c) : SignedArith<false>(c) {} };
template <>
struct Arith<16> : SignedArith<false, 16> { __device__ __forceinline__ Arith(const PrimeCtx &c) : SignedArith<false, 16>(c) {} };
template <>
struct Arith<17> : SignedArith<true, 16> { __device__ __forceinline__ Arith(const PrimeCtx &c) : SignedArith<true, 16>(c) {} };   // (512-point kernel: 2 % faster)
template <class A, class = void>
struct dit_pre_of { static constexpr bool value = false; };
template <class A>
struct dit_pre_of<A, std::enable_if_t<A::DIT_PRE>> { static constexpr bool value = true; };

// ------------------------------------------------------------------------------------
// Radix-8 / 4 / 2 units on registers: stages at point distances 4D, 2D, D.
// w[m] = tw[lo + m*D]; UNIT_D1 marks D == 1 where the twiddles of index 0 equal 1
// (w[1], w[2], w[4]) and their products are skipped.  x[] holds values below KIN*p on entry and
// below KOUT*p on exit (KOUT = KANY: no reduction, the consumer multiplies by a constant).
// ------------------------------------------------------------------------------------
// ZHI: the upper four inputs are known to be zero (the zero-padded half of a linear product), so
// the distance-4D stage degenerates to (a, a*w) and x[4..7] are not read.
template <int H, bool UNIT_D1, bool ZHI = false, int KIN = Arith<H>::KF, int KOUT = Arith<H>::KF>
__device__ __forceinline__ void r8_dif(uint32_t (&x)[8], const uint32_t (&w)[8], const Arith<H> &M) {
  const V<KIN> i0{x[0]}, i1{x[1]}, i2{x[2]}, i3{x[3]}, i4{x[4]}, i5{x[5]}, i6{x[6]}, i7{x[7]};
  // distance 4D
  const auto p04 = [&] {
    if constexpr (ZHI) return mkpair(i0, M.mul(i0, w[4]));
    else if constexpr (UNIT_D1) return M.bf1(i0, i4);
    else return M.dif(i0, i4, w[4]);
  }();
  const auto p15 = [&] { if constexpr (ZHI) return mkpair(i1, M.mul(i1, w[5])); else return M.dif(i1, i5, w[5]); }();
  const auto p26 = [&] { if constexpr (ZHI) return mkpair(i2, M.mul(i2, w[6])); else return M.dif(i2, i6, w[6]); }();
  const auto p37 = [&] { if constexpr (ZHI) return mkpair(i3, M.mul(i3, w[7])); else return M.dif(i3, i7, w[7]); }();
  // distance 2D
  const auto q02 = [&] { if constexpr (UNIT_D1) return M.bf1(p04.a, p26.a); else return M.dif(p04.a, p26.a, w[2]); }();
  const auto q13 = M.dif(p15.a, p37.a, w[3]);
  const auto q46 = [&] { if constexpr (UNIT_D1) return M.bf1(p04.b, p26.b); else return M.dif(p04.b, p26.b, w[2]); }();
  const auto q57 = M.dif(p15.b, p37.b, w[3]);
  // distance D
  const auto r01 = [&] { if constexpr (UNIT_D1) return M.bf1(q02.a, q13.a); else return M.dif(q02.a, q13.a, w[1]); }();
  const auto r23 = [&] { if constexpr (UNIT_D1) return M.bf1(q02.b, q13.b); else return M.dif(q02.b, q13.b, w[1]); }();
  const auto r45 = [&] { if constexpr (UNIT_D1) return M.bf1(q46.a, q57.a); else return M.dif(q46.a, q57.a, w[1]); }();
  const auto r67 = [&] { if constexpr (UNIT_D1) return M.bf1(q46.b, q57.b); else return M.dif(q46.b, q57.b, w[1]); }();
  x[0] = M.template out<KOUT>(r01.a); x[1] = M.template out<KOUT>(r01.b); x[2] = M.template out<KOUT>(r23.a); x[3] = M.template out<KOUT>(r23.b);
  x[4] = M.template out<KOUT>(r45.a); x[5] = M.template out<KOUT>(r45.b); x[6] = M.template out<KOUT>(r67.a); x[7] = M.template out<KOUT>(r67.b);
}

template <int H, bool UNIT_D1, int KIN = Arith<H>::KI, int KOUT = Arith<H>::KI>
__device__ __forceinline__ void r8_dit(uint32_t (&x)[8], const uint32_t (&w)[8], const Arith<H> &M) {
  // Every output is x[0] plus or minus three products, so with headroom to spare reducing x[0] alone bounds them all by
  // four multiples (the other inputs are multiplied before they are added to anything).
  constexpr bool PRE = dit_pre_of<Arith<H>>::value && !UNIT_D1 && KOUT >= 4 && KIN + 3 > KOUT;
  const V<KIN> i1{x[1]}, i2{x[2]}, i3{x[3]}, i4{x[4]}, i5{x[5]}, i6{x[6]}, i7{x[7]};
  const auto i0 = [&] { if constexpr (PRE) return M.template reduce<KOUT - 3>(V<KIN>{x[0]}); else return V<KIN>{x[0]}; }();
  // distance D
  const auto p01 = [&] { if constexpr (UNIT_D1) return M.bf1(i0, i1); else return M.dit(i0, i1, w[1]); }();
  const auto p23 = [&] { if constexpr (UNIT_D1) return M.bf1(i2, i3); else return M.dit(i2, i3, w[1]); }();
  const auto p45 = [&] { if constexpr (UNIT_D1) return M.bf1(i4, i5); else return M.dit(i4, i5, w[1]); }();
  const auto p67 = [&] { if constexpr (UNIT_D1) return M.bf1(i6, i7); else return M.dit(i6, i7, w[1]); }();
  // distance 2D
  const auto q02 = [&] { if constexpr (UNIT_D1) return M.bf1(p01.a, p23.a); else return M.dit(p01.a, p23.a, w[2]); }();
  const auto q13 = M.dit(p01.b, p23.b, w[3]);
  const auto q46 = [&] { if constexpr (UNIT_D1) return M.bf1(p45.a, p67.a); else return M.dit(p45.a, p67.a, w[2]); }();
  const auto q57 = M.dit(p45.b, p67.b, w[3]);
  // distance 4D
  const auto r04 = [&] { if constexpr (UNIT_D1) return M.bf1(q02.a, q46.a); else return M.dit(q02.a, q46.a, w[4]); }();
  const auto r15 = M.dit(q13.a, q57.a, w[5]);
  const auto r26 = M.dit(q02.b, q46.b, w[6]);
  const auto r37 = M.dit(q13.b, q57.b, w[7]);
  x[0] = M.template out<KOUT>(r04.a); x[4] = M.template out<KOUT>(r04.b); x[1] = M.template out<KOUT>(r15.a); x[5] = M.template out<KOUT>(r15.b);
  x[2] = M.template out<KOUT>(r26.a); x[6] = M.template out<KOUT>(r26.b); x[3] = M.template out<KOUT>(r37.a); x[7] = M.template out<KOUT>(r37.b);
}

// Radix-16 units: one more stage at distance 8D around two radix-8 units (w[8 + j] = tw[lo + (8 + j) D] are
// its twiddles, the two halves share w[1 .. 7]).  Used where 16 points of a thread meet anyway: the outermost and the
// distance-1 round of rows of 2^13 and 2^14 points, which then need four trips through LDS instead of five.
template <int H, bool UNIT_D1, int KIN, int KOUT>
__device__ __forceinline__ void r16_dif(uint32_t (&x)[16], const uint32_t (&w)[16], const Arith<H> &M) {
  using PR = decltype(M.dif(V<KIN>{0u}, V<KIN>{0u}, 0u));
  constexpr int KLO = decltype(PR::a)::bound, KHI = decltype(PR::b)::bound;
  uint32_t lo[8], hi[8], w8[8];
  if constexpr (UNIT_D1) {
    const auto pr = M.bf1(V<KIN>{x[0]}, V<KIN>{x[8]});
    static_assert(decltype(pr.a)::bound <= KLO, "sum bound");
    lo[0] = pr.a.v; hi[0] = M.template reduce<KHI>(pr.b).v;
  } else {
    const auto pr = M.dif(V<KIN>{x[0]}, V<KIN>{x[8]}, w[8]);
    lo[0] = pr.a.v; hi[0] = pr.b.v;
  }
#pragma unroll
  for (int j = 1; j < 8; ++j) {
    const auto pr = M.dif(V<KIN>{x[j]}, V<KIN>{x[j + 8]}, w[8 + j]);
    lo[j] = pr.a.v; hi[j] = pr.b.v;
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) w8[m] = w[m];
  r8_dif<H, UNIT_D1, false, KLO, KOUT>(lo, w8, M);
  r8_dif<H, UNIT_D1, false, KHI, KOUT>(hi, w8, M);
#pragma unroll
  for (int j = 0; j < 8; ++j) { x[j] = lo[j]; x[j + 8] = hi[j]; }
}
template <int H, bool UNIT_D1, int KIN, int KOUT>
__device__ __forceinline__ void r16_dit(uint32_t (&x)[16], const uint32_t (&w)[16], const Arith<H> &M) {
  // bound left by the two radix-8 halves: one below KOUT where the class can hold it there for free, so that the last
  // stage's sums need no reduction
  constexpr int KI = (dit_pre_of<Arith<H>>::value && KOUT > 4 && KOUT - 1 < Arith<H>::KI) ? KOUT - 1 : Arith<H>::KI;
  uint32_t lo[8], hi[8], w8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { lo[j] = x[j]; hi[j] = x[j + 8]; w8[j] = w[j]; }
  r8_dit<H, UNIT_D1, KIN, KI>(lo, w8, M);
  r8_dit<H, UNIT_D1, KIN, KI>(hi, w8, M);
  if constexpr (UNIT_D1) {
    const auto pr = M.bf1(V<KI>{lo[0]}, V<KI>{hi[0]});
    x[0] = M.template out<KOUT>(pr.a); x[8] = M.template out<KOUT>(pr.b);
  } else {
    const auto pr = M.dit(V<KI>{lo[0]}, V<KI>{hi[0]}, w[8]);
    x[0] = M.template out<KOUT>(pr.a); x[8] = M.template out<KOUT>(pr.b);
  }
#pragma unroll
  for (int j = 1; j < 8; ++j) {
    const auto pr = M.dit(V<KI>{lo[j]}, V<KI>{hi[j]}, w[8 + j]);
    x[j] = M.template out<KOUT>(pr.a); x[j + 8] = M.template out<KOUT>(pr.b);
  }
}

template <int H, bool UNIT_D1, bool INV, int KIN, int KOUT>
__device__ __forceinline__ void r4_unit(uint32_t (&x)[4], const uint32_t (&w)[4], const Arith<H> &M) {
  const V<KIN> i0{x[0]}, i1{x[1]}, i2{x[2]}, i3{x[3]};
  if constexpr (!INV) {
    const auto p02 = [&] { if constexpr (UNIT_D1) return M.bf1(i0, i2); else return M.dif(i0, i2, w[2]); }();
    const auto p13 = M.dif(i1, i3, w[3]);
    const auto q01 = [&] { if constexpr (UNIT_D1) return M.bf1(p02.a, p13.a); else return M.dif(p02.a, p13.a, w[1]); }();
    const auto q23 = [&] { if constexpr (UNIT_D1) return M.bf1(p02.b, p13.b); else return M.dif(p02.b, p13.b, w[1]); }();
    x[0] = M.template out<KOUT>(q01.a); x[1] = M.template out<KOUT>(q01.b); x[2] = M.template out<KOUT>(q23.a); x[3] = M.template out<KOUT>(q23.b);
  } else {
    const auto p01 = [&] { if constexpr (UNIT_D1) return M.bf1(i0, i1); else return M.dit(i0, i1, w[1]); }();
    const auto p23 = [&] { if constexpr (UNIT_D1) return M.bf1(i2, i3); else return M.dit(i2, i3, w[1]); }();
    const auto q02 = [&] { if constexpr (UNIT_D1) return M.bf1(p01.a, p23.a); else return M.dit(p01.a, p23.a, w[2]); }();
    const auto q13 = M.dit(p01.b, p23.b, w[3]);
    x[0] = M.template out<KOUT>(q02.a); x[2] = M.template out<KOUT>(q02.b); x[1] = M.template out<KOUT>(q13.a); x[3] = M.template out<KOUT>(q13.b);
  }
}

template <int H, bool UNIT_D1, bool INV, int KIN, int KOUT>
__device__ __forceinline__ void r2_unit(uint32_t (&x)[2], const uint32_t (&w)[2], const Arith<H> &M) {
  const V<KIN> i0{x[0]}, i1{x[1]};
  if constexpr (UNIT_D1) {
    const auto r = M.bf1(i0, i1);
    x[0] = M.template out<KOUT>(r.a); x[1] = M.template out<KOUT>(r.b);
  } else if constexpr (!INV) {
    const auto r = M.dif(i0, i1, w[1]);
    x[0] = M.template out<KOUT>(r.a); x[1] = M.template out<KOUT>(r.b);
  } else {
    const auto r = M.dit(i0, i1, w[1]);
    x[0] = M.template out<KOUT>(r.a); x[1] = M.template out<KOUT>(r.b);
  }
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
// ROLLED: keep the loop over a thread's units rolled (kernels whose other live registers leave no
// room for two units in flight).
template <class L, int R, int LOGD, bool INV, int H, int NV, bool ZHI, int KIN, int KOUT, bool ROLLED = false>
__device__ __forceinline__ void ntt_round(uint32_t *const (&s)[NV], const uint32_t *tw, const Arith<H> &M) {
  constexpr int D = 1 << LOGD;
  constexpr int LOG_UNITS = L::LOGT - R;  // log2(units per transform)
  constexpr int TOTAL = L::NT << LOG_UNITS;
  constexpr int NP = 1 << R;
  auto unit = [&](int g) {
    int c, u;
    L::template split<LOG_UNITS>(g, c, u);
    const int lo = u & (D - 1), hi = u >> LOGD;
    const int base = (hi << (LOGD + R)) + lo;
    const int a0 = L::addr(c, base);
    uint32_t w[NP];
    w[0] = 0;
#pragma unroll
    for (int m = 1; m < NP; ++m) w[m] = tw[lo + m * D];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      uint32_t x[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q) x[q] = (ZHI && q >= NP / 2) ? 0u : s[v][a0 + L::padj(q * D)];
      if constexpr (R == 4) {
        static_assert(!ZHI, "no zero-half variant of the radix-16 unit");
        if constexpr (!INV) r16_dif<H, D == 1, KIN, KOUT>(x, w, M); else r16_dit<H, D == 1, KIN, KOUT>(x, w, M);
      } else if constexpr (R == 3) {
        if constexpr (!INV) r8_dif<H, D == 1, ZHI, KIN, KOUT>(x, w, M); else r8_dit<H, D == 1, KIN, KOUT>(x, w, M);
      } else if constexpr (R == 2) {
        r4_unit<H, D == 1, INV, KIN, KOUT>(x, w, M);
      } else {
        r2_unit<H, D == 1, INV, KIN, KOUT>(x, w, M);
      }
#pragma unroll
      for (int q = 0; q < NP; ++q) s[v][a0 + L::padj(q * D)] = x[q];
    }
  };
  if constexpr (ROLLED) {
#pragma unroll 1
    for (int g = threadIdx.x; g < TOTAL; g += L::NTHR) unit(g);
  } else {
    for (int g = threadIdx.x; g < TOTAL; g += L::NTHR) unit(g);
  }
}

// Forward (DIF) rounds: largest distance first.  REM = log2 of the sub-transform still to do.
// KIN bounds the values found in the tile before the first round, KOUT the values left by the
// last one; between rounds the forward contract KF holds.  ZTOP: upper half of the input is zero.
template <class L, int REM, int H, int NV, bool ZTOP, int KIN, int KOUT>
__device__ __forceinline__ void dif_rounds(uint32_t *const (&s)[NV], const uint32_t *tw, const Arith<H> &M) {
  constexpr int KF = Arith<H>::KF;
  constexpr int kin = (REM == L::LOGT) ? KIN : KF;
  constexpr bool zt = ZTOP && REM == L::LOGT;
  if constexpr (REM > 3) {
    ntt_round<L, 3, REM - 3, false, H, NV, zt, kin, KF>(s, tw, M);
    __syncthreads();
    dif_rounds<L, REM - 3, H, NV, ZTOP, KIN, KOUT>(s, tw, M);
  } else {
    ntt_round<L, REM, 0, false, H, NV, zt, kin, KOUT>(s, tw, M);
  }
}

// Inverse (DIT) rounds: the same partition walked from the smallest distance up.
template <class L, int REM, int H, int NV, int KIN, int KOUT>
__device__ __forceinline__ void dit_rounds(uint32_t *const (&s)[NV], const uint32_t *tw, const Arith<H> &M) {
  constexpr int KI = Arith<H>::KI;
  constexpr int kout = (REM == L::LOGT) ? KOUT : KI;
  if constexpr (REM > 3) {
    dit_rounds<L, REM - 3, H, NV, KIN, KOUT>(s, tw, M);
    __syncthreads();
    ntt_round<L, 3, REM - 3, true, H, NV, false, KI, kout>(s, tw, M);
  } else {
    ntt_round<L, REM, 0, true, H, NV, false, KIN, kout>(s, tw, M);
  }
}

// Fused product on a tile: the last forward round, the pointwise product and the first inverse round
// all act on the same 2^RL points at distance 1 of a thread's units, so they run on registers
// between the LDS-exchanged rounds (saves two barriers and three LDS passes per product).
template <class L, int REM, int H>
__device__ __forceinline__ void dif_head(uint32_t *const (&s)[2], const uint32_t *tw, const Arith<H> &M) {
  constexpr int KF = Arith<H>::KF;
  if constexpr (REM > 3) {
    ntt_round<L, 3, REM - 3, false, H, 2, false, KF, KF>(s, tw, M);   // inputs below 2p = KF*p
    __syncthreads();
    dif_head<L, REM - 3, H>(s, tw, M);
  }
}
template <class L, int REM, int H>
__device__ __forceinline__ void dit_tail(uint32_t *const (&s)[1], const uint32_t *tw, const Arith<H> &M) {
  constexpr int KI = Arith<H>::KI;
  if constexpr (REM > 3) {
    dit_tail<L, REM - 3, H>(s, tw, M);
    __syncthreads();
    ntt_round<L, 3, REM - 3, true, H, 1, false, KI, KI>(s, tw, M);
  }
}
template <class L, int RL, int H>
__device__ __forceinline__ void edge_product(uint32_t *sa, const uint32_t *sb, const uint32_t *twf, const uint32_t *twi,
                                             const Arith<H> &M, uint32_t scale) {
  constexpr int NP = 1 << RL;
  constexpr int LOG_UNITS = L::LOGT - RL;
  constexpr int TOTAL = L::NT << LOG_UNITS;
  constexpr int KF = Arith<H>::KF, KI = Arith<H>::KI, KP = Arith<H>::KP;
  uint32_t wf[NP], wi[NP];
  wf[0] = wi[0] = 0;
#pragma unroll
  for (int m = 1; m < NP; ++m) { wf[m] = twf[m]; wi[m] = twi[m]; }
  for (int g = threadIdx.x; g < TOTAL; g += L::NTHR) {
    int c, u;
    L::template split<LOG_UNITS>(g, c, u);
    const int a0 = L::addr(c, u << RL);
    uint32_t xa[NP], xb[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) { xa[q] = sa[a0 + L::padj(q)]; xb[q] = sb[a0 + L::padj(q)]; }
    if constexpr (RL == 3) { r8_dif<H, true, false, KF, KP>(xa, wf, M); r8_dif<H, true, false, KF, KP>(xb, wf, M); }
    else if constexpr (RL == 2) { r4_unit<H, true, false, KF, KP>(xa, wf, M); r4_unit<H, true, false, KF, KP>(xb, wf, M); }
    else { r2_unit<H, true, false, KF, KP>(xa, wf, M); r2_unit<H, true, false, KF, KP>(xb, wf, M); }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      auto pr = M.mul2(V<KP>{xa[q]}, V<KP>{xb[q]});
      if (scale) pr = M.mul(pr, scale);
      xa[q] = pr.v;
    }
    constexpr int KM = Arith<H>::KM;
    if constexpr (RL == 3) r8_dit<H, true, KM, KI>(xa, wi, M);
    else if constexpr (RL == 2) r4_unit<H, true, true, KM, KI>(xa, wi, M);
    else r2_unit<H, true, true, KM, KI>(xa, wi, M);
#pragma unroll
    for (int q = 0; q < NP; ++q) sa[a0 + L::padj(q)] = xa[q];
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
  // negacyclic products (TWIST kernels, rows = whole vectors): inputs are multiplied by psi_in[i] = psi^i on
  // the way in and outputs by psi_out[i] = psi^-i on the way out (Montgomery form, T words each)
  const uint32_t *psi_in, *psi_out;
};

template <class L, int H = 2, bool TW = false>
__device__ __forceinline__ void row_load(uint32_t *s, const uint32_t *in, long long row0, long long nrows,
                                         const uint32_t *psi = nullptr, const Arith<H> *M = nullptr) {
  constexpr int T = L::T;
  constexpr int KANY = Arith<H>::KANY;
  if constexpr (T >= 4) {
    constexpr int NV4 = L::NT * T / 4;
    for (int v = threadIdx.x; v < NV4; v += L::NTHR) {
      const int c = (v * 4) >> L::LOGT, j = (v * 4) & (T - 1);
      uint4 x = make_uint4(0, 0, 0, 0);
      if (row0 + c < nrows) {
        x = *reinterpret_cast<const uint4 *>(in + (row0 + c) * T + j);
        if constexpr (TW) {
          const uint4 f = *reinterpret_cast<const uint4 *>(psi + j);
          x.x = M->mul(V<KANY>{x.x}, f.x).v; x.y = M->mul(V<KANY>{x.y}, f.y).v;
          x.z = M->mul(V<KANY>{x.z}, f.z).v; x.w = M->mul(V<KANY>{x.w}, f.w).v;
        }
      }
      *reinterpret_cast<uint4 *>(s + L::addr(c, j)) = x;
    }
  } else {
    for (int v = threadIdx.x; v < L::NT * T; v += L::NTHR) {
      const int c = v / T, j = v % T;
      uint32_t x = (row0 + c < nrows) ? in[(row0 + c) * T + j] : 0u;
      if constexpr (TW) x = M->mul(V<KANY>{x}, psi[j]).v;
      s[L::addr(c, j)] = x;
    }
  }
}

// TW: stored values are first multiplied by psi[j] (and then always fully reduced)
template <class L, int H, bool TW = false>
__device__ __forceinline__ void row_store(const uint32_t *s, uint32_t *out, long long row0, long long nrows,
                                          const Arith<H> &M, bool canon, const uint32_t *psi = nullptr) {
  // stored values are below KI*p (the last round of either direction is asked for that)
  constexpr int T = L::T;
  constexpr int KI = Arith<H>::KI;
  if constexpr (T >= 4) {
    constexpr int NV4 = L::NT * T / 4;
    for (int v = threadIdx.x; v < NV4; v += L::NTHR) {
      const int c = (v * 4) >> L::LOGT, j = (v * 4) & (T - 1);
      if (row0 + c < nrows) {
        uint4 x = *reinterpret_cast<const uint4 *>(s + L::addr(c, j));
        if constexpr (TW) {
          const uint4 f = *reinterpret_cast<const uint4 *>(psi + j);
          x.x = M.canon(M.mul(V<KI>{x.x}, f.x)); x.y = M.canon(M.mul(V<KI>{x.y}, f.y));
          x.z = M.canon(M.mul(V<KI>{x.z}, f.z)); x.w = M.canon(M.mul(V<KI>{x.w}, f.w));
        } else if (canon) { x.x = M.canon(V<KI>{x.x}); x.y = M.canon(V<KI>{x.y}); x.z = M.canon(V<KI>{x.z}); x.w = M.canon(V<KI>{x.w}); }
        *reinterpret_cast<uint4 *>(out + (row0 + c) * T + j) = x;
      }
    }
  } else {
    for (int v = threadIdx.x; v < L::NT * T; v += L::NTHR) {
      const int c = v / T, j = v % T;
      if (row0 + c < nrows) {
        uint32_t x = s[L::addr(c, j)];
        if constexpr (TW) x = M.canon(M.mul(V<KI>{x}, psi[j]));
        else if (canon) x = M.canon(V<KI>{x});
        out[(row0 + c) * T + j] = x;
      }
    }
  }
}

template <int LOGT, int MODE, int H, bool TWIST = false>
__global__ void __launch_bounds__(WG) row_kernel(RowArgs A) {
  using L = RowLayout<LOGT>;
  constexpr int T = L::T;
  constexpr int NVEC = (MODE == ROW_MUL) ? 2 : 1;
  __shared__ __attribute__((aligned(16))) uint32_t lds[NVEC * L::WORDS + (MODE == ROW_MUL ? 2 : 1) * T];
  uint32_t *sa = lds, *sb = lds + (NVEC - 1) * L::WORDS;
  uint32_t *twf = lds + NVEC * L::WORDS;
  uint32_t *twi = twf + (MODE == ROW_MUL ? T : 0);
  const Arith<H> M(A.ctx);

  if (MODE != ROW_INV) for (int i = threadIdx.x; i < T; i += WG) twf[i] = A.tw_fwd[i];
  if (MODE != ROW_FWD) for (int i = threadIdx.x; i < T; i += WG) twi[i] = A.tw_inv[i];

  for (long long row0 = (long long)blockIdx.x * L::NT; row0 < A.nrows; row0 += (long long)gridDim.x * L::NT) {
    __syncthreads();  // previous tile fully stored before LDS is overwritten
    row_load<L, H, TWIST>(sa, A.in0, row0, A.nrows, A.psi_in, &M);
    if (MODE == ROW_MUL) row_load<L, H, TWIST>(sb, A.in1, row0, A.nrows, A.psi_in, &M);
    __syncthreads();
    constexpr int KFc = Arith<H>::KF, KIc = Arith<H>::KI;
    if (MODE == ROW_FWD) {         // inputs canonical or fresh from a constant multiply (< 2p)
      uint32_t *const v[1] = {sa};
      dif_rounds<L, LOGT, H, 1, false, KFc, KIc>(v, twf, M);
    } else if (MODE == ROW_INV) {  // inputs of unknown history: the widest inverse bound
      uint32_t *const v[1] = {sa};
      dit_rounds<L, LOGT, H, 1, KIc, KIc>(v, twi, M);
    } else {
      constexpr int RL = (LOGT - 1) % 3 + 1;    // stages of the distance-1 round
      uint32_t *const v2[2] = {sa, sb};
      dif_head<L, LOGT, H>(v2, twf, M);
      edge_product<L, RL, H>(sa, sb, twf, twi, M, A.scale);   // + pointwise (and the optional n^{-1} R^2 factor)
      if constexpr (LOGT > 3) __syncthreads();
      uint32_t *const v[1] = {sa};
      dit_tail<L, LOGT, H>(v, twi, M);
    }
    __syncthreads();
    row_store<L, H, TWIST>(sa, A.out, row0, A.nrows, M, A.canon_out != 0, A.psi_out);
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

#ifndef NTTX_ROW9_WAVES
#define NTTX_ROW9_WAVES 4
#endif
// CANON: outputs fully reduced (single-pass products); otherwise they are left unreduced for the
// inverse column pass, which multiplies them by its four-step constants first.
// TWIST (with CANON): negacyclic product of 512-point vectors, the psi factors applied in registers.
// PROBE (measurement builds of the same kernel, results unchanged): what moving a lane's 8 values between lanes costs
// here, by either mechanism.  1: one more trip of both vectors through the wave-private LDS buffer (8 ds_write_b32 +
// 8 ds_read_b32 per vector, what each exchange between rounds costs now).  2: 24 DPP moves
// per vector (8 registers x the three lane distances 8, 16, 32 that a shuffle-based round 2 would cross, at one move
// per register and distance, its minimum), issued as pairs of row rotations by 8, i.e. the identity.
template <int H, bool CANON, bool TWIST = false, int PROBE = 0>
__global__ void __launch_bounds__(WG, NTTX_ROW9_WAVES) row_mul9_kernel(RowArgs A) {
  static_assert(!TWIST || CANON, "the twisted product is a single-pass product");
  __shared__ __attribute__((aligned(16))) uint32_t lds[(WG / 64) * 2 * ROW9_PAD];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: row addresses stay in SGPRs
  uint32_t *sa = lds + wave * 2 * ROW9_PAD, *sb = sa + ROW9_PAD;
  const Arith<H> M(A.ctx);
  constexpr int KF = Arith<H>::KF, KI = Arith<H>::KI, KP = Arith<H>::KP, KANY = Arith<H>::KANY;
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
  // na/nb: the row being loaded; round 1 runs in place on them, then they are refilled for the next row
  uint32_t na[8], nb[8];
  if (row < A.nrows) {
    const uint32_t *pa = A.in0 + row * 512, *pb = A.in1 + row * 512;
#pragma unroll
    for (int q = 0; q < 8; ++q) { na[q] = pa[lane + 64 * q]; nb[q] = pb[lane + 64 * q]; }
  }
  for (; row < A.nrows; row += nwaves) {
    if constexpr (TWIST) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint32_t f = A.psi_in[lane + 64 * q];
        na[q] = M.mul(V<KANY>{na[q]}, f).v; nb[q] = M.mul(V<KANY>{nb[q]}, f).v;
      }
    }
    // forward round 1 (distances 256, 128, 64)
    r8_dif<H, false, false, Arith<H>::KROW, KF>(na, wf1, M);
    r8_dif<H, false, false, Arith<H>::KROW, KF>(nb, wf1, M);
#pragma unroll
    for (int q = 0; q < 8; ++q) { sa[o1 + 72 * q] = na[q]; sb[o1 + 72 * q] = nb[q]; }
    const long long nrow = row + nwaves;
    if (nrow < A.nrows) {
      const uint32_t *pa = A.in0 + nrow * 512, *pb = A.in1 + nrow * 512;
#pragma unroll
      for (int q = 0; q < 8; ++q) { na[q] = pa[lane + 64 * q]; nb[q] = pb[lane + 64 * q]; }
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = sa[o2 + 8 * j]; b[j] = sb[o2 + 8 * j]; }
    if constexpr (PROBE == 1) {
      // (through the round-1 slots of the lane, whose contents have been consumed; a memory clobber keeps the compiler
      // from forwarding the stored values)
#pragma unroll
      for (int j = 0; j < 8; ++j) { sa[o1 + 72 * j] = a[j]; sb[o1 + 72 * j] = b[j]; }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[j] = sa[o1 + 72 * j]; b[j] = sb[o1 + 72 * j]; }
      asm volatile("" ::: "memory");
    } else if constexpr (PROBE == 2) {
      // row_ror:8 (DPP control 0x128) twice is the identity on a row of 16 lanes: 3 such pairs on registers 0..3 and
      // ... 24 moves per vector in all, every register rotated an even number of times
      auto ror8 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); };
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a[j] = ror8(ror8(a[j])); b[j] = ror8(ror8(b[j]));
        if (j < 4) { a[j] = ror8(ror8(a[j])); b[j] = ror8(ror8(b[j])); }
      }
    }
    // forward round 2 (32, 16, 8)
    r8_dif<H, false, false, KF, KF>(a, wf2, M);
    r8_dif<H, false, false, KF, KF>(b, wf2, M);
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
    r8_dif<H, true, false, KF, KP>(a, wf3, M);
    r8_dif<H, true, false, KF, KP>(b, wf3, M);
    if (A.scale) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = M.mul(M.mul2(V<KP>{a[j]}, V<KP>{b[j]}), A.scale).v;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = M.mul2(V<KP>{a[j]}, V<KP>{b[j]}).v;
    }
    r8_dit<H, true, Arith<H>::KM, KI>(a, wi3, M);
    *reinterpret_cast<uint4 *>(sa + o3) = make_uint4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<uint4 *>(sa + o3 + 4) = make_uint4(a[4], a[5], a[6], a[7]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = sa[o2 + 8 * j];
    // inverse round 2 (8, 16, 32)
    r8_dit<H, false, KI, KI>(a, wi2, M);
#pragma unroll
    for (int j = 0; j < 8; ++j) sa[o2 + 8 * j] = a[j];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = sa[o1 + 72 * q];
    // inverse round 3 (64, 128, 256)
    r8_dit<H, false, KI, ((CANON && !TWIST) ? 1 : KANY)>(a, wi1, M);
    uint32_t *po = A.out + row * 512;
    if constexpr (TWIST) {
#pragma unroll
      for (int q = 0; q < 8; ++q) po[lane + 64 * q] = M.canon(M.mul(V<KANY>{a[q]}, A.psi_out[lane + 64 * q]));
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) po[lane + 64 * q] = CANON ? M.fix1(a[q]) : M.raw_out(a[q]);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------
// Long rows (2^10 .. 2^14 points): one row per workgroup of 128 .. 1024 threads, so lengths up to
// 2^23 need only one column pass around the row pass (two trips through HBM per transform
// instead of three) and lengths up to 2^14 a single pass.  The fused product sends a and then b through ONE LDS buffer: the outermost
// forward round runs on registers straight from global memory, the distance-1 round of both
// directions and the pointwise product run on registers too (a's values wait there while b uses
// the buffer), and the outermost inverse round stores to global memory.  Stage twiddles of the
// rounds at distance <= 64 come from an LDS copy of the first 512 table words, the outer rounds
// read the table through the caches.
// ------------------------------------------------------------------------------------
// WIDE: sixteen points per thread (the fused product, whose outermost round is a radix-16 unit per thread)
template <int LOGT_, bool WIDE = false>
struct LongRowLayout {
  static constexpr int LOGT = LOGT_;
  static constexpr int T = 1 << LOGT;
  static constexpr int NT = 1;
  static constexpr int NTHR = WIDE ? (T / 16) : (LOGT <= 11) ? (T / 8) : (LOGT <= 13) ? 512 : 1024;
  static constexpr int TP = T + (T >> 6) * 8;
  static constexpr int WORDS = TP;
  __device__ static constexpr int padj(int j) { return j + ((j >> 6) << 3); }
  __device__ static __forceinline__ int addr(int, int j) { return padj(j); }
  template <int LOGU>
  __device__ static __forceinline__ void split(int g, int &c, int &u) { c = 0; u = g; }
};
constexpr int LONG_TWL = 512;   // leading words of each stage table kept in LDS

// forward rounds through LDS between the outermost round and the distance-1 round, stages LOGD_HI - 1 down to LOGD_LO:
// radix-8 units, led by one round of radix-16 units when the number of stages is 1 mod 3
template <class L, int LOGD_HI, int LOGD_LO, int H>
__device__ __forceinline__ void long_head(uint32_t *s, const uint32_t *twg, const uint32_t *twl, const Arith<H> &M) {
  constexpr int KF = Arith<H>::KF;
  static_assert(LOGD_HI >= LOGD_LO && (LOGD_HI - LOGD_LO) % 3 != 2, "rounds of three or four stages");
  if constexpr (LOGD_HI > LOGD_LO) {
    constexpr int R = ((LOGD_HI - LOGD_LO) % 3 == 1) ? 4 : 3;
    constexpr int LOGD = LOGD_HI - R;
    uint32_t *const v[1] = {s};
    ntt_round<L, R, LOGD, false, H, 1, false, KF, KF, true>(v, ((1 << R) << LOGD) <= LONG_TWL ? twl : twg, M);
    __syncthreads();
    long_head<L, LOGD, LOGD_LO, H>(s, twg, twl, M);
  }
}
// the same stages of the inverse, smallest distance first
template <class L, int LOGD_LO, int LOGD_HI, int H>
__device__ __forceinline__ void long_tail(uint32_t *s, const uint32_t *twg, const uint32_t *twl, const Arith<H> &M) {
  constexpr int KI = Arith<H>::KI;
  if constexpr (LOGD_LO < LOGD_HI) {
    constexpr int R = ((LOGD_HI - LOGD_LO) % 3 == 1) ? 4 : 3;
    uint32_t *const v[1] = {s};
    ntt_round<L, R, LOGD_LO, true, H, 1, false, KI, KI, true>(v, ((1 << R) << LOGD_LO) <= LONG_TWL ? twl : twg, M);
    __syncthreads();
    long_tail<L, LOGD_LO + R, LOGD_HI, H>(s, twg, twl, M);
  }
}
// distance-1 round on registers: r[u*NP + q] = point q of the thread's u-th unit
template <class L, int RL, bool INV, int H, int KIN, int KOUT, int PT>
__device__ __forceinline__ void long_edge(uint32_t *s, const uint32_t (&w)[1 << RL], const Arith<H> &M, uint32_t (&r)[PT]) {
  constexpr int NP = 1 << RL, NU = PT / NP;
  static_assert((L::T >> RL) == NU * L::NTHR, "PT points per thread");
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int a0 = L::padj((threadIdx.x + u * L::NTHR) << RL);
    uint32_t x[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) x[q] = INV ? r[u * NP + q] : s[a0 + q];
    if constexpr (RL == 4) { if constexpr (!INV) r16_dif<H, true, KIN, KOUT>(x, w, M); else r16_dit<H, true, KIN, KOUT>(x, w, M); }
    else if constexpr (RL == 3) { if constexpr (!INV) r8_dif<H, true, false, KIN, KOUT>(x, w, M); else r8_dit<H, true, KIN, KOUT>(x, w, M); }
    else if constexpr (RL == 2) r4_unit<H, true, INV, KIN, KOUT>(x, w, M);
    else r2_unit<H, true, INV, KIN, KOUT>(x, w, M);
#pragma unroll
    for (int q = 0; q < NP; ++q) { if (INV) s[a0 + q] = x[q]; else r[u * NP + q] = x[q]; }
  }
}

// TWIST: negacyclic product of whole vectors (single pass, canonical output), psi factors applied in registers.
// Rounds (stages per round; the first runs on registers straight from global memory, the last on registers too):
//   2^10: 3 | 3 3 | 1    2^11: 3 | 3 3 | 2    2^12: 3 | 3 3 | 3    2^13: 4 | 3 3 | 3    2^14: 4 | 3 3 | 4
// From 2^13 points on a thread owns sixteen points and the outermost round is one radix-16 unit per thread (at 2^14 the
// distance-1 round as well), which keeps those rows at three trips through LDS per transform instead of four.  Shorter
// rows measured no faster that way (2^10: -2 %, 2^12: +3 %) and lose parallelism when a launch has few rows
// (2^20 as 512 rows of 2^11: 16.0 against 11.6 us), so they keep eight points per thread.
constexpr bool long_row_wide(int logt) { return logt >= 13; }
template <int LOGT, int H, bool TWIST = false>
__global__ void __launch_bounds__((LongRowLayout<LOGT, long_row_wide(LOGT)>::NTHR)) longrow_mul_kernel(RowArgs A) {
  using L = LongRowLayout<LOGT, long_row_wide(LOGT)>;
  constexpr int T = L::T, NTHR = L::NTHR;
  constexpr int RO = long_row_wide(LOGT) ? 4 : 3;                                          // stages of the outermost round
  constexpr int RL = !long_row_wide(LOGT) ? (LOGT - 1) % 3 + 1 : ((LOGT - RO) % 3 == 0) ? 3 : 4;   // ... of the distance-1 round
  constexpr int NPO = 1 << RO;
  constexpr int D1 = T >> RO, LOGD1 = LOGT - RO;  // outermost round
  constexpr int PT = T / NTHR;                    // points per thread
  constexpr int KF = Arith<H>::KF, KI = Arith<H>::KI, KP = Arith<H>::KP, KM = Arith<H>::KM, KANY = Arith<H>::KANY;
  static_assert(D1 == NTHR && LOGD1 >= 6 && (LOGD1 - RL) % 3 != 2, "row length does not fit this kernel's rounds");
  __shared__ __attribute__((aligned(16))) uint32_t lds[L::WORDS + 2 * LONG_TWL];
  uint32_t *sa = lds, *twf = lds + L::WORDS, *twi = twf + LONG_TWL;
  const Arith<H> M(A.ctx);
  for (int i = threadIdx.x; i < LONG_TWL; i += NTHR) { twf[i] = A.tw_fwd[i]; twi[i] = A.tw_inv[i]; }
  // twiddles of the distance-1 rounds: the same for every unit, read once through the scalar cache
  uint32_t ef[1 << RL], ei[1 << RL];
  ef[0] = ei[0] = 0;
#pragma unroll
  for (int m = 1; m < (1 << RL); ++m) { ef[m] = A.tw_fwd[m]; ei[m] = A.tw_inv[m]; }
  auto outer_dif = [&](uint32_t (&x)[NPO], const uint32_t (&w)[NPO]) {
    if constexpr (RO == 4) r16_dif<H, false, Arith<H>::KROW, KF>(x, w, M); else r8_dif<H, false, false, Arith<H>::KROW, KF>(x, w, M);
  };

  for (long long row = blockIdx.x; row < A.nrows; row += gridDim.x) {
    const uint32_t *pa = A.in0 + row * T, *pb = A.in1 + row * T;
    // The table bases get an opaque scalar zero added in every trip: otherwise the per-thread table addresses (row
    // invariant, 64 bits each) are hoisted out of the row loop and held in up to 60 vector registers, although a
    // workgroup rarely sees a second row; this way they are scalar base + the 32-bit offsets the row data uses too.
    unsigned zero = 0;
    asm volatile("" : "+s"(zero));               // (an offset rather than the pointers: those must stay provably global)
    const uint32_t *gtf = A.tw_fwd + zero, *gti = A.tw_inv + zero, *gpi = A.psi_in + zero, *gpo = A.psi_out + zero;
    // b's outermost-round points are fetched together with a's where the registers allow it; the longest
    // rows fetch them right before their butterflies (a row takes tens of microseconds, so the
    // exposed latency is small next to what spilled registers would cost)
    constexpr bool EARLY_B = (NTHR == 512);
    const unsigned g = threadIdx.x;                // the thread's unit of the outermost rounds: points g + q * D1
    uint32_t xa[NPO], xb[NPO], w1[NPO];
    uint32_t ra[PT], rb[PT];
    w1[0] = 0;
    if (EARLY_B) {
#pragma unroll
      for (int q = 0; q < NPO; ++q) xb[q] = pb[g + q * D1];
    }
    __syncthreads();                               // previous row (and the twiddle copy) settled
    // ---- a: forward
#pragma unroll
    for (int q = 0; q < NPO; ++q) xa[q] = pa[g + q * D1];
#pragma unroll
    for (int m = 1; m < NPO; ++m) w1[m] = gtf[g + m * D1];
    if constexpr (TWIST) {
#pragma unroll
      for (int q = 0; q < NPO; ++q) xa[q] = M.mul(V<KANY>{xa[q]}, gpi[g + q * D1]).v;
    }
    outer_dif(xa, w1);
#pragma unroll
    for (int q = 0; q < NPO; ++q) sa[L::padj(g + q * D1)] = xa[q];
    __syncthreads();
    long_head<L, LOGD1, RL, H>(sa, A.tw_fwd, twf, M);
    long_edge<L, RL, false, H, KF, KP, PT>(sa, ef, M, ra);
    __syncthreads();
    // ---- b: forward
    if (!EARLY_B) {
#pragma unroll
      for (int q = 0; q < NPO; ++q) xb[q] = pb[g + q * D1];
    }
    if constexpr (TWIST) {
#pragma unroll
      for (int q = 0; q < NPO; ++q) xb[q] = M.mul(V<KANY>{xb[q]}, gpi[g + q * D1]).v;
    }
    outer_dif(xb, w1);
#pragma unroll
    for (int q = 0; q < NPO; ++q) sa[L::padj(g + q * D1)] = xb[q];
    __syncthreads();
    long_head<L, LOGD1, RL, H>(sa, A.tw_fwd, twf, M);
    long_edge<L, RL, false, H, KF, KP, PT>(sa, ef, M, rb);
    // ---- pointwise product, inverse
    if (A.scale) {
#pragma unroll
      for (int k = 0; k < PT; ++k) ra[k] = M.mul(M.mul2(V<KP>{ra[k]}, V<KP>{rb[k]}), A.scale).v;
    } else {
#pragma unroll
      for (int k = 0; k < PT; ++k) ra[k] = M.mul2(V<KP>{ra[k]}, V<KP>{rb[k]}).v;
    }
    long_edge<L, RL, true, H, KM, KI, PT>(sa, ei, M, ra);   // each thread rewrites only the points it read
    __syncthreads();
    long_tail<L, RL, LOGD1, H>(sa, A.tw_inv, twi, M);
    uint32_t *po = A.out + row * T;
    uint32_t x[NPO], w[NPO];
    w[0] = 0;
#pragma unroll
    for (int m = 1; m < NPO; ++m) w[m] = gti[g + m * D1];
#pragma unroll
    for (int q = 0; q < NPO; ++q) x[q] = sa[L::padj(g + q * D1)];
    auto outer_dit = [&](auto kout) {
      constexpr int KO = decltype(kout)::value;
      if constexpr (RO == 4) r16_dit<H, false, KI, KO>(x, w, M); else r8_dit<H, false, KI, KO>(x, w, M);
    };
    if constexpr (TWIST) {
      outer_dit(std::integral_constant<int, KANY>{});
#pragma unroll
      for (int q = 0; q < NPO; ++q) po[g + q * D1] = M.canon(M.mul(V<KANY>{x[q]}, gpo[g + q * D1]));
    } else if (A.canon_out) {
      outer_dit(std::integral_constant<int, 1>{});
#pragma unroll
      for (int q = 0; q < NPO; ++q) po[g + q * D1] = M.fix1(x[q]);
    } else {
      outer_dit(std::integral_constant<int, KANY>{});
#pragma unroll
      for (int q = 0; q < NPO; ++q) po[g + q * D1] = M.raw_out(x[q]);
    }
  }
}

// single transforms on long rows (the transform API): every round through LDS
template <int LOGT, int MODE, int H>
__global__ void __launch_bounds__(LongRowLayout<LOGT>::NTHR) longrow_kernel(RowArgs A) {
  using L = LongRowLayout<LOGT>;
  __shared__ __attribute__((aligned(16))) uint32_t lds[L::WORDS];
  const Arith<H> M(A.ctx);
  constexpr int KF = Arith<H>::KF, KI = Arith<H>::KI;
  for (long long row = blockIdx.x; row < A.nrows; row += gridDim.x) {
    __syncthreads();
    row_load<L>(lds, A.in0, row, A.nrows);
    __syncthreads();
    uint32_t *const v[1] = {lds};
    if (MODE == ROW_FWD) dif_rounds<L, LOGT, H, 1, false, KF, KI>(v, A.tw_fwd, M);
    else dit_rounds<L, LOGT, H, 1, KI, KI>(v, A.tw_inv, M);
    __syncthreads();
    row_store<L, H>(lds, A.out, row, A.nrows, M, A.canon_out != 0);
  }
}

// One u64 coefficient modulo a 32-bit prime (the residue split of main.cc:292-303): x = hi*2^32 + lo, so
// x mod m = hi*R + lo mod m with R = 2^32: two Montgomery products, REDC(hi * R^2) and REDC(lo * R), no 64-bit division.
__device__ __forceinline__ uint32_t split_one(uint64_t x, const PrimeCtx &c) {
  const uint32_t h = csub(mont_mul_lazy((uint32_t)(x >> 32), c.r2, c.p, c.npinv), c.p);   // hi * 2^32 mod m
  const uint32_t l = csub(mont_mul_lazy((uint32_t)x, c.one, c.p, c.npinv), c.p);          // lo mod m
  return csub(h + l, c.p);
}
// Coefficients are canonical in the reference's harness (fRead main.cc:78-95 reads residues), but main.cc:300-301
// accepts any 64-bit word; one at or above the caller's modulus is reduced modulo it first, so that the
// exactness bound n*(p-1)^2 of the CRT path holds for every input (a rare, divergent branch).  The reduction is
// Barrett's with mu = floor(2^64 / pmod): q = floor(x * mu / 2^64) is the true quotient or up to two less, so two
// conditional subtractions finish it (a 64-bit division would pull a long software routine, and its registers,
// into every kernel that reads coefficients).
struct CoefMod { uint64_t pmod, mu; };   // pmod == 0: words are taken as they are
__device__ __forceinline__ uint64_t split_in(uint64_t x, const CoefMod &m) {
  if (m.pmod != 0 && x >= m.pmod) {
    x -= __umul64hi(x, m.mu) * m.pmod;
    if (x >= m.pmod) x -= m.pmod;
    if (x >= m.pmod) x -= m.pmod;
  }
  return x;
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
  const uint32_t *fs;     // four-step table, N words (FS2 kernels: the low-part table, T x 2^logC2 words)
  const uint32_t *fsa;    // FS2 kernels: the high-part table, T x S/2^logC2 words
  int logC2;
  PrimeCtx ctx;
  int logS;               // S = 2^logS >= columns per tile
  int canon_out;
  long long nblocks;      // batch * L / N
  long long nvalid;       // ZHI kernels: words of each block that hold data; the rest reads as zero
  const uint32_t *psi;    // TWIST kernels (outermost pass of a negacyclic product): psi^i forward, psi^-i inverse, N words
  // PIECE kernels: every block (a vector of N words) is stored as pieces of 2^piece_log matrix rows; piece q of block
  // r starts at q * piece_stride + r * block_stride (the layout an all-to-all between processes leaves: [source][row][part])
  int piece_log;
  long long piece_stride, block_stride;
  // SRC64 kernels (first forward pass of a CRT product): block r is read from the caller's u64 coefficients
  // in64[z] + r * nvalid (nvalid = coefficients per vector, no padding stored) and reduced modulo this pass's prime
  const uint64_t *in64[2];
  CoefMod cm;
  int vec64;              // every 4-coefficient piece of in64 is 16-byte aligned (nvalid even, bases aligned)
};

// Each thread moves PER = T*COLS/(4*NTHR) 16-byte pieces of a tile; LPR = COLS/4 lanes cover one
// row segment.  The next tile's pieces are fetched into registers before the butterflies of the
// current tile start, so HBM latency overlaps with compute inside one workgroup.
// ZHI (forward only): each block is a zero-padded vector whose first nvalid <= N/2 words hold data.
// Rows from T/2 on are never read (half the input traffic) and the first butterfly stage degenerates.
// FS2: the four-step factor of column s = h 2^logC2 + l is the product fsa[j][h] * fs[j][l] of two
// small tables (one more multiplication per point) instead of an N-word table read: for the long
// outer pass of three-pass lengths, where that table would double the pass's HBM traffic.
// TWIST: the pass is the outermost one of a negacyclic product: forward inputs are multiplied by psi^i as
// they are read, inverse outputs by psi^-i before the canonical store.
// PIECE: 1 = the input (forward pass), 2 = the output (inverse pass) is in the pieces layout of ColArgs, so that rows
// handed over by an all-to-all are transformed where they arrive and results leave in the layout the way back sends.
// SRC64 (zero-half forward passes only): the residue split of a CRT product (main.cc:292-303) happens as the tile is
// read: the input is the caller's [vectors][nvalid] u64 coefficient array and every word is reduced modulo this pass's
// prime on its way into LDS, so no split pass writes residues that this pass would read straight back.
struct Raw64 { ulonglong2 a, b; };
template <int LOGT, bool INV, int H, int LOGC, int NTHR, bool ZHI = false, bool FS2 = false, bool TWIST = false, int PIECE = 0,
          bool SRC64 = false>
__global__ void __launch_bounds__(NTHR, (NTHR == 1024 ? 8 : 1)) col_kernel(ColArgs A) {
  static_assert(!SRC64 || (ZHI && !INV && !FS2 && !TWIST && PIECE == 0), "u64 source: plain zero-half forward passes");
  static_assert(!TWIST || (!ZHI && !FS2), "twisted passes read whole vectors and use the plain four-step table");
  static_assert(PIECE == 0 || (!ZHI && !FS2 && !TWIST && PIECE == (INV ? 2 : 1)), "pieces layout: plain passes, input forward / output inverse");
  using L = ColLayout<LOGT, LOGC, NTHR>;
  constexpr int T = L::T, COLS = L::COLS;
  constexpr int LPR = COLS / 4, LOG_LPR = LOGC - 2;
  constexpr int PER = (T * LPR) / NTHR;
  constexpr int RSTEP = NTHR / LPR;              // rows covered by one piece index
  constexpr int PERL = ZHI ? (PER + 1) / 2 : PER;  // pieces actually loaded (rows below T/2)
  static_assert(PER >= 1 && PER * NTHR == T * LPR, "tile must be a whole number of pieces per thread");
  static_assert(!ZHI || (!INV && (PER % 2 == 0 || PER == 1)), "zero-half variant is forward only");
  __shared__ __attribute__((aligned(16))) uint32_t lds[L::WORDS + T];
  uint32_t *s = lds, *tw = lds + L::WORDS;
  const Arith<H> M(A.ctx);
  const uint32_t *in = A.in[blockIdx.z];
  uint32_t *out = A.out[blockIdx.z];
  const int logS = A.logS;
  const int log_tpb = logS - LOGC;               // log2(tiles per block)
  const long long ntiles = A.nblocks << log_tpb;
  const int j0 = threadIdx.x >> LOG_LPR, c4 = (threadIdx.x & (LPR - 1)) * 4;

  // LAT: narrow tiles serve launches of few tiles, where one tile's latency is the whole kernel: the four-step factors
  // are fetched into registers together with the data (inverse) or before the butterflies (forward) instead of when
  // they are needed, so a tile pays one trip to memory instead of two or three.
  constexpr bool LAT = (LOGC <= 3) && !FS2 && !TWIST;

  auto tile_base = [&](long long t) -> long long {
    const long long blk = t >> log_tpb;
    const long long ct = t & ((1ll << log_tpb) - 1);
    return (blk << (logS + LOGT)) + (ct << LOGC);
  };
  auto fs_base = [&](long long t) -> long long { return (t & ((1ll << log_tpb) - 1)) << LOGC; };
  // the same tile in the pieces layout: block r at r * block_stride, matrix row j in piece j >> piece_log
  auto piece_base = [&](long long t) -> long long {
    return (t >> log_tpb) * A.block_stride + ((t & ((1ll << log_tpb) - 1)) << LOGC);
  };
  auto piece_row = [&](int j) -> long long {
    return (long long)(j >> A.piece_log) * A.piece_stride + ((long long)(j & ((1 << A.piece_log) - 1)) << logS);
  };
  auto fs_load = [&](long long fbase, int j) -> uint4 {
    return *reinterpret_cast<const uint4 *>(A.fs + fbase + ((long long)j << logS) + c4);
  };
  auto fs_apply = [&](uint4 &x, const uint4 &f) {
    x.x = M.mul(V<Arith<H>::KANY>{x.x}, f.x).v; x.y = M.mul(V<Arith<H>::KANY>{x.y}, f.y).v;
    x.z = M.mul(V<Arith<H>::KANY>{x.z}, f.z).v; x.w = M.mul(V<Arith<H>::KANY>{x.w}, f.w).v;
  };
  // x * (four-step factor of row j, columns fbase + c4 .. + 3)
  auto fs_mul = [&](uint4 &x, long long fbase, int j) {
    uint4 f;
    if constexpr (!FS2) {
      f = *reinterpret_cast<const uint4 *>(A.fs + fbase + ((long long)j << logS) + c4);
    } else {
      const long long s0 = fbase + c4;
      f = *reinterpret_cast<const uint4 *>(A.fs + ((long long)j << A.logC2) + (s0 & ((1ll << A.logC2) - 1)));
      const uint32_t fa = A.fsa[((long long)j << (logS - A.logC2)) + (s0 >> A.logC2)];
      x.x = M.mul(V<Arith<H>::KANY>{x.x}, fa).v; x.y = M.mul(V<Arith<H>::KANY>{x.y}, fa).v;
      x.z = M.mul(V<Arith<H>::KANY>{x.z}, fa).v; x.w = M.mul(V<Arith<H>::KANY>{x.w}, fa).v;
    }
    x.x = M.mul(V<Arith<H>::KANY>{x.x}, f.x).v; x.y = M.mul(V<Arith<H>::KANY>{x.y}, f.y).v;
    x.z = M.mul(V<Arith<H>::KANY>{x.z}, f.z).v; x.w = M.mul(V<Arith<H>::KANY>{x.w}, f.w).v;
  };

  // one 16-byte piece of the tile at row j; ZHI masks words at or beyond nvalid of the block
  const bool ragged = ZHI && A.nvalid != ((long long)(T / 2) << logS);
  auto load_piece = [&](long long base, long long t_, int j) -> uint4 {
    if constexpr (PIECE == 1) {
      return *reinterpret_cast<const uint4 *>(in + piece_base(t_) + piece_row(j) + c4);
    } else if constexpr (!ZHI) {
      return *reinterpret_cast<const uint4 *>(in + base + ((long long)j << logS) + c4);
    } else {
      // every loaded piece lies in a row below T/2; when exactly the lower half holds data no word needs masking
      if (!ragged) return *reinterpret_cast<const uint4 *>(in + base + ((long long)j << logS) + c4);
      const long long ct = t_ & ((1ll << log_tpb) - 1);
      const long long w0 = ((long long)j << logS) + (ct << LOGC) + c4;   // word index inside the block
      uint4 x = make_uint4(0, 0, 0, 0);
      if (j < T / 2 && w0 < A.nvalid) {
        x = *reinterpret_cast<const uint4 *>(in + base + ((long long)j << logS) + c4);
        if (w0 + 3 >= A.nvalid) { if (w0 + 1 >= A.nvalid) x.y = 0; if (w0 + 2 >= A.nvalid) x.z = 0; x.w = 0; }
      }
      return x;
    }
  };
  // the same piece from the u64 source: four coefficients of block t_ >> log_tpb, words at or beyond nvalid read as zero
  const uint64_t *in64 = SRC64 ? A.in64[blockIdx.z] : nullptr;
  auto load_piece64 = [&](long long t_, int j) -> Raw64 {
    Raw64 r; r.a = make_ulonglong2(0, 0); r.b = make_ulonglong2(0, 0);
    const long long ct = t_ & ((1ll << log_tpb) - 1);
    const long long w0 = ((long long)j << logS) + (ct << LOGC) + c4;   // word index inside the block
    if (j < T / 2 && w0 < A.nvalid) {
      const uint64_t *q = in64 + (t_ >> log_tpb) * A.nvalid + w0;
      if (A.vec64 && w0 + 3 < A.nvalid) {
        r.a = *reinterpret_cast<const ulonglong2 *>(q); r.b = *reinterpret_cast<const ulonglong2 *>(q + 2);
      } else {
        r.a.x = q[0];
        if (w0 + 1 < A.nvalid) r.a.y = q[1];
        if (w0 + 2 < A.nvalid) r.b.x = q[2];
        if (w0 + 3 < A.nvalid) r.b.y = q[3];
      }
    }
    return r;
  };
  auto reduce64 = [&](const Raw64 &r) -> uint4 {
    return make_uint4(split_one(split_in(r.a.x, A.cm), A.ctx), split_one(split_in(r.a.y, A.cm), A.ctx),
                      split_one(split_in(r.b.x, A.cm), A.ctx), split_one(split_in(r.b.y, A.cm), A.ctx));
  };
  // Tiles narrower than a 128-byte line: the 2, 4 or 8 tiles that share a line go to workgroups 8 block slots apart,
  // which the dispatcher places on the same XCD, so that XCD's L2 merges their partial-line writes and serves the
  // later partial-line reads (a placement heuristic for speed; any placement is correct).
  auto remap = [&](long long t0) -> long long {
    if constexpr (LOGC < 5) {
      constexpr int G = 1 << (5 - LOGC), CH = 8 * G;
      if ((t0 | (CH - 1)) < ntiles) { const long long r = t0 & (CH - 1); return (t0 & ~(long long)(CH - 1)) + (r & 7) * G + (r >> 3); }
    }
    return t0;
  };
  uint4 nx[SRC64 ? 1 : PERL];
  Raw64 nx64[SRC64 ? PERL : 1];
  uint4 fx[(LAT && INV) ? PERL : 1];   // LAT, inverse: the factors of the pieces in nx
  long long t0 = blockIdx.x;
  if (t0 < ntiles) {
    const long long t = remap(t0), base = tile_base(t);
#pragma unroll
    for (int k = 0; k < PERL; ++k) {
      if constexpr (SRC64) nx64[k] = load_piece64(t, j0 + k * RSTEP);
      else nx[k] = load_piece(base, t, j0 + k * RSTEP);
    }
    if constexpr (LAT && INV) {
#pragma unroll
      for (int k = 0; k < PERL; ++k) fx[k] = fs_load(fs_base(t), j0 + k * RSTEP);
    }
  }
  for (int i = threadIdx.x; i < T; i += NTHR) tw[i] = A.tw[i];   // (after the first tile's loads are in flight)
  for (; t0 < ntiles; t0 += gridDim.x) {
    const long long t = remap(t0);
    const long long gbase = tile_base(t), fbase = fs_base(t);
    __syncthreads();   // previous tile's LDS reads are done
#pragma unroll
    for (int k = 0; k < PERL; ++k) {
      uint4 x;
      if constexpr (SRC64) x = reduce64(nx64[k]); else x = nx[k];
      const int j = j0 + k * RSTEP;
      if constexpr (LAT && INV) fs_apply(x, fx[k]);
      else if (INV) fs_mul(x, fbase, j);
      if constexpr (TWIST && !INV) {
        const uint4 f = *reinterpret_cast<const uint4 *>(A.psi + fbase + ((long long)j << logS) + c4);
        x.x = M.mul(V<Arith<H>::KANY>{x.x}, f.x).v; x.y = M.mul(V<Arith<H>::KANY>{x.y}, f.y).v;
        x.z = M.mul(V<Arith<H>::KANY>{x.z}, f.z).v; x.w = M.mul(V<Arith<H>::KANY>{x.w}, f.w).v;
      }
      *reinterpret_cast<uint4 *>(s + L::addr(c4, j)) = x;
    }
    if (t0 + gridDim.x < ntiles) {
      const long long tn = remap(t0 + gridDim.x), base = tile_base(tn);
#pragma unroll
      for (int k = 0; k < PERL; ++k) {
        if constexpr (SRC64) nx64[k] = load_piece64(tn, j0 + k * RSTEP);
        else nx[k] = load_piece(base, tn, j0 + k * RSTEP);
      }
      if constexpr (LAT && INV) {
#pragma unroll
        for (int k = 0; k < PERL; ++k) fx[k] = fs_load(fs_base(tn), j0 + k * RSTEP);
      }
    }
    uint4 fo[(LAT && !INV) ? PER : 1];   // LAT, forward: this tile's output factors, in flight during the butterflies
    if constexpr (LAT && !INV) {
#pragma unroll
      for (int k = 0; k < PER; ++k) fo[k] = fs_load(fbase, j0 + k * RSTEP);
    }
    __syncthreads();
    uint32_t *const vec[1] = {s};
    // forward: inputs are canonical or fresh from a constant multiply (< 2p); outputs go straight into
    // the four-step multiply, so the last round leaves them unreduced.  inverse: inputs were just
    // multiplied (< 2p); outputs are kept below KI*p for the canonical store.
    if constexpr (!INV) dif_rounds<L, LOGT, H, 1, ZHI, Arith<H>::KM, Arith<H>::KANY>(vec, tw, M);
    else dit_rounds<L, LOGT, H, 1, Arith<H>::KM, Arith<H>::KI>(vec, tw, M);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int j = j0 + k * RSTEP;
      uint4 x = *reinterpret_cast<const uint4 *>(s + L::addr(c4, j));
      if constexpr (LAT && !INV) fs_apply(x, fo[k]);
      else if (!INV) fs_mul(x, fbase, j);
      if constexpr (TWIST && INV) {
        const uint4 f = *reinterpret_cast<const uint4 *>(A.psi + fbase + ((long long)j << logS) + c4);
        x.x = M.canon(M.mul(V<Arith<H>::KI>{x.x}, f.x)); x.y = M.canon(M.mul(V<Arith<H>::KI>{x.y}, f.y));
        x.z = M.canon(M.mul(V<Arith<H>::KI>{x.z}, f.z)); x.w = M.canon(M.mul(V<Arith<H>::KI>{x.w}, f.w));
      } else if (A.canon_out) {   // forward results are < 2p after the multiply, inverse results < KI*p
        constexpr int KS = INV ? Arith<H>::KI : Arith<H>::KM;
        x.x = M.canon(V<KS>{x.x}); x.y = M.canon(V<KS>{x.y}); x.z = M.canon(V<KS>{x.z}); x.w = M.canon(V<KS>{x.w});
      }
      if constexpr (PIECE == 2) *reinterpret_cast<uint4 *>(out + piece_base(t) + piece_row(j) + c4) = x;
      else *reinterpret_cast<uint4 *>(out + gbase + ((long long)j << logS) + c4) = x;
    }
  }
}

// ------------------------------------------------------------------------------------
// Element-wise helpers around the transforms
// ------------------------------------------------------------------------------------

// out[i] = in[bitrev_logn(i)] within each length-2^logn vector (stand-alone transform API:
// the reference's transform is natural-order in and out, main.cc:146-152).
static __global__ void bitrev_kernel(const uint32_t *in, uint32_t *out, int logn, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long vec = i >> logn;
    const uint32_t k = (uint32_t)(i & ((1ll << logn) - 1));
    const uint32_t r = logn ? (__brev(k) >> (32 - logn)) : 0u;
    out[i] = in[(vec << logn) + r];
  }
}

// u64 coefficients -> u32 residues mod up to 4 primes (residue split, main.cc:292-303).
// src is [batch][n]; dst[k] is [batch][len].  x = hi*2^32 + lo, so x mod m = hi*R + lo mod m with
// R = 2^32: two Montgomery products, REDC(hi * R^2) and REDC(lo * R), instead of a 64-bit division.
// Words n..zfill-1 of every vector are zeroed (zfill = len for single-pass lengths, whose kernels
// read the padding; zfill = n where the column pass never reads it).
struct SplitArgs {
  const uint64_t *src[2];   // a and b (blockIdx.z)
  uint32_t *dst[2][4];
  PrimeCtx ctx[4];
  int nmod;
  int n, len, zfill;
  int vec;                  // n % 4 == 0 and all pointers 16-byte aligned
  int batch;                // vectors per input; gridDim.y strides over them
  CoefMod cm;               // pmod != 0: coefficients at or above it are first reduced modulo it (the caller's modulus)
};
static __global__ void split_kernel(SplitArgs A) {
 for (long long b = blockIdx.y; b < A.batch; b += gridDim.y) {
  const uint64_t *src = A.src[blockIdx.z] + b * A.n;
  if (A.vec) {   // rows are 16-byte aligned: four coefficients per thread, 16-byte accesses
    for (int j = (blockIdx.x * blockDim.x + threadIdx.x) * 4; j < A.zfill; j += gridDim.x * blockDim.x * 4) {
      ulonglong2 x01 = make_ulonglong2(0, 0), x23 = make_ulonglong2(0, 0);
      if (j < A.n) {
        x01 = *reinterpret_cast<const ulonglong2 *>(src + j);
        x23 = *reinterpret_cast<const ulonglong2 *>(src + j + 2);
        x01.x = split_in(x01.x, A.cm); x01.y = split_in(x01.y, A.cm);
        x23.x = split_in(x23.x, A.cm); x23.y = split_in(x23.y, A.cm);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k < A.nmod) {
          const PrimeCtx c = A.ctx[k];
          const uint4 r = make_uint4(split_one(x01.x, c), split_one(x01.y, c), split_one(x23.x, c), split_one(x23.y, c));
          *reinterpret_cast<uint4 *>(A.dst[blockIdx.z][k] + b * A.len + j) = r;
        }
      }
    }
    continue;
  }
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < A.zfill; j += gridDim.x * blockDim.x) {
    const uint64_t x = (j < A.n) ? split_in(src[j], A.cm) : 0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < A.nmod) A.dst[blockIdx.z][k][b * A.len + j] = split_one(x, A.ctx[k]);
    }
  }
 }
}

// i32 / u32 coefficients -> zero-padded canonical residues; src [batch][n] -> dst [batch][len]
struct PadArgs {
  const uint32_t *src[2];
  uint32_t *dst[2];
  PrimeCtx ctx;
  int n, len, zfill;
  int batch;
};
static __global__ void pad_kernel(PadArgs A) {
  for (long long b = blockIdx.y; b < A.batch; b += gridDim.y) {
    const uint32_t *src = A.src[blockIdx.z] + b * A.n;
    uint32_t *dst = A.dst[blockIdx.z] + b * A.len;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < A.zfill; j += gridDim.x * blockDim.x) {
      const uint32_t x = (j < A.n) ? src[j] : 0u;
      // inputs are canonical in the reference's harness; anything larger is reduced like main.cc:300 does
      dst[j] = x >= A.ctx.p ? csub(mont_mul_lazy(x, A.ctx.one, A.ctx.p, A.ctx.npinv), A.ctx.p) : x;
    }
  }
}

// dst[z][b][i] = src[z][b][i] * tab[i] mod p, canonical (tab in Montgomery form): the psi^i / psi^-i
// twists that turn a cyclic product into a negacyclic one
struct TwistArgs {
  const uint32_t *src[2];
  uint32_t *dst[2];
  const uint32_t *tab;
  PrimeCtx ctx;
  int len;
  int batch;
};
static __global__ void twist_kernel(TwistArgs A) {
  for (long long b = blockIdx.y; b < A.batch; b += gridDim.y) {
    const uint32_t *src = A.src[blockIdx.z] + b * A.len;
    uint32_t *dst = A.dst[blockIdx.z] + b * A.len;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < A.len; i += gridDim.x * blockDim.x)
      dst[i] = mont_mul(src[i], A.tab[i], A.ctx.p, A.ctx.npinv);
  }
}

// dst[i] = src[i] on 16-byte words, one word per lane (one trip wherever the grid covers the buffer): the streaming copy whose rate is the ceiling a
// read-once / write-once pass can reach on this part (measurement aid: benchmarks quote the passes against it)
static __global__ void __launch_bounds__(256) copy16_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, long long n16) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}

// gather the first nout words of every length-len vector (u32 -> u32 or u64)
template <class OUT>
__global__ void unpad_kernel(const uint32_t *src, OUT *dst, int nout, int len, int batch) {
  for (long long b = blockIdx.y; b < batch; b += gridDim.y)
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nout; j += gridDim.x * blockDim.x)
      dst[b * nout + j] = (OUT)src[b * len + j];
}

// Multi-prime CRT recombination (replaces the 128-bit sum/% M loop of main.cc:323-337):
// Garner mixed-radix digits v_k, then Horner evaluation of
//   x = v0 + m0 (v1 + m1 (v2 + m2 v3))  modulo the caller's p on 64-bit Montgomery lanes.
// The moduli are in ascending order, so a digit v_q < m_q is already reduced modulo every later m_k.
struct CrtArgs {
  const uint32_t *res[4];   // [batch][len] canonical residues per prime
  uint64_t *out;            // [batch][nout]
  PrimeCtx ctx[4];
  uint32_t inv[4][4];       // inv[i][j] = m_i^{-1} mod m_j in Montgomery form (i < j)
  // caller modulus p = 2^e * q, q odd: the value is evaluated modulo q on Montgomery lanes and modulo
  // 2^e by wrapping 64-bit arithmetic, then joined (e = 0 for the odd moduli of the reference's inputs)
  PrimeCtx64 pctx;          // q (unused when q == 1)
  uint64_t cR[4];           // (m_0 ... m_{k-1}) * 2^64 mod q, canonical; cR[0] = 2^64 mod q
  uint64_t qinv;            // q^{-1} mod 2^64
  int e;                    // p = 2^e * q
  int nmod, nout, len;
  int vec;                  // len % 4 == 0 and every residue pointer 16-byte aligned: 16-byte loads
  int batch;                // vectors; gridDim.y strides over them
};
// x = v0 + m0 v1 + m0 m1 v2 + m0 m1 m2 v3 with 31-bit digits: S = sum v_k cR[k] < 2^33 q is reduced by
// ONE 128-bit REDC (S / 2^64 < q / 2^31, so the result is at most q and one subtraction finishes).
__device__ __forceinline__ uint64_t crt_one(const CrtArgs &A, const uint32_t (&res)[4]) {
  const uint64_t q = A.pctx.p;
  uint32_t v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k < A.nmod) {
      const uint32_t mk = A.ctx[k].p;
      uint32_t t = res[k];
#pragma unroll
      for (int i = 0; i < k; ++i) {
        const uint32_t d = t - v[i];
        t = mont_mul(t >= v[i] ? d : d + mk, A.inv[i][k], mk, A.ctx[k].npinv);
      }
      v[k] = t;
    } else {
      v[k] = 0;
    }
  }
  uint64_t r = 0;
  if (q > 1) {
    unsigned __int128 S = (unsigned __int128)(uint64_t)v[0] * A.cR[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) S += (unsigned __int128)(uint64_t)v[k] * A.cR[k];
    const uint64_t lo = (uint64_t)S, hi = (uint64_t)(S >> 64);
    const uint64_t m = lo * A.pctx.npinv;
    r = hi + mulhi64(m, q) + (lo != 0 ? 1u : 0u);
    if (r >= q) r -= q;
  }
  if (A.e > 0) {   // even modulus: x mod 2^e from wrapping arithmetic, then x = r + q * ((x2 - r) / q mod 2^e)
    uint64_t x2 = v[3];
    for (int k = 2; k >= 0; --k) x2 = x2 * A.ctx[k].p + v[k];
    const uint64_t mask = (1ull << A.e) - 1;
    r += q * (((x2 - r) * A.qinv) & mask);
  }
  return r;
}
// four consecutive indices per thread (16-byte residue loads) when the rows allow it
static __global__ void crt_kernel(CrtArgs A) {
 for (long long b = blockIdx.y; b < A.batch; b += gridDim.y) {
  uint64_t *out = A.out + b * A.nout;
  if (A.vec) {
    for (int j = (blockIdx.x * blockDim.x + threadIdx.x) * 4; j < A.nout; j += gridDim.x * blockDim.x * 4) {
      uint4 x[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = (k < A.nmod) ? *reinterpret_cast<const uint4 *>(A.res[k] + b * A.len + j) : make_uint4(0, 0, 0, 0);
      const uint32_t r0[4] = {x[0].x, x[1].x, x[2].x, x[3].x}, r1[4] = {x[0].y, x[1].y, x[2].y, x[3].y};
      const uint32_t r2[4] = {x[0].z, x[1].z, x[2].z, x[3].z}, r3[4] = {x[0].w, x[1].w, x[2].w, x[3].w};
      out[j] = crt_one(A, r0);
      if (j + 1 < A.nout) out[j + 1] = crt_one(A, r1);
      if (j + 2 < A.nout) out[j + 2] = crt_one(A, r2);
      if (j + 3 < A.nout) out[j + 3] = crt_one(A, r3);
    }
    continue;
  }
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < A.nout; j += gridDim.x * blockDim.x) {
    uint32_t r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = (k < A.nmod) ? A.res[k][b * A.len + j] : 0u;
    out[j] = crt_one(A, r);
  }
 }
}

}  // namespace nttx
