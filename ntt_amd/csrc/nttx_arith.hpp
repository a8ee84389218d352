// Modular arithmetic shared by the host-side plan builder and the gfx950 kernels.
//
// Residues are 32-bit lanes, Montgomery radix R = 2^32, odd primes p < 2^31.
// Data stays in ordinary (non-Montgomery) form; only constants (twiddles,
// scale factors) are stored as w*R mod p, so mont_mul(x, wR) == x*w mod p.
//
// Reference arithmetic this replaces: BarrettReduction::mul_mod main.cc:51-57
// (two 64x64->128 products per modular multiply) and MontMul::REDC
// monttry2.cpp:91-100 (R = 2^30 on 64-bit lanes).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NTTX_HD __host__ __device__ __forceinline__
#else
#define NTTX_HD inline
#endif

namespace nttx {

struct PrimeCtx {
  uint32_t p;      // modulus
  uint32_t npinv;  // -p^{-1} mod 2^32
  uint32_t one;    // R mod p   (Montgomery form of 1)
  uint32_t r2;     // R^2 mod p (to-Montgomery multiplier)
};

// Montgomery product a*b*R^{-1} mod p, result in [0, 2p).
// Precondition: a*b < 2^32 * p  (any 32-bit a with canonical b < p qualifies).
// Three multiplier-pipe instructions on gfx950: v_mad_u64_u32, v_mul_lo_u32,
// v_mad_u64_u32 (the second mad folds the 64-bit add of REDC).
NTTX_HD uint32_t mont_mul_lazy(uint32_t a, uint32_t b, uint32_t p, uint32_t npinv) {
  uint64_t t = (uint64_t)a * b;
  uint32_t m = (uint32_t)t * npinv;
  return (uint32_t)((t + (uint64_t)m * p) >> 32);
}

// x in [0, 2m) -> [0, m) without a branch: unsigned min picks the non-wrapped value.
NTTX_HD uint32_t csub(uint32_t x, uint32_t m) {
  uint32_t y = x - m;
  return x < y ? x : y;
}

NTTX_HD uint32_t mont_mul(uint32_t a, uint32_t b, uint32_t p, uint32_t npinv) {
  return csub(mont_mul_lazy(a, b, p, npinv), p);
}

// ---- 64-bit lanes: REDC with R = 2^64 for moduli up to 2^63 (CRT recombination) ----
struct PrimeCtx64 {
  uint64_t p;
  uint64_t npinv;  // -p^{-1} mod 2^64
  uint64_t r2;     // R^2 mod p
};

NTTX_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// a*b*2^{-64} mod p, canonical. Precondition a*b < 2^64 * p, p odd < 2^63.
NTTX_HD uint64_t mont_mul64(uint64_t a, uint64_t b, uint64_t p, uint64_t npinv) {
  uint64_t lo = a * b, hi = mulhi64(a, b);
  uint64_t m = lo * npinv;
  uint64_t t = hi + mulhi64(m, p) + (lo != 0 ? 1u : 0u);  // low words cancel to 2^64 iff lo != 0
  return t >= p ? t - p : t;
}

// ---------------- host-only helpers for plan construction ----------------
inline uint32_t h_mulmod(uint32_t a, uint32_t b, uint32_t p) { return (uint32_t)((uint64_t)a * b % p); }
inline uint32_t h_powmod(uint32_t a, uint64_t e, uint32_t p) {
  uint64_t r = 1, x = a % p;
  while (e) { if (e & 1) r = r * x % p; x = x * x % p; e >>= 1; }
  return (uint32_t)r;
}
inline uint64_t h_mulmod64(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t)((unsigned __int128)a * b % p); }
inline uint64_t h_powmod64(uint64_t a, uint64_t e, uint64_t p) {
  uint64_t r = 1 % p, x = a % p;
  while (e) { if (e & 1) r = h_mulmod64(r, x, p); x = h_mulmod64(x, x, p); e >>= 1; }
  return r;
}
inline PrimeCtx make_ctx(uint32_t p) {
  PrimeCtx c; c.p = p;
  uint32_t inv = p;                                   // Newton: inv = p^{-1} mod 2^32
  for (int i = 0; i < 5; ++i) inv *= 2u - p * inv;
  c.npinv = 0u - inv;
  c.one = (uint32_t)((1ull << 32) % p);
  c.r2 = (uint32_t)((uint64_t)c.one * c.one % p);
  return c;
}
inline PrimeCtx64 make_ctx64(uint64_t p) {
  PrimeCtx64 c; c.p = p;
  uint64_t inv = p;
  for (int i = 0; i < 6; ++i) inv *= 2ull - p * inv;
  c.npinv = 0ull - inv;
  uint64_t r = (uint64_t)((((unsigned __int128)1) << 64) % p);
  c.r2 = h_mulmod64(r, r, p);
  return c;
}
inline uint32_t to_mont(uint32_t x, const PrimeCtx &c) { return (uint32_t)(((uint64_t)x << 32) % c.p); }
inline bool is_prime_u64(uint64_t n) {               // deterministic Miller-Rabin for 64-bit
  if (n < 2) return false;
  for (uint64_t q : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
    if (n % q == 0) return n == q;
  }
  uint64_t d = n - 1; int s = 0;
  while ((d & 1) == 0) { d >>= 1; ++s; }
  for (uint64_t a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
    uint64_t x = h_powmod64(a, d, n);
    if (x == 1 || x == n - 1) continue;
    bool comp = true;
    for (int r = 1; r < s; ++r) { x = h_mulmod64(x, x, n); if (x == n - 1) { comp = false; break; } }
    if (comp) return false;
  }
  return true;
}

}  // namespace nttx
