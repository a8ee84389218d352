// Host side of the 64-bit-lane direct path: plans (decomposition + Montgomery tables, R = 2^64),
// workspace and kernel launches for nttx_kernels64.hpp.  The orchestration is that of the
// single-prime product get_result (monttry2.cpp:319-343: forward a, forward b, pointwise,
// inverse, scale by n^-1) on a modulus the reference can only reach through four 32-bit
// primes and CRT (main.cc:286-338).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>
#include <cstring>

#include "../../include/nttx.h"
#include "nttx64.hpp"
#include "nttx64_launch.hpp"
#include "nttx_arith.hpp"
#include "nttx_kernels64.hpp"

namespace nttx {

static thread_local std::string g_err64;
static int fail64(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err64 = buf;
  return code;
}
const char *engine64_error() { return g_err64.c_str(); }
#define HIPCHK64(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { (void)hipGetLastError(); \
  return fail64(NTTX_EDEVICE, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } } while (0)

constexpr int kMaxRow64 = 13, kMinCol64 = 6, kMaxCol64 = 9, kMaxLog64 = 29;
static int env_int64(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
// Longest row pass for modulus p.  Rows of 2^12 and 2^13 points take one whole workgroup each
// (NTTX_LONGROW64=0: the plans without them, for A/B runs).  In the lazy class (p < 2^51) a forward
// pass of k stages on values below 2p ends below 2^(k+1) p, which has to fit 64 bits: k <= 63 - bits(p).
// Moduli of the floating-point class (nttx_kernels64.hpp: ArithFP): 2^31 <= p < 1.35e15, where three and a
// quarter multiples of p stay below 2^52.  NTTX_FP64=0 switches the class off (A/B runs).
bool engine64_fp_ok(uint64_t p) {
  static const bool off = [] { const char *e = getenv("NTTX_FP64"); return e && atoi(e) == 0; }();
  return !off && p >= (1ull << 31) && p < 1350000000000000ull;
}

static int max_row64(uint64_t p, bool fp) {
  static const int m = [] { const char *e = getenv("NTTX_LONGROW64"); return (e && atoi(e) == 0) ? 11 : kMaxRow64; }();
  if (fp || p >= (1ull << 51)) return m;                 // values never grow: floating-point class, Harvey butterflies
  int bits = 0; while (bits < 64 && (p >> bits)) ++bits;   // p < 2^bits
  return std::max(11, std::min(m, 63 - bits));
}

int engine64_decompose(int k, uint64_t p, int lt[3], bool fp) {
  const int maxrow = max_row64(p, fp);
  if (k <= maxrow) { lt[0] = k; return 1; }
  if (k <= kMaxCol64 + maxrow) {
    lt[0] = std::min(kMaxCol64, k / 2);
    lt[1] = k - lt[0];
    return 2;
  }
  int row = std::max((k + 2) / 3, k - 2 * kMaxCol64);
  if (row > 11) row = 11;
  const int rem = k - row;
  lt[0] = (rem + 1) / 2; lt[1] = rem - lt[0]; lt[2] = row;
  return 3;
}

struct Plan64 {
  uint64_t p = 0; int logL = 0; bool conj = false; bool safe = false, plo1 = false;
  bool fp = false;            // floating-point class: tables hold (w, w/p) pairs of doubles, no Montgomery form
  uint64_t pre_q = 0;         // fp: n^-1 / p as a double (pre_single holds n^-1 as a double)
  PrimeArgs64 ctx{};
  int npass = 0; int lt[3] = {0, 0, 0};
  uint64_t *tw_fwd[3] = {nullptr, nullptr, nullptr}, *tw_inv[3] = {nullptr, nullptr, nullptr};
  uint64_t *fs_fwd[2] = {nullptr, nullptr}, *fs_inv[2] = {nullptr, nullptr};
  uint64_t pre_single = 0;   // n^-1 R^2 mod p
  ~Plan64() {
    for (auto q : tw_fwd) if (q) (void)hipFree(q);
    for (auto q : tw_inv) if (q) (void)hipFree(q);
    for (auto q : fs_fwd) if (q) (void)hipFree(q);
    for (auto q : fs_inv) if (q) (void)hipFree(q);
  }
};

struct Buf64 {
  void *ptr = nullptr; size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (ptr) { HIPCHK64(hipFree(ptr)); ptr = nullptr; cap = 0; }
    const size_t want = (bytes + 255) & ~(size_t)255;
    HIPCHK64(hipMalloc(&ptr, want));
    cap = want;
    return 0;
  }
  ~Buf64() { if (ptr) (void)hipFree(ptr); }
};

// psi^i and psi^-i, i < 2^logL, Montgomery form: psi a primitive 2^(logL+1)-th root of unity
struct Twist64 {
  uint64_t *fwd = nullptr, *inv = nullptr;
  ~Twist64() { if (fwd) (void)hipFree(fwd); if (inv) (void)hipFree(inv); }
};

struct Engine64 {
  std::map<std::tuple<uint64_t, int, bool, bool>, std::unique_ptr<Plan64>> plans;
  std::map<std::tuple<uint64_t, int, bool>, std::unique_ptr<Twist64>> twists;
  Buf64 wa, wb;
  Buf64 ta, tb;      // twisted copies of the inputs of a negacyclic product
};

Engine64 *engine64_create() { return new Engine64; }
void engine64_destroy(Engine64 *e) { delete e; }

// ---- host arithmetic ---------------------------------------------------------------------
static uint64_t find_nonresidue64(uint64_t p) {   // the reference's root 3 first (main.cc:241)
  const uint64_t cand[] = {3, 2, 5, 6, 7, 10, 11, 13, 14, 15, 17, 19, 21, 22, 23, 26, 29, 31, 37, 41, 43, 47};
  for (uint64_t g : cand) if (g % p && h_powmod64(g, (p - 1) / 2, p) == p - 1) return g;
  for (uint64_t g = 48; g < 100000; ++g) if (h_powmod64(g, (p - 1) / 2, p) == p - 1) return g;
  return 0;
}

bool engine64_ok(uint64_t p, int logL) {
  if (p < (1ull << 31) || p >= (1ull << 62) || (p & 1) == 0) return false;
  if (logL < 1 || logL > kMaxLog64) return false;
  if (((p - 1) & ((1ull << logL) - 1)) != 0) return false;
  if (!is_prime_u64(p)) return false;
  return find_nonresidue64(p) != 0;
}

struct HostMont64 {
  uint64_t p, npinv, one, r2;
  uint64_t mul(uint64_t a, uint64_t b) const { return mont_mul64(a, b, p, npinv); }   // canonical
  uint64_t to(uint64_t x) const { return mul(x % p, r2); }
};

static inline uint32_t bitrev64(uint32_t x, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

// tab[h + j] = w^(j * T/(2h)), Montgomery form, h = 1, 2, ..., T/2
static std::vector<uint64_t> stage_table64(uint64_t w, int logT, const HostMont64 &M) {
  const int T = 1 << logT;
  std::vector<uint64_t> tab(std::max(T, 2), 0);
  for (int h = 1; h < T; h <<= 1) {
    const uint64_t step = M.to(h_powmod64(w, (uint64_t)(T / (2 * h)), M.p));
    uint64_t cur = M.one;
    for (int j = 0; j < h; ++j) { tab[h + j] = cur; cur = M.mul(cur, step); }
  }
  return tab;
}

// tab[j*S + s] = extra * wN^(s * bitrev_T(j)), Montgomery form
static std::vector<uint64_t> fourstep_table64(uint64_t wN, int logT, int logS, uint64_t extra, const HostMont64 &M) {
  const size_t T = (size_t)1 << logT, S = (size_t)1 << logS;
  std::vector<uint64_t> tab(T * S);
  const uint64_t e0 = M.to(extra);
  for (size_t j = 0; j < T; ++j) {
    const uint64_t step = M.to(h_powmod64(wN, bitrev64((uint32_t)j, logT), M.p));
    uint64_t cur = e0;
    uint64_t *row = &tab[j * S];
    for (size_t s = 0; s < S; ++s) { row[s] = cur; cur = M.mul(cur, step); }
  }
  return tab;
}

static int upload64(uint64_t **dst, const std::vector<uint64_t> &h) {
  HIPCHK64(hipMalloc((void **)dst, h.size() * sizeof(uint64_t)));
  HIPCHK64(hipMemcpy(*dst, h.data(), h.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  return 0;
}

// (w, w/p) pairs of doubles for the floating-point class, from plain residues
static uint64_t dbits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static std::vector<uint64_t> fp_pairs(const std::vector<uint64_t> &plain, uint64_t p) {
  std::vector<uint64_t> out(2 * plain.size());
  for (size_t i = 0; i < plain.size(); ++i) {
    out[2 * i] = dbits((double)plain[i]);
    out[2 * i + 1] = dbits((double)plain[i] / (double)p);
  }
  return out;
}
// plain-residue versions of the two table builders (the Montgomery "one" is 1 and products are ordinary)
struct HostPlain64 {
  uint64_t p;
  uint64_t mul(uint64_t a, uint64_t b) const { return h_mulmod64(a, b, p); }
};
static std::vector<uint64_t> stage_table_plain(uint64_t w, int logT, uint64_t p) {
  const int T = 1 << logT;
  const HostPlain64 M{p};
  std::vector<uint64_t> tab(std::max(T, 2), 0);
  for (int h = 1; h < T; h <<= 1) {
    const uint64_t step = h_powmod64(w, (uint64_t)(T / (2 * h)), p);
    uint64_t cur = 1;
    for (int j = 0; j < h; ++j) { tab[h + j] = cur; cur = M.mul(cur, step); }
  }
  return tab;
}
static std::vector<uint64_t> fourstep_table_plain(uint64_t wN, int logT, int logS, uint64_t extra, uint64_t p) {
  const size_t T = (size_t)1 << logT, S = (size_t)1 << logS;
  const HostPlain64 M{p};
  std::vector<uint64_t> tab(T * S);
  for (size_t j = 0; j < T; ++j) {
    const uint64_t step = h_powmod64(wN, bitrev64((uint32_t)j, logT), p);
    uint64_t cur = extra % p;
    uint64_t *row = &tab[j * S];
    for (size_t s = 0; s < S; ++s) { row[s] = cur; cur = M.mul(cur, step); }
  }
  return tab;
}

static int get_plan64(Engine64 *e, uint64_t p, int logL, bool conj, Plan64 **out, bool fp = false) {
  auto key = std::make_tuple(p, logL, conj, fp);
  auto it = e->plans.find(key);
  if (it != e->plans.end()) { *out = it->second.get(); return 0; }
  if (!engine64_ok(p, logL))
    return fail64(NTTX_EINVAL, "modulus %llu does not support a direct 64-bit length-2^%d transform", (unsigned long long)p, logL);
  std::unique_ptr<Plan64> pl(new Plan64);
  pl->p = p; pl->logL = logL; pl->conj = conj; pl->fp = fp;
  pl->safe = p >= (1ull << 51);
  pl->plo1 = (uint32_t)p == 1u;
  const PrimeCtx64 c = make_ctx64(p);
  HostMont64 M{p, c.npinv, (uint64_t)((((unsigned __int128)1) << 64) % p), c.r2};
  pl->ctx = PrimeArgs64{p, c.npinv, M.one, M.r2};
  pl->npass = engine64_decompose(logL, p, pl->lt, fp);
  const uint64_t g = find_nonresidue64(p);
  uint64_t wL = h_powmod64(g, (p - 1) >> logL, p);                 // primitive 2^logL-th root (main.cc:156)
  if (conj) wL = h_powmod64(wL, p - 2, p);                         // main.cc:157
  const uint64_t wLi = h_powmod64(wL, p - 2, p);
  const uint64_t Lfull = 1ull << logL;
  const uint64_t ninv = h_powmod64(Lfull % p, p - 2, p);           // main.cc:204
  if (fp) { pl->pre_single = dbits((double)ninv); pl->pre_q = dbits((double)ninv / (double)p); }
  else pl->pre_single = M.mul(M.to(ninv), M.r2);                   // n^-1 R^2 mod p
  int logN = logL;
  for (int lv = 0; lv < pl->npass; ++lv) {
    const int lt = pl->lt[lv];
    const uint64_t wN = h_powmod64(wL, Lfull >> logN, p), wNi = h_powmod64(wLi, Lfull >> logN, p);
    const uint64_t wT = h_powmod64(wN, (1ull << logN) >> lt, p), wTi = h_powmod64(wNi, (1ull << logN) >> lt, p);
    if (fp) {
      if (int rc = upload64(&pl->tw_fwd[lv], fp_pairs(stage_table_plain(wT, lt, p), p))) return rc;
      if (int rc = upload64(&pl->tw_inv[lv], fp_pairs(stage_table_plain(wTi, lt, p), p))) return rc;
    } else {
      if (int rc = upload64(&pl->tw_fwd[lv], stage_table64(wT, lt, M))) return rc;
      if (int rc = upload64(&pl->tw_inv[lv], stage_table64(wTi, lt, M))) return rc;
    }
    if (lv < pl->npass - 1) {
      const int logS = logN - lt;
      if (fp) {
        if (int rc = upload64(&pl->fs_fwd[lv], fp_pairs(fourstep_table_plain(wN, lt, logS, 1, p), p))) return rc;
        if (int rc = upload64(&pl->fs_inv[lv], fp_pairs(fourstep_table_plain(wNi, lt, logS, lv == 0 ? ninv : 1, p), p))) return rc;
      } else {
        if (int rc = upload64(&pl->fs_fwd[lv], fourstep_table64(wN, lt, logS, 1, M))) return rc;
        if (int rc = upload64(&pl->fs_inv[lv], fourstep_table64(wNi, lt, logS, lv == 0 ? ninv : 1, M))) return rc;
      }
      logN = logS;
    }
  }
  *out = pl.get();
  e->plans[key] = std::move(pl);
  return 0;
}

// ---- launches ------------------------------------------------------------------------------
static int grid_for64(long long tiles, int cap) { return (int)std::max<long long>(1, std::min<long long>(tiles, cap)); }

template <int MODE, bool SAFE, bool PLO1>
static int launch_row64_t(int logT, const RowArgs64 &A, hipStream_t st, bool fp = false) {
#define NTTX_ROW64_CASE(K) case K: { \
    using Lyt = RowLayout64<K>; \
    const long long tiles = (A.nrows + Lyt::NT - 1) / Lyt::NT; \
    const int cap = K >= 12 ? (1 << 20) : cap_row;   /* long rows: one row per workgroup */ \
    if constexpr (MODE == 2) { \
      if (fp) { if constexpr (!SAFE && !PLO1) { \
        if (A.psi_in) hipLaunchKernelGGL((rowmul64_kernel<K, false, false, true, true>), dim3(grid_for64(tiles, cap)), dim3(Lyt::NTHR), 0, st, A); \
        else hipLaunchKernelGGL((rowmul64_kernel<K, false, false, false, true>), dim3(grid_for64(tiles, cap)), dim3(Lyt::NTHR), 0, st, A); } } \
      else if (A.psi_in) hipLaunchKernelGGL((rowmul64_kernel<K, SAFE, PLO1, true>), dim3(grid_for64(tiles, cap)), dim3(Lyt::NTHR), 0, st, A); \
      else hipLaunchKernelGGL((rowmul64_kernel<K, SAFE, PLO1>), dim3(grid_for64(tiles, cap)), dim3(Lyt::NTHR), 0, st, A); } \
    else hipLaunchKernelGGL((row64_kernel<K, MODE, SAFE, PLO1>), dim3(grid_for64(tiles, cap)), dim3(Lyt::NTHR), 0, st, A); \
    break; }
  static const int cap_row = env_int64("NTTX_GRID_ROW64", 4096);
  switch (logT) {
    NTTX_ROW64_CASE(1) NTTX_ROW64_CASE(2) NTTX_ROW64_CASE(3) NTTX_ROW64_CASE(4) NTTX_ROW64_CASE(5) NTTX_ROW64_CASE(6)
    NTTX_ROW64_CASE(7) NTTX_ROW64_CASE(8) NTTX_ROW64_CASE(9) NTTX_ROW64_CASE(10) NTTX_ROW64_CASE(11)
    NTTX_ROW64_CASE(12) NTTX_ROW64_CASE(13)
    default: return fail64(NTTX_ERANGE, "row pass of 2^%d points unsupported", logT);
  }
#undef NTTX_ROW64_CASE
  HIPCHK64(hipGetLastError());
  return 0;
}
template <int MODE>
static int launch_row64_m(const Plan64 *pl, int logT, const RowArgs64 &A, hipStream_t st) {
  if (pl->fp) return launch_row64_t<MODE, false, false>(logT, A, st, true);
  if (pl->safe) return pl->plo1 ? launch_row64_t<MODE, true, true>(logT, A, st) : launch_row64_t<MODE, true, false>(logT, A, st);
  return pl->plo1 ? launch_row64_t<MODE, false, true>(logT, A, st) : launch_row64_t<MODE, false, false>(logT, A, st);
}
static int launch_row64(const Plan64 *pl, int mode, int logT, const RowArgs64 &A, hipStream_t st, const ProfHook64 *prof) {
  if (prof) prof->begin(prof->self, mode == 2 ? NTTX_K_ROW_MUL : NTTX_K_ROW_FWD, st);
  // (mode 1, a decimation-in-time row pass of its own, has no caller: inverse transforms run mode 0 on the conjugate root)
  const int rc = mode == 0 ? launch_row64_m<0>(pl, logT, A, st) : mode == 2 ? launch_row64_m<2>(pl, logT, A, st)
                                                                           : fail64(NTTX_EINVAL, "row pass mode %d unsupported", mode);
  if (prof) prof->end(prof->self, st);
  return rc;
}

// column passes are compiled in their own translation unit (nttx64_col.hip)
static int launch_col64(const Plan64 *pl, bool inv, int logT, const ColArgs64 &A, int nvec, hipStream_t st, const ProfHook64 *prof) {
  if (logT < kMinCol64 || logT > kMaxCol64 || A.logS < 4) return fail64(NTTX_ERANGE, "column pass shape unsupported (logT=%d logS=%d)", logT, A.logS);
  if (prof) prof->begin(prof->self, inv ? NTTX_K_COL_INV : NTTX_K_COL_FWD, st);
  const char *msg = nullptr;
  int rc = launch_col64_kernels(inv, pl->safe, pl->plo1, pl->fp, logT, A, nvec, st, &msg);
  if (prof) prof->end(prof->self, st);
  if (rc) rc = fail64(rc, "%s (logT=%d logS=%d)", msg ? msg : "column pass failed", logT, A.logS);
  return rc;
}

int engine64_reserve(Engine64 *e, int logL, int batch) {
  const size_t vec = ((size_t)batch << logL) * 8;
  if (int rc = e->wa.ensure(vec)) return rc;
  return e->wb.ensure(vec);
}

// psi_in / psi_out (one- and two-pass plans): the product becomes the negacyclic one, a and b multiplied by psi_in[i] as
// the first pass reads them and c by psi_out[i] as the last pass writes it
static int cyclic_multiply64(Engine64 *e, const uint64_t *a, const uint64_t *b, uint64_t *c, int logL, uint64_t p,
                             int batch, long long nvalid, hipStream_t st, const ProfHook64 *prof,
                             const uint64_t *psi_in, const uint64_t *psi_out, bool psi_fp = false) {
  Plan64 *pl;
  // plain products of moduli below 1.35e15 run on the floating-point class; inputs there must be canonical
  const bool fp = engine64_fp_ok(p) && (!psi_in || psi_fp);
  if (int rc = get_plan64(e, p, logL, false, &pl, fp)) return rc;
  const long long len = 1ll << logL;
  if (psi_in && pl->npass > 2) return fail64(NTTX_ERANGE, "fused twist needs a one- or two-pass plan");
  if (pl->npass == 1) {
    RowArgs64 R{}; R.in0 = a; R.in1 = b; R.out = c; R.tw_fwd = pl->tw_fwd[0]; R.tw_inv = pl->tw_inv[0];
    R.ctx = pl->ctx; R.pre = pl->pre_single; R.pre_q = pl->pre_q; R.int_in = 1; R.canon_out = 1; R.nrows = batch;
    R.psi_in = psi_in; R.psi_out = psi_out;
    return launch_row64(pl, 2, pl->lt[0], R, st, prof);
  }
  if (int rc = engine64_reserve(e, logL, batch)) return rc;
  uint64_t *wa = (uint64_t *)e->wa.ptr, *wb = (uint64_t *)e->wb.ptr;
  const int ncol = pl->npass - 1;
  int logN = logL;
  for (int lv = 0; lv < ncol; ++lv) {
    ColArgs64 C{}; C.in[0] = lv == 0 ? a : wa; C.in[1] = lv == 0 ? b : wb; C.out[0] = wa; C.out[1] = wb;
    C.tw = pl->tw_fwd[lv]; C.fs = pl->fs_fwd[lv]; C.ctx = pl->ctx; C.logS = logN - pl->lt[lv]; C.canon_out = 0;
    C.nblocks = ((long long)batch << logL) >> logN;
    C.nvalid = (lv == 0 && nvalid > 0 && 2 * nvalid <= len) ? nvalid : 0;
    if (lv == 0 && psi_in) { C.psi = psi_in; C.nvalid = 0; }
    C.int_in = (lv == 0);
    if (int rc = launch_col64(pl, false, pl->lt[lv], C, 2, st, prof)) return rc;
    logN = C.logS;
  }
  {
    RowArgs64 R{}; R.in0 = wa; R.in1 = wb; R.out = c; R.tw_fwd = pl->tw_fwd[ncol]; R.tw_inv = pl->tw_inv[ncol];
    R.ctx = pl->ctx; R.pre = pl->fp ? 0 : pl->ctx.r2; R.canon_out = 0; R.nrows = ((long long)batch << logL) >> pl->lt[ncol];
    if (int rc = launch_row64(pl, 2, pl->lt[ncol], R, st, prof)) return rc;
  }
  for (int lv = ncol - 1; lv >= 0; --lv) {
    int logNl = logL; for (int q = 0; q < lv; ++q) logNl -= pl->lt[q];
    ColArgs64 C{}; C.in[0] = c; C.out[0] = c; C.tw = pl->tw_inv[lv]; C.fs = pl->fs_inv[lv]; C.ctx = pl->ctx;
    C.logS = logNl - pl->lt[lv]; C.canon_out = (lv == 0); C.nblocks = ((long long)batch << logL) >> logNl;
    if (lv == 0 && psi_out) C.psi = psi_out;
    if (int rc = launch_col64(pl, true, pl->lt[lv], C, 1, st, prof)) return rc;
  }
  return 0;
}

int engine64_cyclic_multiply(Engine64 *e, const uint64_t *a, const uint64_t *b, uint64_t *c, int logL, uint64_t p,
                             int batch, long long nvalid, hipStream_t st, const ProfHook64 *prof) {
  return cyclic_multiply64(e, a, b, c, logL, p, batch, nvalid, st, prof, nullptr, nullptr);
}

static dim3 ew_grid64(long long per_vec, int batch, int z) {
  return dim3((unsigned)std::max<long long>(1, std::min<long long>((per_vec + 255) / 256, 1024)), (unsigned)std::min(batch, 65535), (unsigned)z);
}

static int get_twist64(Engine64 *e, Plan64 *pl, Twist64 **out) {
  auto key = std::make_tuple(pl->p, pl->logL, pl->fp);
  auto it = e->twists.find(key);
  if (it != e->twists.end()) { *out = it->second.get(); return 0; }
  const uint64_t p = pl->p;
  const int logL = pl->logL;
  if (logL + 1 > kMaxLog64 || ((p - 1) & ((2ull << logL) - 1)) != 0)
    return fail64(NTTX_EINVAL, "modulus %llu has no 2^%d-th root of unity for a negacyclic product of length 2^%d",
                  (unsigned long long)p, logL + 1, logL);
  const HostMont64 M{p, pl->ctx.npinv, pl->ctx.one, pl->ctx.r2};
  const uint64_t g = find_nonresidue64(p);
  const uint64_t psi = h_powmod64(g, (p - 1) >> (logL + 1), p), psii = h_powmod64(psi, p - 2, p);
  const size_t len = (size_t)1 << logL;
  std::vector<uint64_t> f(len), b(len);
  if (pl->fp) {                       // plain residues, then (value, value / p) pairs of doubles
    uint64_t cf = 1, cb = 1;
    for (size_t i = 0; i < len; ++i) { f[i] = cf; b[i] = cb; cf = h_mulmod64(cf, psi, p); cb = h_mulmod64(cb, psii, p); }
    f = fp_pairs(f, p); b = fp_pairs(b, p);
  } else {
    const uint64_t sf = M.to(psi), sb = M.to(psii);
    uint64_t cf = M.one, cb = M.one;
    for (size_t i = 0; i < len; ++i) { f[i] = cf; b[i] = cb; cf = M.mul(cf, sf); cb = M.mul(cb, sb); }
  }
  std::unique_ptr<Twist64> tw(new Twist64);
  if (int rc = upload64(&tw->fwd, f)) return rc;
  if (int rc = upload64(&tw->inv, b)) return rc;
  *out = tw.get();
  e->twists[key] = std::move(tw);
  return 0;
}

// c = a b mod (x^len + 1, p): a and b times psi^i, cyclic product, times psi^-i
int engine64_negacyclic_multiply(Engine64 *e, const uint64_t *a, const uint64_t *b, uint64_t *c, int logL, uint64_t p,
                                 int batch, hipStream_t st, const ProfHook64 *prof) {
  Plan64 *pl; Twist64 *tw;
  if (logL < 1) return fail64(NTTX_EINVAL, "length must be at least 2");
  // one- and two-pass plans apply psi^i inside their first and last pass (on the floating-point class where the
  // modulus allows it); three-pass plans use element-wise passes around the product on the integer classes
  static const bool unfused = [] { const char *e_ = getenv("NTTX_TWIST_UNFUSED"); return e_ && atoi(e_) != 0; }();
  if (engine64_fp_ok(p) && !unfused) {
    if (int rc = get_plan64(e, p, logL, false, &pl, true)) return rc;
    if (pl->npass <= 2) {
      if (int rc = get_twist64(e, pl, &tw)) return rc;
      return cyclic_multiply64(e, a, b, c, logL, p, batch, 0, st, prof, tw->fwd, tw->inv, true);
    }
  }
  if (int rc = get_plan64(e, p, logL, false, &pl)) return rc;
  if (int rc = get_twist64(e, pl, &tw)) return rc;
  if (pl->npass <= 2 && !unfused) return cyclic_multiply64(e, a, b, c, logL, p, batch, 0, st, prof, tw->fwd, tw->inv);
  const size_t vec = ((size_t)batch << logL) * 8;
  if (int rc = e->ta.ensure(vec)) return rc;
  if (int rc = e->tb.ensure(vec)) return rc;
  uint64_t *ta = (uint64_t *)e->ta.ptr, *tb = (uint64_t *)e->tb.ptr;
  Twist64Args T{}; T.src[0] = a; T.src[1] = b; T.dst[0] = ta; T.dst[1] = tb; T.tab = tw->fwd; T.ctx = pl->ctx; T.len = 1 << logL; T.canon = 0; T.batch = batch;
  if (prof) prof->begin(prof->self, NTTX_K_MISC, st);
  if (pl->plo1) hipLaunchKernelGGL((twist64_kernel<true>), ew_grid64(T.len / 2, batch, 2), dim3(256), 0, st, T);
  else hipLaunchKernelGGL((twist64_kernel<false>), ew_grid64(T.len / 2, batch, 2), dim3(256), 0, st, T);
  if (prof) prof->end(prof->self, st);
  HIPCHK64(hipGetLastError());
  if (int rc = engine64_cyclic_multiply(e, ta, tb, c, logL, p, batch, 0, st, prof)) return rc;
  Twist64Args U{}; U.src[0] = c; U.dst[0] = c; U.tab = tw->inv; U.ctx = pl->ctx; U.len = 1 << logL; U.canon = 1; U.batch = batch;
  if (prof) prof->begin(prof->self, NTTX_K_MISC, st);
  if (pl->plo1) hipLaunchKernelGGL((twist64_kernel<true>), ew_grid64(U.len / 2, batch, 1), dim3(256), 0, st, U);
  else hipLaunchKernelGGL((twist64_kernel<false>), ew_grid64(U.len / 2, batch, 1), dim3(256), 0, st, U);
  if (prof) prof->end(prof->self, st);
  HIPCHK64(hipGetLastError());
  return 0;
}

int engine64_forward(Engine64 *e, uint64_t *x, int logL, uint64_t p, int inverse, int batch, hipStream_t st,
                     const ProfHook64 *prof) {
  Plan64 *pl;
  if (int rc = get_plan64(e, p, logL, inverse != 0, &pl)) return rc;
  int logN = logL;
  const int ncol = pl->npass - 1;
  for (int lv = 0; lv < ncol; ++lv) {
    ColArgs64 C{}; C.in[0] = x; C.out[0] = x; C.tw = pl->tw_fwd[lv]; C.fs = pl->fs_fwd[lv]; C.ctx = pl->ctx;
    C.logS = logN - pl->lt[lv]; C.canon_out = 0; C.nblocks = ((long long)batch << logL) >> logN;
    if (int rc = launch_col64(pl, false, pl->lt[lv], C, 1, st, prof)) return rc;
    logN = C.logS;
  }
  RowArgs64 R{}; R.in0 = x; R.out = x; R.tw_fwd = pl->tw_fwd[ncol]; R.ctx = pl->ctx; R.canon_out = 1;
  R.nrows = ((long long)batch << logL) >> pl->lt[ncol];
  return launch_row64(pl, 0, pl->lt[ncol], R, st, prof);
}


int engine64_pad(Engine64 *e, const uint64_t *a, const uint64_t *b, uint64_t *pa, uint64_t *pb, int n, int logL,
                 uint64_t p, int batch, hipStream_t st) {
  Plan64 *pl;     // the plan the product will take: single-pass plans read the whole zero-padded row
  if (int rc = get_plan64(e, p, logL, false, &pl, engine64_fp_ok(p))) return rc;
  PadArgs64 P{}; P.src[0] = a; P.src[1] = b; P.dst[0] = pa; P.dst[1] = pb; P.ctx = pl->ctx; P.n = n; P.len = 1 << logL;
  P.zfill = pl->npass == 1 ? P.len : n; P.batch = batch;
  if (pl->plo1) hipLaunchKernelGGL((pad64_kernel<true>), ew_grid64(P.zfill, batch, 2), dim3(256), 0, st, P);
  else hipLaunchKernelGGL((pad64_kernel<false>), ew_grid64(P.zfill, batch, 2), dim3(256), 0, st, P);
  HIPCHK64(hipGetLastError());
  return 0;
}

int engine64_unpad(const uint64_t *src, uint64_t *dst, int nout, int logL, int batch, hipStream_t st) {
  hipLaunchKernelGGL(unpad64_kernel, ew_grid64(nout, batch, 1), dim3(256), 0, st, src, dst, nout, 1 << logL, batch);
  HIPCHK64(hipGetLastError());
  return 0;
}

}  // namespace nttx
