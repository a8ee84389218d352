// libnttx.so: host-side plans, workspace and the C ABI (include/nttx.h) over the
// gfx950 kernels of nttx_kernels.hpp.
//
// What the host side replaces in the reference: the per-call orchestration of
// main.cc:283-341 (allocate, residue split, four single-prime products, CRT) and of
// monttry2.cpp:375-393 (single-prime product), plus the once-per-process constants of
// main.cc:245-270.  Twiddles, which the reference regenerates inside every butterfly
// block by a serial w *= wn chain (main.cc:156-178), live in a plan cache keyed by
// (p, length) and are built once.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/nttx.h"
#include "nttx_arith.hpp"
#include "nttx_kernels.hpp"
#include "nttx64.hpp"
#include "nttx_launch.hpp"

namespace nttx {

// The reference's CRT primes (main.cc:245-248); all have 3 as a quadratic non-residue and
// 2^21 | p-1, so four-prime products support lengths up to 2^21.
static const uint32_t kCrtPrimes[4] = {469762049u, 998244353u, 1004535809u, 1224736769u};

// Scratch of the entry point that allocates nothing (nttx_dev_poly_multiply_u32_static): the two transformed copies
// that the other entry points keep in hipMalloc'ed workspace.  One copy of the array per GPU the library runs on, held
// until the process ends.  Host code never uses the array's name as an address: nttx_init asks the runtime for the
// device address (hipGetSymbolAddress) and the kernels get that pointer as an argument.
constexpr long long kStaticWords = 1ll << 22;            // words per copy: batch * length may not exceed this (16 MiB)
static __device__ __align__(256) uint32_t g_static_scratch[2 * kStaticWords];

static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}
// (a failed call also leaves its code in the runtime's last-error slot; it is cleared here so that a later launch
// check does not report it a second time)
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { (void)hipGetLastError(); \
  return fail(NTTX_EDEVICE, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } } while (0)

// ---- grow-only device buffer ----------------------------------------------------------
struct DevBuf {
  void *ptr = nullptr; size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (ptr) { HIPCHK(hipFree(ptr)); ptr = nullptr; cap = 0; }
    size_t want = (bytes + 255) & ~(size_t)255;
    HIPCHK(hipMalloc(&ptr, want));
    cap = want;
    return 0;
  }
  void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
  template <class T> T *as() const { return reinterpret_cast<T *>(ptr); }
};

// ---- plan: decomposition + device tables for one (p, log2 length, root direction) -----
struct Plan {
  uint32_t p = 0; int logL = 0; bool conj = false; int hclass = 2;  // multiples of p below 2^32: 2, 4 or 9
  PrimeCtx ctx{};
  int npass = 0; int lt[3] = {0, 0, 0};     // lt[0..npass-2] column passes, lt[npass-1] the row pass
  uint32_t *tw_fwd[3] = {nullptr, nullptr, nullptr};  // stage-major tables per pass, root w
  uint32_t *tw_inv[3] = {nullptr, nullptr, nullptr};  // same for w^{-1}
  uint32_t *fs_fwd[2] = {nullptr, nullptr};           // four-step tables per column pass
  uint32_t *fs_inv[2] = {nullptr, nullptr};           // inverse; level 0 also carries n^{-1} R
  // three-pass lengths: the outer pass's factors as a product of two small tables (fs_* = low part)
  uint32_t *fsa_fwd = nullptr, *fsa_inv = nullptr; int logC2 = 0;
  uint32_t scale_single = 0;                          // n^{-1} R^2 mod p (single-pass products)
  ~Plan() {
    for (auto q : tw_fwd) if (q) (void)hipFree(q);
    for (auto q : tw_inv) if (q) (void)hipFree(q);
    for (auto q : fs_fwd) if (q) (void)hipFree(q);
    for (auto q : fs_inv) if (q) (void)hipFree(q);
    if (fsa_fwd) (void)hipFree(fsa_fwd);
    if (fsa_inv) (void)hipFree(fsa_inv);
  }
};

// Tables of one column pass over a slab of columns (a transform shared by several processes):
// the stage table of the length-T column transforms and the four-step factors of the slab.
struct Slab {
  uint32_t *tw = nullptr, *fs = nullptr, *fsa = nullptr;   // fsa != null: factors split over two tables
  int logC2 = 0;
  PrimeCtx ctx{};
  int hclass = 2;
  ~Slab() { if (tw) (void)hipFree(tw); if (fs) (void)hipFree(fs); if (fsa) (void)hipFree(fsa); }
};

// psi^i and psi^-i for i < 2^logL, psi a primitive 2^(logL+1)-th root of unity (negacyclic products)
struct Twist {
  uint32_t *fwd = nullptr, *inv = nullptr;
  ~Twist() { if (fwd) (void)hipFree(fwd); if (inv) (void)hipFree(inv); }
};

// Arithmetic class of the kernels for a modulus: the number of multiples of p known to fit an unsigned lane.
static int arith_class(uint32_t p) { return (9ull * p < (1ull << 32)) ? 9 : (p < (1u << 30)) ? 4 : 2; }
// The fused 512-point row kernel runs on signed lanes where 4p < 2^31 (NTTX_UNSIGNED=1: A/B measurements).
static bool signed_rows(uint32_t p) {
  static const bool unsigned_only = [] { const char *e = getenv("NTTX_UNSIGNED"); return e && atoi(e) != 0; }();
  return p < (1u << 29) && !unsigned_only;
}
// ... and with sixteen multiples of headroom where 16p < 2^31 (NTTX_WIDE=0: the four-multiple kernels for every p < 2^29)
static bool wide_rows(uint32_t p) {
  static const bool narrow_only = [] { const char *e = getenv("NTTX_WIDE"); return e && atoi(e) == 0; }();
  return p < (1u << 27) && signed_rows(p) && !narrow_only;
}

// kMaxLog: no prime below 2^31 has 2^28 | p - 1 (the longest direct transform any modulus of these lanes admits is 2^27,
// modulo 2013265921 = 15 * 2^27 + 1), so the plans stop there.
constexpr int kMaxRowLog = 11, kMinColLog = 6, kMaxColLog = 9, kMaxLog = 27;
constexpr int kMinLongRow = 12, kMaxLongRow = 14;   // rows handled by one whole workgroup (longrow kernels)

// NTTX_LONGROW (A/B measurements): bit 0 = long rows behind one column pass (lengths 2^21 .. 2^23), bit 1 = lengths 2^12 .. 2^14 in one pass
static int longrow_mode() { static const int m = [] { const char *e = getenv("NTTX_LONGROW"); return e ? atoi(e) : 3; }(); return m; }

static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }

static int decompose(int k, int lt[3]) {
  if (k <= kMaxRowLog) { lt[0] = k; return 1; }
  if (k <= kMaxLongRow && (longrow_mode() & 2)) { lt[0] = k; return 1; }
  if (k <= 9 + kMaxRowLog) {
    // 2^9-point column passes run on full 128-byte tiles (2^10-point ones only on 64-byte tiles), and the
    // 512-point row pass is the wave-per-row kernel: lengths 2^15 .. 2^18 put 9 stages in the row pass
    // (2 % faster at 2^15 and 2^16 than an even split), longer ones in the column pass
    lt[0] = (k <= 18) ? std::max(kMinColLog, k - 9) : 9;
    lt[1] = k - lt[0];
    return 2;
  }
  if (longrow_mode() & 1) {
    // (a 2^10-point column pass around 2^14-point rows measured slower at 2^24 than three passes)
    if (k <= 9 + kMaxLongRow) { lt[0] = 9; lt[1] = k - 9; return 2; }
  } else if (k <= kMaxColLog + kMaxRowLog) {
    lt[0] = std::min(kMaxColLog, k / 2); lt[1] = k - lt[0];
    return 2;
  }
  // the 512-point row pass is the fused wave-per-row kernel: prefer it while both column passes stay >= 2^7 (the outer
  // pass uses the split factor tables, which exist for 2^7 .. 2^9 points)
  int row = (k + 2) / 3; if (row > kMaxRowLog) row = kMaxRowLog;
  if (row < 9 && k - 9 >= 2 * 7) row = 9;
  int rem = k - row;
  // the smaller half goes to the outer pass, whose rows lie far apart (2^26 as 2^8 | 2^9 | 2^9: 964 us against 1018 us
  // with the 2^9-point pass outside, whose split-table kernel has half the occupancy; 2^25 as 2^8 | 2^8 | 2^9 either way)
  lt[0] = rem / 2; lt[1] = rem - lt[0]; lt[2] = row;
  return 3;
}

// per-kernel timing with HIP events on the launch stream (bench.py's roofline figure)
enum ProfKind { PK_COL_FWD = 0, PK_ROW_MUL, PK_COL_INV, PK_ROW_FWD, PK_ROW_INV, PK_SPLIT, PK_CRT, PK_MISC, PK_COUNT };
struct ProfRec { hipEvent_t e0, e1; int kind; };
struct Profiler {
  bool on = false;
  std::vector<ProfRec> pending;
  std::vector<hipEvent_t> pool;
  double ms[PK_COUNT] = {0};
  long long launches[PK_COUNT] = {0};
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
  }
  void begin(int kind, hipStream_t st) {
    if (!on) return;
    ProfRec r{get(), get(), kind};
    (void)hipEventRecord(r.e0, st);
    pending.push_back(r);
  }
  void end(hipStream_t st) { if (on && !pending.empty()) (void)hipEventRecord(pending.back().e1, st); }
  void fold() {
    for (auto &r : pending) {
      (void)hipEventSynchronize(r.e1);
      float t = 0; if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) { ms[r.kind] += t; launches[r.kind] += 1; }
      pool.push_back(r.e0); pool.push_back(r.e1);
    }
    pending.clear();
  }
  void reset() { fold(); for (int i = 0; i < PK_COUNT; ++i) { ms[i] = 0; launches[i] = 0; } }
  void destroy() { fold(); for (auto e : pool) (void)hipEventDestroy(e); pool.clear(); }
};

struct Context {
  int device = -1;
  Profiler prof;
  std::map<std::tuple<uint32_t, int, bool>, std::unique_ptr<Plan>> plans;
  std::map<std::tuple<uint32_t, int, int, long long, int, bool>, std::unique_ptr<struct Slab>> slabs;
  std::map<std::pair<uint32_t, int>, std::unique_ptr<struct Twist>> twists;
  // workspace
  DevBuf wa, wb;            // transformed copies of a and b (multi-pass direct path) [batch][len]
  DevBuf in_a, in_b;        // raw host inputs staged on device
  DevBuf pa, pb, pc;        // padded residues / product [nprimes][batch][len]
  DevBuf outb;              // compacted output staged for the copy back
  DevBuf tmp;               // transform API scratch
  Engine64 *e64 = nullptr;  // plans and workspace of the 64-bit-lane direct path
  // stream of the last call that touched the workspace, and the event that orders the next stream behind it
  hipStream_t last_stream = nullptr; bool have_last = false; hipEvent_t ws_event = nullptr;
  uint32_t *static_ws = nullptr;   // device address of g_static_scratch on this context's GPU
  // CRT route: the products modulo the second to fourth prime run on these streams beside the first prime's
  hipStream_t side[3] = {nullptr, nullptr, nullptr};
  hipEvent_t fork_ev = nullptr, join_ev[3] = {nullptr, nullptr, nullptr};
  DevBuf wsx;                      // their transformed copies (the first prime uses wa / wb)
};
// One context per GPU of this process (nttx_init: one; nttx_init_devices: a list).  g_ctx is the context the calls of
// this process act on (nttx_set_device switches it); nttx_poly_multiply_batch_u32 deals its products over all of them.
static std::vector<Context *> g_ctxs;
static Context *g_ctx = nullptr;
static std::mutex g_init_mu;

// Every call of the context shares one workspace (transformed copies, residue buffers, staging, the 64-bit engine's
// buffers), and the mutex only serialises the host side.  A call enqueued on another stream than the previous one
// therefore first waits, on the device, for everything the previous call's stream holds.
static int enter_stream(Context *cx, hipStream_t st) {
  if (cx->have_last && st != cx->last_stream) {
    HIPCHK(hipEventRecord(cx->ws_event, cx->last_stream));
    HIPCHK(hipStreamWaitEvent(st, cx->ws_event, 0));
  }
  cx->last_stream = st; cx->have_last = true;
  return 0;
}

// smallest quadratic non-residue, preferring the reference's root 3 (main.cc:241)
static uint32_t find_nonresidue(uint32_t p) {
  const uint32_t cand[] = {3, 2, 5, 6, 7, 10, 11, 13, 14, 15, 17, 19, 21, 22, 23, 26, 29, 31, 37, 41, 43, 47};
  for (uint32_t g : cand) if (g % p && h_powmod(g, (p - 1) / 2, p) == p - 1) return g;
  for (uint32_t g = 48; g < p && g < 100000; ++g) if (h_powmod(g, (p - 1) / 2, p) == p - 1) return g;
  return 0;
}

static bool direct_ok(uint64_t p, int logL) {
  if (p < 3 || p >= (1ull << 31) || (p & 1) == 0) return false;
  if (logL > kMaxLog) return false;
  if (((p - 1) & ((1ull << logL) - 1)) != 0) return false;
  if (!is_prime_u64(p)) return false;
  return find_nonresidue((uint32_t)p) != 0;
}

static int upload(uint32_t **dst, const std::vector<uint32_t> &h) {
  HIPCHK(hipMalloc((void **)dst, h.size() * sizeof(uint32_t)));
  HIPCHK(hipMemcpy(*dst, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  return 0;
}

// stage-major table for a length-T transform with primitive T-th root w (ordinary form):
// tab[h + j] = w^(j * T/(2h)) in Montgomery form, h = 1, 2, ..., T/2
static std::vector<uint32_t> stage_table(uint32_t w, int logT, const PrimeCtx &c) {
  const int T = 1 << logT;
  std::vector<uint32_t> tab(std::max(T, 2), 0);
  for (int h = 1; h < T; h <<= 1) {
    const uint32_t step = h_powmod(w, (uint64_t)(T / (2 * h)), c.p);
    uint32_t cur = 1;
    for (int j = 0; j < h; ++j) { tab[h + j] = to_mont(cur, c); cur = h_mulmod(cur, step, c.p); }
  }
  return tab;
}

static inline uint32_t bitrev(uint32_t x, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

// four-step table for blocks of N = T*S words: tab[j*S + s] = wN^(s * bitrev_T(j)) * extra, Montgomery form
// col0 > 0: the table of columns col0 .. col0 + S - 1 of a wider matrix (s counts from col0)
static std::vector<uint32_t> fourstep_table(uint32_t wN, int logT, int logS, uint32_t extra, const PrimeCtx &c,
                                            uint64_t col0 = 0) {
  const size_t T = (size_t)1 << logT, S = (size_t)1 << logS;
  std::vector<uint32_t> tab(T * S);
  for (size_t j = 0; j < T; ++j) {
    const uint32_t wj = h_powmod(wN, bitrev((uint32_t)j, logT), c.p);
    const uint32_t step = to_mont(wj, c);
    uint32_t cur = to_mont(h_mulmod(extra, h_powmod(wj, col0, c.p), c.p), c);  // Montgomery form of extra * wj^(col0+s)
    uint32_t *row = &tab[j * S];
    for (size_t s = 0; s < S; ++s) { row[s] = cur; cur = mont_mul(cur, step, c.p, c.npinv); }
  }
  return tab;
}

static int get_plan(Context *cx, uint32_t p, int logL, bool conj, Plan **out) {
  auto key = std::make_tuple(p, logL, conj);
  auto it = cx->plans.find(key);
  if (it != cx->plans.end()) { *out = it->second.get(); return 0; }
  if (!direct_ok(p, logL)) return fail(NTTX_EINVAL, "modulus %u does not support a direct length-2^%d transform", p, logL);
  std::unique_ptr<Plan> pl(new Plan);
  pl->p = p; pl->logL = logL; pl->conj = conj;
  pl->hclass = arith_class(p);
  pl->ctx = make_ctx(p);
  pl->npass = decompose(logL, pl->lt);
  const uint32_t g = find_nonresidue(p);
  uint32_t wL = h_powmod(g, (uint64_t)(p - 1) >> logL, p);        // primitive 2^logL-th root (main.cc:156)
  if (conj) wL = h_powmod(wL, p - 2, p);                          // main.cc:157
  const uint32_t wLi = h_powmod(wL, p - 2, p);
  const uint64_t Lfull = 1ull << logL;
  const uint32_t ninv = h_powmod((uint32_t)(Lfull % p), p - 2, p);  // main.cc:204
  const uint32_t ninvR = to_mont(ninv, pl->ctx);                    // n^{-1} R mod p
  pl->scale_single = to_mont(ninvR, pl->ctx);                       // n^{-1} R^2 mod p
  int logN = logL;                                                  // block length handled by this level
  for (int lv = 0; lv < pl->npass; ++lv) {
    const int lt = pl->lt[lv];
    const uint32_t wN = h_powmod(wL, Lfull >> logN, p), wNi = h_powmod(wLi, Lfull >> logN, p);
    const uint32_t wT = h_powmod(wN, (1ull << logN) >> lt, p), wTi = h_powmod(wNi, (1ull << logN) >> lt, p);
    if (int rc = upload(&pl->tw_fwd[lv], stage_table(wT, lt, pl->ctx))) return rc;
    if (int rc = upload(&pl->tw_inv[lv], stage_table(wTi, lt, pl->ctx))) return rc;
    if (lv < pl->npass - 1) {
      const int logS = logN - lt;
      if (lv == 0 && pl->npass == 3) {
        // w^(s r) = w^(h C r) * w^(l r) for s = h C + l: T*(S/C + C) words instead of T*S
        const int c2 = (logS + 1) / 2;
        pl->logC2 = c2;
        const uint32_t wC = h_powmod(wN, 1ull << c2, p), wCi = h_powmod(wNi, 1ull << c2, p);
        if (int rc = upload(&pl->fs_fwd[lv], fourstep_table(wN, lt, c2, 1, pl->ctx))) return rc;
        if (int rc = upload(&pl->fs_inv[lv], fourstep_table(wNi, lt, c2, 1, pl->ctx))) return rc;
        if (int rc = upload(&pl->fsa_fwd, fourstep_table(wC, lt, logS - c2, 1, pl->ctx))) return rc;
        if (int rc = upload(&pl->fsa_inv, fourstep_table(wCi, lt, logS - c2, ninvR, pl->ctx))) return rc;
      } else {
        if (int rc = upload(&pl->fs_fwd[lv], fourstep_table(wN, lt, logS, 1, pl->ctx))) return rc;
        if (int rc = upload(&pl->fs_inv[lv], fourstep_table(wNi, lt, logS, lv == 0 ? ninvR : 1, pl->ctx))) return rc;
      }
      logN = logS;
    }
  }
  *out = pl.get();
  cx->plans[key] = std::move(pl);
  return 0;
}

// ---- kernel dispatch -------------------------------------------------------------------
static int grid_for(long long tiles, int cap) { return (int)std::max<long long>(1, std::min<long long>(tiles, cap)); }

template <int MODE, int H, bool TWIST = false>
static int launch_row_t(int logT, const RowArgs &A, hipStream_t st) {
  if constexpr (MODE == ROW_MUL && !TWIST) {
    if (A.psi_in) return launch_row_t<MODE, H, true>(logT, A, st);   // negacyclic product: psi factors fused
  }
  if (MODE == ROW_MUL && logT == 9) {   // wave-per-row fused kernel
    const long long wgs = (A.nrows + (WG / 64) - 1) / (WG / 64);
    static const int cap9 = env_int("NTTX_GRID_ROW", 8192);   // 4 rows per wave at batch 256: measured optimum
    const dim3 grid(grid_for(wgs, cap9));
    if constexpr (TWIST) {
      if (H == 9 && wide_rows(A.ctx.p)) hipLaunchKernelGGL((row_mul9_kernel<17, true, true>), grid, dim3(WG), 0, st, A);
      else if (H == 9 && signed_rows(A.ctx.p)) hipLaunchKernelGGL((row_mul9_kernel<1, true, true>), grid, dim3(WG), 0, st, A);
      else hipLaunchKernelGGL((row_mul9_kernel<H, true, true>), grid, dim3(WG), 0, st, A);
    } else if (H == 9 && wide_rows(A.ctx.p)) {
      if (A.canon_out) hipLaunchKernelGGL((row_mul9_kernel<17, true>), grid, dim3(WG), 0, st, A);
      else hipLaunchKernelGGL((row_mul9_kernel<17, false>), grid, dim3(WG), 0, st, A);
    } else if (H == 9 && signed_rows(A.ctx.p)) {
      if (A.canon_out) hipLaunchKernelGGL((row_mul9_kernel<1, true>), grid, dim3(WG), 0, st, A);
#ifdef NTTX_BUILD_PROBES
      // measurement builds only (make CXXFLAGS+=-DNTTX_BUILD_PROBES): NTTX_ROW9_PROBE=1 / 2 selects the probe variants of
      // this kernel (one more LDS trip / 24 DPP moves per vector); the shipped library does not contain them
      else if (env_int("NTTX_ROW9_PROBE", 0) == 1) hipLaunchKernelGGL((row_mul9_kernel<1, false, false, 1>), grid, dim3(WG), 0, st, A);
      else if (env_int("NTTX_ROW9_PROBE", 0) == 2) hipLaunchKernelGGL((row_mul9_kernel<1, false, false, 2>), grid, dim3(WG), 0, st, A);
#endif
      else hipLaunchKernelGGL((row_mul9_kernel<1, false>), grid, dim3(WG), 0, st, A);
    } else if (A.canon_out) hipLaunchKernelGGL((row_mul9_kernel<H, true>), grid, dim3(WG), 0, st, A);
    else hipLaunchKernelGGL((row_mul9_kernel<H, false>), grid, dim3(WG), 0, st, A);
    HIPCHK(hipGetLastError());
    return 0;
  }
  // one row per workgroup: fused products from 2^10 points on (faster there than the tiled kernel
  // below: one LDS buffer, outer rounds on registers), single transforms from 2^12 on
  if (logT >= (MODE == ROW_MUL ? 10 : kMinLongRow) && logT <= kMaxLongRow) {
    static const int capl = env_int("NTTX_GRID_LROW", 1 << 20);
    const int grid = grid_for(A.nrows, capl);
#define NTTX_LROW_CASE(K) case K: { \
      constexpr int NT_ = LongRowLayout<K>::NTHR, NTM_ = LongRowLayout<K, long_row_wide(K)>::NTHR; \
      if constexpr (MODE != ROW_MUL) { if constexpr (K >= kMinLongRow) hipLaunchKernelGGL((longrow_kernel<K, ROW_FWD, H>), dim3(grid), dim3(NT_), 0, st, A); } \
      else if (H == 9 && wide_rows(A.ctx.p)) hipLaunchKernelGGL((longrow_mul_kernel<K, 16, TWIST>), dim3(grid), dim3(NTM_), 0, st, A); \
      else if (H == 9 && signed_rows(A.ctx.p)) hipLaunchKernelGGL((longrow_mul_kernel<K, 0, TWIST>), dim3(grid), dim3(NTM_), 0, st, A); \
      else hipLaunchKernelGGL((longrow_mul_kernel<K, H, TWIST>), dim3(grid), dim3(NTM_), 0, st, A); \
      break; }
    switch (logT) { NTTX_LROW_CASE(10) NTTX_LROW_CASE(11) NTTX_LROW_CASE(12) NTTX_LROW_CASE(13) NTTX_LROW_CASE(14) default: break; }
#undef NTTX_LROW_CASE
    HIPCHK(hipGetLastError());
    return 0;
  }
  /* (fused products of 2^9 points and more never get here: the kernels above take them) */
#define NTTX_ROW_CASE(K) case K: { \
    using Lyt = RowLayout<K>; \
    const long long tiles = (A.nrows + Lyt::NT - 1) / Lyt::NT; \
    if constexpr (MODE != ROW_MUL || K < 9) hipLaunchKernelGGL((row_kernel<K, MODE, H, TWIST>), dim3(grid_for(tiles, capg)), dim3(WG), 0, st, A); \
    break; }
  static const int capg = env_int("NTTX_GRID_ROWG", 8192);
  switch (logT) {
    NTTX_ROW_CASE(1) NTTX_ROW_CASE(2) NTTX_ROW_CASE(3) NTTX_ROW_CASE(4) NTTX_ROW_CASE(5) NTTX_ROW_CASE(6)
    NTTX_ROW_CASE(7) NTTX_ROW_CASE(8) NTTX_ROW_CASE(9) NTTX_ROW_CASE(10) NTTX_ROW_CASE(11)
    default: return fail(NTTX_ERANGE, "row pass of 2^%d points unsupported", logT);
  }
#undef NTTX_ROW_CASE
  HIPCHK(hipGetLastError());
  return 0;
}
template <int MODE>
static int launch_row_m(int h, int logT, const RowArgs &A, hipStream_t st) {
  return h == 9 ? launch_row_t<MODE, 9>(logT, A, st) : h == 4 ? launch_row_t<MODE, 4>(logT, A, st) : launch_row_t<MODE, 2>(logT, A, st);
}
// (an inverse transform runs the forward kernels with the conjugate root's tables, so no caller asks for ROW_INV)
static int launch_row_raw(int mode, int h, int logT, const RowArgs &A, hipStream_t st) {
  if (mode == ROW_FWD) return launch_row_m<ROW_FWD>(h, logT, A, st);
  if (mode == ROW_MUL) return launch_row_m<ROW_MUL>(h, logT, A, st);
  return fail(NTTX_EINVAL, "row pass mode %d unsupported", mode);
}

// column passes are compiled in their own translation unit (nttx_col.hip)
static int launch_col_raw(bool inv, int h, int logT, const ColArgs &A, int nvec, hipStream_t st) {
  const char *msg = nullptr;
  const int rc = launch_col_kernels(inv, h, logT, A, nvec, st, &msg);
  return rc ? fail(rc, "%s (logT=%d logS=%d)", msg ? msg : "column pass failed", logT, A.logS) : 0;
}

static int launch_row(Context *cx, int mode, int h, int logT, const RowArgs &A, hipStream_t st) {
  cx->prof.begin(mode == ROW_MUL ? PK_ROW_MUL : mode == ROW_FWD ? PK_ROW_FWD : PK_ROW_INV, st);
  const int rc = launch_row_raw(mode, h, logT, A, st);
  cx->prof.end(st);
  return rc;
}
static int launch_col(Context *cx, bool inv, int h, int logT, const ColArgs &A0, int nvec, hipStream_t st) {
  const ColArgs &A = A0;
  cx->prof.begin(inv ? PK_COL_INV : PK_COL_FWD, st);
  const int rc = launch_col_raw(inv, h, logT, A, nvec, st);
  cx->prof.end(st);
  return rc;
}

static bool aligned16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
static void set_vec(SplitArgs &S) {
  bool ok = (S.n & 3) == 0 && aligned16(S.src[0]) && aligned16(S.src[1]);
  for (int k = 0; k < S.nmod; ++k) ok = ok && aligned16(S.dst[0][k]) && aligned16(S.dst[1][k]);
  S.vec = ok ? 1 : 0;
}
static void set_vec(CrtArgs &A) {
  bool ok = (A.len & 3) == 0;
  for (int k = 0; k < A.nmod; ++k) ok = ok && aligned16(A.res[k]);
  A.vec = ok ? 1 : 0;
}
// grid of the element-wise kernels: x covers one vector, y the batch (the kernels stride over it, so batches beyond
// the 65535 a grid dimension may hold are legal)
static dim3 ew_grid2(long long per_vec, int batch, int z = 1) {
  return dim3((unsigned)std::max<long long>(1, std::min<long long>((per_vec + 255) / 256, 1024)), (unsigned)std::min(batch, 65535), (unsigned)z);
}
static int ew_grid(long long total) { return grid_for((total + 255) / 256, 256 * 16); }

// ---- device-resident building blocks ---------------------------------------------------

// c = cyclic product of a and b, all [batch][2^logL] u32 on the device
// nvalid > 0: only the first nvalid words of every input vector are read, the rest counts as zero
// (the zero padding of a linear product, main.cc:280-281); needs nvalid <= len/2 to take effect.
// psi_in / psi_out (one- and two-pass plans only): the product becomes the negacyclic one, a and b multiplied by
// psi_in[i] as the first pass reads them and c by psi_out[i] as the last pass writes it
// ws != nullptr: the two transformed copies go to ws[0 .. batch*len) and ws[batch*len .. 2*batch*len) instead of the
// context's grow-only workspace (the caller has checked the size), and nothing is allocated
// src64 != nullptr (two-pass plans of linear products): a and b are not read; the first forward pass takes the caller's
// [batch][nvalid] u64 coefficients instead and reduces them modulo the plan's prime as it loads them (the residue
// split of main.cc:292-303 without a pass of its own)
struct Src64 { const uint64_t *a, *b; CoefMod cm; };
static CoefMod coef_mod(uint64_t pmod) {
  CoefMod m{pmod, 0};
  if (pmod) m.mu = (uint64_t)((((unsigned __int128)1) << 64) / pmod);
  return m;
}
// plans whose first forward pass has a u64-source variant: full-line zero-half tiles of 2^6 .. 2^9 points
static bool src64_ok(const Plan *pl) { return pl->npass == 2 && pl->lt[0] >= 6 && pl->lt[0] <= 9 && pl->logL - pl->lt[0] >= 5; }
static int dev_cyclic_multiply(Context *cx, Plan *pl, const uint32_t *a, const uint32_t *b, uint32_t *c, int batch,
                               hipStream_t st, long long nvalid = 0, const uint32_t *psi_in = nullptr,
                               const uint32_t *psi_out = nullptr, uint32_t *ws = nullptr, int pieces = 0,
                               int out_slices = 0, long long out_slice_stride = 0, const Src64 *src64 = nullptr) {
  const long long len = 1ll << pl->logL;
  if (src64 && (!src64_ok(pl) || nvalid <= 0 || 2 * nvalid > len || psi_in || pieces > 0))
    return fail(NTTX_ERANGE, "u64 source needs a linear product on a two-pass plan");
  if (psi_in && pl->npass > 2) return fail(NTTX_ERANGE, "fused twist needs a one- or two-pass plan");
  // pieces > 0: a, b and c are [pieces][batch][len / pieces] arrays (vector r = the concatenation of its pieces), the
  // layout in which an all-to-all between processes delivers rows and takes them back: the first forward pass reads
  // it and the last inverse pass writes it, so no reordering copy is needed on either side
  // out_slices > 0 (CRT residues for several processes): only c is in pieces form, slice q of every vector going to
  // c + q * out_slice_stride (slice q = what process q recombines); a and b are ordinary [batch][len] arrays
  int piece_log = 0;
  if (pieces > 0) {
    int lg = 0; while ((1 << lg) < pieces) ++lg;
    if ((1 << lg) != pieces || pl->npass != 2 || lg > pl->lt[0] || pl->lt[0] > 8 || psi_in)
      return fail(NTTX_ERANGE, "pieces layout needs a two-pass length (2^15 .. 2^17) and a power-of-two number of pieces up to 2^%d", pl->lt[0]);
    piece_log = pl->lt[0] - lg;
    out_slices = pieces; out_slice_stride = (long long)batch * (len / pieces);
  } else if (out_slices > 0) {
    int lg = 0; while ((1 << lg) < out_slices) ++lg;
    if ((1 << lg) != out_slices || pl->npass != 2 || lg > pl->lt[0] || psi_in)
      return fail(NTTX_ERANGE, "sliced output needs a two-pass length and a power-of-two number of slices up to 2^%d", pl->lt[0]);
    piece_log = pl->lt[0] - lg;
  }
  if (pl->npass == 1) {
    RowArgs R{}; R.in0 = a; R.in1 = b; R.out = c; R.tw_fwd = pl->tw_fwd[0]; R.tw_inv = pl->tw_inv[0];
    R.ctx = pl->ctx; R.scale = pl->scale_single; R.canon_out = 1; R.nrows = batch;
    R.psi_in = psi_in; R.psi_out = psi_out;
    return launch_row(cx, ROW_MUL, pl->hclass, pl->lt[0], R, st);
  }
  // One launch per pass over the whole resident batch.  (Cutting the batch into cache-sized launch
  // groups, or issuing groups on two streams so column and row passes overlap, both measured slower
  // on MI355X: short launches lose more than residency or overlap wins.)
  const size_t wsz = (size_t)batch * len * 4;
  uint32_t *wa = ws, *wb = ws ? ws + (size_t)batch * len : nullptr;
  if (!ws) {
    if (int rc = cx->wa.ensure(wsz)) return rc;
    if (int rc = cx->wb.ensure(wsz)) return rc;
    wa = cx->wa.as<uint32_t>(); wb = cx->wb.as<uint32_t>();
  }
  const int ncol = pl->npass - 1;
  int logN = pl->logL;
  for (int lv = 0; lv < ncol; ++lv) {           // forward column passes on a and b
    ColArgs C{}; C.in[0] = lv == 0 ? a : wa; C.in[1] = lv == 0 ? b : wb; C.out[0] = wa; C.out[1] = wb;
    C.tw = pl->tw_fwd[lv]; C.fs = pl->fs_fwd[lv]; C.ctx = pl->ctx; C.logS = logN - pl->lt[lv]; C.canon_out = 0;
    if (lv == 0) { C.fsa = pl->fsa_fwd; C.logC2 = pl->logC2; }
    C.nblocks = ((long long)batch << pl->logL) >> logN;
    C.nvalid = (lv == 0 && nvalid > 0 && 2 * nvalid <= len) ? nvalid : 0;
    if (lv == 0 && psi_in) { C.psi = psi_in; C.nvalid = 0; }
    if (lv == 0 && pieces > 0) {
      C.piece_log = piece_log; C.block_stride = len / pieces; C.piece_stride = (long long)batch * C.block_stride; C.nvalid = 0;
    }
    if (lv == 0 && src64) {
      C.in[0] = C.in[1] = nullptr; C.in64[0] = src64->a; C.in64[1] = src64->b; C.cm = src64->cm;
      C.vec64 = ((nvalid & 1) == 0 && aligned16(src64->a) && aligned16(src64->b)) ? 1 : 0;
    }
    if (int rc = launch_col(cx, false, pl->hclass, pl->lt[lv], C, 2, st)) return rc;
    logN = C.logS;
  }
  {                                              // rows: forward a, forward b, pointwise, inverse
    RowArgs R{}; R.in0 = wa; R.in1 = wb; R.out = out_slices > 0 ? wa : c; R.tw_fwd = pl->tw_fwd[ncol]; R.tw_inv = pl->tw_inv[ncol];
    R.ctx = pl->ctx; R.scale = 0; R.canon_out = 0; R.nrows = ((long long)batch << pl->logL) >> pl->lt[ncol];
    if (int rc = launch_row(cx, ROW_MUL, pl->hclass, pl->lt[ncol], R, st)) return rc;
  }
  for (int lv = ncol - 1; lv >= 0; --lv) {       // inverse column passes, innermost level first
    int logNl = pl->logL; for (int q = 0; q < lv; ++q) logNl -= pl->lt[q];
    ColArgs C{}; C.in[0] = out_slices > 0 ? wa : c; C.in[1] = nullptr; C.out[0] = c; C.out[1] = nullptr;
    if (lv == 0 && out_slices > 0) { C.piece_log = piece_log; C.block_stride = len / out_slices; C.piece_stride = out_slice_stride; }
    C.tw = pl->tw_inv[lv]; C.fs = pl->fs_inv[lv]; C.ctx = pl->ctx; C.logS = logNl - pl->lt[lv]; C.canon_out = (lv == 0);
    if (lv == 0) { C.fsa = pl->fsa_inv; C.logC2 = pl->logC2; }
    C.nblocks = ((long long)batch << pl->logL) >> logNl;
    if (lv == 0 && psi_out) C.psi = psi_out;
    if (int rc = launch_col(cx, true, pl->hclass, pl->lt[lv], C, 1, st)) return rc;
  }
  return 0;
}

// forward transform (root w of the plan), natural in -> bit-reversed out, in place on x
static int dev_forward(Context *cx, Plan *pl, uint32_t *x, int batch, bool canon, hipStream_t st) {
  int logN = pl->logL;
  const int ncol = pl->npass - 1;
  for (int lv = 0; lv < ncol; ++lv) {
    ColArgs C{}; C.in[0] = x; C.out[0] = x; C.tw = pl->tw_fwd[lv]; C.fs = pl->fs_fwd[lv]; C.ctx = pl->ctx;
    if (lv == 0) { C.fsa = pl->fsa_fwd; C.logC2 = pl->logC2; }
    C.logS = logN - pl->lt[lv]; C.canon_out = 0; C.nblocks = ((long long)batch << pl->logL) >> logN;
    if (int rc = launch_col(cx, false, pl->hclass, pl->lt[lv], C, 1, st)) return rc;
    logN = C.logS;
  }
  RowArgs R{}; R.in0 = x; R.out = x; R.tw_fwd = pl->tw_fwd[ncol]; R.ctx = pl->ctx; R.canon_out = canon;
  R.nrows = ((long long)batch << pl->logL) >> pl->lt[ncol];
  return launch_row(cx, ROW_FWD, pl->hclass, pl->lt[ncol], R, st);
}

// ---- CRT constants (main.cc:256-270 computes M, K_i = M/m_i, K_i^{-1} mod m_i; the Garner
// form needs m_i^{-1} mod m_j for i < j instead) ------------------------------------------
struct CrtConst { PrimeCtx ctx[4]; uint32_t inv[4][4]; unsigned __int128 M; };
static const CrtConst &crt_const() {
  static CrtConst cc = [] {
    CrtConst c{}; c.M = 1;
    for (int i = 0; i < 4; ++i) { c.ctx[i] = make_ctx(kCrtPrimes[i]); c.M *= kCrtPrimes[i]; }
    for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) {
      const uint32_t mj = kCrtPrimes[j];
      c.inv[i][j] = to_mont(h_powmod(kCrtPrimes[i] % mj, mj - 2, mj), c.ctx[j]);
    }
    return c;
  }();
  return cc;
}

// constants for recombining residues modulo kCrtPrimes[sel[0..ns-1]] (ascending)
static int fill_crt_args(CrtArgs &A, uint64_t p, const int *sel, int ns) {
  const CrtConst &cc = crt_const();
  for (int k = 0; k < ns; ++k) { A.ctx[k] = cc.ctx[sel[k]]; for (int j = 0; j < ns; ++j) A.inv[k][j] = cc.inv[sel[k]][sel[j]]; }
  for (int k = ns; k < 4; ++k) A.ctx[k] = cc.ctx[3];   // unused slots: any valid modulus
  A.nmod = ns;
  if (p < 2 || p >= (1ull << 63)) return fail(NTTX_EINVAL, "modulus out of range");
  int e = 0; uint64_t q = p;
  while ((q & 1) == 0) { q >>= 1; ++e; }             // p = 2^e * q, q odd
  A.e = e;
  A.pctx = make_ctx64(q);                            // for q == 1 the odd part is skipped by the kernel
  A.qinv = 0ull - A.pctx.npinv;                      // q^{-1} mod 2^64
  if (q > 1) {
    uint64_t c = (uint64_t)((((unsigned __int128)1) << 64) % q);     // 2^64 mod q
    for (int k = 0; k < ns; ++k) { A.cR[k] = c; c = h_mulmod64(c, kCrtPrimes[sel[k]] % q, q); }
  }
  return 0;
}
static int fill_crt_args(CrtArgs &A, uint64_t p) { const int all[4] = {0, 1, 2, 3}; return fill_crt_args(A, p, all, 4); }

// exactness bound of the CRT path: every coefficient of the integer product, at most
// n*(amax)*(bmax) with amax,bmax <= p-1, must stay below the product of the moduli used
// (cf. "sum % M" main.cc:330).  The moduli are kCrtPrimes[first..first+count-1].
static bool crt_bound_ok(uint64_t p, int n, int first = 0, int count = 4) {
  unsigned __int128 lim = 1;                                  // at most 2^119.6
  for (int k = first; k < first + count; ++k) lim *= kCrtPrimes[k];
  if (p - 1 >= (1ull << 60)) return false;                    // (p-1)^2 >= 2^120 > lim
  const unsigned __int128 sq = (unsigned __int128)(p - 1) * (p - 1);
  return sq <= (lim - 1) / (unsigned)n;                       // n * sq < lim, exactly
}
// the fewest primes that keep the product exact: the smallest ones where they suffice (their transforms are the
// cheapest: more lazy-reduction headroom, signed lanes), else the largest ones
static void crt_pick(uint64_t p, int n, int *first, int *count) {
  static const bool four_only = [] { const char *e = getenv("NTTX_CRT_FOUR"); return e && atoi(e) != 0; }();
  *first = 0; *count = 4;
  if (four_only) return;
  for (int ns = 1; ns < 4; ++ns) {
    if (crt_bound_ok(p, n, 0, ns)) { *first = 0; *count = ns; return; }
    if (crt_bound_ok(p, n, 4 - ns, ns)) { *first = 4 - ns; *count = ns; return; }
  }
}

static int log2_ceil_len(int n) {  // main.cc:278-279: smallest power of two >= 2n
  int k = 0;
  while ((1ll << k) < 2ll * n) ++k;
  return k;
}

// The single-prime products of the CRT route (the reference runs them on one thread per prime, main.cc:306-311 /
// main_pthread.cc:413-428) run side by side: prime 0 on the caller's stream, the others on side streams forked from it
// and joined back before the recombination, so the start and the tail of one prime's passes overlap another's.
// Measured on MI355X at n = 2^17, four primes: 8 products 19.3 -> 17.0 us each, 32: 13.5 -> 12.7, 128: 11.5 -> 11.3,
// 256: 11.2 -> 11.0; below about 6 products the forks and joins cost more than they win (one product 66 -> 110 us),
// so those stay on one stream.  NTTX_CRT_STREAMS=0 / 1 forces one stream / side streams (A/B measurements).
static bool crt_side_by_side(size_t words, int ns, int npass) {
  static const int mode = env_int("NTTX_CRT_STREAMS", -1);
  if (ns < 2 || npass < 2 || mode == 0) return false;
  return mode == 1 || (words >= ((size_t)6 << 18) && words <= ((size_t)512 << 18));
}
static int crt_products(Context *cx, Plan *const pl[4], int ns, const uint32_t *pa, const uint32_t *pb, uint32_t *const out[4],
                        int batch, int n, hipStream_t st, int out_slices = 0, long long out_slice_stride = 0,
                        const Src64 *src64 = nullptr) {
  const long long len = 1ll << pl[0]->logL;
  const size_t words = (size_t)batch * len;
  const bool par = crt_side_by_side(words, ns, pl[0]->npass);
  if (par) {
    if (int rc = cx->wsx.ensure((size_t)(ns - 1) * 2 * words * 4)) return rc;
    HIPCHK(hipEventRecord(cx->fork_ev, st));
  }
  // From the first fork on, every way out joins the side streams back into the caller's stream (a failed product
  // included: the next call reuses pa / pb / pc / wsx and must not meet side-stream work still in flight).
  int forked = 0, rc = 0;
  for (int k = 0; k < ns && !rc; ++k) {
    hipStream_t sk = st;
    uint32_t *ws = nullptr;
    if (par && k > 0) {
      sk = cx->side[k - 1];
      ws = cx->wsx.as<uint32_t>() + (size_t)(k - 1) * 2 * words;
      const hipError_t e = hipStreamWaitEvent(sk, cx->fork_ev, 0);
      if (e != hipSuccess) { (void)hipGetLastError(); rc = fail(NTTX_EDEVICE, "hipStreamWaitEvent failed: %s", hipGetErrorString(e)); break; }
      forked = k;
    }
    rc = dev_cyclic_multiply(cx, pl[k], src64 ? nullptr : pa + (size_t)k * words, src64 ? nullptr : pb + (size_t)k * words, out[k], batch,
                             sk, n, nullptr, nullptr, ws, 0, out_slices, out_slice_stride, src64);
  }
  for (int k = 1; k <= forked; ++k) {
    hipError_t e = hipEventRecord(cx->join_ev[k - 1], cx->side[k - 1]);
    if (e == hipSuccess) e = hipStreamWaitEvent(st, cx->join_ev[k - 1], 0);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      (void)hipStreamSynchronize(cx->side[k - 1]);       // the join could not be enqueued: wait for the side stream on the host
      if (!rc) rc = fail(NTTX_EDEVICE, "joining a side stream failed: %s", hipGetErrorString(e));
    }
  }
  return rc;
}

// Residue split of a CRT product.  Two-pass lengths read the u64 coefficients inside their first forward pass
// (fused = true, nothing launched here; NTTX_CRT_FUSED_SPLIT=0 keeps the split as a pass of its own: A/B measurements);
// other lengths run split_kernel into pa / pb.
static bool crt_fused_split(const Plan *pl) {
  static const int on = env_int("NTTX_CRT_FUSED_SPLIT", 1);
  return on != 0 && src64_ok(pl);
}
static int crt_split(Context *cx, Plan *const pl[4], int ns, const uint64_t *d_a, const uint64_t *d_b, int n, uint64_t pmod,
                     int batch, hipStream_t st, bool *fused) {
  const long long len = 1ll << pl[0]->logL;
  *fused = crt_fused_split(pl[0]);
  if (*fused) return 0;
  const size_t vec = (size_t)batch * len * 4;
  if (int rc = cx->pa.ensure(ns * vec)) return rc;
  if (int rc = cx->pb.ensure(ns * vec)) return rc;
  SplitArgs S{}; S.src[0] = d_a; S.src[1] = d_b; S.nmod = ns; S.n = n; S.len = (int)len;
  S.zfill = pl[0]->npass == 1 ? (int)len : n;   // multi-pass products never read the padding
  S.batch = batch; S.cm = coef_mod(pmod);
  for (int k = 0; k < ns; ++k) {
    S.dst[0][k] = cx->pa.as<uint32_t>() + (size_t)k * batch * len;
    S.dst[1][k] = cx->pb.as<uint32_t>() + (size_t)k * batch * len;
    S.ctx[k] = pl[k]->ctx;
  }
  cx->prof.begin(PK_SPLIT, st);
  set_vec(S);
  hipLaunchKernelGGL(split_kernel, ew_grid2(S.zfill, batch, 2), dim3(256), 0, st, S);
  cx->prof.end(st);
  HIPCHK(hipGetLastError());
  return 0;
}

// d_ab[b][0..2n-2] from u64 device inputs through the CRT path: the reference's four primes
// (main.cc:245-248), or only three, two or one of them where n*(p-1)^2 stays below their product
// (three for every modulus below 2^36 at n = 2^17, two below 2^21): fewer transforms, the same
// coefficients
static int dev_crt_multiply(Context *cx, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_ab, int n, uint64_t p,
                            int batch, hipStream_t st) {
  const int logL = log2_ceil_len(n);
  if (logL > 21) return fail(NTTX_ERANGE, "length 2^%d exceeds the CRT prime set (2^21)", logL);
  if (!crt_bound_ok(p, n)) return fail(NTTX_ERANGE, "n*(p-1)^2 exceeds the product of the CRT primes");
  int first, ns;
  crt_pick(p, n, &first, &ns);
  int sel[4];
  for (int k = 0; k < ns; ++k) sel[k] = first + k;
  const long long len = 1ll << logL;
  const size_t vec = (size_t)batch * len * 4;
  if (int rc = cx->pc.ensure(ns * vec)) return rc;
  Plan *pl[4];
  for (int k = 0; k < ns; ++k) if (int rc = get_plan(cx, kCrtPrimes[sel[k]], logL, false, &pl[k])) return rc;
  bool fused = false;
  if (int rc = crt_split(cx, pl, ns, d_a, d_b, n, p, batch, st, &fused)) return rc;
  const Src64 src{d_a, d_b, coef_mod(p)};
  CrtArgs A{};
  if (int rc = fill_crt_args(A, p, sel, ns)) return rc;
  uint32_t *outs[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int k = 0; k < ns; ++k) { outs[k] = cx->pc.as<uint32_t>() + (size_t)k * batch * len; A.res[k] = outs[k]; }
  if (int rc = crt_products(cx, pl, ns, cx->pa.as<uint32_t>(), cx->pb.as<uint32_t>(), outs, batch, n, st, 0, 0, fused ? &src : nullptr)) return rc;
  A.out = d_ab; A.nout = 2 * n - 1; A.len = (int)len; A.batch = batch;
  cx->prof.begin(PK_CRT, st);
  set_vec(A);
  hipLaunchKernelGGL(crt_kernel, ew_grid2(A.nout, batch), dim3(256), 0, st, A);
  cx->prof.end(st);
  HIPCHK(hipGetLastError());
  return 0;
}

// residues of the products modulo the CRT primes selected by prime_mask (bit k = kCrtPrimes[k]):
// d_res gets [selected primes, ascending k][batch][len] canonical u32 (the per-prime half of
// main.cc:292-311, for processes that each own some of the primes)
// slices > 1: d_res is [slice][selected primes][batch][len / slices] instead (slice q = indices q*len/slices ..), the send
// buffer of the all-to-all that hands every process its slice of all residues
static int dev_crt_residues(Context *cx, const uint64_t *d_a, const uint64_t *d_b, uint32_t *d_res, int n, int prime_mask,
                            int batch, hipStream_t st, int slices = 1) {
  const int logL = log2_ceil_len(n);
  if (logL > 21) return fail(NTTX_ERANGE, "length 2^%d exceeds the CRT prime set (2^21)", logL);
  if (prime_mask <= 0 || prime_mask > 15) return fail(NTTX_EINVAL, "prime mask must select 1..4 of the CRT primes");
  const long long len = 1ll << logL;
  int sel[4], ns = 0;
  for (int k = 0; k < 4; ++k) if (prime_mask >> k & 1) sel[ns++] = k;
  Plan *pl[4];
  for (int i = 0; i < ns; ++i) if (int rc = get_plan(cx, kCrtPrimes[sel[i]], logL, false, &pl[i])) return rc;
  bool fused = false;   // no caller modulus here: words are taken as they are (main.cc:300-301)
  if (int rc = crt_split(cx, pl, ns, d_a, d_b, n, 0, batch, st, &fused)) return rc;
  const Src64 src{d_a, d_b, coef_mod(0)};
  const Src64 *sp = fused ? &src : nullptr;
  uint32_t *outs[4] = {nullptr, nullptr, nullptr, nullptr};
  if (slices > 1) {
    const long long sl = len / slices;
    for (int i = 0; i < ns; ++i) outs[i] = d_res + (size_t)i * batch * sl;
    return crt_products(cx, pl, ns, cx->pa.as<uint32_t>(), cx->pb.as<uint32_t>(), outs, batch, n, st, slices, (long long)ns * batch * sl, sp);
  }
  for (int i = 0; i < ns; ++i) outs[i] = d_res + (size_t)i * batch * len;
  return crt_products(cx, pl, ns, cx->pa.as<uint32_t>(), cx->pb.as<uint32_t>(), outs, batch, n, st, 0, 0, sp);
}

// out[i] = (the integer below M with residues r_k[i]) mod p, i < count (main.cc:323-337 on a slice)
static int dev_crt_recombine(Context *cx, const uint32_t *const r[4], uint64_t *d_out, long long count, uint64_t p, hipStream_t st) {
  if (count < 1 || count > (1ll << 30)) return fail(NTTX_EINVAL, "count out of range");
  CrtArgs A{};
  if (int rc = fill_crt_args(A, p)) return rc;
  for (int k = 0; k < 4; ++k) A.res[k] = r[k];
  A.out = d_out; A.nout = (int)count; A.len = (int)count; A.batch = 1;
  cx->prof.begin(PK_CRT, st);
  set_vec(A);
  hipLaunchKernelGGL(crt_kernel, ew_grid2(count, 1), dim3(256), 0, st, A);
  cx->prof.end(st);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- one column pass over a slab of columns (transform shared by several processes) -----
// The length-L vector is the T x S matrix x[j][s] = v[j*S + s]; this process holds columns
// col0 .. col0+cols-1 as a dense [T][cols] matrix.
//   forward:  every column goes through the length-T decimation-in-frequency transform (rows come
//             out in bit-reversed order) and row j' is multiplied by w_L^((col0+s) * bitrev_T(j'))
//   inverse:  the inverse factors (times T^{-1}) first, then the decimation-in-time transform
// A full transform is: forward slabs, transpose between processes, length-S transforms of the rows.
static int get_slab(Context *cx, uint32_t p, int logL, int logT, long long col0, int logC, bool inverse, Slab **out) {
  auto key = std::make_tuple(p, logL, logT, col0, logC, inverse);
  auto it = cx->slabs.find(key);
  if (it != cx->slabs.end()) { *out = it->second.get(); return 0; }
  if (!direct_ok(p, logL)) return fail(NTTX_EINVAL, "modulus %u does not support a direct length-2^%d transform", p, logL);
  std::unique_ptr<Slab> sl(new Slab);
  sl->ctx = make_ctx(p);
  sl->hclass = arith_class(p);
  const uint32_t g = find_nonresidue(p);
  uint32_t wL = h_powmod(g, (uint64_t)(p - 1) >> logL, p);
  if (inverse) wL = h_powmod(wL, p - 2, p);
  const uint32_t wT = h_powmod(wL, 1ull << (logL - logT), p);
  const uint32_t tinv = h_powmod((uint32_t)((1ull << logT) % p), p - 2, p);
  if (int rc = upload(&sl->tw, stage_table(wT, logT, sl->ctx))) return rc;
  const bool split_ok = logC >= 13 && logT >= 7 && logT <= 9;   // shapes with a split-table kernel
  if (split_ok) {
    // w^((col0 + s) r) with s = h C + l and col0 a multiple of C: (w^C)^((col0/C + h) r) * w^(l r)
    const int c2 = (logC + 1) / 2;
    sl->logC2 = c2;
    const uint32_t wC = h_powmod(wL, 1ull << c2, p);
    if (int rc = upload(&sl->fs, fourstep_table(wL, logT, c2, 1, sl->ctx))) return rc;
    if (int rc = upload(&sl->fsa, fourstep_table(wC, logT, logC - c2, inverse ? tinv : 1, sl->ctx, (uint64_t)col0 >> c2))) return rc;
  } else {
    if (int rc = upload(&sl->fs, fourstep_table(wL, logT, logC, inverse ? tinv : 1, sl->ctx, (uint64_t)col0))) return rc;
  }
  *out = sl.get();
  cx->slabs[key] = std::move(sl);
  return 0;
}

static int dev_colpass(Context *cx, uint32_t *d_x, int logL, int logT, long long col0, int cols, uint32_t p, bool inverse,
                       bool zero_hi, hipStream_t st) {
  if (logL < 2 || logL > kMaxLog || logT < kMinColLog || logT > kMaxColLog || logT >= logL)
    return fail(NTTX_ERANGE, "column pass of 2^%d points in a 2^%d transform unsupported", logT, logL);
  const long long S = 1ll << (logL - logT);
  int logC = 0; while ((1ll << logC) < cols) ++logC;
  if (cols < 16 || (1ll << logC) != cols || col0 < 0 || col0 % cols || col0 + cols > S)
    return fail(NTTX_EINVAL, "slab must be a power-of-two number (>= 16) of columns aligned to its own width");
  Slab *sl;
  if (int rc = get_slab(cx, p, logL, logT, col0, logC, inverse, &sl)) return rc;
  ColArgs C{}; C.in[0] = d_x; C.out[0] = d_x; C.tw = sl->tw; C.fs = sl->fs; C.fsa = sl->fsa; C.logC2 = sl->logC2; C.ctx = sl->ctx; C.logS = logC;
  C.canon_out = 1; C.nblocks = 1;
  C.nvalid = (!inverse && zero_hi) ? ((1ll << logT) / 2) * cols : 0;
  return launch_col(cx, inverse, sl->hclass, logT, C, 1, st);
}

// ---- negacyclic product: c = a*b mod (x^len + 1, p) ---------------------------------------
// a'(x) = a(psi x), b'(x) = b(psi x) with psi^len = -1 turns the wrap-around sign into the cyclic
// product's plain wrap-around; c(x) = c'(x / psi).  Three element-wise passes around the cyclic
// product (no counterpart in the reference, which only multiplies zero-padded polynomials).
static int get_twist(Context *cx, uint32_t p, int logL, Twist **out) {
  auto key = std::make_pair(p, logL);
  auto it = cx->twists.find(key);
  if (it != cx->twists.end()) { *out = it->second.get(); return 0; }
  if (!direct_ok(p, logL + 1)) return fail(NTTX_EINVAL, "modulus %u has no 2^%d-th root of unity for a negacyclic product of length 2^%d", p, logL + 1, logL);
  const PrimeCtx c = make_ctx(p);
  const uint32_t g = find_nonresidue(p);
  const uint32_t psi = h_powmod(g, (uint64_t)(p - 1) >> (logL + 1), p), psii = h_powmod(psi, p - 2, p);
  const size_t len = (size_t)1 << logL;
  std::vector<uint32_t> f(len), b(len);
  const uint32_t sf = to_mont(psi, c), sb = to_mont(psii, c);
  uint32_t cf = c.one, cb = c.one;
  for (size_t i = 0; i < len; ++i) { f[i] = cf; b[i] = cb; cf = mont_mul(cf, sf, c.p, c.npinv); cb = mont_mul(cb, sb, c.p, c.npinv); }
  std::unique_ptr<Twist> tw(new Twist);
  if (int rc = upload(&tw->fwd, f)) return rc;
  if (int rc = upload(&tw->inv, b)) return rc;
  *out = tw.get();
  cx->twists[key] = std::move(tw);
  return 0;
}

static int dev_negacyclic_multiply(Context *cx, const uint32_t *a, const uint32_t *b, uint32_t *c, int logL, uint32_t p,
                                   int batch, hipStream_t st) {
  Plan *pl; Twist *tw;
  if (int rc = get_plan(cx, p, logL, false, &pl)) return rc;
  if (int rc = get_twist(cx, p, logL, &tw)) return rc;
  // one- and two-pass plans apply psi^i inside their first and last pass (two-pass plans use full-line column tiles,
  // the only shape with a twisted variant); three-pass plans go through separate element-wise passes
  static const bool unfused = [] { const char *e = getenv("NTTX_TWIST_UNFUSED"); return e && atoi(e) != 0; }();
  // (a two-pass plan whose column pass is longer than 2^9 points has no twisted column kernel: plans chosen under NTTX_LONGROW=0)
  const bool fusable = pl->npass == 1 || (pl->npass == 2 && pl->lt[0] <= 9 && pl->logL - pl->lt[0] >= 5);
  if (fusable && !unfused) return dev_cyclic_multiply(cx, pl, a, b, c, batch, st, 0, tw->fwd, tw->inv);
  const long long len = 1ll << logL;
  const size_t vec = (size_t)batch * len * 4;
  if (int rc = cx->pa.ensure(vec)) return rc;
  if (int rc = cx->pb.ensure(vec)) return rc;
  TwistArgs T{}; T.src[0] = a; T.src[1] = b; T.dst[0] = cx->pa.as<uint32_t>(); T.dst[1] = cx->pb.as<uint32_t>();
  T.tab = tw->fwd; T.ctx = pl->ctx; T.len = (int)len; T.batch = batch;
  cx->prof.begin(PK_MISC, st);
  hipLaunchKernelGGL(twist_kernel, ew_grid2(len, batch, 2), dim3(256), 0, st, T);
  cx->prof.end(st);
  HIPCHK(hipGetLastError());
  if (int rc = dev_cyclic_multiply(cx, pl, cx->pa.as<uint32_t>(), cx->pb.as<uint32_t>(), c, batch, st)) return rc;
  TwistArgs U{}; U.src[0] = c; U.dst[0] = c; U.tab = tw->inv; U.ctx = pl->ctx; U.len = (int)len; U.batch = batch;
  cx->prof.begin(PK_MISC, st);
  hipLaunchKernelGGL(twist_kernel, ew_grid2(len, batch, 1), dim3(256), 0, st, U);
  cx->prof.end(st);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- 64-bit-lane direct path (nttx64.hip) ------------------------------------------------
static void prof64_begin(void *self, int kind, hipStream_t st) { static_cast<Profiler *>(self)->begin(kind, st); }
static void prof64_end(void *self, hipStream_t st) { static_cast<Profiler *>(self)->end(st); }
static ProfHook64 prof_hook(Context *cx) { return ProfHook64{&cx->prof, prof64_begin, prof64_end}; }
static int fail_from64(int rc) { return rc < 0 ? fail(rc, "%s", engine64_error()) : rc; }

}  // namespace nttx

// =========================================================================================
// C ABI
// =========================================================================================
using namespace nttx;

// One mutex serialises every call of the process, nttx_init and nttx_shutdown included, so a
// shutdown cannot pull the context from under a call in flight on another thread.
#define NTTX_ENTER() \
  std::lock_guard<std::mutex> lock_(g_init_mu); \
  if (!g_ctx) return fail(NTTX_ENOINIT, "nttx_init has not been called"); \
  Context *cx = g_ctx; \
  HIPCHK(hipSetDevice(cx->device));
// entry points that use the shared workspace on stream st (host-buffer calls: the null stream)
#define NTTX_ENTER_ON(st) \
  NTTX_ENTER(); \
  if (int rc_ = enter_stream(cx, (hipStream_t)(st))) return rc_;

// Host-buffer entry points enqueue copies from and to the caller's arrays on the null stream.  Every way out of such a
// call waits for that stream: the normal one through finish(), an error in between through the destructor, so no copy
// touches the caller's memory after the call has returned.
struct HostCallGuard {
  Context *cx;
  bool armed = true;
  explicit HostCallGuard(Context *c) : cx(c) {}
  int finish() { armed = false; HIPCHK(hipStreamSynchronize(nullptr)); return 0; }
  ~HostCallGuard() {
    if (!armed) return;
    if (cx && cx->side[0]) (void)hipStreamSynchronize(cx->side[0]);   // (the second operand's copy runs there)
    (void)hipStreamSynchronize(nullptr);
    (void)hipGetLastError();
  }
};

// The two operands of a host-buffer call go to the device on two streams (a on the null stream, b on a side stream that the
// null stream then waits for), so the fixed cost of a copy-engine transfer is paid side by side instead of twice in a row
// (two 1 MiB operands: about 45 instead of 54 us on MI355X).  The staging buffers belong to the host-buffer calls alone,
// and each of those ends with a wait for the null stream, so nothing can still be reading them here.
static int host_in2(Context *cx, void *d_a, const void *a, void *d_b, const void *b, size_t bytes) {
  HIPCHK(hipMemcpyAsync(d_b, b, bytes, hipMemcpyHostToDevice, cx->side[0]));
  HIPCHK(hipEventRecord(cx->join_ev[0], cx->side[0]));
  HIPCHK(hipMemcpyAsync(d_a, a, bytes, hipMemcpyHostToDevice, nullptr));
  HIPCHK(hipStreamWaitEvent(nullptr, cx->join_ev[0], 0));
  return 0;
}

extern "C" {

const char *nttx_last_error(void) { return g_err.c_str(); }

// everything a context owns (also what a half-built one owns after a failed nttx_init)
static void destroy_context(Context *c) {
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  c->plans.clear();
  c->slabs.clear();
  c->twists.clear();
  if (c->e64) engine64_destroy(c->e64);
  c->prof.destroy();
  if (c->ws_event) (void)hipEventDestroy(c->ws_event);
  if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
  for (int k = 0; k < 3; ++k) {
    if (c->join_ev[k]) (void)hipEventDestroy(c->join_ev[k]);
    if (c->side[k]) (void)hipStreamDestroy(c->side[k]);
  }
  for (DevBuf *b : {&c->wa, &c->wb, &c->in_a, &c->in_b, &c->pa, &c->pb, &c->pc, &c->outb, &c->tmp, &c->wsx}) b->release();
  delete c;
}

static int build_context(Context *c) {
  HIPCHK(hipSetDevice(c->device));
  c->e64 = engine64_create();
  HIPCHK(hipEventCreateWithFlags(&c->ws_event, hipEventDisableTiming));
  // the static scratch: every GPU holds its own copy of the array at its own address, so the address is asked for
  // with this context's device current, once per context
  HIPCHK(hipGetSymbolAddress((void **)&c->static_ws, HIP_SYMBOL(g_static_scratch)));
  HIPCHK(hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
  for (int k = 0; k < 3; ++k) {
    HIPCHK(hipStreamCreateWithFlags(&c->side[k], hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->join_ev[k], hipEventDisableTiming));
  }
  return 0;
}

static int init_devices_locked(const int *devices, int ndev) {
  int have = 0;
  hipError_t e = hipGetDeviceCount(&have);
  if (e != hipSuccess || have <= 0) return fail(NTTX_EDEVICE, "no HIP device available: %s", hipGetErrorString(e));
  if (!devices || ndev < 1 || ndev > 64) return fail(NTTX_EINVAL, "device list must hold 1..64 entries");
  for (int i = 0; i < ndev; ++i) {
    if (devices[i] < 0 || devices[i] >= have) return fail(NTTX_EINVAL, "device %d out of range (0..%d)", devices[i], have - 1);
    // NTTX_REHEARSE_DEVICES=1: the list may name one device several times (a context each), which rehearses the
    // several-device code path, host threads and dealing included, on a machine with one GPU
    static const bool rehearse = [] { const char *e_ = getenv("NTTX_REHEARSE_DEVICES"); return e_ && atoi(e_) != 0; }();
    for (int j = 0; j < i && !rehearse; ++j) if (devices[j] == devices[i]) return fail(NTTX_EINVAL, "device %d listed twice", devices[i]);
  }
  if (!g_ctxs.empty()) {
    bool same = (int)g_ctxs.size() == ndev;
    for (int i = 0; same && i < ndev; ++i) same = g_ctxs[i]->device == devices[i];
    if (same) return 0;
    return fail(NTTX_EINVAL, "already initialised on device %d%s; call nttx_shutdown first", g_ctxs[0]->device,
                g_ctxs.size() > 1 ? " and others" : "");
  }
  // built aside and published only when complete: a failure below leaves the library uninitialised, not half-built
  std::vector<Context *> built;
  for (int i = 0; i < ndev; ++i) {
    Context *c = new Context;
    c->device = devices[i];
    built.push_back(c);
    if (int rc = build_context(c)) { for (Context *q : built) destroy_context(q); return rc; }
  }
  (void)crt_const();
  g_ctxs = built;
  g_ctx = g_ctxs[0];
  (void)hipSetDevice(g_ctx->device);
  return 0;
}

int nttx_init(int device) {
  std::lock_guard<std::mutex> lk(g_init_mu);
  return init_devices_locked(&device, 1);
}

int nttx_init_devices(const int *devices, int ndev) {
  std::lock_guard<std::mutex> lk(g_init_mu);
  return init_devices_locked(devices, ndev);
}

int nttx_device_count(void) {
  std::lock_guard<std::mutex> lk(g_init_mu);
  return (int)g_ctxs.size();
}

int nttx_set_device(int device) {
  std::lock_guard<std::mutex> lk(g_init_mu);
  if (g_ctxs.empty()) return fail(NTTX_ENOINIT, "nttx_init has not been called");
  for (Context *c : g_ctxs) if (c->device == device) { g_ctx = c; HIPCHK(hipSetDevice(device)); return 0; }
  return fail(NTTX_EINVAL, "device %d is not among the devices this process initialised", device);
}

int nttx_deal_batch(int batch, int ndev, int k, int *lo, int *hi) {
  if (batch < 0 || ndev < 1 || k < 0 || k >= ndev || !lo || !hi) return fail(NTTX_EINVAL, "bad arguments");
  const int base = batch / ndev, rem = batch % ndev;      // the first `rem` devices take one product more
  *lo = k * base + std::min(k, rem);
  *hi = *lo + base + (k < rem ? 1 : 0);
  return 0;
}

void nttx_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_init_mu);
  for (Context *c : g_ctxs) destroy_context(c);
  g_ctxs.clear();
  g_ctx = nullptr;
}

void *nttx_host_alloc(size_t bytes) {
  std::lock_guard<std::mutex> lock_(g_init_mu);
  if (!g_ctx) { (void)fail(NTTX_ENOINIT, "nttx_init has not been called"); return nullptr; }
  void *q = nullptr;
  if (bytes == 0 || hipSetDevice(g_ctx->device) != hipSuccess || hipHostMalloc(&q, bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    (void)fail(NTTX_EDEVICE, "page-locked allocation of %zu bytes failed", bytes);
    return nullptr;
  }
  return q;
}

void nttx_host_free(void *ptr) {
  if (!ptr) return;
  std::lock_guard<std::mutex> lock_(g_init_mu);
  if (g_ctx) (void)hipSetDevice(g_ctx->device);
  if (hipHostFree(ptr) != hipSuccess) (void)hipGetLastError();
}

int nttx_path_for(uint64_t p, int n) {
  if (n <= 0 || p < 2) return fail(NTTX_EINVAL, "bad arguments");
  const int k = log2_ceil_len(n);
  return direct_ok(p, k) ? NTTX_PATH_DIRECT32 : engine64_ok(p, k) ? NTTX_PATH_DIRECT64 : NTTX_PATH_CRT4;
}

int nttx_plan_info_u64(int len, uint64_t p, int *logt) {
  if (len < 2 || (len & (len - 1))) return fail(NTTX_EINVAL, "length must be a power of two >= 2");
  int k = 0; while ((1 << k) < len) ++k;
  int lt[3] = {0, 0, 0};
  const int np = engine64_decompose(k, p, lt, engine64_fp_ok(p));
  if (logt) for (int i = 0; i < np; ++i) logt[i] = lt[i];
  return np;
}

int nttx_plan_info(int len, int *logt) {
  if (len < 2 || (len & (len - 1))) return fail(NTTX_EINVAL, "length must be a power of two >= 2");
  int k = 0; while ((1 << k) < len) ++k;
  int lt[3] = {0, 0, 0};
  const int np = decompose(k, lt);
  if (logt) for (int i = 0; i < np; ++i) logt[i] = lt[i];
  return np;
}

int nttx_profile_enable(int on) {
  NTTX_ENTER();
  cx->prof.fold();
  cx->prof.on = on != 0;
  return 0;
}

int nttx_profile_reset(void) {
  NTTX_ENTER();
  cx->prof.reset();
  return 0;
}

int nttx_profile_read(int kind, double *total_ms, long long *launches) {
  NTTX_ENTER();
  if (kind < 0 || kind >= PK_COUNT) return fail(NTTX_EINVAL, "kernel kind out of range");
  cx->prof.fold();
  if (total_ms) *total_ms = cx->prof.ms[kind];
  if (launches) *launches = cx->prof.launches[kind];
  return 0;
}

int nttx_dev_stream_copy(void *d_dst, const void *d_src, size_t bytes, void *stream) {
  NTTX_ENTER();
  if (!d_dst || !d_src || bytes == 0 || (bytes & 15) || !aligned16(d_dst) || !aligned16(d_src))
    return fail(NTTX_EINVAL, "stream copy needs 16-byte aligned pointers and a multiple of 16 bytes");
  const long long n16 = (long long)(bytes >> 4);
  // one 16-byte word per lane, one workgroup per 4 KB: measured fastest on MI355X (6.0 TB/s on 2 x 268 MB against
  // 4.3 - 5.1 TB/s with 4 .. 64 words per lane)
  const int grid = grid_for((n16 + 255) / 256, 1 << 22);
  hipLaunchKernelGGL(copy16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4 *)d_src, (uint4 *)d_dst, n16);
  HIPCHK(hipGetLastError());
  return 0;
}

int nttx_reserve(int len, int batch, int nprimes) {
  NTTX_ENTER();
  if (len < 2 || batch < 1 || nprimes < 1 || nprimes > 4) return fail(NTTX_EINVAL, "bad arguments");
  const size_t vec = (size_t)batch * len * 4;
  if (int rc = cx->wa.ensure(vec)) return rc;
  if (int rc = cx->wb.ensure(vec)) return rc;
  if (nprimes > 1) {
    if (int rc = cx->pa.ensure(nprimes * vec)) return rc;
    if (int rc = cx->pb.ensure(nprimes * vec)) return rc;
    if (int rc = cx->pc.ensure(nprimes * vec)) return rc;
    if (crt_side_by_side((size_t)batch * len, nprimes, 2))
      if (int rc = cx->wsx.ensure((size_t)(nprimes - 1) * 2 * vec)) return rc;
  }
  return 0;
}

int nttx_dev_cyclic_multiply_u32(const void *d_a, const void *d_b, void *d_c, int len, uint32_t p, int batch, void *stream) {
  NTTX_ENTER_ON(stream);
  if (!d_a || !d_b || !d_c || batch < 1 || len < 2 || (len & (len - 1))) return fail(NTTX_EINVAL, "bad arguments");
  int k = 0; while ((1ll << k) < len) ++k;
  Plan *pl;
  if (int rc = get_plan(cx, p, k, false, &pl)) return rc;
  return dev_cyclic_multiply(cx, pl, (const uint32_t *)d_a, (const uint32_t *)d_b, (uint32_t *)d_c, batch, (hipStream_t)stream);
}

int nttx_dev_cyclic_multiply_pieces_u32(const void *d_a, const void *d_b, void *d_c, int len, int pieces, int rows, uint32_t p,
                                        void *stream) {
  NTTX_ENTER_ON(stream);
  if (!d_a || !d_b || !d_c || rows < 1 || pieces < 1 || len < 2 || (len & (len - 1))) return fail(NTTX_EINVAL, "bad arguments");
  int k = 0; while ((1ll << k) < len) ++k;
  Plan *pl;
  if (int rc = get_plan(cx, p, k, false, &pl)) return rc;
  return dev_cyclic_multiply(cx, pl, (const uint32_t *)d_a, (const uint32_t *)d_b, (uint32_t *)d_c, rows, (hipStream_t)stream, 0,
                             nullptr, nullptr, nullptr, pieces);
}

int nttx_dev_negacyclic_multiply_u32(const void *d_a, const void *d_b, void *d_c, int len, uint32_t p, int batch, void *stream) {
  NTTX_ENTER_ON(stream);
  if (!d_a || !d_b || !d_c || batch < 1 || len < 2 || (len & (len - 1))) return fail(NTTX_EINVAL, "bad arguments");
  int k = 0; while ((1ll << k) < len) ++k;
  return dev_negacyclic_multiply(cx, (const uint32_t *)d_a, (const uint32_t *)d_b, (uint32_t *)d_c, k, p, batch, (hipStream_t)stream);
}

int nttx_dev_poly_multiply_u32(const void *d_a, const void *d_b, void *d_c, int n, uint32_t p, int batch, void *stream) {
  NTTX_ENTER_ON(stream);
  if (!d_a || !d_b || !d_c || batch < 1 || n < 1) return fail(NTTX_EINVAL, "bad arguments");
  Plan *pl;
  if (int rc = get_plan(cx, p, log2_ceil_len(n), false, &pl)) return rc;
  return dev_cyclic_multiply(cx, pl, (const uint32_t *)d_a, (const uint32_t *)d_b, (uint32_t *)d_c, batch, (hipStream_t)stream, n);
}

int nttx_dev_poly_multiply_u32_static(const void *d_a, const void *d_b, void *d_c, int n, uint32_t p, int batch, void *stream) {
  NTTX_ENTER();
  if (!d_a || !d_b || !d_c || batch < 1 || n < 1) return fail(NTTX_EINVAL, "bad arguments");
  const int logL = log2_ceil_len(n);
  const long long words = (long long)batch << logL;
  if (logL > 22 || words > kStaticWords)
    return fail(NTTX_ERANGE, "batch %d x length 2^%d = %lld words exceeds the static scratch of %lld words per copy; "
                "use nttx_dev_poly_multiply_u32", batch, logL, words, kStaticWords);
  auto it = cx->plans.find(std::make_tuple(p, logL, false));
  if (it == cx->plans.end())
    return fail(NTTX_ENOTREADY, "no tables yet for modulus %u and length 2^%d: building them allocates, so call "
                "nttx_dev_poly_multiply_u32 once with this modulus and length at set-up", p, logL);
  if (!cx->static_ws) return fail(NTTX_EDEVICE, "static scratch address unknown");
  // no per-kernel event timing on this path (event creation allocates); the setting is restored afterwards
  const bool prof_was = cx->prof.on;
  cx->prof.on = false;
  const int rc = dev_cyclic_multiply(cx, it->second.get(), (const uint32_t *)d_a, (const uint32_t *)d_b, (uint32_t *)d_c, batch,
                                     (hipStream_t)stream, n, nullptr, nullptr, cx->static_ws);
  cx->prof.on = prof_was;
  return rc;
}

int nttx_dev_crt_multiply_u64(const void *d_a, const void *d_b, void *d_ab, int n, uint64_t p, int batch, void *stream) {
  NTTX_ENTER_ON(stream);
  if (!d_a || !d_b || !d_ab || batch < 1 || n < 1) return fail(NTTX_EINVAL, "bad arguments");
  return dev_crt_multiply(cx, (const uint64_t *)d_a, (const uint64_t *)d_b, (uint64_t *)d_ab, n, p, batch, (hipStream_t)stream);
}

int nttx_dev_crt_residues_u64(const void *d_a, const void *d_b, void *d_res, int n, int prime_mask, int batch, void *stream) {
  NTTX_ENTER_ON(stream);
  if (!d_a || !d_b || !d_res || batch < 1 || n < 1) return fail(NTTX_EINVAL, "bad arguments");
  return dev_crt_residues(cx, (const uint64_t *)d_a, (const uint64_t *)d_b, (uint32_t *)d_res, n, prime_mask, batch, (hipStream_t)stream);
}

int nttx_dev_crt_residues_sliced_u64(const void *d_a, const void *d_b, void *d_res, int n, int prime_mask, int batch, int slices,
                                     void *stream) {
  NTTX_ENTER_ON(stream);
  if (!d_a || !d_b || !d_res || batch < 1 || n < 1 || slices < 1) return fail(NTTX_EINVAL, "bad arguments");
  return dev_crt_residues(cx, (const uint64_t *)d_a, (const uint64_t *)d_b, (uint32_t *)d_res, n, prime_mask, batch, (hipStream_t)stream, slices);
}

int nttx_dev_crt_recombine_u64(const void *d_r0, const void *d_r1, const void *d_r2, const void *d_r3, void *d_out,
                               long long count, uint64_t p, void *stream) {
  NTTX_ENTER();
  if (!d_r0 || !d_r1 || !d_r2 || !d_r3 || !d_out) return fail(NTTX_EINVAL, "bad arguments");
  const uint32_t *r[4] = {(const uint32_t *)d_r0, (const uint32_t *)d_r1, (const uint32_t *)d_r2, (const uint32_t *)d_r3};
  return dev_crt_recombine(cx, r, (uint64_t *)d_out, count, p, (hipStream_t)stream);
}

int nttx_dev_colpass_u32(void *d_x, int log_len, int log_t, long long col0, int cols, uint32_t p, int inverse, int zero_hi,
                         void *stream) {
  NTTX_ENTER();
  if (!d_x) return fail(NTTX_EINVAL, "bad arguments");
  return dev_colpass(cx, (uint32_t *)d_x, log_len, log_t, col0, cols, p, inverse != 0, zero_hi != 0, (hipStream_t)stream);
}

int nttx_crt_primes_for(uint64_t p, int n, uint32_t primes[4]) {
  if (n < 1 || p < 2 || p >= (1ull << 63) || !primes) return fail(NTTX_EINVAL, "bad arguments");
  if (!crt_bound_ok(p, n)) return fail(NTTX_ERANGE, "n*(p-1)^2 exceeds the product of the CRT primes");
  int first, ns;
  crt_pick(p, n, &first, &ns);
  for (int k = 0; k < ns; ++k) primes[k] = kCrtPrimes[first + k];
  return ns;
}

int nttx_crt_prime(int k, uint32_t *prime) {
  if (k < 0 || k >= 4 || !prime) return fail(NTTX_EINVAL, "CRT prime index out of range");
  *prime = kCrtPrimes[k];
  return 0;
}

static int dev_ntt_impl(Context *cx, uint32_t *d_x, int len, uint32_t p, int inverse, int bit_reversed_out, int batch,
                        hipStream_t st) {
  if (len == 1) return 0;
  int k = 0; while ((1ll << k) < len) ++k;
  Plan *pl;
  if (int rc = get_plan(cx, p, k, inverse != 0, &pl)) return rc;
  if (int rc = dev_forward(cx, pl, d_x, batch, true, st)) return rc;
  if (!bit_reversed_out) {
    const long long total = (long long)batch * len;
    if (int rc = cx->tmp.ensure((size_t)total * 4)) return rc;
    HIPCHK(hipMemcpyAsync(cx->tmp.ptr, d_x, (size_t)total * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(bitrev_kernel, dim3(ew_grid(total)), dim3(256), 0, st, cx->tmp.as<uint32_t>(), d_x, k, total);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

int nttx_dev_ntt_u32(void *d_x, int len, uint32_t p, int inverse, int bit_reversed_out, int batch, void *stream) {
  NTTX_ENTER_ON(stream);
  if (!d_x || batch < 1 || len < 1 || (len & (len - 1))) return fail(NTTX_EINVAL, "bad arguments");
  return dev_ntt_impl(cx, (uint32_t *)d_x, len, p, inverse, bit_reversed_out, batch, (hipStream_t)stream);
}

int nttx_ntt_u32(uint32_t *x, int len, uint32_t p, int inverse) {
  NTTX_ENTER_ON(nullptr);
  if (!x || len < 1 || (len & (len - 1))) return fail(NTTX_EINVAL, "length must be a power of two");
  if (int rc = cx->in_a.ensure((size_t)len * 4)) return rc;
  HostCallGuard guard_(cx);
  HIPCHK(hipMemcpyAsync(cx->in_a.ptr, x, (size_t)len * 4, hipMemcpyHostToDevice, nullptr));
  if (int rc = dev_ntt_impl(cx, cx->in_a.as<uint32_t>(), len, p, inverse, 0, 1, nullptr)) return rc;
  HIPCHK(hipMemcpyAsync(x, cx->in_a.ptr, (size_t)len * 4, hipMemcpyDeviceToHost, nullptr));
  return guard_.finish();
}

int nttx_cyclic_multiply_u32(const uint32_t *a, const uint32_t *b, uint32_t *c, int len, uint32_t p) {
  NTTX_ENTER_ON(nullptr);
  if (!a || !b || !c || len < 2 || (len & (len - 1))) return fail(NTTX_EINVAL, "bad arguments");
  int k = 0; while ((1ll << k) < len) ++k;
  Plan *pl;
  if (int rc = get_plan(cx, p, k, false, &pl)) return rc;
  const size_t bytes = (size_t)len * 4;
  if (int rc = cx->pa.ensure(bytes)) return rc;
  if (int rc = cx->pb.ensure(bytes)) return rc;
  if (int rc = cx->pc.ensure(bytes)) return rc;
  HostCallGuard guard_(cx);
  HIPCHK(hipMemcpyAsync(cx->pa.ptr, a, bytes, hipMemcpyHostToDevice, nullptr));
  HIPCHK(hipMemcpyAsync(cx->pb.ptr, b, bytes, hipMemcpyHostToDevice, nullptr));
  if (int rc = dev_cyclic_multiply(cx, pl, cx->pa.as<uint32_t>(), cx->pb.as<uint32_t>(), cx->pc.as<uint32_t>(), 1, nullptr)) return rc;
  HIPCHK(hipMemcpyAsync(c, cx->pc.ptr, bytes, hipMemcpyDeviceToHost, nullptr));
  return guard_.finish();
}

int nttx_crt_combine_u64(const uint32_t *const res[4], uint64_t *ab, int len, int nout, uint64_t p) {
  NTTX_ENTER_ON(nullptr);
  if (!res || !ab || len < 1 || nout < 1 || nout > len) return fail(NTTX_EINVAL, "bad arguments");
  const size_t bytes = (size_t)len * 4;
  if (int rc = cx->pc.ensure(4 * bytes)) return rc;
  if (int rc = cx->outb.ensure((size_t)nout * 8)) return rc;
  CrtArgs A{};
  if (int rc = fill_crt_args(A, p)) return rc;
  for (int k = 0; k < 4; ++k) if (!res[k]) return fail(NTTX_EINVAL, "null residue vector");
  HostCallGuard guard_(cx);
  for (int k = 0; k < 4; ++k) {
    uint32_t *d = cx->pc.as<uint32_t>() + (size_t)k * len;
    HIPCHK(hipMemcpyAsync(d, res[k], bytes, hipMemcpyHostToDevice, nullptr));
    A.res[k] = d;
  }
  A.out = cx->outb.as<uint64_t>(); A.nout = nout; A.len = len; A.batch = 1;
  set_vec(A);
  hipLaunchKernelGGL(crt_kernel, ew_grid2(nout, 1), dim3(256), 0, 0, A);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(ab, cx->outb.ptr, (size_t)nout * 8, hipMemcpyDeviceToHost, nullptr));
  return guard_.finish();
}

// Host-buffer entry points: the copies in and out are enqueued on the null stream like the kernels between them and the
// call waits once, after the copy back (page-locked buffers then move by DMA back to back; pageable ones are staged by
// the runtime, which keeps the caller's buffers safe to reuse on return either way).
// shared body of the host-buffer direct path: raw u32 coefficients [batch][n] -> [batch][2n-1]
static int host_direct_u32(Context *cx, const uint32_t *a, const uint32_t *b, uint32_t *ab, int n, uint32_t p, int batch) {
  const int logL = log2_ceil_len(n);
  Plan *pl;
  if (int rc = get_plan(cx, p, logL, false, &pl)) return rc;
  const long long len = 1ll << logL;
  const size_t raw = (size_t)batch * n * 4, vec = (size_t)batch * len * 4;
  const int nout = 2 * n - 1;
  if (int rc = cx->in_a.ensure(raw)) return rc;
  if (int rc = cx->in_b.ensure(raw)) return rc;
  if (int rc = cx->pa.ensure(vec)) return rc;
  if (int rc = cx->pb.ensure(vec)) return rc;
  if (int rc = cx->pc.ensure(vec)) return rc;
  if (int rc = cx->outb.ensure((size_t)batch * nout * 4)) return rc;
  HostCallGuard guard_(cx);
  if (int rc = host_in2(cx, cx->in_a.ptr, a, cx->in_b.ptr, b, raw)) return rc;
  {
    PadArgs P{}; P.src[0] = cx->in_a.as<uint32_t>(); P.src[1] = cx->in_b.as<uint32_t>();
    P.dst[0] = cx->pa.as<uint32_t>(); P.dst[1] = cx->pb.as<uint32_t>(); P.ctx = pl->ctx; P.n = n; P.len = (int)len;
    P.zfill = pl->npass == 1 ? (int)len : n; P.batch = batch;
    hipLaunchKernelGGL(pad_kernel, ew_grid2(P.zfill, batch, 2), dim3(256), 0, 0, P);
    HIPCHK(hipGetLastError());
  }
  if (int rc = dev_cyclic_multiply(cx, pl, cx->pa.as<uint32_t>(), cx->pb.as<uint32_t>(), cx->pc.as<uint32_t>(), batch, nullptr, n)) return rc;
  const long long tout = (long long)batch * nout;
  hipLaunchKernelGGL((unpad_kernel<uint32_t>), ew_grid2(nout, batch), dim3(256), 0, 0, cx->pc.as<uint32_t>(), cx->outb.as<uint32_t>(), nout, (int)len, batch);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(ab, cx->outb.ptr, (size_t)tout * 4, hipMemcpyDeviceToHost, nullptr));
  return guard_.finish();
}

// With several devices initialised (nttx_init_devices) the products are dealt to them in contiguous shares
// (nttx_deal_batch), one host thread per device, each on its own device's context, workspace and null stream: the
// thread-per-modulus and thread-per-index-chunk split of main_pthread.cc:413-428 / 435-465 with GPUs in place of the
// threads.  No data moves between the devices: independent products need no exchange.
int nttx_poly_multiply_batch_u32(const uint32_t *a, const uint32_t *b, uint32_t *ab, int n, uint32_t p, int batch) {
  NTTX_ENTER_ON(nullptr);
  if (!a || !b || !ab || n < 1 || batch < 1 || p < 2) return fail(NTTX_EINVAL, "bad arguments");
  if (!direct_ok(p, log2_ceil_len(n))) return fail(NTTX_EINVAL, "modulus %u has no direct transform of this length; use the u64 entry point", p);
  const int ndev = (int)g_ctxs.size();
  if (ndev == 1 || batch < ndev) return host_direct_u32(cx, a, b, ab, n, p, batch);
  std::vector<int> rcs(ndev, 0);
  std::vector<std::string> errs(ndev);
  std::vector<std::thread> workers;
  const size_t in_stride = (size_t)n, out_stride = 2 * (size_t)n - 1;
  for (int k = 0; k < ndev; ++k) {
    workers.emplace_back([&, k] {
      int lo = 0, hi = 0;
      (void)nttx_deal_batch(batch, ndev, k, &lo, &hi);
      Context *c = g_ctxs[k];
      if (hi <= lo) return;
      if (hipSetDevice(c->device) != hipSuccess) { rcs[k] = NTTX_EDEVICE; errs[k] = "hipSetDevice failed"; return; }
      int rc = enter_stream(c, nullptr);
      if (!rc) rc = host_direct_u32(c, a + lo * in_stride, b + lo * in_stride, ab + lo * out_stride, n, p, hi - lo);
      if (rc) { rcs[k] = rc; errs[k] = g_err; }            // (g_err is per thread: carried back to the caller's)
    });
  }
  for (auto &w : workers) w.join();
  (void)hipSetDevice(cx->device);
  for (int k = 0; k < ndev; ++k) if (rcs[k]) return fail(rcs[k], "device %d: %s", g_ctxs[k]->device, errs[k].c_str());
  return 0;
}

int nttx_poly_multiply_i32(const int32_t *a, const int32_t *b, int32_t *ab, int n, int32_t p) {
  if (!a || !b || !ab || n < 1 || p < 2) return fail(NTTX_EINVAL, "bad arguments");
  if (nttx_path_for((uint64_t)p, n) == NTTX_PATH_DIRECT32) {
    NTTX_ENTER_ON(nullptr);
    return host_direct_u32(cx, (const uint32_t *)a, (const uint32_t *)b, (uint32_t *)ab, n, (uint32_t)p, 1);
  }
  std::vector<uint64_t> wa(n), wb(n), wc(2 * (size_t)n - 1);
  for (int i = 0; i < n; ++i) { wa[i] = (uint32_t)a[i]; wb[i] = (uint32_t)b[i]; }
  if (int rc = nttx_poly_multiply_crt_u64(wa.data(), wb.data(), wc.data(), n, (uint64_t)p)) return rc;
  for (size_t i = 0; i < wc.size(); ++i) ab[i] = (int32_t)wc[i];
  return 0;
}

int nttx_poly_multiply_crt_u64(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p) {
  NTTX_ENTER_ON(nullptr);
  if (!a || !b || !ab || n < 1 || p < 2) return fail(NTTX_EINVAL, "bad arguments");
  const size_t raw = (size_t)n * 8, nout = 2 * (size_t)n - 1;
  if (int rc = cx->in_a.ensure(raw)) return rc;
  if (int rc = cx->in_b.ensure(raw)) return rc;
  if (int rc = cx->outb.ensure(nout * 8)) return rc;
  HostCallGuard guard_(cx);
  if (int rc = host_in2(cx, cx->in_a.ptr, a, cx->in_b.ptr, b, raw)) return rc;
  if (int rc = dev_crt_multiply(cx, cx->in_a.as<uint64_t>(), cx->in_b.as<uint64_t>(), cx->outb.as<uint64_t>(), n, p, 1, nullptr)) return rc;
  HIPCHK(hipMemcpyAsync(ab, cx->outb.ptr, nout * 8, hipMemcpyDeviceToHost, nullptr));
  return guard_.finish();
}

int nttx_poly_multiply_direct_u64(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p) {
  if (!a || !b || !ab || n < 1 || p < 2) return fail(NTTX_EINVAL, "bad arguments");
  NTTX_ENTER_ON(nullptr);
  const int logL = log2_ceil_len(n);
  if (!engine64_ok(p, logL)) return fail(NTTX_EINVAL, "modulus %llu has no direct 64-bit transform of length 2^%d", (unsigned long long)p, logL);
  const long long len = 1ll << logL;
  const size_t raw = (size_t)n * 8, vec = (size_t)len * 8;
  const int nout = 2 * n - 1;
  if (int rc = cx->in_a.ensure(raw)) return rc;
  if (int rc = cx->in_b.ensure(raw)) return rc;
  if (int rc = cx->pa.ensure(vec)) return rc;
  if (int rc = cx->pb.ensure(vec)) return rc;
  if (int rc = cx->pc.ensure(vec)) return rc;
  if (int rc = cx->outb.ensure((size_t)nout * 8)) return rc;
  HostCallGuard guard_(cx);
  if (int rc = host_in2(cx, cx->in_a.ptr, a, cx->in_b.ptr, b, raw)) return rc;
  const ProfHook64 ph = prof_hook(cx);
  if (int rc = fail_from64(engine64_pad(cx->e64, cx->in_a.as<uint64_t>(), cx->in_b.as<uint64_t>(), cx->pa.as<uint64_t>(),
                                        cx->pb.as<uint64_t>(), n, logL, p, 1, nullptr))) return rc;
  if (int rc = fail_from64(engine64_cyclic_multiply(cx->e64, cx->pa.as<uint64_t>(), cx->pb.as<uint64_t>(), cx->pc.as<uint64_t>(),
                                                    logL, p, 1, n, nullptr, &ph))) return rc;
  if (int rc = fail_from64(engine64_unpad(cx->pc.as<uint64_t>(), cx->outb.as<uint64_t>(), nout, logL, 1, nullptr))) return rc;
  HIPCHK(hipMemcpyAsync(ab, cx->outb.ptr, (size_t)nout * 8, hipMemcpyDeviceToHost, nullptr));
  return guard_.finish();
}

int nttx_dev_negacyclic_multiply_u64(const void *d_a, const void *d_b, void *d_c, int len, uint64_t p, int batch, void *stream) {
  NTTX_ENTER_ON(stream);
  if (!d_a || !d_b || !d_c || batch < 1 || len < 2 || (len & (len - 1))) return fail(NTTX_EINVAL, "bad arguments (len must be a power of two >= 2)");
  int k = 0; while ((1 << k) < len) ++k;
  const ProfHook64 ph = prof_hook(cx);
  return fail_from64(engine64_negacyclic_multiply(cx->e64, (const uint64_t *)d_a, (const uint64_t *)d_b, (uint64_t *)d_c, k, p, batch,
                                                  (hipStream_t)stream, &ph));
}

int nttx_dev_direct_multiply_u64(const void *d_a, const void *d_b, void *d_c, int n, uint64_t p, int batch, void *stream) {
  NTTX_ENTER_ON(stream);
  if (!d_a || !d_b || !d_c || batch < 1 || n < 1) return fail(NTTX_EINVAL, "bad arguments");
  const ProfHook64 ph = prof_hook(cx);
  return fail_from64(engine64_cyclic_multiply(cx->e64, (const uint64_t *)d_a, (const uint64_t *)d_b, (uint64_t *)d_c,
                                              log2_ceil_len(n), p, batch, n, (hipStream_t)stream, &ph));
}

int nttx_dev_ntt_u64(void *d_x, int len, uint64_t p, int inverse, int batch, void *stream) {
  NTTX_ENTER_ON(stream);
  if (!d_x || batch < 1 || len < 2 || (len & (len - 1))) return fail(NTTX_EINVAL, "bad arguments");
  int k = 0; while ((1ll << k) < len) ++k;
  const ProfHook64 ph = prof_hook(cx);
  return fail_from64(engine64_forward(cx->e64, (uint64_t *)d_x, k, p, inverse, batch, (hipStream_t)stream, &ph));
}

int nttx_reserve_u64(int len, int batch) {
  NTTX_ENTER();
  if (len < 2 || (len & (len - 1)) || batch < 1) return fail(NTTX_EINVAL, "bad arguments");
  int k = 0; while ((1ll << k) < len) ++k;
  return fail_from64(engine64_reserve(cx->e64, k, batch));
}

int nttx_poly_multiply_u64(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p) {
  if (!a || !b || !ab || n < 1 || p < 2) return fail(NTTX_EINVAL, "bad arguments");
  const int path = nttx_path_for(p, n);
  if (path == NTTX_PATH_DIRECT64) return nttx_poly_multiply_direct_u64(a, b, ab, n, p);
  if (path != NTTX_PATH_DIRECT32) return nttx_poly_multiply_crt_u64(a, b, ab, n, p);
  NTTX_ENTER_ON(nullptr);
  // direct path on 32-bit lanes: reduce the u64 coefficients mod p on the device (one modulus)
  const int logL = log2_ceil_len(n);
  Plan *pl;
  if (int rc = get_plan(cx, (uint32_t)p, logL, false, &pl)) return rc;
  const long long len = 1ll << logL;
  const size_t raw = (size_t)n * 8, vec = (size_t)len * 4;
  const int nout = 2 * n - 1;
  if (int rc = cx->in_a.ensure(raw)) return rc;
  if (int rc = cx->in_b.ensure(raw)) return rc;
  if (int rc = cx->pa.ensure(vec)) return rc;
  if (int rc = cx->pb.ensure(vec)) return rc;
  if (int rc = cx->pc.ensure(vec)) return rc;
  if (int rc = cx->outb.ensure((size_t)nout * 8)) return rc;
  HostCallGuard guard_(cx);
  if (int rc = host_in2(cx, cx->in_a.ptr, a, cx->in_b.ptr, b, raw)) return rc;
  if (crt_fused_split(pl)) {     // two-pass lengths: the first column pass reads the u64 coefficients itself
    const Src64 src{cx->in_a.as<uint64_t>(), cx->in_b.as<uint64_t>(), coef_mod(0)};
    if (int rc = dev_cyclic_multiply(cx, pl, nullptr, nullptr, cx->pc.as<uint32_t>(), 1, nullptr, n, nullptr, nullptr, nullptr, 0, 0, 0, &src)) return rc;
  } else {
    SplitArgs S{}; S.src[0] = cx->in_a.as<uint64_t>(); S.src[1] = cx->in_b.as<uint64_t>(); S.nmod = 1; S.n = n; S.len = (int)len;
    S.zfill = pl->npass == 1 ? (int)len : n; S.batch = 1; S.cm = coef_mod(0);
    S.dst[0][0] = cx->pa.as<uint32_t>(); S.dst[1][0] = cx->pb.as<uint32_t>(); S.ctx[0] = pl->ctx;
    set_vec(S);
    hipLaunchKernelGGL(split_kernel, ew_grid2(S.zfill, 1, 2), dim3(256), 0, 0, S);
    HIPCHK(hipGetLastError());
    if (int rc = dev_cyclic_multiply(cx, pl, cx->pa.as<uint32_t>(), cx->pb.as<uint32_t>(), cx->pc.as<uint32_t>(), 1, nullptr, n)) return rc;
  }
  hipLaunchKernelGGL((unpad_kernel<uint64_t>), ew_grid2(nout, 1), dim3(256), 0, 0, cx->pc.as<uint32_t>(), cx->outb.as<uint64_t>(), nout, (int)len, 1);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(ab, cx->outb.ptr, (size_t)nout * 8, hipMemcpyDeviceToHost, nullptr));
  return guard_.finish();
}

}  // extern "C"
