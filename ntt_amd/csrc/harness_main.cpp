// nttx_harness: the reference's driver loop on the GPU engine.
//
// Same flow, file formats and stdout lines as main() of main.cc:272-346: for every test id
// fRead -> poly_multiply -> fCheck -> "average latency ..." -> fWrite.  The only host code
// between fRead and fCheck is the call through the C ABI (include/nttx.h).
//
//   NTT_DATA_DIR   directory of <id>.in / <id>.out   (the reference hard-codes /nttdata, main.cc:81)
//   NTT_OUT_DIR    directory for <id>.out dumps       (default "files", main.cc:100)
//   argv[1..2]     first and last test id             (default 0 4, main.cc:240)
//   NTT_DEVICE     GPU ordinal                        (default 0)
//
// Two flavours from this one source, as the reference has two harness shapes:
//   nttx_harness      (T, P) = (uint64_t, uint64_t): main.cc:78-128, 272-346 (every modulus, id 4 included)
//   nttx_harness_int  (T, P) = (int, int), built with -DNTTX_HARNESS_INT: the course harness of
//                     monttry2.cpp:134-196, 344-400 and pthread_normal.cpp:417-468, whose declared entry point is
//                     void poly_multiply(int *a, int *b, int *ab, int n, int p); it prints the latency with the
//                     reference's "(us)" label (monttry2.cpp:396: the figure is milliseconds there too) and, like
//                     the reference, cannot represent id 4's 51-bit modulus (reported as an error here, garbage there)
#include <chrono>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nttx.h"

#ifdef NTTX_HARNESS_INT
typedef int coef_t;
typedef int mod_t;
#define NTTX_LATENCY_UNIT "(us)"
#else
typedef uint64_t coef_t;
typedef uint64_t mod_t;
#define NTTX_LATENCY_UNIT "(ms)"
#endif

static std::string data_dir() { const char *d = getenv("NTT_DATA_DIR"); return d ? d : "/nttdata"; }
static std::string out_dir() { const char *d = getenv("NTT_OUT_DIR"); return d ? d : "files"; }

static bool slurp(const std::string &path, std::vector<char> &buf) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  buf.resize((size_t)sz + 1);
  size_t got = fread(buf.data(), 1, (size_t)sz, f);
  fclose(f);
  buf[got] = 0;
  return true;
}

// whitespace-separated unsigned decimals; returns false at end of text
static bool next_u64(const char *&s, uint64_t &v) {
  while (*s && (*s < '0' || *s > '9')) ++s;
  if (!*s) return false;
  uint64_t x = 0;
  while (*s >= '0' && *s <= '9') x = x * 10 + (uint64_t)(*s++ - '0');
  v = x;
  return true;
}

// n p a[0..n) b[0..n)   (fRead, main.cc:78-95)
// The arrays a, b, ab (globals of 300000 words in main.cc:228 / monttry2.cpp:352) are page-locked buffers here, sized
// per test: the library then moves them by DMA at PCIe speed.
struct HostArray {
  coef_t *ptr = nullptr; size_t count = 0;
  bool resize(size_t n) {
    if (ptr) nttx_host_free(ptr);
    count = n;
    ptr = static_cast<coef_t *>(nttx_host_alloc((n ? n : 1) * sizeof(coef_t)));
    return ptr != nullptr;
  }
  ~HostArray() { if (ptr) nttx_host_free(ptr); }
  coef_t &operator[](size_t i) { return ptr[i]; }
  const coef_t &operator[](size_t i) const { return ptr[i]; }
  coef_t *data() { return ptr; }
};

static bool fRead(HostArray &a, HostArray &b, int *n, mod_t *p, int id) {
  std::vector<char> buf;
  if (!slurp(data_dir() + "/" + std::to_string(id) + ".in", buf)) return false;
  const char *s = buf.data();
  uint64_t nn, pp, x;
  if (!next_u64(s, nn) || !next_u64(s, pp)) return false;
#ifdef NTTX_HARNESS_INT
  if (pp > (uint64_t)INT_MAX) {
    fprintf(stderr, "%d.in: modulus %llu does not fit the (int, int) harness flavour\n", id, (unsigned long long)pp);
    return false;
  }
#endif
  *n = (int)nn;
  *p = (mod_t)pp;
  if (!a.resize(nn) || !b.resize(nn)) { fprintf(stderr, "%s\n", nttx_last_error()); return false; }
  for (uint64_t i = 0; i < nn; ++i) { if (!next_u64(s, x)) return false; a[i] = (coef_t)x; }
  for (uint64_t i = 0; i < nn; ++i) { if (!next_u64(s, x)) return false; b[i] = (coef_t)x; }
  return true;
}

// 2n-1 lines (fWrite, main.cc:97-109)
static void fWrite(const HostArray &ab, int n, int id) {
  FILE *f = fopen((out_dir() + "/" + std::to_string(id) + ".out").c_str(), "w");
  if (!f) return;
  std::string text;
  text.reserve((size_t)(2 * n) * 12);
  char tmp[24];
  for (int i = 0; i < 2 * n - 1; ++i) {
    int len = snprintf(tmp, sizeof tmp, "%llu\n", (unsigned long long)ab[i]);
    text.append(tmp, (size_t)len);
  }
  fwrite(text.data(), 1, text.size(), f);
  fclose(f);
}

// compare with <id>.out (fCheck, main.cc:111-128); also returns the verdict
static bool fCheck(const HostArray &ab, int n, int id) {
  std::vector<char> buf;
  bool ok = slurp(data_dir() + "/" + std::to_string(id) + ".out", buf);
  const char *s = ok ? buf.data() : "";
  for (int i = 0; ok && i < 2 * n - 1; ++i) {
    uint64_t x;
    if (!next_u64(s, x) || x != (uint64_t)ab[i]) ok = false;
  }
  puts(ok ? "多项式乘法结果正确" : "多项式乘法结果错误");
  return ok;
}

#ifdef NTTX_HARNESS_INT
// the harness-declared entry point exactly as monttry2.cpp:344 / pthread_normal.cpp:352 declare it
void poly_multiply(int *a, int *b, int *ab, int n, int p) {
  if (nttx_poly_multiply_i32(a, b, ab, n, p) != 0) {
    fprintf(stderr, "poly_multiply failed: %s\n", nttx_last_error());
    exit(2);
  }
}
#else
// the same entry point in the 64-bit flavour of main.cc (its timed block main.cc:283-341)
void poly_multiply(uint64_t *a, uint64_t *b, uint64_t *ab, int n, uint64_t p) {
  if (nttx_poly_multiply_u64(a, b, ab, n, p) != 0) {
    fprintf(stderr, "poly_multiply failed: %s\n", nttx_last_error());
    exit(2);
  }
}
#endif

int main(int argc, char **argv) {
  int test_begin = argc > 1 ? atoi(argv[1]) : 0, test_end = argc > 2 ? atoi(argv[2]) : 4;
  const char *dv = getenv("NTT_DEVICE");
  if (nttx_init(dv ? atoi(dv) : 0) != 0) {
    fprintf(stderr, "nttx_init failed: %s\n", nttx_last_error());
    return 2;
  }
  printf("nttx: MI355X HIP engine, device %d\n", dv ? atoi(dv) : 0);
  bool all_ok = true;
  for (int id = test_begin; id <= test_end; ++id) {
    HostArray a, b, ab;
    int n_ = 0;
    mod_t p_ = 0;
    if (!fRead(a, b, &n_, &p_, id)) {
      fprintf(stderr, "cannot read %s/%d.in\n", data_dir().c_str(), id);
      all_ok = false;
      continue;
    }
    if (!ab.resize((size_t)2 * n_)) { fprintf(stderr, "%s\n", nttx_last_error()); return 2; }
    auto t0 = std::chrono::high_resolution_clock::now();
    poly_multiply(a.data(), b.data(), ab.data(), n_, p_);
    auto t1 = std::chrono::high_resolution_clock::now();
    double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    all_ok &= fCheck(ab, n_, id);
    printf("average latency for n = %d p = %llu : %g " NTTX_LATENCY_UNIT "\n", n_, (unsigned long long)p_, ms);
    fWrite(ab, n_, id);
  }
  nttx_shutdown();
  return all_ok ? 0 : 1;
}
