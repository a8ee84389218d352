// Builds the REAL reference programs, from their sources where they lie under
// /root/reference, into callable shared objects / executables under oracle/_ref/.
//
// TEST INFRASTRUCTURE ONLY (checker for oracle/ntt_oracle.c and source of the golden
// fixtures in tests/golden).  No reference source text is copied: the reference file is
// pulled in by #include at build time (REF_SRC is an absolute path given on the command
// line) with two pre-processor hooks:
//   * main is renamed ref_main, so the program's driver loop becomes a callable function;
//   * sprintf is routed through a helper that swaps the hard-coded "/nttdata/" prefix
//     (main.cc:81,114; main_pthread.cc:111,144) for $NTT_REF_DATA_DIR.
// The Montgomery programs build their paths with std::string; their main() is never
// called here, only MontMul / get_result / ntt_iter_* which touch no files.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <cstdint>
#include <iostream>
#include <fstream>
#include <vector>
#include <stdexcept>

static int ref_path_sprintf(char *buf, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  const char *dir = getenv("NTT_REF_DATA_DIR");
  int n;
  if (dir && strncmp(fmt, "/nttdata/", 9) == 0) {
    std::string f = std::string(dir) + "/" + (fmt + 9);
    n = vsprintf(buf, f.c_str(), ap);
  } else {
    n = vsprintf(buf, fmt, ap);
  }
  va_end(ap);
  return n;
}

#define main ref_main
#define sprintf ref_path_sprintf
#include REF_SRC
#undef sprintf
#undef main

#if defined(REF_KIND_CRT)   // main.cc or main_pthread.cc
extern "C" {
void ref_ntt_iter_barrett(uint32_t *a, int n, uint64_t p, int root, int invert) {
  std::vector<uint32_t> v(a, a + n);
  BarrettReduction bar(p);
  ntt_iter_barrett(v, p, root, invert != 0, bar);
  std::copy(v.begin(), v.end(), a);
}
uint64_t ref_barrett_reduce(uint64_t mod, uint64_t a) { BarrettReduction bar(mod); return bar.reduce(a); }
uint64_t ref_barrett_mul_mod(uint64_t mod, uint64_t a, uint64_t b) { BarrettReduction bar(mod); return bar.mul_mod(a, b); }
#if defined(REF_HAS_COMPUTE_OMP)
void ref_ntt_compute(const uint32_t *a, const uint32_t *b, uint32_t *res, int len, uint64_t p, int root) {
  std::vector<uint32_t> va(a, a + len), vb(b, b + len), vr;
  BarrettReduction bar(p);
  ntt_compute_omp(va, vb, vr, p, root, bar);
  std::copy(vr.begin(), vr.end(), res);
}
#endif
// whole driver loop over test ids 0..4: reads $NTT_REF_DATA_DIR/<id>.in, checks against
// <id>.out, prints the latency line, writes files/<id>.out under the current directory
int ref_run_main(void) { return ref_main(0, nullptr); }
}
#endif

#if defined(REF_KIND_MONT)  // monttry2.cpp or pthread_normal.cpp
extern "C" {
// returns 0, or 1/2 when the constructor throws invalid_argument / runtime_error
int ref_mont_probe(uint64_t R, uint64_t N) {
  try { MontMul m(R, N); (void)m; } catch (const std::invalid_argument &) { return 1; } catch (const std::runtime_error &) { return 2; }
  return 0;
}
uint64_t ref_mont_to(uint64_t R, uint64_t N, uint64_t a) { MontMul m(R, N); return m.toMont(a); }
uint64_t ref_mont_from(uint64_t R, uint64_t N, uint64_t a) { MontMul m(R, N); return m.fromMont(a); }
uint64_t ref_mont_mul(uint64_t R, uint64_t N, uint64_t a, uint64_t b) { MontMul m(R, N); return m.mulMont(a, b); }
int ref_mont_modmul(uint64_t R, uint64_t N, uint64_t a, uint64_t b, uint64_t *out) {
  try { MontMul m(R, N); *out = m.ModMul(a, b); } catch (const std::invalid_argument &) { return 1; }
  return 0;
}
// main() body of monttry2.cpp:369-393 / pthread_normal.cpp:434-461 for one test on caller buffers
int ref_poly_multiply_mont(const int *pa, const int *pb, int *pab, int n, int p) {
  int len = 1;
  while (len < 2 * n) len <<= 1;
  std::vector<long long> va(len, 0), vb(len, 0);
  for (int i = 0; i < n; ++i) { va[i] = pa[i]; vb[i] = pb[i]; }
  MontMul mont(1LL << 30, p);
  for (int i = 0; i < len; ++i) { va[i] = mont.toMont(va[i]); vb[i] = mont.toMont(vb[i]); }
#if defined(REF_GET_RESULT_PTHREAD)
  std::vector<long long> cr = get_result_pthread(va, vb, p, 3, mont);
#else
  std::vector<long long> cr = get_result(va, vb, p, 3, mont);
#endif
  for (int i = 0; i < 2 * n - 1; ++i) pab[i] = (int)mont.fromMont(cr[i]);
  return 0;
}
void ref_poly_multiply_naive(int *pa, int *pb, int *pab, int n, int p) { poly_multiply(pa, pb, pab, n, p); }
}
#endif

#if defined(REF_EXE)
#undef main
int main(int argc, char **argv) { return ref_main(argc, argv); }
#endif
