// PCIe-inclusive latency of the host-buffer entry point as the harness uses it: long-lived arrays a, b, ab
// (main.cc:228), one product per call (main.cc:283-341).  Never the benchmark's `value`.
//   g++ -O2 -std=c++17 tools/host_latency.cpp -Iinclude -Lntt_amd/csrc -lnttx -Wl,-rpath,$PWD/ntt_amd/csrc -o /tmp/host_latency
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "nttx.h"

static uint64_t sm(uint64_t &s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

static double median_ms(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p, int reps) {
  std::vector<double> t;
  for (int r = 0; r < reps + 3; ++r) {
    auto t0 = std::chrono::high_resolution_clock::now();
    if (nttx_poly_multiply_u64(a, b, ab, n, p)) { fprintf(stderr, "%s\n", nttx_last_error()); return -1; }
    auto t1 = std::chrono::high_resolution_clock::now();
    if (r >= 3) t.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  if (nttx_init(0)) { fprintf(stderr, "%s\n", nttx_last_error()); return 1; }
  const int n = 131072;
  static uint64_t a[300000], b[300000], ab[300000];     // the harness's arrays (main.cc:228)
  uint64_t *pa = (uint64_t *)nttx_host_alloc(sizeof a), *pb = (uint64_t *)nttx_host_alloc(sizeof b), *pab = (uint64_t *)nttx_host_alloc(sizeof ab);
  if (!pa || !pb || !pab) { fprintf(stderr, "host_alloc: %s\n", nttx_last_error()); return 1; }
  for (uint64_t p : {469762049ull, 1337006139375617ull, 2147483647ull}) {
    uint64_t s = 1;
    for (int i = 0; i < n; ++i) { a[i] = pa[i] = sm(s) % p; b[i] = pb[i] = sm(s) % p; }
    const double pageable = median_ms(a, b, ab, n, p, 30);
    const double pinned = median_ms(pa, pb, pab, n, p, 30);
    bool same = true; for (int i = 0; i < 2 * n - 1; ++i) same = same && ab[i] == pab[i];
    printf("nttx_poly_multiply_u64 n=%d p=%llu path=%d: pageable arrays %.3f ms, page-locked arrays %.3f ms per call (median of 30)%s\n",
           n, (unsigned long long)p, nttx_path_for(p, n), pageable, pinned, same ? "" : "  RESULTS DIFFER");
  }
  nttx_host_free(pa); nttx_host_free(pb); nttx_host_free(pab);
  nttx_shutdown();
  return 0;
}
