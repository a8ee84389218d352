// Prototype of a modular product on FP64 lanes for p < 2^51, checked against exact integer arithmetic on
// the host, and a butterfly-chain timing against the 64-bit-lane Montgomery product of the library.
//   r = x*w - q*p,  q = rndne(x * wq),  wq = w/p rounded to double, |x| <= K p  =>  |r| <= (0.5 + K p 2^-52) p
// h = fl(x*w), l = fma(x, w, -h) give x*w = h + l exactly; h - q*p is an integer below 2^53, so the fused
// multiply-add returns it exactly, and adding l stays exact.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../ntt_amd/csrc/nttx_arith.hpp"
#include "../ntt_amd/csrc/nttx_kernels64.hpp"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

__device__ __forceinline__ double fpmod(double x, double w, double wq, double p) {
  const double h = x * w;
  const double l = fma(x, w, -h);
  const double q = rint(x * wq);
  return fma(-q, p, h) + l;
}
// x - rndne(x/p) p: |r| <= (0.5 + |x| 2^-52) p
__device__ __forceinline__ double fpred(double x, double pinv, double p) { return fma(-rint(x * pinv), p, x); }

__global__ void k_check(const double *x, const double *w, const double *wq, double *r, double p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) r[i] = fpmod(x[i], w[i], wq[i], p);
}

constexpr int CHAIN = 4096;
// Gentleman-Sande butterflies in a dependent chain: (a, b) -> (a + b, (a - b) w), sums reduced each time
__global__ void __launch_bounds__(256) k_chain_fp(double *out, double p, double pinv, double w, double wq) {
  double a[4], b[4];
  for (int k = 0; k < 4; ++k) { a[k] = threadIdx.x + 3.0 * k; b[k] = 7.0 * threadIdx.x + k; }
  for (int it = 0; it < CHAIN; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double s = fpred(a[k] + b[k], pinv, p), d = fpmod(a[k] - b[k], w, wq, p);
      a[k] = s; b[k] = d;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a[0] + a[1] + a[2] + a[3] + b[0] + b[1] + b[2] + b[3];
}
__global__ void __launch_bounds__(256) k_chain_int(uint64_t *out, uint64_t p, uint64_t npinv, uint64_t w) {
  uint64_t a[4], b[4];
  for (int k = 0; k < 4; ++k) { a[k] = threadIdx.x + 3 * k; b[k] = 7 * threadIdx.x + k; }
  const uint64_t p2 = 2 * p;
  for (int it = 0; it < CHAIN; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {      // the Harvey form (values below 2p) so that the chain can run for ever
      const uint64_t s = nttx::csub64(a[k] + b[k], p2), d = nttx::mont64_lazy<true>(a[k] - b[k] + p2, w, p, npinv);
      a[k] = s; b[k] = d;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3];
}

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main() {
  const uint64_t primes[] = {1337006139375617ull, 2251756864012289ull, 1125899437080577ull, 263882790666241ull, 8588886017ull};
  for (uint64_t p : primes) {
    const int n = 1 << 22;
    const double kmax = std::min(3.3, 4503599627370496.0 / (double)p * 0.99);     // |x| < 2^52
    std::vector<double> x(n), w(n), wq(n), r(n);
    for (int i = 0; i < n; ++i) {
      const uint64_t wi = (i < 8) ? (i & 1 ? p - 1 - (i >> 1) : (uint64_t)(i >> 1)) : rng() % p;
      const double lim = kmax * (double)p;
      int64_t xi = (int64_t)(rng() % (uint64_t)(2 * lim)) - (int64_t)lim;
      if (i < 64) xi = (i & 1 ? 1 : -1) * (int64_t)(lim - (i >> 1));                // the extremes too
      x[i] = (double)xi; w[i] = (double)wi; wq[i] = (double)wi / (double)p;
    }
    double *dx, *dw, *dq, *dr;
    CHECK(hipMalloc(&dx, n * 8)); CHECK(hipMalloc(&dw, n * 8)); CHECK(hipMalloc(&dq, n * 8)); CHECK(hipMalloc(&dr, n * 8));
    CHECK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dw, w.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dq, wq.data(), n * 8, hipMemcpyHostToDevice));
    k_check<<<n / 256, 256>>>(dx, dw, dq, dr, (double)p, n);
    CHECK(hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost));
    long long bad = 0; double worst = 0;
    for (int i = 0; i < n; ++i) {
      const __int128 exact = (__int128)(int64_t)x[i] * (__int128)(int64_t)w[i];
      const __int128 diff = exact - (__int128)(int64_t)r[i];
      if (r[i] != std::floor(r[i]) || diff % (__int128)p != 0) ++bad;
      worst = std::max(worst, std::fabs(r[i]) / (double)p);
    }
    printf("p = %llu (2^%.2f): %d products with |x| < %.2f p: %lld wrong, max |r|/p = %.4f (bound %.4f)\n", (unsigned long long)p,
           std::log2((double)p), n, kmax, bad, worst, 0.5 + kmax * (double)p / 4503599627370496.0);
    CHECK(hipFree(dx)); CHECK(hipFree(dw)); CHECK(hipFree(dq)); CHECK(hipFree(dr));
  }
  // timing: 256 CUs x 8 blocks, 4 butterflies per thread and iteration
  const uint64_t p = primes[0];
  const nttx::PrimeCtx64 c = nttx::make_ctx64(p);
  double *od; uint64_t *oi;
  const int blocks = 256 * 8;
  CHECK(hipMalloc(&od, blocks * 256 * 8)); CHECK(hipMalloc(&oi, blocks * 256 * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    float ms_fp, ms_int;
    CHECK(hipEventRecord(e0)); k_chain_fp<<<blocks, 256>>>(od, (double)p, 1.0 / (double)p, 123456789012345.0, 123456789012345.0 / (double)p);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_fp, e0, e1));
    CHECK(hipEventRecord(e0)); k_chain_int<<<blocks, 256>>>(oi, p, c.npinv, 123456789012345ull);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_int, e0, e1));
    const double bf = (double)blocks * 256 * 4 * CHAIN;
    printf("butterfly chains: FP64 %.3f ms (%.2f ps per butterfly), 64-bit Montgomery %.3f ms (%.2f ps per butterfly), ratio %.2f\n",
           ms_fp, ms_fp * 1e9 / bf, ms_int, ms_int * 1e9 / bf, ms_int / ms_fp);
  }
  return 0;
}
