// Montgomery-product formulations in a dependent butterfly chain (what the NTT kernels run):
//  A: mad_u64(a,b,0); mul_lo(lo,-pinv); mad_u64(m,p,T) -> hi                       (3 ops)
//  C: mad_u64(a,b,0); mul_lo(lo, pinv); mul_hi(m,p); hi - h2 + p                    (5 ops, no 64-bit add)
//  D: mul_lo(a,b'); mul_hi(a,b); mul_hi(m,p); hi - h2 + p   with b' = b*pinv precomputed (Shoup-style, 5 ops)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
constexpr int ITERS = 4096;
template <int VAR>
__device__ __forceinline__ uint32_t mm(uint32_t a, uint32_t b, uint32_t b2, uint32_t p, uint32_t np, uint32_t pinv) {
  if (VAR == 0) { uint64_t t = (uint64_t)a * b; uint32_t m = (uint32_t)t * np; return (uint32_t)((t + (uint64_t)m * p) >> 32); }
  if (VAR == 1) { uint64_t t = (uint64_t)a * b; uint32_t m = (uint32_t)t * pinv; return (uint32_t)(t >> 32) - __umulhi(m, p) + p; }
  uint32_t m = a * b2; return __umulhi(a, b) - __umulhi(m, p) + p;
}
template <int VAR, int NCH>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed, uint32_t p, uint32_t np, uint32_t pinv) {
  uint32_t x[NCH], y[NCH];
  for (int c = 0; c < NCH; ++c) { x[c] = seed + threadIdx.x * (c + 3); y[c] = seed * 7 + threadIdx.x + c; }
  uint32_t w = seed * 2654435761u + threadIdx.x, w2 = w * pinv;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      uint32_t t = mm<VAR>(y[c], w, w2, p, np, pinv);
      uint32_t s = x[c] + t, d = x[c] - t + 2 * p;
      x[c] = s; y[c] = d;
    }
  }
  uint32_t acc = 0;
  for (int c = 0; c < NCH; ++c) acc ^= x[c] ^ y[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int VAR, int NCH>
static void run(int bpc) {
  const int blocks = 256 * bpc;
  uint32_t *d; CHECK(hipMalloc(&d, blocks * 256 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<VAR, NCH><<<blocks, 256>>>(d, 1u, 469762049u, 469762047u, 3825205249u); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k<VAR, NCH><<<blocks, 256>>>(d, 2u, 469762049u, 469762047u, 3825205249u);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)ITERS * NCH * bpc);
  printf("variant %c chains %d waves/SIMD %d  %6.2f cycles@2.4GHz per butterfly per SIMD\n", "ACD"[VAR], NCH, bpc, cyc);
  CHECK(hipFree(d));
}
int main() {
  for (int bpc : {4, 8}) {
    run<0, 1>(bpc); run<0, 4>(bpc); run<1, 1>(bpc); run<1, 4>(bpc); run<2, 1>(bpc); run<2, 4>(bpc); printf("\n");
  }
  return 0;
}
