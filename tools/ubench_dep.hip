// Dependent-chain cost of the Montgomery butterfly on gfx950: one serial chain per lane
// (as hipcc schedules it) versus 2 and 4 interleaved chains, at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
constexpr int ITERS = 4096;

__device__ __forceinline__ uint32_t mm(uint32_t a, uint32_t b, uint32_t p, uint32_t np) {
  uint64_t t = (uint64_t)a * b; uint32_t m = (uint32_t)t * np; return (uint32_t)((t + (uint64_t)m * p) >> 32);
}
template <int NCH>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed, uint32_t p, uint32_t np) {
  uint32_t x[NCH], y[NCH];
  for (int c = 0; c < NCH; ++c) { x[c] = seed + threadIdx.x * (c + 3); y[c] = seed * 7 + threadIdx.x + c; }
  uint32_t w = seed * 2654435761u + threadIdx.x;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {            // DIT butterfly: serial chain through the product
      uint32_t t = mm(y[c], w, p, np);
      uint32_t s = x[c] + t, d = x[c] - t + 2 * p;
      x[c] = s; y[c] = d;
    }
    if (NCH > 1) { asm volatile("" ::: "memory"); }
  }
  uint32_t acc = 0;
  for (int c = 0; c < NCH; ++c) acc ^= x[c] ^ y[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int NCH>
static void run(int bpc) {
  const int blocks = 256 * bpc;
  uint32_t *d; CHECK(hipMalloc(&d, blocks * 256 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<NCH><<<blocks, 256>>>(d, 1u, 469762049u, 469762047u); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k<NCH><<<blocks, 256>>>(d, 2u, 469762049u, 469762047u);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bf_per_wave = (double)ITERS * NCH;
  const double cyc = ms * 1e-3 * 2.4e9 / (bf_per_wave * bpc);   // bpc waves per SIMD
  printf("chains/lane %d  waves/SIMD %d  %7.3f ms  %6.2f cycles@2.4GHz per butterfly per SIMD (6 VALU: mad,mul,mad,add,sub,add)\n", NCH, bpc, ms, cyc);
  CHECK(hipFree(d));
}
int main() {
  for (int bpc : {1, 2, 4, 8}) { run<1>(bpc); run<2>(bpc); run<4>(bpc); run<8>(bpc); printf("\n"); }
  return 0;
}
