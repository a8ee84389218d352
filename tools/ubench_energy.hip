// Which modular-product formulation yields the most butterflies per second when the WHOLE chip runs it for seconds,
// i.e. at the socket's power cap, where the clock the part holds depends on the energy of the instruction mix and not
// only on how many issue slots it fills (tools/ubench_mont.hip measures the latter in launches of a few milliseconds).
//   0  unsigned REDC, lazy [0, 2p):  mad_u64_u32, mul_lo, mad_u64_u32; butterfly add, sub, add(2p)     (the column kernels)
//   1  signed REDC (-p, p):          mad_i64_i32, mul_lo, mad_i64_i32; butterfly add, sub               (the row kernels)
//   2  unsigned, high halves only:   mad_u64_u32, mul_lo, mul_hi, sub, add(p); add, sub, add(2p)
//   3  unsigned, twiddle pair (w, w * p^-1): mul_lo, mul_hi, mul_hi, sub, add(p); add, sub, add(2p)
//   4  signed, twiddle pair:         mul_lo, mul_hi_i32, mul_hi_i32, sub; add, sub
// Every variant runs dependent butterflies on NCH independent chains per lane, 2048 workgroups of 256 threads, launch after
// launch for about four seconds; the rate is taken over the second half.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_energy.hip -o tools/ubench_energy     Run: tools/ubench_energy [variant]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 1 << 13, NCH = 8, BLOCKS = 2048, THREADS = 256;

template <int VAR>
__device__ __forceinline__ void butterfly(uint32_t &x, uint32_t &y, uint32_t w, uint32_t w2, uint32_t p, uint32_t np, uint32_t pinv) {
  if constexpr (VAR == 0) {
    const uint64_t t = (uint64_t)y * w;
    const uint32_t m = (uint32_t)t * np;
    const uint32_t r = (uint32_t)((t + (uint64_t)m * p) >> 32);
    const uint32_t s = x + r, d = x - r + 2 * p;
    x = s; y = d;
  } else if constexpr (VAR == 1) {
    const int64_t t = (int64_t)(int32_t)y * (int64_t)(int32_t)w;
    const uint32_t m = (uint32_t)t * pinv;
    const int64_t r64 = (int64_t)(int32_t)m * (int64_t)(-(int32_t)p) + t;
    const uint32_t r = (uint32_t)((uint64_t)r64 >> 32);
    const uint32_t s = x + r, d = x - r;
    x = s; y = d;
  } else if constexpr (VAR == 2) {
    const uint64_t t = (uint64_t)y * w;
    const uint32_t m = (uint32_t)t * pinv;
    const uint32_t r = (uint32_t)(t >> 32) - __umulhi(m, p) + p;
    const uint32_t s = x + r, d = x - r + 2 * p;
    x = s; y = d;
  } else if constexpr (VAR == 3) {
    const uint32_t m = y * w2;
    const uint32_t r = __umulhi(y, w) - __umulhi(m, p) + p;
    const uint32_t s = x + r, d = x - r + 2 * p;
    x = s; y = d;
  } else {
    const uint32_t m = y * w2;
    const uint32_t r = (uint32_t)__mulhi((int)y, (int)w) - (uint32_t)__mulhi((int)m, (int)p);
    const uint32_t s = x + r, d = x - r;
    x = s; y = d;
  }
}

template <int VAR>
__global__ void __launch_bounds__(THREADS) chain_kernel(uint32_t *out, uint32_t seed, uint32_t p, uint32_t np, uint32_t pinv) {
  uint32_t x[NCH], y[NCH], w[4], w2[4];
  for (int c = 0; c < NCH; ++c) { x[c] = (seed + threadIdx.x * (c + 3)) % p; y[c] = (seed * 7 + threadIdx.x + c) % p; }
  for (int i = 0; i < 4; ++i) { w[i] = (seed * 2654435761u + threadIdx.x * 97u + i * 1315423911u) % p; w2[i] = w[i] * pinv; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      butterfly<VAR>(x[c], y[c], w[c & 3], w2[c & 3], p, np, pinv);   // (values wrap; only the instruction stream matters here)
    }
  }
  uint32_t acc = 0;
  for (int c = 0; c < NCH; ++c) acc ^= x[c] ^ y[c];
  out[blockIdx.x * THREADS + threadIdx.x] = acc;
}

template <int VAR>
static void run(const char *name, uint32_t *d) {
  const uint32_t p = 469762049u, pinv = 3825205249u, np = 0u - pinv;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&] { hipLaunchKernelGGL((chain_kernel<VAR>), dim3(BLOCKS), dim3(THREADS), 0, 0, d, 1u, p, np, pinv); };
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float first = 0; CK(hipEventElapsedTime(&first, e0, e1));
  const int reps = (int)(2000.0f / first) + 1;          // about two seconds per half
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bf = (double)BLOCKS * THREADS * NCH * ITERS * reps;
  printf("variant %d %-44s first launch %7.2f ms, sustained %7.2f ms per launch, %6.3f T butterflies/s sustained (%5.3f T/s in the first launch)\n",
         VAR, name, first, ms / reps, bf / (ms * 1e-3) / 1e12, bf / reps / (first * 1e-3) / 1e12);
  fflush(stdout);
}

int main(int argc, char **argv) {
  CK(hipSetDevice(0));
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  uint32_t *d; CK(hipMalloc(&d, (size_t)BLOCKS * THREADS * 4));
  if (only < 0 || only == 0) run<0>("unsigned REDC (mad64, mul_lo, mad64)", d);
  if (only < 0 || only == 1) run<1>("signed REDC (mad_i64, mul_lo, mad_i64)", d);
  if (only < 0 || only == 2) run<2>("unsigned, mad64 + mul_lo + mul_hi", d);
  if (only < 0 || only == 3) run<3>("unsigned, twiddle pair (mul_lo, 2 mul_hi)", d);
  if (only < 0 || only == 4) run<4>("signed, twiddle pair (mul_lo, 2 mul_hi_i32)", d);
  CK(hipFree(d));
  return 0;
}
