// Micro-benchmark: issue rate of the integer VALU instructions the NTT
// butterflies are built from, on gfx950.  Prints giga lane-ops per second
// per opcode with every CU busy (8 independent chains per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed, uint32_t p, uint32_t npinv) {
  uint32_t x[CHAINS];
  uint32_t w = seed * 2654435761u + threadIdx.x;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) x[c] = seed + threadIdx.x * 17 + c * 1315423911u;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (OP == 0) x[c] = x[c] + w;                                 // v_add_u32
      if (OP == 1) x[c] = x[c] * w;                                 // v_mul_lo_u32
      if (OP == 2) x[c] = __umulhi(x[c], w);                        // v_mul_hi_u32
      if (OP == 3) { uint64_t t = (uint64_t)x[c] * w + x[c]; x[c] = (uint32_t)(t >> 32) ^ (uint32_t)t; } // v_mad_u64_u32 (+xor)
      if (OP == 4) x[c] = __umul24(x[c], w) ;       // v_mul_u32_u24
      if (OP == 5) {                                                // Montgomery multiply (3 multiplies)
        uint64_t t = (uint64_t)x[c] * w;
        uint32_t m = (uint32_t)t * npinv;
        x[c] = (uint32_t)((t + (uint64_t)m * p) >> 32);
      }
      if (OP == 6) { uint32_t s = x[c] + w; uint32_t d = s - p; x[c] = s < d ? s : d; } // add, sub, min
      if (OP == 7) {                                                // full lazy DIF butterfly on chain pairs
        if ((c & 1) == 0) {
          uint32_t a = x[c], b = x[c + 1];
          uint32_t s = a + b, s2 = s - 2 * p; s = s < s2 ? s : s2;
          uint32_t d = a - b + 2 * p;
          uint64_t t = (uint64_t)d * w;
          uint32_t m = (uint32_t)t * npinv;
          x[c] = s; x[c + 1] = (uint32_t)((t + (uint64_t)m * p) >> 32);
        }
      }
      if (OP == 8) { double f = __hiloint2double(x[c], w); f = fma(f, 1.0000001, 0.5); x[c] = __double2hiint(f); } // v_fma_f64
    }
  }
  uint32_t acc = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) acc ^= x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
static void run(const char *name, double ops_per_chain_iter) {
  const int blocks = 256 * 8, threads = 256;
  uint32_t *d; CHECK(hipMalloc(&d, blocks * threads * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<OP><<<blocks, threads>>>(d, 12345u, 469762049u, 469762047u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) k<OP><<<blocks, threads>>>(d, 12345u + r, 469762049u, 469762047u);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  double laneops = (double)blocks * threads * ITERS * CHAINS * ops_per_chain_iter;
  printf("%-28s %8.3f ms  %9.1f G lane-ops/s  (%.2f cycles/wave-instr/SIMD @2.4GHz)\n", name, ms,
         laneops / ms * 1e-6, 1024.0 * 64 * 2.4e9 / (laneops / (ms * 1e-3)) );
  CHECK(hipFree(d));
}

int main() {
  hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
  printf("device %s CUs=%d clock=%d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
  run<0>("v_add_u32", 1);
  run<1>("v_mul_lo_u32", 1);
  run<2>("v_mul_hi_u32", 1);
  run<3>("v_mad_u64_u32(+xor)", 1);
  run<4>("v_mul_u32_u24", 1);
  run<5>("montmul32 (per montmul)", 1);
  run<6>("add+sub+min (per triple)", 1);
  run<7>("lazy DIF butterfly (per bf)", 0.5);
  run<8>("v_fma_f64", 1);
  return 0;
}
