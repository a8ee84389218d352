// Issue cost of individual gfx950 VALU opcodes (inline asm, 8 independent chains per lane,
// 8 waves per SIMD).  Prints cycles per wave-instruction per SIMD at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
constexpr int ITERS = 2048;

#define REP8(OPSTR) \
  asm volatile(OPSTR(0) "\n" OPSTR(1) "\n" OPSTR(2) "\n" OPSTR(3) "\n" OPSTR(4) "\n" OPSTR(5) "\n" OPSTR(6) "\n" OPSTR(7) \
    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(w), "s"(sp) );

#define DEFK(NAME, OPSTR) \
__global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed, uint32_t sp, unsigned long long *clk) { \
  uint32_t x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 9, x5 = x0 * 11, x6 = x0 * 13, x7 = x0 * 15; \
  uint32_t w = seed * 2654435761u + threadIdx.x; \
  unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
  for (int it = 0; it < ITERS; ++it) { REP8(OPSTR) REP8(OPSTR) REP8(OPSTR) REP8(OPSTR) } \
  unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7; \
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0; }


// mixes: how does a fast-class op (v_add_u32) issue when interleaved with multiplier-class ops?
#define S_ADD(i) "v_add_u32 %" #i ", %" #i ", %8"
#define S_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8"
#define S_MIN(i) "v_min_u32 %" #i ", %" #i ", %8"
// mul then add on the same register (dependent pair): 2 instructions per slot
#define S_MULADD_DEP(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n v_add_u32 %" #i ", %" #i ", %8"
// add on even chains, mul on odd chains
#define S_ALT(i) "v_add_u32 %" #i ", %" #i ", %8"
#define S_ADD2(i) "v_add_u32 %" #i ", %" #i ", %8\n v_add_u32 %" #i ", %" #i ", %8"
#define S_MUL_ADD3(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n v_add_u32 %" #i ", %" #i ", %8\n v_sub_u32 %" #i ", %" #i ", %8\n v_add_u32 %" #i ", %" #i ", %8"
#define S_MIN_ADD(i) "v_min_u32 %" #i ", %" #i ", %8\n v_add_u32 %" #i ", %" #i ", %8"
DEFK(k_add, S_ADD) DEFK(k_mullo, S_MULLO) DEFK(k_muladd_dep, S_MULADD_DEP) DEFK(k_add2, S_ADD2) DEFK(k_mul_add3, S_MUL_ADD3) DEFK(k_min_add, S_MIN_ADD)

__global__ void __launch_bounds__(256) k_alt(uint32_t *out, uint32_t seed, uint32_t sp, unsigned long long *clk) {
  uint32_t x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 9, x5 = x0 * 11, x6 = x0 * 13, x7 = x0 * 15;
  uint32_t w = seed * 2654435761u + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    for (int r = 0; r < 4; ++r)
      asm volatile("v_mul_lo_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                   "v_mul_lo_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(w), "s"(sp));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(256) k_inter(uint32_t *out, uint32_t seed, uint32_t sp, unsigned long long *clk) {
  uint32_t x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 9, x5 = x0 * 11, x6 = x0 * 13, x7 = x0 * 15;
  uint32_t w = seed * 2654435761u + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    for (int r = 0; r < 2; ++r)
      asm volatile("v_mul_lo_u32 %0, %0, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_mul_lo_u32 %1, %1, %8\n v_add_u32 %7, %7, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_mul_lo_u32 %2, %2, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n v_add_u32 %4, %4, %8\n v_mul_lo_u32 %3, %3, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(w), "s"(sp));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(256) k_grouped(uint32_t *out, uint32_t seed, uint32_t sp, unsigned long long *clk) {
  uint32_t x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 9, x5 = x0 * 11, x6 = x0 * 13, x7 = x0 * 15;
  uint32_t w = seed * 2654435761u + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    for (int r = 0; r < 2; ++r)
      asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(w), "s"(sp));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(256) k_p4(uint32_t *out, uint32_t seed, uint32_t sp, unsigned long long *clk) {
  uint32_t x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 9, x5 = x0 * 11, x6 = x0 * 13, x7 = x0 * 15;
  uint32_t w = seed * 2654435761u + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    for (int r = 0; r < 2; ++r)
      asm volatile("v_mul_lo_u32 %0, %0, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n v_mul_lo_u32 %1, %1, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n v_mul_lo_u32 %2, %2, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n v_mul_lo_u32 %3, %3, %8"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(w), "s"(sp));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(256) k_gmin(uint32_t *out, uint32_t seed, uint32_t sp, unsigned long long *clk) {
  uint32_t x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 9, x5 = x0 * 11, x6 = x0 * 13, x7 = x0 * 15;
  uint32_t w = seed * 2654435761u + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    for (int r = 0; r < 2; ++r)
      asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(w), "s"(sp));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(256) k_imin(uint32_t *out, uint32_t seed, uint32_t sp, unsigned long long *clk) {
  uint32_t x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 9, x5 = x0 * 11, x6 = x0 * 13, x7 = x0 * 15;
  uint32_t w = seed * 2654435761u + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    for (int r = 0; r < 2; ++r)
      asm volatile("v_mul_lo_u32 %0, %0, %8\n v_add_u32 %4, %4, %8\n v_min_u32 %4, %4, %8\n v_add_u32 %4, %4, %8\n v_mul_lo_u32 %1, %1, %8\n v_add_u32 %5, %5, %8\n v_min_u32 %5, %5, %8\n v_add_u32 %5, %5, %8\n v_mul_lo_u32 %2, %2, %8\n v_add_u32 %6, %6, %8\n v_min_u32 %6, %6, %8\n v_add_u32 %6, %6, %8\n v_mul_lo_u32 %3, %3, %8\n v_add_u32 %7, %7, %8\n v_min_u32 %7, %7, %8\n v_add_u32 %7, %7, %8"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(w), "s"(sp));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <class K>
static void run(const char *name, K kern, int blocks_per_cu, double per_slot) {
  const int blocks = 256 * blocks_per_cu, threads = 256;
  uint32_t *d; unsigned long long *clk;
  CHECK(hipMalloc(&d, blocks * threads * 4)); CHECK(hipMalloc(&clk, blocks * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  kern<<<blocks, threads>>>(d, 12345u, 469762049u, clk);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  kern<<<blocks, threads>>>(d, 777u, 469762049u, clk);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[256]; CHECK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
  double avgclk = 0; for (int i = 0; i < 256; ++i) avgclk += (double)h[i]; avgclk /= 256;
  const double instr_per_wave = (double)ITERS * 32 * per_slot;
  const double waves_per_simd = blocks_per_cu * 4 / 4.0;   // 4 waves per block over 4 SIMDs
  // memtime ticks at 100 MHz; convert with the wall time of the launch instead
  const double cyc = (ms * 1e-3 * 2.4e9) / (instr_per_wave * waves_per_simd);
  printf("%-16s waves/SIMD=%2.0f  %8.3f ms  %6.2f cycles@2.4GHz per wave-instr per SIMD   (memtime ticks/wave %.0f)\n", name, waves_per_simd, ms, cyc, avgclk);
  CHECK(hipFree(d)); CHECK(hipFree(clk));
}

int main() {
  for (int bpc : {2, 4, 8}) {
    run("add", k_add, bpc, 1); run("mul_lo", k_mullo, bpc, 1); run("add,add dep", k_add2, bpc, 2);
    run("mul,add dep", k_muladd_dep, bpc, 2); run("mul|add indep", k_alt, bpc, 1);
    run("mul,add,sub,add", k_mul_add3, bpc, 4); run("min,add dep", k_min_add, bpc, 2);
    run("m,a,a,a x4", k_inter, bpc, 1); run("m x4, a x12", k_grouped, bpc, 1); run("m,a x4 runs", k_p4, bpc, 1);
    run("m4,min4,a8", k_gmin, bpc, 1); run("m,a,min,a x4", k_imin, bpc, 1);
    printf("\n");
  }
  return 0;
}
