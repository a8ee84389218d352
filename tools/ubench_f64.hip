// Issue cost of the FP64 VALU opcodes a floating-point modular multiplication would use, next to
// v_mad_u64_u32 (inline asm, 8 independent chains per lane).  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
constexpr int ITERS = 2048;

#define REP8(OPSTR) \
  asm volatile(OPSTR(0) "\n" OPSTR(1) "\n" OPSTR(2) "\n" OPSTR(3) "\n" OPSTR(4) "\n" OPSTR(5) "\n" OPSTR(6) "\n" OPSTR(7) \
    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(w), "v"(w2) );

#define DEFK(NAME, OPSTR) \
__global__ void __launch_bounds__(256) NAME(double *out, double seed, unsigned long long *clk) { \
  double x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 9, x5 = x0 * 11, x6 = x0 * 13, x7 = x0 * 15; \
  double w = 1.0 + 1e-9 * threadIdx.x, w2 = 1e-7 * seed; \
  unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
  for (int it = 0; it < ITERS; ++it) { REP8(OPSTR) REP8(OPSTR) REP8(OPSTR) REP8(OPSTR) } \
  unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; \
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0; }

#define S_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9"
#define S_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8"
#define S_ADD64(i) "v_add_f64 %" #i ", %" #i ", %9"
#define S_RND64(i) "v_rndne_f64 %" #i ", %" #i
DEFK(k_fma64, S_FMA64) DEFK(k_mul64, S_MUL64) DEFK(k_add64, S_ADD64) DEFK(k_rnd64, S_RND64)

__global__ void __launch_bounds__(256) k_mad64(double *out, double seed, unsigned long long *clk) {
  unsigned long long x0 = (unsigned long long)seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 9, x5 = x0 * 11, x6 = x0 * 13, x7 = x0 * 15;
  uint32_t w = 2654435761u + threadIdx.x, sp = 469762049u;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define M64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i
  for (int it = 0; it < ITERS; ++it) {
    for (int r = 0; r < 4; ++r)
      asm volatile(M64(0) "\n" M64(1) "\n" M64(2) "\n" M64(3) "\n" M64(4) "\n" M64(5) "\n" M64(6) "\n" M64(7)
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(w), "v"(sp) : "vcc");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7);
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <class K>
static void run(const char *name, K kern, int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu, threads = 256;
  double *d; unsigned long long *clk;
  CHECK(hipMalloc(&d, blocks * threads * 8)); CHECK(hipMalloc(&clk, blocks * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  kern<<<blocks, threads>>>(d, 12345.0, clk);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  kern<<<blocks, threads>>>(d, 777.0, clk);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double instr_per_wave = (double)ITERS * 32;
  const double waves_per_simd = blocks_per_cu * 4 / 4.0;
  const double cyc = (ms * 1e-3 * 2.4e9) / (instr_per_wave * waves_per_simd);
  printf("%-16s waves/SIMD=%2.0f  %8.3f ms  %6.2f cycles@2.4GHz per wave-instr per SIMD\n", name, waves_per_simd, ms, cyc);
  CHECK(hipFree(d)); CHECK(hipFree(clk));
}

int main() {
  for (int bpc : {1, 4, 8}) {
    run("v_fma_f64", k_fma64, bpc); run("v_mul_f64", k_mul64, bpc); run("v_add_f64", k_add64, bpc);
    run("v_rndne_f64", k_rnd64, bpc); run("v_mad_u64_u32", k_mad64, bpc);
    printf("\n");
  }
  return 0;
}
