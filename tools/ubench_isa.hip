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

#define S_ADD(i) "v_add_u32 %" #i ", %" #i ", %8"
#define S_SUB(i) "v_sub_u32 %" #i ", %" #i ", %8"
#define S_MIN(i) "v_min_u32 %" #i ", %" #i ", %8"
#define S_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8"
#define S_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8"
#define S_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8"
#define S_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %" #i
#define S_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9"
#define S_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8"
#define S_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8"
#define S_MIN3(i) "v_min3_u32 %" #i ", %" #i ", %8, %9"
#define S_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %" #i
#define S_MADU32(i) "v_mad_u32_u16 %" #i ", %" #i ", %8, %" #i
#define S_SUBS(i) "v_sub_u32 %" #i ", %" #i ", %9"
#define S_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc"
#define S_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8"
#define S_MULHI24(i) "v_mul_hi_u32_u24 %" #i ", %" #i ", %8"
#define S_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 17"
#define S_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9"
#define S_MOVDPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define S_ADDDPP(i) "v_add_u32_dpp %" #i ", %" #i ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"

DEFK(k_add, S_ADD) DEFK(k_sub, S_SUB) DEFK(k_min, S_MIN) DEFK(k_mullo, S_MULLO) DEFK(k_mulhi, S_MULHI)
DEFK(k_mul24, S_MUL24) DEFK(k_mad24, S_MAD24) DEFK(k_add3, S_ADD3) DEFK(k_lshladd, S_LSHLADD) DEFK(k_xor, S_XOR)
DEFK(k_min3, S_MIN3) DEFK(k_fma, S_FMA) DEFK(k_subs, S_SUBS) DEFK(k_cnd, S_CNDMASK) DEFK(k_addco, S_ADDCO)
DEFK(k_mulhi24, S_MULHI24) DEFK(k_bfe, S_BFE) DEFK(k_perm, S_PERM) DEFK(k_movdpp, S_MOVDPP) DEFK(k_adddpp, S_ADDDPP)

// v_mad_u64_u32: 64-bit destination pairs, separate kernel
__global__ void __launch_bounds__(256) k_mad64(uint32_t *out, uint32_t seed, uint32_t sp, unsigned long long *clk) {
  unsigned long long x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 9, x5 = x0 * 11, x6 = x0 * 13, x7 = x0 * 15;
  uint32_t w = seed * 2654435761u + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define M64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i
  for (int it = 0; it < ITERS; ++it) {
    for (int r = 0; r < 4; ++r)
      asm volatile(M64(0) "\n" M64(1) "\n" M64(2) "\n" M64(3) "\n" M64(4) "\n" M64(5) "\n" M64(6) "\n" M64(7)
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(w), "s"(sp) : "vcc");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <class K>
static void run(const char *name, K kern, int blocks_per_cu) {
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
  const double instr_per_wave = (double)ITERS * 32;
  const double waves_per_simd = blocks_per_cu * 4 / 4.0;   // 4 waves per block over 4 SIMDs
  // memtime ticks at 100 MHz; convert with the wall time of the launch instead
  const double cyc = (ms * 1e-3 * 2.4e9) / (instr_per_wave * waves_per_simd);
  printf("%-16s waves/SIMD=%2.0f  %8.3f ms  %6.2f cycles@2.4GHz per wave-instr per SIMD   (memtime ticks/wave %.0f)\n", name, waves_per_simd, ms, cyc, avgclk);
  CHECK(hipFree(d)); CHECK(hipFree(clk));
}

int main() {
  for (int bpc : {1, 2, 8}) {
    run("v_add_u32", k_add, bpc); run("v_sub_u32", k_sub, bpc); run("v_min_u32", k_min, bpc);
    run("v_mul_lo_u32", k_mullo, bpc); run("v_mul_hi_u32", k_mulhi, bpc); run("v_mad_u64_u32", k_mad64, bpc);
    run("v_mul_u32_u24", k_mul24, bpc); run("v_mad_u32_u24", k_mad24, bpc); run("v_mul_hi_u32_u24", k_mulhi24, bpc);
    run("v_add3_u32", k_add3, bpc); run("v_lshl_add_u32", k_lshladd, bpc); run("v_xor_b32", k_xor, bpc);
    run("v_min3_u32", k_min3, bpc); run("v_fma_f32", k_fma, bpc); run("v_sub_u32(sgpr)", k_subs, bpc);
    run("v_cndmask_b32", k_cnd, bpc); run("v_add_co_u32", k_addco, bpc); run("v_bfe_u32", k_bfe, bpc);
    run("v_perm_b32", k_perm, bpc); run("v_mov_b32_dpp", k_movdpp, bpc); run("v_add_u32_dpp", k_adddpp, bpc);
    printf("\n");
  }
  return 0;
}
