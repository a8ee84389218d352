// Achievable HBM bandwidth of the column-pass access pattern: tiles of ROWS rows x SEG bytes,
// rows STRIDE bytes apart, read into registers and written back to a second buffer with the
// same shape.  Compared with a plain contiguous copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

// vector = 2^18 words (1 MiB) viewed as 512 rows x 512 words; batch of such vectors.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int LOGC, int NTHR, int MODE, int NT = 0, int PAIR = 0>   // MODE 0: read+write, 1: read only, 2: write only; NT: 1 = nontemporal loads, 2 = stores, 3 = both
__global__ void __launch_bounds__(NTHR) tile_copy(const uint32_t *in, uint32_t *out, long long ntiles) {
  constexpr int COLS = 1 << LOGC, LPR = COLS / 4, LOG_LPR = LOGC - 2;
  constexpr int PER = 512 * LPR / NTHR, RSTEP = NTHR / LPR;
  const int log_tpb = 9 - LOGC;
  const int j0 = threadIdx.x >> LOG_LPR, c4 = (threadIdx.x & (LPR - 1)) * 4;
  uint32_t acc = 0;
  for (long long t0 = blockIdx.x; t0 < ntiles; t0 += gridDim.x) {
    long long t = t0;
    if (PAIR) {   // blocks b and b+8 (same XCD under round-robin dispatch) take the two halves of a 128-byte line
      const long long g = t0 >> 4, r = t0 & 15;          // groups of 16 consecutive block slots
      t = (g << 4) + ((r & 7) << 1) + (r >> 3);
    }
    const long long blk = t >> log_tpb, ct = t & ((1 << log_tpb) - 1);
    const long long base = (blk << 18) + (ct << LOGC) + c4;
    uint4 x[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (MODE != 2) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(in + base + ((long long)(j0 + k * RSTEP) << 9));
        u32x4 v = (NT & 1) ? __builtin_nontemporal_load(src) : *src;
        x[k] = make_uint4(v.x, v.y, v.z, v.w);
      }
      else x[k] = make_uint4(k, t, 3, 4);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (MODE != 1) {
        u32x4 v = {x[k].x, x[k].y, x[k].z, x[k].w};
        u32x4 *dst = reinterpret_cast<u32x4 *>(out + base + ((long long)(j0 + k * RSTEP) << 9));
        if (NT & 2) __builtin_nontemporal_store(v, dst); else *dst = v;
      }
      else acc ^= x[k].x ^ x[k].y ^ x[k].z ^ x[k].w;
    }
  }
  if (MODE == 1 && acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <int MODE>
__global__ void __launch_bounds__(256) flat_copy(const uint4 *in, uint4 *out, long long n16) {
  uint32_t acc = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
    uint4 x = MODE != 2 ? in[i] : make_uint4(i, 2, 3, 4);
    if (MODE != 1) out[i] = x; else acc ^= x.x ^ x.y ^ x.z ^ x.w;
  }
  if (MODE == 1 && acc == 0x12345678u) out[threadIdx.x] = make_uint4(acc, 0, 0, 0);
}

template <class F>
static double timeit(F f) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  f(); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) f();
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5;
}

int main() {
  const int batch = 512;                         // 512 MiB per buffer
  const long long words = (long long)batch << 18;
  uint32_t *a, *b; CHECK(hipMalloc(&a, words * 4)); CHECK(hipMalloc(&b, words * 4));
  CHECK(hipMemset(a, 1, words * 4)); CHECK(hipMemset(b, 2, words * 4));
  const double gb = words * 4 / 1e9;
  const char *mn[3] = {"read+write", "read only", "write only"};
  for (int grid : {1024, 2048, 4096}) {
    printf("grid %d\n", grid);
#define FLAT(M) { double ms = timeit([&] { flat_copy<M><<<grid, 256>>>((const uint4 *)a, (uint4 *)b, words / 4); }); \
      printf("  flat 16B/lane      %-10s %7.3f ms  %7.1f GB/s\n", mn[M], ms, gb * (M == 0 ? 2 : 1) / ms * 1e3); }
    FLAT(0) FLAT(1) FLAT(2)
#define TILE(LC, NT, M) { long long nt = (long long)batch << (9 - LC); \
      double ms = timeit([&] { tile_copy<LC, NT, M><<<grid, NT>>>(a, b, nt); }); \
      printf("  tile seg %3d B thr %3d %-10s %7.3f ms  %7.1f GB/s\n", 4 << LC, NT, mn[M], ms, gb * (M == 0 ? 2 : 1) / ms * 1e3); }
    TILE(4, 256, 0) TILE(4, 256, 1) TILE(4, 256, 2)
    TILE(5, 256, 0) TILE(5, 512, 0) TILE(5, 512, 1) TILE(5, 512, 2)
    TILE(6, 512, 0) TILE(6, 1024, 0) TILE(6, 512, 1) TILE(6, 512, 2)
    TILE(7, 1024, 0)
#define TILENT(LC, NT_, M, NTF) { long long nt = (long long)batch << (9 - LC); \
      double ms = timeit([&] { tile_copy<LC, NT_, M, NTF><<<grid, NT_>>>(a, b, nt); }); \
      printf("  tile seg %3d B thr %3d %-10s nt=%d %7.3f ms  %7.1f GB/s\n", 4 << LC, NT_, mn[M], NTF, ms, gb * (M == 0 ? 2 : 1) / ms * 1e3); }
#define TILEP(LC, NT_, M) { long long nt = (long long)batch << (9 - LC); \
      double ms = timeit([&] { tile_copy<LC, NT_, M, 0, 1><<<grid, NT_>>>(a, b, nt); }); \
      printf("  tile seg %3d B thr %3d %-10s XCD-paired %7.3f ms  %7.1f GB/s\n", 4 << LC, NT_, mn[M], ms, gb * (M == 0 ? 2 : 1) / ms * 1e3); }
    TILEP(4, 256, 0) TILEP(4, 256, 1) TILEP(4, 256, 2)
    TILENT(5, 512, 0, 1) TILENT(5, 512, 0, 2) TILENT(5, 512, 0, 3) TILENT(5, 512, 1, 1) TILENT(5, 512, 2, 2)
  }
  return 0;
}
