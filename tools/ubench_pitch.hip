// Column-tile copy (512 rows x 128-byte segments per tile, as the 2^9-point column pass moves them) with the rows of a
// vector PITCH words apart instead of 512: does a row pitch that is not a power of two move faster?  (The intermediate
// arrays between the passes of a product are workspace whose layout is free.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

template <int NTHR, int MODE>   // MODE 0: read+write, 1: read only, 2: write only
__global__ void __launch_bounds__(NTHR) tile_copy(const uint32_t *in, uint32_t *out, long long ntiles, int pin, int pout) {
  constexpr int LPR = 8, PER = 512 * LPR / NTHR, RSTEP = NTHR / LPR;
  const int j0 = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
  uint32_t acc = 0;
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long long blk = t >> 4, ct = t & 15;
    const uint32_t *src = in + blk * 512 * pin + (ct << 5) + c4;
    uint32_t *dst = out + blk * 512 * pout + (ct << 5) + c4;
    uint4 x[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) x[k] = MODE != 2 ? *reinterpret_cast<const uint4 *>(src + (long long)(j0 + k * RSTEP) * pin) : make_uint4(k, (int)t, 3, 4);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (MODE != 1) *reinterpret_cast<uint4 *>(dst + (long long)(j0 + k * RSTEP) * pout) = x[k];
      else acc ^= x[k].x ^ x[k].y ^ x[k].z ^ x[k].w;
    }
  }
  if (MODE == 1 && acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <class F>
static double timeit(F f) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  f(); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 10; ++r) f();
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 10;
}

// one long vector viewed as 512 rows x cols words: tiles of 512 rows x 32 columns, rows `pitch` words apart
template <int MODE>
__global__ void __launch_bounds__(1024) big_copy(const uint32_t *in, uint32_t *out, long long ntiles, long long pin, long long pout) {
  constexpr int PER = 4, RSTEP = 128;
  const int j0 = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
  uint32_t acc = 0;
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint32_t *src = in + (t << 5) + c4;
    uint32_t *dst = out + (t << 5) + c4;
    uint4 x[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) x[k] = MODE != 2 ? *reinterpret_cast<const uint4 *>(src + (long long)(j0 + k * RSTEP) * pin) : make_uint4(k, (int)t, 3, 4);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (MODE != 1) *reinterpret_cast<uint4 *>(dst + (long long)(j0 + k * RSTEP) * pout) = x[k];
      else acc ^= x[k].x ^ x[k].y ^ x[k].z ^ x[k].w;
    }
  }
  if (MODE == 1 && acc == 0x12345678u) out[threadIdx.x] = acc;
}

int main() {
  const int batch = 512;
  const long long cap = (long long)batch * 512 * 640;          // room for pitches up to 640
  uint32_t *a, *b; CHECK(hipMalloc(&a, cap * 4)); CHECK(hipMalloc(&b, cap * 4));
  CHECK(hipMemset(a, 1, cap * 4)); CHECK(hipMemset(b, 2, cap * 4));
  const double gb = (double)batch * 512 * 512 * 4 / 1e9;
  const long long nt = (long long)batch * 16;
  const int pitches[][2] = {{512, 512}, {544, 544}, {528, 528}, {520, 520}, {516, 516}, {576, 576}, {512, 544}, {544, 512}, {640, 640}};
  for (auto &pp : pitches) {
    for (int grid : {8192}) {
      double m0 = timeit([&] { tile_copy<1024, 0><<<grid, 1024>>>(a, b, nt, pp[0], pp[1]); });
      double m1 = timeit([&] { tile_copy<1024, 1><<<grid, 1024>>>(a, b, nt, pp[0], pp[1]); });
      double m2 = timeit([&] { tile_copy<1024, 2><<<grid, 1024>>>(a, b, nt, pp[0], pp[1]); });
      double m3 = timeit([&] { tile_copy<512, 0><<<grid, 512>>>(a, b, nt, pp[0], pp[1]); });
      printf("pitch in %3d out %3d grid %d: r+w %7.1f GB/s  read %7.1f  write %7.1f | 512 thr r+w %7.1f\n", pp[0], pp[1], grid,
             2 * gb / m0 * 1e3, gb / m1 * 1e3, gb / m2 * 1e3, 2 * gb / m3 * 1e3);
    }
  }
  // the outer pass of a three-pass product: tiles of 512 (or 256) rows whose rows lie 2^16 / 2^17 words apart, several
  // vectors per launch (more than the memory-side cache holds), input and output pitch chosen separately
  for (int logc : {16, 17}) {
    const long long cols = 1ll << logc;
    const int nvec = logc == 16 ? 4 : 2;
    const double gbb = 512.0 * cols * 4 / 1e9 * nvec;
    const long long pads[][2] = {{0, 0}, {0, 32}, {32, 0}, {32, 32}, {0, 96}, {96, 0}, {0, 544}, {544, 0}};
    for (auto &pd : pads) {
      const long long pin = cols + pd[0], pout = cols + pd[1];
      if (512 * std::max(pin, pout) * nvec > cap) continue;
      double m0 = timeit([&] {
        for (int v = 0; v < nvec; ++v) big_copy<0><<<(int)(cols / 32), 1024>>>(a + v * 512 * pin, b + v * 512 * pout, cols / 32, pin, pout); });
      printf("%d vectors of 512 rows x 2^%d words, pitch in +%lld out +%lld: r+w %7.1f GB/s\n", nvec, logc, pd[0], pd[1], 2 * gbb / m0 * 1e3);
    }
  }
  for (int logc : {12, 15, 16, 17}) {      // columns per row: 2^21, 2^24, 2^25, 2^26-point vectors (the last one beyond the memory-side cache)
    const long long cols = 1ll << logc;
    const double gbb = 512.0 * cols * 4 / 1e9;
    for (long long pad : {0ll, 32ll, 96ll, 544ll}) {
      const long long pitch = cols + pad;
      if (512 * pitch > cap) continue;
      double m0 = timeit([&] { big_copy<0><<<(int)(cols / 32), 1024>>>(a, b, cols / 32, pitch, pitch); });
      double m1 = timeit([&] { big_copy<1><<<(int)(cols / 32), 1024>>>(a, b, cols / 32, pitch, pitch); });
      double m2 = timeit([&] { big_copy<2><<<(int)(cols / 32), 1024>>>(a, b, cols / 32, pitch, pitch); });
      printf("one vector, 512 rows x 2^%d words, pitch +%lld: r+w %7.1f GB/s  read %7.1f  write %7.1f\n", logc, pad, 2 * gbb / m0 * 1e3, gbb / m1 * 1e3, gbb / m2 * 1e3);
    }
  }
  return 0;
}
