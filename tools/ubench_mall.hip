// Does a working set that fits the 256 MiB Infinity Cache stream faster than HBM?
// Ping-pong copy a->b, b->a over buffers of increasing size; same kernel, same grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
template <int MODE>
__global__ void __launch_bounds__(256) flat_copy(const uint4 *in, uint4 *out, long long n16) {
  uint32_t acc = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
    uint4 x = MODE != 2 ? in[i] : make_uint4(i, 2, 3, 4);
    if (MODE != 1) out[i] = x; else acc ^= x.x ^ x.y ^ x.z ^ x.w;
  }
  if (MODE == 1 && acc == 0x12345678u) out[threadIdx.x] = make_uint4(acc, 0, 0, 0);
}
int main() {
  const size_t maxb = (size_t)1 << 30;
  uint4 *a, *b; CHECK(hipMalloc(&a, maxb)); CHECK(hipMalloc(&b, maxb));
  CHECK(hipMemset(a, 1, maxb)); CHECK(hipMemset(b, 2, maxb));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (size_t mib : {4, 8, 16, 32, 64, 96, 128, 192, 256, 512, 1024}) {
    const long long n16 = (long long)mib * 1024 * 1024 / 16;
    const int reps = (int)(8192 / mib) + 2;
    for (int mode = 0; mode < 3; ++mode) {
      auto go = [&](int r) {
        const uint4 *src = (r & 1) ? b : a; uint4 *dst = (r & 1) ? a : b;
        if (mode == 0) flat_copy<0><<<2048, 256>>>(src, dst, n16);
        if (mode == 1) flat_copy<1><<<2048, 256>>>(a, b, n16);
        if (mode == 2) flat_copy<2><<<2048, 256>>>(a, b, n16);
      };
      go(0); go(1); CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) go(r);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      const double bytes = (double)mib * 1048576 * (mode == 0 ? 2 : 1);
      printf("%5zu MiB/buffer  %-10s %8.4f ms  %8.1f GB/s\n", mib, mode == 0 ? "copy r+w" : mode == 1 ? "read" : "write", ms, bytes / ms * 1e-6);
    }
  }
  return 0;
}
