// How should one product's operands cross PCIe?  Page-locked host arrays of the harness's size (two inputs of 1 MiB,
// one result of 2 MiB) moved by the copy engine (hipMemcpyAsync) against kernels that read / write the page-locked
// arrays themselves (16 bytes per lane).  Wall time from the first call to the return of the synchronize, median.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_hostmem.hip -o tools/ubench_hostmem
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, long long n16) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void tiny(unsigned *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static double med(F f, int reps = 60) {
  std::vector<double> v;
  for (int r = 0; r < reps + 5; ++r) { const double t0 = now_us(); f(); const double t = now_us() - t0; if (r >= 5) v.push_back(t); }
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main() {
  CK(hipSetDevice(0));
  const size_t MB = 1 << 20;
  void *ha, *hb, *hc, *da, *db, *dc;
  CK(hipHostMalloc(&ha, MB, hipHostMallocDefault)); CK(hipHostMalloc(&hb, MB, hipHostMallocDefault)); CK(hipHostMalloc(&hc, 2 * MB, hipHostMallocDefault));
  CK(hipMalloc(&da, MB)); CK(hipMalloc(&db, MB)); CK(hipMalloc(&dc, 2 * MB));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (hipStream_t s : {(hipStream_t) nullptr, st}) {
    const char *sn = s ? "non-blocking stream" : "null stream";
    printf("-- %s\n", sn);
    printf("synchronize on an idle stream                              %7.2f us\n", med([&] { (void)hipStreamSynchronize(s); }));
    printf("one trivial kernel + synchronize                           %7.2f us\n", med([&] { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, (unsigned *)da); (void)hipStreamSynchronize(s); }));
    printf("copy engine, 2 x 1 MiB host -> device + synchronize        %7.2f us\n",
           med([&] { (void)hipMemcpyAsync(da, ha, MB, hipMemcpyHostToDevice, s); (void)hipMemcpyAsync(db, hb, MB, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); }));
    printf("copy engine, 1 x 2 MiB device -> host + synchronize        %7.2f us\n",
           med([&] { (void)hipMemcpyAsync(hc, dc, 2 * MB, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); }));
    printf("copy engine, in + out as above, one synchronize            %7.2f us\n",
           med([&] { (void)hipMemcpyAsync(da, ha, MB, hipMemcpyHostToDevice, s); (void)hipMemcpyAsync(db, hb, MB, hipMemcpyHostToDevice, s);
                     (void)hipMemcpyAsync(hc, dc, 2 * MB, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); }));
    for (int grid : {64, 256, 1024}) {
      printf("kernel reads 2 x 1 MiB of host memory (%4d workgroups)     %7.2f us\n", grid,
             med([&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, s, (const uint4 *)ha, (uint4 *)da, (long long)(MB / 16));
                       hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, s, (const uint4 *)hb, (uint4 *)db, (long long)(MB / 16)); (void)hipStreamSynchronize(s); }));
      printf("kernel writes 2 MiB of host memory     (%4d workgroups)     %7.2f us\n", grid,
             med([&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, s, (const uint4 *)dc, (uint4 *)hc, (long long)(2 * MB / 16)); (void)hipStreamSynchronize(s); }));
    }
    printf("kernels read 2 x 1 MiB and write 2 MiB of host memory      %7.2f us\n",
           med([&] { hipLaunchKernelGGL(copy16, dim3(256), dim3(256), 0, s, (const uint4 *)ha, (uint4 *)da, (long long)(MB / 16));
                     hipLaunchKernelGGL(copy16, dim3(256), dim3(256), 0, s, (const uint4 *)hb, (uint4 *)db, (long long)(MB / 16));
                     hipLaunchKernelGGL(copy16, dim3(256), dim3(256), 0, s, (const uint4 *)dc, (uint4 *)hc, (long long)(2 * MB / 16)); (void)hipStreamSynchronize(s); }));
  }
  return 0;
}
