// Launch and synchronisation costs on MI355X that decide how a product of a few microseconds should be cut into kernels:
//   1. three dependent trivial kernels, eager: back-to-back rate and launch-to-completion latency of one triple
//   2. the same three kernels as a HIP graph: replay latency and back-to-back replay rate
//   3. a grid-wide barrier inside one resident launch (one monotonic counter, agent-scope atomics, bounded polling),
//      bare and with the release / acquire fences a data hand-off needs
//   4. the same barrier between the workgroups of ONE XCD only (a counter per XCD, membership read from XCC_ID)
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_sync.hip -o tools/ubench_sync     Run: tools/ubench_sync
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void tiny_kernel(unsigned *p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += 1;
}

constexpr int kMaxSpin = 1 << 22;   // every poll loop ends after this many polls whatever the counters hold

__device__ __forceinline__ unsigned xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 15u;
}

// rounds barriers among `members` workgroups on counter c (monotonic: round r waits for r * members arrivals).
// FENCE: lane 0 runs an agent-scope release before arriving and an agent-scope acquire after the wait.
template <bool FENCE>
__device__ __forceinline__ bool barrier_rounds(unsigned *c, unsigned members, int rounds, unsigned *failed) {
  for (int r = 1; r <= rounds; ++r) {
    __syncthreads();
    if (threadIdx.x == 0) {
      if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)r * members;
      int spin = 0;
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spin < kMaxSpin) __builtin_amdgcn_s_sleep(1);
      if (spin >= kMaxSpin) { *failed = 1; }
      if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  return true;
}

// ctr[0]: grid counter; ctr[16 + x]: members registered on XCD x; ctr[32 + x]: barrier counter of XCD x; ctr[64]: failure flag
template <bool FENCE, bool XCD>
__global__ void barrier_kernel(unsigned *ctr, int rounds) {
  __shared__ unsigned members;
  unsigned *c = ctr;
  if (XCD) {
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) {
      // register on the XCD, wait until every workgroup of the grid has registered, then read the XCD's head count
      __hip_atomic_fetch_add(ctr + 16 + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spin = 0;
      while (__hip_atomic_load(ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spin < kMaxSpin) __builtin_amdgcn_s_sleep(1);
      if (spin >= kMaxSpin) ctr[64] = 1;
      members = __hip_atomic_load(ctr + 16 + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    c = ctr + 32 + x;
  } else {
    if (threadIdx.x == 0) members = gridDim.x;
    __syncthreads();
  }
  barrier_rounds<FENCE>(c, members, rounds, ctr + 64);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <bool FENCE, bool XCD>
static int run_barrier(const char *name, unsigned *ctr, int grid, hipStream_t st) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> per;
  for (int rep = 0; rep < 7; ++rep) {
    for (int rounds : {20, 220}) {
      CK(hipMemsetAsync(ctr, 0, 128 * sizeof(unsigned), st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL((barrier_kernel<FENCE, XCD>), dim3(grid), dim3(256), 0, st, ctr, rounds);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      per.push_back(ms * 1e3);
    }
  }
  unsigned h[128]; CK(hipMemcpy(h, ctr, sizeof h, hipMemcpyDeviceToHost));
  // per barrier = (time of 220 rounds - time of 20 rounds) / 200, medians over the repetitions
  std::vector<double> a, b;
  for (size_t i = 0; i < per.size(); i += 2) { a.push_back(per[i]); b.push_back(per[i + 1]); }
  printf("%-58s %6.2f us per barrier (%d workgroups%s%s)\n", name, (median(b) - median(a)) / 200.0, grid,
         h[64] ? ", A POLL LOOP RAN OUT" : "", XCD ? "; members per XCD:" : "");
  if (XCD) { printf("   "); for (int x = 0; x < 8; ++x) printf(" %u", h[16 + x]); printf("\n"); }
  return 0;
}

int main() {
  CK(hipSetDevice(0));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned *buf; CK(hipMalloc(&buf, 1 << 20)); CK(hipMemset(buf, 0, 1 << 20));
  const int n = 256 * 256;
  // ---- 1. eager triples
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, st, buf, n);
  CK(hipStreamSynchronize(st));
  {
    const int R = 2000;
    const double t0 = now_us();
    for (int i = 0; i < 3 * R; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, st, buf, n);
    CK(hipStreamSynchronize(st));
    printf("%-58s %6.2f us per triple (%d triples back to back)\n", "three dependent trivial kernels, eager, back to back", (now_us() - t0) / R, R);
    std::vector<double> lat;
    for (int r = 0; r < 200; ++r) {
      const double t1 = now_us();
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, st, buf, n);
      CK(hipStreamSynchronize(st));
      lat.push_back(now_us() - t1);
    }
    printf("%-58s %6.2f us (first launch call to synchronize return, median of 200)\n", "one eager triple on an idle stream", median(lat));
  }
  // ---- 2. the same triple as a graph
  {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, st, buf, n);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const int R = 2000;
    const double t0 = now_us();
    for (int i = 0; i < R; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    printf("%-58s %6.2f us per replay (%d replays back to back)\n", "the triple as a HIP graph, back to back", (now_us() - t0) / R, R);
    std::vector<double> lat;
    for (int r = 0; r < 200; ++r) {
      const double t1 = now_us();
      CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      lat.push_back(now_us() - t1);
    }
    printf("%-58s %6.2f us (launch call to synchronize return, median of 200)\n", "one graph replay on an idle stream", median(lat));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  // ---- 3 / 4. barriers inside one resident launch (256 and 512 workgroups of 256 threads: at most two per CU, all resident)
  unsigned *ctr = buf + (1 << 16);
  for (int grid : {256, 512}) {
    if (run_barrier<false, false>("grid-wide barrier, bare counter", ctr, grid, st)) return 1;
    if (run_barrier<true, false>("grid-wide barrier, release + acquire fences by lane 0", ctr, grid, st)) return 1;
    if (run_barrier<false, true>("XCD-local barrier, bare counter", ctr, grid, st)) return 1;
    if (run_barrier<true, true>("XCD-local barrier, release + acquire fences by lane 0", ctr, grid, st)) return 1;
  }
  CK(hipFree(buf));
  return 0;
}
