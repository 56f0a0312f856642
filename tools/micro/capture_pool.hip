// tools/micro/capture_pool.hip -- does a HIP graph that holds a stream-ordered allocation (hipMallocFromPoolAsync ... hipFreeAsync
// captured between two kernels) replay correctly on this runtime?  Plain HIP, none of this library's kernels.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/capture_pool tools/micro/capture_pool.hip && /tmp/capture_pool
// Graph: alloc p (N doubles) -> fill(p, seed read from device memory) -> consume(p -> out = sum of what it finds, mismatches) -> free p.
// Between replays the same pool serves EAGER allocations of the same size that another kernel fills with a different
// pattern (same stream, and a second stream), i.e. what a caller of f16_mpc_batch does around a captured call.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 2; } } while (0)

__global__ void fill(double *p, size_t n, const double *seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = *seed + (double)(i & 1023);
}
__global__ void consume(const double *p, size_t n, const double *seed, unsigned long long *bad) {
  unsigned long long b = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b += (p[i] != *seed + (double)(i & 1023));
  if (b) atomicAdd(bad, b);
}
__global__ void scribble(double *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = -7.0;
}

int main() {
  const size_t n = (size_t)40 << 20;                      // 320 MB, the size class of a B = 4096 QP workspace
  hipMemPool_t pool;
  hipMemPoolProps props = {};
  props.allocType = hipMemAllocationTypePinned; props.handleTypes = hipMemHandleTypeNone;
  props.location.type = hipMemLocationTypeDevice; props.location.id = 0;
  CK(hipMemPoolCreate(&pool, &props));
  uint64_t keep = 4ull << 30;
  CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
  hipStream_t s, s2;
  CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
  double *seed; unsigned long long *bad;
  CK(hipMalloc(&seed, 8)); CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
  double h = 1.0; CK(hipMemcpy(seed, &h, 8, hipMemcpyHostToDevice));
  // eager use of the pool first (as the library does before anyone captures)
  for (int k = 0; k < 3; ++k) { double *q; CK(hipMallocFromPoolAsync((void **)&q, n * 8, pool, s)); scribble<<<1024, 256, 0, s>>>(q, n); CK(hipFreeAsync(q, s)); }
  CK(hipStreamSynchronize(s));
  hipGraph_t g; hipGraphExec_t ge;
  double *p = nullptr;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  CK(hipMallocFromPoolAsync((void **)&p, n * 8, pool, s));
  fill<<<1024, 256, 0, s>>>(p, n, seed);
  consume<<<1024, 256, 0, s>>>(p, n, seed, bad);
  CK(hipFreeAsync(p, s));
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  printf("captured: workspace pointer baked into the kernel nodes = %p\n", (void *)p);
  unsigned long long total_bad = 0;
  for (int rep = 0; rep < 12; ++rep) {
    h = 100.0 * (rep + 1); CK(hipMemcpyAsync(seed, &h, 8, hipMemcpyHostToDevice, s));
    CK(hipGraphLaunch(ge, s));
    // eager traffic on the same pool while / after the replay: same stream (ordered) and a second stream (concurrent)
    double *q, *q2;
    CK(hipMallocFromPoolAsync((void **)&q, n * 8, pool, s)); scribble<<<1024, 256, 0, s>>>(q, n); CK(hipFreeAsync(q, s));
    CK(hipMallocFromPoolAsync((void **)&q2, n * 8, pool, s2)); scribble<<<1024, 256, 0, s2>>>(q2, n); CK(hipFreeAsync(q2, s2));
    CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
    unsigned long long b; CK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost));
    printf("replay %2d: mismatching elements seen by the graph's consumer so far %llu (eager blocks at %p, %p)\n", rep, b, (void *)q, (void *)q2);
    total_bad = b;
  }
  printf("RESULT: %s\n", total_bad ? "graph replay of a stream-ordered allocation is NOT reliable here" : "plain-HIP case replays correctly");
  return total_bad ? 1 : 0;
}
