// Microbenchmark (diagnostic): the rate at which ONE CU streams a private buffer (the KKT-inverse stream of k_mpc_big) by load
// width: 8 bytes per lane (global_load_dwordx2, what half_symv_t issues) against 16 bytes per lane (dwordx4), 512 lanes per
// workgroup, one workgroup per CU, every workgroup its own `bytes`-byte buffer read `reps` times.
// hipcc --offload-arch=gfx950 -O3 -o cu_stream cu_stream.hip && ./cu_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int W>   // W = doubles per lane and load
__global__ __launch_bounds__(512) void k(const double *buf, size_t doubles_per_wg, int reps, double *out) {
  typedef double vec_t __attribute__((ext_vector_type(W)));
  const vec_t *p = reinterpret_cast<const vec_t *>(buf + (size_t)blockIdx.x * doubles_per_wg);
  const size_t nvec = doubles_per_wg / W;
  double acc = 0.0;
  for (int r = 0; r < reps; ++r) {
    for (size_t i = threadIdx.x; i + 7 * 512 < nvec; i += 8 * 512) {
      vec_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * 512) ;
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int c = 0; c < W; ++c) acc += v[u][c];
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int W>
void run(int wgs, size_t bytes, int reps) {
  double *buf, *out;
  const size_t d = bytes / 8;
  hipMalloc(&buf, d * 8 * wgs); hipMalloc(&out, 8 * 512 * wgs);
  hipMemset(buf, 0, d * 8 * wgs);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<W>, dim3(wgs), dim3(512), 0, 0, buf, d, 2, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<W>, dim3(wgs), dim3(512), 0, 0, buf, d, reps, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%2d bytes per lane, %3d workgroups x %.2f MB: %.1f GB/s per CU (%.2f TB/s in all)\n", 8 * W, wgs, bytes / 1e6, bytes * (double)reps / (ms * 1e-3) / 1e9,
         bytes * (double)reps * wgs / (ms * 1e-3) / 1e12);
  hipFree(buf); hipFree(out);
}
int main() {
  for (int wgs : {8, 64, 256}) {            // 8: one per XCD, L2-resident; 64: Infinity-Cache resident; 256: every CU
    run<1>(wgs, 917504, 400); run<2>(wgs, 917504, 400);
  }
  run<1>(64, 425984, 800); run<2>(64, 425984, 800);
  return 0;
}
