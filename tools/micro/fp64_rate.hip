// Microbenchmark (diagnostic): issue rate of v_fma_f64 / v_mul_f64 / v_add_f64 / v_fma_f32 on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o fp64_rate fp64_rate.hip && ./fp64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OP, int NC>
__global__ void k(double *out, int iters) {
  double a[NC];
  float f[NC];
  for (int i = 0; i < NC; ++i) { a[i] = threadIdx.x * 1e-3 + i; f[i] = (float)a[i]; }
  const double m = 1.0000001, c = 1e-9;
  const float mf = 1.0000001f, cf = 1e-9f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 32 / NC; ++r)
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        if (OP == 0) a[i] = __builtin_fma(a[i], m, c);
        if (OP == 1) a[i] = a[i] * m;
        if (OP == 2) a[i] = a[i] + c;
        if (OP == 3) f[i] = __builtin_fmaf(f[i], mf, cf);
      }
  }
  double s = 0;
  for (int i = 0; i < NC; ++i) s += a[i] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP, int NC>
void run(const char *name, int block, int grid) {
  double *d; hipMalloc(&d, sizeof(double) * block * grid);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP, NC>), dim3(grid), dim3(block), 0, 0, d, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP, NC>), dim3(grid), dim3(block), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double inst_per_wave = 32.0 * iters;                 // per wave
  const double waves_per_simd = (double)block / 64 * grid / (256.0 * 4);
  const double ns_per_inst_per_simd = ms * 1e6 / (inst_per_wave * waves_per_simd);
  printf("%-10s chains %2d block %4d grid %5d: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, NC, block, grid, ms,
         ns_per_inst_per_simd, ns_per_inst_per_simd * 2.4);
  hipFree(d);
}
int main() {
  for (int wps = 1; wps <= 4; wps *= 2) {       // waves per SIMD
    run<0, 4>("fma_f64", 256, 256 * wps); run<0, 8>("fma_f64", 256, 256 * wps); run<0, 16>("fma_f64", 256, 256 * wps); run<0, 32>("fma_f64", 256, 256 * wps);
    run<2, 16>("add_f64", 256, 256 * wps); run<3, 16>("fma_f32", 256, 256 * wps);
  }
  return 0;
}
