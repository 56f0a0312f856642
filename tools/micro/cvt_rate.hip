// Microbenchmark (diagnostic): issue cost of v_cvt_f64_i32 against v_add_f64 / v_mov_b32 + v_add_f64 (the exact int -> double
// conversion by the 2^52 bias: hi word 0x43300000, low word k ^ 0x80000000, minus (2^52 + 2^31)) on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o cvt_rate cvt_rate.hip && ./cvt_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OP>
__global__ void k(double *out, int iters) {
  int v[16];
  double a[16];
  for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x * 7 + i; a[i] = 0.0; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (OP == 0) { double d; asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d) : "v"(v[i])); a[i] = d; }
        if (OP == 1) { double d; asm volatile("v_add_f64 %0, %1, %1" : "=v"(d) : "v"(a[i])); a[i] = d; }
        if (OP == 2) { int h; asm volatile("v_mov_b32 %0, 0x43300000" : "=v"(h)); double d = __hiloint2double(h, v[i]);
                       double e; asm volatile("v_add_f64 %0, %1, %2" : "=v"(e) : "v"(d), "v"(-4503601774854144.0)); a[i] = e; }
        if (OP == 3) { double d; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(__int_as_float(v[i]))); a[i] = d; }
      }
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, int block, int grid, double per_iter) {
  double *d; hipMalloc(&d, sizeof(double) * block * grid);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(block), 0, 0, d, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(block), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = (double)block / 64 * grid / (256.0 * 4);
  const double ns = ms * 1e6 / (32.0 * iters * waves_per_simd);
  printf("%-28s waves/SIMD %.0f: %.3f ms, %.2f ns per conversion per SIMD (= %.2f cycles at 2.4 GHz; %g instructions each)\n", name, waves_per_simd, ms,
         ns, ns * 2.4, per_iter);
  hipFree(d);
}
int main() {
  for (int wps = 1; wps <= 2; wps *= 2) {
    run<0>("v_cvt_f64_i32", 256, 256 * wps, 1); run<1>("v_add_f64", 256, 256 * wps, 1); run<2>("v_mov_b32 + v_add_f64 (bias)", 256, 256 * wps, 2);
    run<3>("v_cvt_f64_f32", 256, 256 * wps, 1);
  }
  return 0;
}
