// Microbenchmark (diagnostic): the state-recursion form of a Toeplitz stage inside the one-wavefront-per-aircraft mapping.
//   x_{k+1} = A x_k + B u_k, z_k = x_{k+1} (9 states, 3 inputs, N = 30 steps): 108 FMAs per step, 3,240 per stage, against the
//   18 k of the convolution form -- but a serial chain of 30 steps with ONE wavefront per SIMD and nothing to hide it behind.
// Lanes 0..8 hold the state rows; the new state reaches every row through v_readlane (SGPR operands of the next step's FMAs);
// the 12 products of a row run as three independent partial sums.  One wavefront per SIMD, as in k_mpc_wave.
// hipcc --offload-arch=gfx950 -O3 -o recursion_chain recursion_chain.hip && ./recursion_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ double bcast(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__global__ __launch_bounds__(64, 1) void k(const double *AB, const double *u, double *out, unsigned long long *cyc, int reps) {
  __shared__ double us[96], zs[9 * 32];
  const int l = threadIdx.x, r = l < 9 ? l : 8;
  double a[12];
  for (int c = 0; c < 12; ++c) a[c] = AB[r * 12 + c];
  for (int i = l; i < 96; i += 64) us[i] = u[i];
  __syncthreads();
  double acc = 0.0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
    double x = 1e-3 * r + acc * 1e-300;
    for (int k2 = 0; k2 < 30; ++k2) {
      double v[12];
#pragma unroll
      for (int c = 0; c < 9; ++c) v[c] = bcast(x, c);
#pragma unroll
      for (int c = 0; c < 3; ++c) v[9 + c] = us[3 * k2 + c];
      double p0 = 0.0, p1 = 0.0, p2 = 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) { p0 = fma(a[c], v[c], p0); p1 = fma(a[4 + c], v[4 + c], p1); p2 = fma(a[8 + c], v[8 + c], p2); }
      x = (p0 + p1) + p2;
      if (l < 9) zs[9 * k2 + l] = x;                       // the stage's output row (what the projection then reads)
    }
    acc += x;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * 64 + l] = acc + zs[l];
}
int main() {
  double hAB[108], hu[96];
  for (int i = 0; i < 108; ++i) hAB[i] = (i % 13 == 0 ? 0.9 : 0.01) * ((i % 3) - 1);
  for (int i = 0; i < 96; ++i) hu[i] = 0.1 * (i % 7);
  double *AB, *u, *out; unsigned long long *cyc, h;
  hipMalloc(&AB, sizeof hAB); hipMalloc(&u, sizeof hu); hipMalloc(&out, 1024 * 64 * 8); hipMalloc(&cyc, 8);
  hipMemcpy(AB, hAB, sizeof hAB, hipMemcpyHostToDevice); hipMemcpy(u, hu, sizeof hu, hipMemcpyHostToDevice);
  const int reps = 2000;
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, AB, u, out, cyc, reps);
  hipDeviceSynchronize();
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("state-recursion stage, one wavefront per SIMD (1024 wavefronts): %.0f cycles per 30-step stage (s_memtime), %.1f per step\n",
         (double)h / reps, (double)h / reps / 30);
  printf("for comparison: a convolution-form stage of k_mpc_wave (288 FMAs per lane, 64 lanes) measures 2.5-2.8 k cycles incl. the right-hand side\n");
  return 0;
}
