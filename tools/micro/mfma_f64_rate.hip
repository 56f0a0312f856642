// Microbenchmark (diagnostic): what v_mfma_f64_16x16x4_f64 costs one wavefront on gfx950 -- back-to-back independent products (eight
// accumulator tiles in a[0:63], round robin), a dependent chain on one tile, and products with N independent fp64 FMAs between them
// (does vector work run in the shadow of the matrix pipe?).  Explicit registers in one asm block per loop body: the builtin (and an
// asm with an "a" constraint) keeps the tiles in VGPRs across the loop and copies every tile in and out around its product.
// One wavefront per SIMD (LDS pad: four 64-lane workgroups per CU) and four per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

#define MF(t) "v_mfma_f64_16x16x4_f64 a[" #t "], v[64:65], v[66:67], a[" #t "]\n"
#define F4 "v_fma_f64 v[32:33], v[64:65], v[66:67], v[32:33]\n v_fma_f64 v[34:35], v[64:65], v[66:67], v[34:35]\n" \
           "v_fma_f64 v[36:37], v[64:65], v[66:67], v[36:37]\n v_fma_f64 v[38:39], v[64:65], v[66:67], v[38:39]\n"
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v64","v65","v66","v67", \
  "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23", \
  "a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47", \
  "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"

template <int MODE, int LDS_KB>
__global__ __launch_bounds__(64) void k(unsigned long long *out, int iters) {
  __shared__ double lds[LDS_KB * 128];
  for (int i = threadIdx.x; i < LDS_KB * 128; i += 64) lds[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  const double a = lds[threadIdx.x] * 1e-3, b = lds[threadIdx.x + 64];
  asm volatile("v_mov_b32 v64, %0\n v_mov_b32 v65, %1\n v_mov_b32 v66, %2\n v_mov_b32 v67, %3\n"
               :: "v"(__double2loint(a)), "v"(__double2hiint(a)), "v"(__double2loint(b)), "v"(__double2hiint(b)) : CLOB);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) asm volatile(MF(0:7) MF(8:15) MF(16:23) MF(24:31) MF(32:39) MF(40:47) MF(48:55) MF(56:63) ::: CLOB);
    if (MODE == 1) asm volatile(MF(0:7) MF(0:7) MF(0:7) MF(0:7) MF(0:7) MF(0:7) MF(0:7) MF(0:7) ::: CLOB);
    if (MODE == 2) asm volatile(MF(0:7) F4 MF(8:15) F4 MF(16:23) F4 MF(24:31) F4 MF(32:39) F4 MF(40:47) F4 MF(48:55) F4 MF(56:63) F4 ::: CLOB);
    if (MODE == 3) asm volatile(MF(0:7) F4 F4 MF(8:15) F4 F4 MF(16:23) F4 F4 MF(24:31) F4 F4 MF(32:39) F4 F4 MF(40:47) F4 F4 MF(48:55) F4 F4 MF(56:63) F4 F4 ::: CLOB);
    if (MODE == 4) asm volatile(MF(0:7) F4 F4 F4 MF(8:15) F4 F4 F4 MF(16:23) F4 F4 F4 MF(24:31) F4 F4 F4 MF(32:39) F4 F4 F4 MF(40:47) F4 F4 F4 MF(48:55) F4 F4 F4 MF(56:63) F4 F4 F4 ::: CLOB);
    if (MODE == 5) asm volatile(MF(0:7) MF(8:15) MF(0:7) MF(8:15) MF(0:7) MF(8:15) MF(0:7) MF(8:15) ::: CLOB);      // two tiles alternating
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE, int LDS_KB>
static void run(const char *name, int blocks) {
  unsigned long long *out;
  (void)hipMalloc(&out, blocks * 8);
  const int iters = 2000;
  k<MODE, LDS_KB><<<blocks, 64>>>(out, iters);
  k<MODE, LDS_KB><<<blocks, 64>>>(out, iters);
  (void)hipDeviceSynchronize();
  unsigned long long *h = (unsigned long long *)malloc(blocks * 8);
  (void)hipMemcpy(h, out, blocks * 8, hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < blocks; ++i) m += (double)h[i];
  printf("%-64s %7.1f cycles per product\n", name, m / blocks / (iters * 8.0));
  free(h); (void)hipFree(out);
}
int main() {
  run<0, 36>("one wave per SIMD: eight independent tiles", 1024);
  run<1, 36>("one wave per SIMD: dependent chain (one tile)", 1024);
  run<5, 36>("one wave per SIMD: two tiles alternating", 1024);
  run<2, 36>("one wave per SIMD: product + 4 fp64 FMAs", 1024);
  run<3, 36>("one wave per SIMD: product + 8 fp64 FMAs", 1024);
  run<4, 36>("one wave per SIMD: product + 12 fp64 FMAs", 1024);
  run<0, 8>("four waves per SIMD: eight independent tiles (per wave)", 4096);
  run<0, 36>("one wave on the whole chip: eight independent tiles", 1);
  return 0;
}
