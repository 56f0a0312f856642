// Microbenchmark (diagnostic): what ONE resident wavefront per SIMD pays per fp64 FMA on gfx950, and what changes it:
// operand sources (three VGPR pairs / one SGPR pair / an inline constant), register banks of the operands, dependent chains, and
// other instruction classes issued between the FMAs (v_accvgpr_read, v_mov, ds_read_b128, v_mov_dpp).  Cycles come from s_memtime
// (shader clock) and from s_memrealtime (100 MHz), so the clock the part really runs at under this load is printed too.
//   hipcc --offload-arch=gfx950 -O3 -o issue_mix issue_mix.hip && ./issue_mix
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// 16 independent accumulators v[32..63]; multiplicands v[64:65] (banks 0,1), v[66:67] (banks 2,3), v[68:69] (banks 0,1)
#define FMA_ALL(A, B) \
  "v_fma_f64 v[32:33], " A ", " B ", v[32:33]\n v_fma_f64 v[34:35], " A ", " B ", v[34:35]\n" \
  "v_fma_f64 v[36:37], " A ", " B ", v[36:37]\n v_fma_f64 v[38:39], " A ", " B ", v[38:39]\n" \
  "v_fma_f64 v[40:41], " A ", " B ", v[40:41]\n v_fma_f64 v[42:43], " A ", " B ", v[42:43]\n" \
  "v_fma_f64 v[44:45], " A ", " B ", v[44:45]\n v_fma_f64 v[46:47], " A ", " B ", v[46:47]\n" \
  "v_fma_f64 v[48:49], " A ", " B ", v[48:49]\n v_fma_f64 v[50:51], " A ", " B ", v[50:51]\n" \
  "v_fma_f64 v[52:53], " A ", " B ", v[52:53]\n v_fma_f64 v[54:55], " A ", " B ", v[54:55]\n" \
  "v_fma_f64 v[56:57], " A ", " B ", v[56:57]\n v_fma_f64 v[58:59], " A ", " B ", v[58:59]\n" \
  "v_fma_f64 v[60:61], " A ", " B ", v[60:61]\n v_fma_f64 v[62:63], " A ", " B ", v[62:63]\n"
// accumulators in bank pair {0,1} only: v[32:33], v[36:37], ... (8 of them, twice)
#define FMA_B01(A, B) \
  "v_fma_f64 v[32:33], " A ", " B ", v[32:33]\n v_fma_f64 v[36:37], " A ", " B ", v[36:37]\n" \
  "v_fma_f64 v[40:41], " A ", " B ", v[40:41]\n v_fma_f64 v[44:45], " A ", " B ", v[44:45]\n" \
  "v_fma_f64 v[48:49], " A ", " B ", v[48:49]\n v_fma_f64 v[52:53], " A ", " B ", v[52:53]\n" \
  "v_fma_f64 v[56:57], " A ", " B ", v[56:57]\n v_fma_f64 v[60:61], " A ", " B ", v[60:61]\n" \
  "v_fma_f64 v[32:33], " A ", " B ", v[32:33]\n v_fma_f64 v[36:37], " A ", " B ", v[36:37]\n" \
  "v_fma_f64 v[40:41], " A ", " B ", v[40:41]\n v_fma_f64 v[44:45], " A ", " B ", v[44:45]\n" \
  "v_fma_f64 v[48:49], " A ", " B ", v[48:49]\n v_fma_f64 v[52:53], " A ", " B ", v[52:53]\n" \
  "v_fma_f64 v[56:57], " A ", " B ", v[56:57]\n v_fma_f64 v[60:61], " A ", " B ", v[60:61]\n"
#define FMA_DEP(A, B) REP16("v_fma_f64 v[32:33], " A ", " B ", v[32:33]\n")
#define FMAC_ALL \
  "v_fmac_f64_e32 v[32:33], v[64:65], v[66:67]\n v_fmac_f64_e32 v[34:35], v[64:65], v[66:67]\n" \
  "v_fmac_f64_e32 v[36:37], v[64:65], v[66:67]\n v_fmac_f64_e32 v[38:39], v[64:65], v[66:67]\n" \
  "v_fmac_f64_e32 v[40:41], v[64:65], v[66:67]\n v_fmac_f64_e32 v[42:43], v[64:65], v[66:67]\n" \
  "v_fmac_f64_e32 v[44:45], v[64:65], v[66:67]\n v_fmac_f64_e32 v[46:47], v[64:65], v[66:67]\n" \
  "v_fmac_f64_e32 v[48:49], v[64:65], v[66:67]\n v_fmac_f64_e32 v[50:51], v[64:65], v[66:67]\n" \
  "v_fmac_f64_e32 v[52:53], v[64:65], v[66:67]\n v_fmac_f64_e32 v[54:55], v[64:65], v[66:67]\n" \
  "v_fmac_f64_e32 v[56:57], v[64:65], v[66:67]\n v_fmac_f64_e32 v[58:59], v[64:65], v[66:67]\n" \
  "v_fmac_f64_e32 v[60:61], v[64:65], v[66:67]\n v_fmac_f64_e32 v[62:63], v[64:65], v[66:67]\n"
// one other instruction after every FMA (16 FMAs + 16 others)
#define MIX(OTHER) \
  "v_fma_f64 v[32:33], v[64:65], v[66:67], v[32:33]\n" OTHER "v_fma_f64 v[34:35], v[64:65], v[66:67], v[34:35]\n" OTHER \
  "v_fma_f64 v[36:37], v[64:65], v[66:67], v[36:37]\n" OTHER "v_fma_f64 v[38:39], v[64:65], v[66:67], v[38:39]\n" OTHER \
  "v_fma_f64 v[40:41], v[64:65], v[66:67], v[40:41]\n" OTHER "v_fma_f64 v[42:43], v[64:65], v[66:67], v[42:43]\n" OTHER \
  "v_fma_f64 v[44:45], v[64:65], v[66:67], v[44:45]\n" OTHER "v_fma_f64 v[46:47], v[64:65], v[66:67], v[46:47]\n" OTHER \
  "v_fma_f64 v[48:49], v[64:65], v[66:67], v[48:49]\n" OTHER "v_fma_f64 v[50:51], v[64:65], v[66:67], v[50:51]\n" OTHER \
  "v_fma_f64 v[52:53], v[64:65], v[66:67], v[52:53]\n" OTHER "v_fma_f64 v[54:55], v[64:65], v[66:67], v[54:55]\n" OTHER \
  "v_fma_f64 v[56:57], v[64:65], v[66:67], v[56:57]\n" OTHER "v_fma_f64 v[58:59], v[64:65], v[66:67], v[58:59]\n" OTHER \
  "v_fma_f64 v[60:61], v[64:65], v[66:67], v[60:61]\n" OTHER "v_fma_f64 v[62:63], v[64:65], v[66:67], v[62:63]\n" OTHER

#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51", \
  "v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75", \
  "a0","a1","a2","a3","s40","s41"

template <int V, int LDS_KB>
__global__ __launch_bounds__(64) void k(unsigned long long *out, int iters) {
  __shared__ double lds[LDS_KB * 128];
  for (int i = threadIdx.x; i < LDS_KB * 128; i += 64) lds[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  const unsigned ldsaddr = (unsigned)(threadIdx.x * 16);
  // operands: |a| < 1 so that nothing overflows
  asm volatile(
      "v_mov_b32 v64, 0\n v_mov_b32 v65, 0x3fe00000\n"          // 0.5
      "v_mov_b32 v66, 0\n v_mov_b32 v67, 0x3fe80000\n"          // 0.75
      "v_mov_b32 v68, 0\n v_mov_b32 v69, 0x3fe40000\n"          // 0.625
      "s_mov_b32 s40, 0\n s_mov_b32 s41, 0x3fe00000\n"
      "v_accvgpr_write_b32 a0, v64\n v_accvgpr_write_b32 a1, v65\n v_accvgpr_write_b32 a2, v66\n v_accvgpr_write_b32 a3, v67\n"
      "v_mov_b32 v32, 0\n v_mov_b32 v33, 0x3ff00000\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0x3ff00000\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0x3ff00000\n"
      "v_mov_b32 v38, 0\n v_mov_b32 v39, 0x3ff00000\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0x3ff00000\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0x3ff00000\n"
      "v_mov_b32 v44, 0\n v_mov_b32 v45, 0x3ff00000\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0x3ff00000\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0x3ff00000\n"
      "v_mov_b32 v50, 0\n v_mov_b32 v51, 0x3ff00000\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0x3ff00000\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0x3ff00000\n"
      "v_mov_b32 v56, 0\n v_mov_b32 v57, 0x3ff00000\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0x3ff00000\n v_mov_b32 v60, 0\n v_mov_b32 v61, 0x3ff00000\n"
      "v_mov_b32 v62, 0\n v_mov_b32 v63, 0x3ff00000\n v_mov_b32 v70, %0\n" ::"v"(ldsaddr) : CLOB);
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (V == 0) asm volatile(REP4(FMA_ALL("v[64:65]", "v[66:67]")) ::: CLOB);            // a {0,1}, b {2,3}, acc alternating
    if (V == 1) asm volatile(REP4(FMA_B01("v[64:65]", "v[68:69]")) ::: CLOB);            // everything in bank pair {0,1}
    if (V == 2) asm volatile(REP4(FMA_B01("v[66:67]", "v[66:67]")) ::: CLOB);            // a = b in {2,3}, acc in {0,1}
    if (V == 3) asm volatile(REP4(FMA_ALL("s[40:41]", "v[66:67]")) ::: CLOB);            // one SGPR operand
    if (V == 4) asm volatile(REP4(FMA_ALL("0.5", "v[66:67]")) ::: CLOB);                 // inline constant
    if (V == 5) asm volatile(REP4(FMA_DEP("v[64:65]", "v[66:67]")) ::: CLOB);            // one dependent chain
    if (V == 6) asm volatile(REP4(FMAC_ALL) ::: CLOB);                                   // VOP2 encoding
    if (V == 7) asm volatile(REP4(MIX("v_accvgpr_read_b32 v72, a0\n")) ::: CLOB);
    if (V == 8) asm volatile(REP4(MIX("v_mov_b32 v72, v73\n")) ::: CLOB);
    if (V == 9) asm volatile(REP4(MIX("ds_read_b128 v[72:75], v70\n")) "s_waitcnt lgkmcnt(0)\n" ::: CLOB);
    if (V == 10) asm volatile(REP4(MIX("v_mov_b32_dpp v72, v73 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")) ::: CLOB);
    if (V == 11) asm volatile(REP4(MIX("v_add_f64 v[72:73], v[64:65], v[66:67]\n")) ::: CLOB);
    if (V == 12) asm volatile(REP4(MIX("s_nop 0\n")) ::: CLOB);
    if (V == 13) asm volatile(REP4(MIX("v_cndmask_b32 v72, v73, v74, vcc\n")) ::: CLOB);
    if (V == 14) asm volatile(REP4(MIX("ds_bpermute_b32 v72, v70, v73\n")) "s_waitcnt lgkmcnt(0)\n" ::: CLOB);
  }
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s;
  asm volatile("v_add_f64 %0, v[32:33], v[62:63]" : "=v"(s)::CLOB);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
  if (s == 123.456) out[0] = 0;
}

template <int V, int LDS_KB>
void run(const char *name, int fmas_per_iter, int others_per_iter) {
  const int grid = 256 * (160 / LDS_KB), iters = 2000;
  unsigned long long *d, h[2];
  hipMalloc(&d, sizeof(unsigned long long) * 2 * grid);
  hipLaunchKernelGGL((k<V, LDS_KB>), dim3(grid), dim3(64), 0, 0, d, 10);
  hipLaunchKernelGGL((k<V, LDS_KB>), dim3(grid), dim3(64), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, d + 2 * (grid / 2), sizeof h, hipMemcpyDeviceToHost);
  const double cyc = (double)h[0] / iters, ns = (double)h[1] * 10.0 / iters;
  printf("%-44s %d waves/SIMD: %7.1f cycles per %d FMA + %d other = %.2f per FMA (%.2f per instruction); %.2f GHz\n", name, 160 / LDS_KB / 4, cyc,
         fmas_per_iter, others_per_iter, cyc / fmas_per_iter, cyc / (fmas_per_iter + others_per_iter), cyc / ns);
  hipFree(d);
}

#define BOTH(V, name, f, o) run<V, 40>(name, f, o); run<V, 20>(name, f, o); run<V, 10>(name, f, o);
int main() {
  BOTH(0, "fma  a{0,1} b{2,3} acc alternating", 64, 0)
  BOTH(1, "fma  all operands in bank pair {0,1}", 64, 0)
  BOTH(2, "fma  a = b {2,3}, acc {0,1}", 64, 0)
  BOTH(3, "fma  one SGPR operand", 64, 0)
  BOTH(4, "fma  one inline constant", 64, 0)
  BOTH(5, "fma  one dependent chain", 64, 0)
  BOTH(6, "fmac (VOP2)", 64, 0)
  BOTH(7, "fma + v_accvgpr_read 1:1", 64, 64)
  BOTH(8, "fma + v_mov_b32 1:1", 64, 64)
  BOTH(9, "fma + ds_read_b128 1:1", 64, 64)
  BOTH(10, "fma + v_mov_b32_dpp 1:1", 64, 64)
  BOTH(11, "fma + v_add_f64 1:1", 64, 64)
  BOTH(12, "fma + s_nop 1:1", 64, 64)
  BOTH(13, "fma + v_cndmask_b32 1:1", 64, 64)
  BOTH(14, "fma + ds_bpermute_b32 1:1", 64, 64)
  return 0;
}
