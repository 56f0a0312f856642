// f16_debug.hip -- test entry points that expose pieces of the device plant on their own.
//
//   k_dbg_table   ONE of the reference's 43 table functions (C/hifi_F16_AeroData.c:109-1861, each a lazy file read +
//                 interpn(), C/mexndinterp.c:97-265) evaluated by the SAME device helpers the dynamics kernels use
//                 (br_load / br_cell / br_axis / lerp / ld4 / bil4 on the node-major LDS image, f16_plant.hpp), one lane
//                 per query point.  The kernels never evaluate a single table -- lookups are fused per table group -- so
//                 without this entry the bracket / interpolation code is only checked through whole-Nlplant sums.
#include <hip/hip_runtime.h>

#include "../../include/f16_hip.h"
#include "f16_ctx.h"
#include "f16_plant.hpp"

namespace f16 {

// table id (enum f16_table_id of f16_tables_data.inc) -> group of the node-major image
struct TabLoc { int off, stride, k, na, dims, dgrid; };     // dims 1: alpha; 2: alpha x beta; 3: + elevator; 0: eta(el)
__host__ __device__ inline TabLoc table_location(int tid) {
  if (tid < 3) return {OFF_G3A, S_G3A, tid, N_A1, 3, 1};
  if (tid < 5) return {OFF_G3B, S_G3B, tid - 3, N_A1, 3, 2};
  if (tid < 12) return {OFF_G2A, S_G2A, tid - 5, N_A1, 2, 0};
  if (tid < 21) return {OFF_G2B, S_G2B, tid - 12, N_A2, 2, 0};
  if (tid < 33) return {OFF_G1A, S_G1A, tid - 21, N_A1, 1, 0};
  if (tid < 42) return {OFF_G1B, S_G1B, tid - 33, N_A2, 1, 0};
  return {OFF_ETA, 1, 0, 0, 0, 1};
}

__global__ __launch_bounds__(64) void k_dbg_table(const double *__restrict__ tab, int tid, const double *alpha,
                                                  const double *beta, const double *el, double *out, int32_t *status, int n) {
  __shared__ __attribute__((aligned(16))) double T[TABLE_IMAGE_DOUBLES];
  {
    const double2 *src = reinterpret_cast<const double2 *>(tab);
    double2 *dst = reinterpret_cast<double2 *>(T);
    for (int i = threadIdx.x; i < TABLE_IMAGE_DOUBLES / 2; i += 64) dst[i] = src[i];
    __syncthreads();
  }
  const TabLoc L = table_location(tid);
  for (int p = blockIdx.x * 64 + threadIdx.x; p < n; p += gridDim.x * 64) {
    const double a = alpha[p], b = beta[p], e = el[p];
    int st = 0;
    // brackets exactly as aero_totals_phased takes them: alpha on ALPHA1 (the ALPHA2 grid is its prefix; beyond its last
    // node the lef tables are clamped: last cell, lambda = 1), beta on BETA1, elevator on DH1 or DH2
    const double *cT = T;
    const BrRaw ra = br_load(cT + OFF_BP_A1, N_A1, alpha_guess(a));
    const BrRaw rb = br_load(cT + OFF_BP_B1, N_B1, beta_guess(b));
    const BrRaw rd = L.dgrid == 2 ? br_load(cT + OFF_BP_D2, N_D2, (int)(e >= 0.0))
                                  : br_load(cT + OFF_BP_D1, N_D1, (e >= -10.0) + (e >= 0.0) + (e >= 10.0));
    bool offa, offb, offd;
    const BrCell ca = br_cell(ra, N_A1, a, offa), cb = br_cell(rb, N_B1, b, offb);
    const BrCell cd = br_cell(rd, L.dgrid == 2 ? N_D2 : N_D1, e, offd);
    Axis a1 = br_axis(ca);
    const Axis bx = br_axis(cb), dx = br_axis(cd);
    int ja = ca.j;
    if (L.na == N_A2) {
      const bool hi_a = ca.j > N_A2 - 2;
      if (hi_a) { ja = N_A2 - 2; a1.j = ja; a1.l = 1.0; a1.m = 0.0; if (a > cT[OFF_BP_A1 + N_A2 - 1]) st |= ST_ALPHA2; }
    }
    const W4 W1 = bil_weights(a1, bx);
    double v;
    if (L.dims == 0) {
      v = lerp(cT[OFF_ETA + cd.j], cT[OFF_ETA + cd.j + 1], dx);
      if (offd) st |= ST_EL;
    } else if (L.dims == 1) {
      const double *g = cT + L.off + ja * L.stride + L.k;
      v = lerp(g[0], g[L.stride], a1);
      if (offa) st |= ST_ALPHA1;
    } else {
      const int sa = L.stride, sb = L.stride * L.na;
      const double *p2 = cT + L.off + (cb.j * L.na + ja) * sa + L.k;
      if (L.dims == 2) {
        v = bil4w(ld4(p2, sa, sb), a1, bx, W1);
      } else {
        const int sd = L.stride * L.na * N_B1;
        v = lerp(bil4w(ld4(p2 + cd.j * sd, sa, sb), a1, bx, W1), bil4w(ld4(p2 + (cd.j + 1) * sd, sa, sb), a1, bx, W1), dx);
        if (offd) st |= ST_EL;
      }
      if (offa) st |= ST_ALPHA1;
      if (offb) st |= ST_BETA;
    }
    out[p] = v;
    if (status) status[p] = st;
  }
}

}  // namespace f16

using namespace f16;

extern "C" int f16_debug_table_lookup(f16_ctx *ctx, int tid, const double *h_alpha, const double *h_beta, const double *h_el,
                                      int n, double *h_out, int32_t *h_status) {
  if (!ctx || tid < 0 || tid > 42 || !h_alpha || !h_beta || !h_el || !h_out || n < 0)
    return set_error(F16_EINVAL, "bad argument to f16_debug_table_lookup");
  if (n == 0) return F16_OK;
  double *d = nullptr;
  int32_t *ds = nullptr;
  int rc;
  if ((rc = hip_check(hipMalloc(&d, 4 * (size_t)n * sizeof(double)), "hipMalloc dbg table"))) return rc;
  if ((rc = hip_check(hipMalloc(&ds, (size_t)n * sizeof(int32_t)), "hipMalloc dbg table"))) { (void)hipFree(d); return rc; }
  const size_t nb = (size_t)n * sizeof(double);
  if (!(rc = hip_check(hipMemcpy(d, h_alpha, nb, hipMemcpyHostToDevice), "copy alpha")) &&
      !(rc = hip_check(hipMemcpy(d + n, h_beta, nb, hipMemcpyHostToDevice), "copy beta")) &&
      !(rc = hip_check(hipMemcpy(d + 2 * (size_t)n, h_el, nb, hipMemcpyHostToDevice), "copy el"))) {
    const int blocks = (n + 63) / 64;
    hipLaunchKernelGGL(k_dbg_table, dim3(blocks < 256 ? blocks : 256), dim3(64), 0, nullptr, ctx->d_tab, tid, d, d + n,
                       d + 2 * (size_t)n, d + 3 * (size_t)n, ds, n);
    rc = hip_check(hipGetLastError(), "f16_debug_table_lookup launch");
    if (!rc) rc = hip_check(hipMemcpy(h_out, d + 3 * (size_t)n, nb, hipMemcpyDeviceToHost), "copy out");
    if (!rc && h_status) rc = hip_check(hipMemcpy(h_status, ds, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost), "copy status");
  }
  (void)hipFree(d);
  (void)hipFree(ds);
  return rc;
}
