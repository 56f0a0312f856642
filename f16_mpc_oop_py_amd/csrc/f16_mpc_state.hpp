// f16_mpc_state.hpp -- the STATE-DEPENDENT vectors of the condensed MPC QP (utils.py:21-167), computed by one wavefront on
// LDS-resident operands: the prediction pred_i = A^(i+1) x (MM x of utils.py:92, never formed as a matrix) and the gradient
// q = -2 CC' QQ (x_ref - MM x) (utils.py:112).  ONE definition for the two callers -- the build kernel k_mpc<true>
// (f16_control.hip: every calc_MPC_action call) and the closed-loop rollout kernel (f16_mpc_wave.hip: every step of every
// aircraft) -- so that the fused loop reproduces the host loop's QPs bit for bit.
#pragma once
#include <hip/hip_runtime.h>

#include "f16_smallmat.hpp"

namespace f16 {

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// out[3j+c] = sum_{i>=j} sum_{r in rows} G_{i-j}[r][c] * v[i*NR + rr]   (CC' v restricted to `rows`)
template <int NR>
__device__ __forceinline__ void conv_adjoint(double *out, const double *G, const double *v, int N, const int *rows) {
  for (int e = lane_id(); e < 3 * N; e += F16_WAVE) {
    const int j = e / 3, c = e - 3 * j;
    double s = 0.0;
#pragma unroll 2
    for (int i = j; i < N; ++i) {              // per-step dot products are independent chains; only the final add is serial
      const double *g = G + (i - j) * 27 + c;
      const double *vi = v + i * NR;
      double t = 0.0;
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) t += g[(NR == 9 ? rr : rows[rr]) * 3] * vi[rr];
      s += t;
    }
    out[e] = s;
  }
}

// A[81], Q[81], Qb[81] (terminal weight), G[27 N] (G_k = A^k B), x9[9], xref[9] in LDS -> pred[9 N], qv[3 N]; wbuf: 9 N doubles of
// scratch (the Q-weighted error).  Ends behind a barrier: every lane may read pred / qv.
__device__ __forceinline__ void mpc_state_vectors(const double *A, const double *Q, const double *Qb, const double *G, const double *x9,
                                                  const double *xref, double *pred, double *wbuf, double *qv, int N) {
  const int l = lane_id(), n = 3 * N;
  {   // pred_i = A pred_(i-1): lane r < 9 carries component r, the operands come by v_readlane
    const int r = l < 9 ? l : 0;
    double ar[9], pv = x9[r];
#pragma unroll
    for (int p = 0; p < 9; ++p) ar[p] = A[r * 9 + p];
    for (int i = 0; i < N; ++i) {
      double sacc = 0.0;
#pragma unroll
      for (int p = 0; p < 9; ++p) sacc += ar[p] * readlane_f64(pv, p);
      pv = sacc;
      if (l < 9) pred[i * 9 + l] = pv;
    }
    __syncthreads();
  }
  // q = -2 CC' QQ (x_ref - MM x)   (utils.py:112)
  for (int e = l; e < 9 * N; e += F16_WAVE) {
    const int i = e / 9, r = e - 9 * i;
    const double *Qi = (i == N - 1) ? Qb : Q;
    double s = 0.0;
    for (int p = 0; p < 9; ++p) s += Qi[r * 9 + p] * (xref[p] - pred[i * 9 + p]);
    wbuf[e] = s;
  }
  __syncthreads();
  conv_adjoint<9>(qv, G, wbuf, N, nullptr);
  __syncthreads();
  for (int e = l; e < n; e += F16_WAVE) qv[e] = -2.0 * qv[e];
  __threadfence_block();
  __syncthreads();
}

}  // namespace f16
