// f16_trim.hip -- batched straight-and-level trim: the reference's F16.trim (env.py:198-292)
// for B flight conditions (h_t, v_t) at once, one wavefront lane per condition.
//
// The reference minimises  cost(UX0) = w . xdot[0:12]^2  over UX0 = (P3, dh, da, dr, alpha) with
// scipy.optimize.minimize(method='Nelder-Mead', tol=1e-10, maxiter=5e4) -- 1,932 sequential _calc_xdot calls
// per aircraft.  Here the same Nelder-Mead iteration (scipy/optimize/_optimize.py:_minimize_neldermead: rho 1,
// chi 2, psi 0.5, sigma 0.5, initial simplex +5 % / 0.00025, termination max|sim[1:]-sim[0]| <= xatol and
// max|f0 - f[1:]| <= fatol) runs as a per-lane state machine in which EVERY loop trip evaluates exactly one
// candidate point, so lanes that reflect, expand, contract or shrink stay convergent on the expensive part
// (the plant evaluation, same device code as the dynamics kernels); the simplex lives in a lane-private LDS column.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/f16_hip.h"
#include "f16_ctx.h"
#include "f16_plant.hpp"

namespace f16 {

struct TrimArgs {
  const double *tab, *lofi, *h, *v;
  double *xtrim, *cost;
  int32_t *iters, *nfev, *status;
  long B, ld;
  double xcg;
  int fi;
  unsigned flags;
  int maxiter;
  double xatol, fatol;
  double x0[5];
};

// obj_func of env.py:217-262
template <typename TP>
__device__ __forceinline__ double trim_cost(TP T, const double *LT, const double *q, double h, double V, double xcg, int fi,
                                            unsigned flags, int &st, double *xfull) {
  const double pi = 3.141592653589793;
  const double P3 = q[0], dh = q[1], da = q[2], dr = q[3], alpha = q[4];
  const double rho0 = 2.377e-3;
  const double tfac = 1 - 0.703e-5 * h;
  double temp = 519 * tfac;
  if (h >= 35000) temp = 390;
  const double rho = rho0 * pow(tfac, 4.14);
  const double qbar = 0.5 * rho * (V * V);
  const double ps = 1715 * rho * temp;
  const double dlef = 1.38 * alpha * 180 / pi - 9.05 * qbar / ps + 1.45;
  double x[18] = {0, 0, h, 0, alpha, 0, V, alpha, 0, 0, 0, 0, P3, dh, da, dr, dlef, -alpha * 180 / pi};
  x[12] = clipd(x[12], 1000, 19000);
  x[13] = clipd(x[13], -25, 25);
  x[14] = clipd(x[14], -21.5, 21.5);
  x[15] = clipd(x[15], -30., 30);
  x[7] = clipd(x[7], -20. * pi / 180, 90 * pi / 180);
  const double u[4] = {x[12], x[13], x[14], x[15]};
  double xd[18];
  calc_xdot(T, LT, x, u, xd, xcg, fi, flags, st);
  const double w[12] = {0, 0, 5, 10, 10, 10, 2, 10, 10, 10, 10, 10};
  double c = 0.0;
#pragma unroll
  for (int k = 0; k < 12; ++k) c += w[k] * (xd[k] * xd[k]);
  if (xfull) {
#pragma unroll
    for (int k = 0; k < 18; ++k) xfull[k] = x[k];
  }
  return c;
}

enum { TS_INIT = 0, TS_REFLECT, TS_EXPAND, TS_CONTRACT_OUT, TS_CONTRACT_IN, TS_SHRINK, TS_DONE };

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_trim(TrimArgs a) {
  __shared__ __attribute__((aligned(16))) double tab[TABLE_IMAGE_DOUBLES];
  __shared__ double simx[36][BLOCK];     // rows p*5+k: simplex vertex p (sorted by cost), coordinate k; rows 30+p: cost
  if (a.fi == 1) {
    const double2 *src = reinterpret_cast<const double2 *>(a.tab);
    double2 *dst = reinterpret_cast<double2 *>(tab);
    for (int i = threadIdx.x; i < TABLE_IMAGE_DOUBLES / 2; i += BLOCK) dst[i] = src[i];
    __syncthreads();
  }
  const int t = threadIdx.x;
#define SIM(p, k) simx[(p) * 5 + (k)][t]
#define FS(p) simx[30 + (p)][t]
  for (long b0 = (long)blockIdx.x * BLOCK; b0 < a.B; b0 += (long)gridDim.x * BLOCK) {
    const long b = b0 + t;
    const bool valid = b < a.B;
    const double h = valid ? a.h[b] : 10000.0, V = valid ? a.v[b] : 700.0;
    // initial simplex (scipy: nonzdelt 0.05, zdelt 0.00025)
    for (int p = 0; p < 6; ++p)
      for (int k = 0; k < 5; ++k) {
        double y = a.x0[k];
        if (p == k + 1) y = (y != 0.0) ? (1 + 0.05) * y : 0.00025;
        SIM(p, k) = y;
      }
    int state = TS_INIT, sub = 0, iters = 0, nfev = 0, st = 0;
    double xr[5], fxr = 0.0, xbar[5];
    bool done = !valid;
    while (true) {
      // ---- termination test / next candidate
      double q[5];
      if (state == TS_REFLECT && !done) {
        double dx = 0.0, df = 0.0;
        for (int p = 1; p < 6; ++p) {
          for (int k = 0; k < 5; ++k) dx = fmax(dx, fabs(SIM(p, k) - SIM(0, k)));
          df = fmax(df, fabs(FS(0) - FS(p)));
        }
        if ((dx <= a.xatol && df <= a.fatol) || iters >= a.maxiter) done = true;
        else {
          ++iters;
          for (int k = 0; k < 5; ++k) {
            double s = SIM(0, k);
            for (int p = 1; p < 5; ++p) s += SIM(p, k);
            xbar[k] = s / 5;
            xr[k] = (1 + 1.0) * xbar[k] - 1.0 * SIM(5, k);
          }
        }
      }
      if (__all(done)) break;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        double v;
        switch (state) {
          case TS_INIT: v = SIM(sub, k); break;
          case TS_REFLECT: v = xr[k]; break;
          case TS_EXPAND: v = (1 + 1.0 * 2.0) * xbar[k] - 1.0 * 2.0 * SIM(5, k); break;
          case TS_CONTRACT_OUT: v = (1 + 0.5 * 1.0) * xbar[k] - 0.5 * 1.0 * SIM(5, k); break;
          case TS_CONTRACT_IN: v = (1 - 0.5) * xbar[k] + 0.5 * SIM(5, k); break;
          default: v = SIM(sub, k); break;   // TS_SHRINK: vertex `sub` was already moved
        }
        q[k] = done ? a.x0[k] : v;
      }
      int stq = 0;
      const double fq = trim_cost((const double *)tab, a.lofi, q, h, V, a.xcg, a.fi, a.flags, stq, nullptr);
      if (done) continue;
      ++nfev;
      // ---- bookkeeping (scipy _minimize_neldermead, one branch per evaluated point)
      bool accept = false, shrink = false, resort = false;
      double xa[5], fa = fq;
      for (int k = 0; k < 5; ++k) xa[k] = q[k];
      if (state == TS_INIT) {
        FS(sub) = fq;
        if (++sub == 6) { resort = true; state = TS_REFLECT; }
      } else if (state == TS_REFLECT) {
        fxr = fq;
        if (fxr < FS(0)) state = TS_EXPAND;
        else if (fxr < FS(4)) accept = true;
        else if (fxr < FS(5)) state = TS_CONTRACT_OUT;
        else state = TS_CONTRACT_IN;
      } else if (state == TS_EXPAND) {
        if (!(fq < fxr)) { for (int k = 0; k < 5; ++k) xa[k] = xr[k]; fa = fxr; }
        accept = true;
      } else if (state == TS_CONTRACT_OUT) {
        if (fq <= fxr) accept = true; else shrink = true;
      } else if (state == TS_CONTRACT_IN) {
        if (fq < FS(5)) accept = true; else shrink = true;
      } else if (state == TS_SHRINK) {
        FS(sub) = fq;
        if (++sub == 6) { resort = true; state = TS_REFLECT; }
        else for (int k = 0; k < 5; ++k) SIM(sub, k) = SIM(0, k) + 0.5 * (SIM(sub, k) - SIM(0, k));
      }
      if (accept) {   // replace the worst vertex, keep the list sorted (stable: after equal costs, like numpy's argsort)
        int pos = 5;
        for (int p = 4; p >= 0; --p) if (fa < FS(p)) pos = p;
        for (int p = 5; p > pos; --p) {
          for (int k = 0; k < 5; ++k) SIM(p, k) = SIM(p - 1, k);
          FS(p) = FS(p - 1);
        }
        for (int k = 0; k < 5; ++k) SIM(pos, k) = xa[k];
        FS(pos) = fa;
        state = TS_REFLECT;
      }
      if (shrink) {
        state = TS_SHRINK; sub = 1;
        for (int k = 0; k < 5; ++k) SIM(1, k) = SIM(0, k) + 0.5 * (SIM(1, k) - SIM(0, k));
      }
      if (resort) {   // insertion sort of the six vertices by cost
        for (int i = 1; i < 6; ++i) {
          double fi_ = FS(i), xi[5];
          for (int k = 0; k < 5; ++k) xi[k] = SIM(i, k);
          int j = i - 1;
          while (j >= 0 && FS(j) > fi_) {
            for (int k = 0; k < 5; ++k) SIM(j + 1, k) = SIM(j, k);
            FS(j + 1) = FS(j);
            --j;
          }
          for (int k = 0; k < 5; ++k) SIM(j + 1, k) = xi[k];
          FS(j + 1) = fi_;
        }
      }
    }
    if (valid) {
      // x_trim of env.py:275-290 (unclipped optimiser output; lef from the formula)
      double q[5], xf[18];
      for (int k = 0; k < 5; ++k) q[k] = SIM(0, k);
      const double c = trim_cost((const double *)tab, a.lofi, q, h, V, a.xcg, a.fi, a.flags, st, xf);
      xf[7] = q[4]; xf[12] = q[0]; xf[13] = q[1]; xf[14] = q[2]; xf[15] = q[3];
#pragma unroll
      for (int k = 0; k < 18; ++k) a.xtrim[k * a.ld + b] = xf[k];
      if (a.cost) a.cost[b] = c;
      if (a.iters) a.iters[b] = iters;
      if (a.nfev) a.nfev[b] = nfev;
      if (a.status) a.status[b] |= st | (iters >= a.maxiter ? F16_ST_QP_MAXITER : 0);
    }
  }
#undef SIM
#undef FS
}

}  // namespace f16

using namespace f16;

extern "C" int f16_trim_batch(f16_ctx *ctx, const double *h, const double *v, double *x_trim, double *cost, int32_t *iters,
                              int32_t *nfev, int32_t *status, long B, long ld, double xcg, int fi_flag, unsigned flags,
                              int maxiter, const double *h_x0, void *stream) {
  if (!ctx || !h || !v || !x_trim || B < 0 || ld < B) return set_error(F16_EINVAL, "bad argument to f16_trim_batch");
  if (B == 0) return F16_OK;
  TrimArgs a{};
  a.tab = ctx->d_tab; a.lofi = ctx->d_lofi; a.h = h; a.v = v; a.xtrim = x_trim; a.cost = cost; a.iters = iters; a.nfev = nfev;
  a.status = status; a.B = B; a.ld = ld; a.xcg = xcg; a.fi = fi_flag; a.flags = flags;
  a.maxiter = maxiter > 0 ? maxiter : 50000;        // env.py:273
  a.xatol = 1e-10; a.fatol = 1e-10;                 // tol=1e-10
  static const double x0_ref[5] = {5000, -0.09, 8.49, -0.01, 0.01};   // env.py:265-271 (order as passed to minimize)
  for (int k = 0; k < 5; ++k) a.x0[k] = h_x0 ? h_x0[k] : x0_ref[k];
  const long blocks = (B + 63) / 64;
  hipLaunchKernelGGL(k_trim<64>, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(64), 0, (hipStream_t)stream, a);
  return hip_check(hipGetLastError(), "f16_trim_batch launch");
}
