// f16_trim.hip -- batched straight-and-level trim: the reference's F16.trim (env.py:198-292)
// for B flight conditions (h_t, v_t) at once, one wavefront lane per condition.
//
// The reference minimises  cost(UX0) = w . xdot[0:12]^2  over UX0 = (P3, dh, da, dr, alpha) with
// scipy.optimize.minimize(method='Nelder-Mead', tol=1e-10, maxiter=5e4) -- 1,932 sequential _calc_xdot calls
// per aircraft.  Here the same Nelder-Mead iteration (scipy/optimize/_optimize.py:_minimize_neldermead: rho 1,
// chi 2, psi 0.5, sigma 0.5, initial simplex +5 % / 0.00025, termination max|sim[1:]-sim[0]| <= xatol and
// max|f0 - f[1:]| <= fatol) with SIXTEEN LANES PER CONDITION (one DPP row): every point an iteration could need depends
// on the simplex at the start of the iteration only -- the reflected, expanded and both contracted points, and the five
// vertices of a shrink -- so the row evaluates all nine at once (lane s takes candidate s) and an iteration costs ONE plant
// evaluation of latency whatever branch scipy's rules then take; the decisions, the accepted coordinates and the
// evaluation COUNT are those of the sequential algorithm (nfev counts what scipy would have evaluated).  The simplex is
// replicated in the registers of the row's lanes (identical bookkeeping on every lane: no LDS, no synchronisation).
// A condition that cannot be trimmed runs to the reference's maxiter = 50,000 iterations as scipy does: 50,000 evaluation
// latencies (~0.15 s) instead of the ~350,000 of the one-evaluation-per-trip form this replaces.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>

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
  int fast_forward;     // stop at a fixed point of the iteration and account for the remaining iterations (F16_TRIM_FASTFORWARD=0: run them)
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

// value of lane `src` of this lane's 16-lane row
__device__ __forceinline__ double row_get(double v, int src) {
  const int lane = (threadIdx.x & 48) | src;                      // within the wavefront
  const int lo = __shfl(__double2loint(v), lane, 64), hi = __shfl(__double2hiint(v), lane, 64);
  return __hiloint2double(hi, lo);
}

// stable sort of the six vertices by cost (numpy argsort on six elements: insertion sort, i.e. stable)
__device__ __forceinline__ void sort_simplex(double (&sim)[6][5], double (&fs)[6]) {
  int rank[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    int r = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) r += (fs[j] < fs[i] || (j < i && fs[j] == fs[i])) ? 1 : 0;
    rank[i] = r;
  }
  double ns[6][5], nf[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    nf[r] = fs[0];
#pragma unroll
    for (int k = 0; k < 5; ++k) ns[r][k] = sim[0][k];
#pragma unroll
    for (int i = 1; i < 6; ++i) {
      const bool here = rank[i] == r;
      nf[r] = here ? fs[i] : nf[r];
#pragma unroll
      for (int k = 0; k < 5; ++k) ns[r][k] = here ? sim[i][k] : ns[r][k];
    }
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    fs[r] = nf[r];
#pragma unroll
    for (int k = 0; k < 5; ++k) sim[r][k] = ns[r][k];
  }
}

__global__ __launch_bounds__(256) void k_trim(TrimArgs a) {
  __shared__ __attribute__((aligned(16))) double tab[TABLE_IMAGE_DOUBLES];
  if (a.fi == 1) {
    const double2 *src = reinterpret_cast<const double2 *>(a.tab);
    double2 *dst = reinterpret_cast<double2 *>(tab);
    for (int i = threadIdx.x; i < TABLE_IMAGE_DOUBLES / 2; i += 256) dst[i] = src[i];
    __syncthreads();
  }
  const int s = threadIdx.x & 15;                                 // candidate slot of this lane
  // one row per condition, four conditions per wavefront; wavefronts stride over the batch independently (no barrier below)
  for (long b = (long)blockIdx.x * 16 + (threadIdx.x >> 4); b < ((a.B + 3) & ~3L); b += (long)gridDim.x * 16) {
    const bool valid = b < a.B;
    const double h = valid ? a.h[b] : 10000.0, V = valid ? a.v[b] : 700.0;
    double sim[6][5], fs[6];
    // initial simplex (scipy: nonzdelt 0.05, zdelt 0.00025)
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        double y = a.x0[k];
        if (p == k + 1) y = (y != 0.0) ? (1 + 0.05) * y : 0.00025;
        sim[p][k] = y;
      }
    int iters = 0, nfev = 6, st = 0;
    {
      double q[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        q[k] = sim[0][k];
#pragma unroll
        for (int p = 1; p < 6; ++p) q[k] = s == p ? sim[p][k] : q[k];
      }
      int stq = 0;
      const double fq = trim_cost((const double *)tab, a.lofi, q, h, V, a.xcg, a.fi, a.flags, stq, nullptr);
#pragma unroll
      for (int p = 0; p < 6; ++p) fs[p] = row_get(fq, p);
      sort_simplex(sim, fs);
    }
    bool done = !valid;
    while (true) {
      if (!done) {
        double dx = 0.0, df = 0.0;
#pragma unroll
        for (int p = 1; p < 6; ++p) {
#pragma unroll
          for (int k = 0; k < 5; ++k) dx = fmax(dx, fabs(sim[p][k] - sim[0][k]));
          df = fmax(df, fabs(fs[0] - fs[p]));
        }
        if ((dx <= a.xatol && df <= a.fatol) || iters >= a.maxiter) done = true;
      }
      if (__all(done)) break;
      if (!done) ++iters;
      // every point this iteration can need, from the simplex as it stands
      double xr[5], xe[5], xc[5], xcc[5], q[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        double sb = sim[0][k];
#pragma unroll
        for (int p = 1; p < 5; ++p) sb += sim[p][k];
        const double xbar = sb / 5;
        xr[k] = (1 + 1.0) * xbar - 1.0 * sim[5][k];
        xe[k] = (1 + 1.0 * 2.0) * xbar - 1.0 * 2.0 * sim[5][k];
        xc[k] = (1 + 0.5 * 1.0) * xbar - 0.5 * 1.0 * sim[5][k];
        xcc[k] = (1 - 0.5) * xbar + 0.5 * sim[5][k];
        double v = s == 1 ? xe[k] : (s == 2 ? xc[k] : (s == 3 ? xcc[k] : xr[k]));
#pragma unroll
        for (int j = 1; j < 6; ++j) v = s == 3 + j ? sim[0][k] + 0.5 * (sim[j][k] - sim[0][k]) : v;
        q[k] = done ? a.x0[k] : v;
      }
      int stq = 0;
      const double fq = trim_cost((const double *)tab, a.lofi, q, h, V, a.xcg, a.fi, a.flags, stq, nullptr);
      const double fxr = row_get(fq, 0), fxe = row_get(fq, 1), fxc = row_get(fq, 2), fxcc = row_get(fq, 3);
      double fsh[6];
#pragma unroll
      for (int j = 1; j < 6; ++j) fsh[j] = row_get(fq, 3 + j);
      if (done) continue;
      // ---- scipy _minimize_neldermead, one iteration (nfev: the evaluations the sequential algorithm makes)
      const int nfev_in = nfev;
      double sim_in[6][5], fs_in[6];                            // (for the fixed-point test below)
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        fs_in[p] = fs[p];
#pragma unroll
        for (int k = 0; k < 5; ++k) sim_in[p][k] = sim[p][k];
      }
      bool accept = false, shrink = false;
      double xa[5], fa = fxr;
#pragma unroll
      for (int k = 0; k < 5; ++k) xa[k] = xr[k];
      nfev += 1;
      if (fxr < fs[0]) {
        nfev += 1;
        if (fxe < fxr) {
          fa = fxe;
#pragma unroll
          for (int k = 0; k < 5; ++k) xa[k] = xe[k];
        }
        accept = true;
      } else if (fxr < fs[4]) {
        accept = true;
      } else if (fxr < fs[5]) {
        nfev += 1;
        if (fxc <= fxr) {
          accept = true; fa = fxc;
#pragma unroll
          for (int k = 0; k < 5; ++k) xa[k] = xc[k];
        } else shrink = true;
      } else {
        nfev += 1;
        if (fxcc < fs[5]) {
          accept = true; fa = fxcc;
#pragma unroll
          for (int k = 0; k < 5; ++k) xa[k] = xcc[k];
        } else shrink = true;
      }
      if (accept) {   // replace the worst vertex, keep the list sorted (stable: after equal costs, like numpy's argsort)
        int pos = 5;
#pragma unroll
        for (int p = 4; p >= 0; --p) pos = fa < fs[p] ? p : pos;
#pragma unroll
        for (int p = 5; p >= 1; --p) {
          const bool shift = p > pos, here = p == pos;
          fs[p] = shift ? fs[p - 1] : (here ? fa : fs[p]);
#pragma unroll
          for (int k = 0; k < 5; ++k) sim[p][k] = shift ? sim[p - 1][k] : (here ? xa[k] : sim[p][k]);
        }
        if (pos == 0) {
          fs[0] = fa;
#pragma unroll
          for (int k = 0; k < 5; ++k) sim[0][k] = xa[k];
        }
      }
      if (shrink) {
        nfev += 5;
#pragma unroll
        for (int j = 1; j < 6; ++j) {
          fs[j] = fsh[j];
#pragma unroll
          for (int k = 0; k < 5; ++k) sim[j][k] = sim[0][k] + 0.5 * (sim[j][k] - sim[0][k]);
        }
        sort_simplex(sim, fs);
      }
      // A FIXED POINT of the iteration: the simplex and its costs come out bit for bit as they went in.  The iteration is a
      // deterministic function of (simplex, costs), so every remaining iteration repeats this one: scipy would run them all
      // to maxiter (env.py:273) and arrive at the same simplex, the same iteration count and nfev + the same increment each
      // time -- fast-forward instead of spending 50,000 plant evaluations of latency on them.  (Seen where the thrust
      // command saturates: the cost is flat in P3, the simplex drifts beyond |P3| ~ 4.5e5 where neighbouring doubles are
      // more than xatol = 1e-10 apart, and a shrink step x0 + 0.5 (xj - x0) of a one-ulp gap rounds back to xj.)
      bool same = true;
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        same = same && (__double_as_longlong(fs[p]) == __double_as_longlong(fs_in[p]));
#pragma unroll
        for (int k = 0; k < 5; ++k) same = same && (__double_as_longlong(sim[p][k]) == __double_as_longlong(sim_in[p][k]));
      }
      if (same && a.fast_forward) {
        const int left = a.maxiter - iters;                   // (the termination test did not fire for this simplex and never will)
        nfev += left * (nfev - nfev_in);
        iters = a.maxiter;
      }
    }
    if (valid && s == 0) {
      // x_trim of env.py:275-290 (unclipped optimiser output; lef from the formula)
      double q[5], xf[18];
#pragma unroll
      for (int k = 0; k < 5; ++k) q[k] = sim[0][k];
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
  { const char *e = getenv("F16_TRIM_FASTFORWARD"); a.fast_forward = !(e && e[0] == '0'); }
  static const double x0_ref[5] = {5000, -0.09, 8.49, -0.01, 0.01};   // env.py:265-271 (order as passed to minimize)
  for (int k = 0; k < 5; ++k) a.x0[k] = h_x0 ? h_x0[k] : x0_ref[k];
  const long blocks = (B + 15) / 16;       // 16 conditions (16 lanes each) per 256-lane workgroup, one workgroup per CU
  hipLaunchKernelGGL(k_trim, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, (hipStream_t)stream, a);
  return hip_check(hipGetLastError(), "f16_trim_batch launch");
}
