// f16_mpc_big.hip -- the OSQP solve of the condensed MPC QP (env.py:420-424) for LONG horizons, 33 <= N <= 150 (the reference
// sweeps N = 1..150, env.py:426-436): one 1024-lane WORKGROUP per aircraft.
//
// Same rules as the other solvers (f16_control.hip: k_mpc; f16_mpc_solve.hip; f16_mpc_wave.hip): Ruiz equilibration on the
// original entries and the running D, E, c; rho vector; x unscaled, zb / yb scaled; termination on the unscaled residuals; rho
// estimate on the scaled ones; primal-infeasibility certificate.  What changes is where things live: at N = 150 the KKT
// matrix is 450 x 450 (1.6 MB), far beyond registers and LDS, so
//   * the Gram matrix A'WA (it does not depend on rho) and the KKT matrix / its inverse are PACKED lower triangles in the
//     per-aircraft HBM workspace (L2 / Infinity-Cache resident: 64 aircraft x 3.3 MB), the inverse is then mirrored into a full
//     n x n matrix so that x~ = K^-1 rhs reads rows contiguously (one wavefront per row group, lanes across the columns);
//   * the inverse is the symmetric sweep (Gauss-Jordan without pivoting, SPD) with one wavefront per row and the pivot column
//     in LDS: n barriers per factorisation;
//   * vectors (G_k, q, pred, x, rhs, E, D, the row vector w) live in LDS (102 KB at N = 150), the per-row values of a lane's
//     (up to two) constraint rows in registers.
// The one-wavefront kernel this replaces for N > 32 (k_mpc<false>, k_mpc<false, true>) ran every loop of the solve on 64
// lanes: four aircraft at N = 150 took 2.2 s.
#include <hip/hip_runtime.h>
#include <math.h>

#include <mutex>

#include "f16_mpc.hpp"
#include "f16_smallmat.hpp"

namespace f16 {
namespace big {

constexpr int BLK = 1024, NW = BLK / 64;
constexpr int TM = (12 * BIG_MAXN + BLK - 1) / BLK;          // constraint rows per lane (2)

// block-wide reduction of one value (sum, or max of non-negative values); every lane receives the result
template <bool SUM>
__device__ __forceinline__ double block_reduce(double v, double *red) {
  v = wave_reduce_dpp<SUM>(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) r = SUM ? r + red[w] : fmax(r, red[w]);
  return r;
}

struct Lds {
  double *G, *qv, *pred, *wbuf, *xs, *xt, *rhs, *tv, *Dg, *E9, *Ec, *Er, *cvec, *red;
};
__host__ __device__ inline size_t lds_doubles(int N) {
  const size_t n = 3 * (size_t)N, m = 12 * (size_t)N;
  auto ev = [](size_t v) { return (v + 1) & ~(size_t)1; };
  return ev(27 * N) + ev(n) + ev(9 * N) + ev(m) + 5 * ev(n) + ev(9 * N) + ev(n) + ev(n + 3) + ev(n) + ev(NW);
}
__device__ __forceinline__ Lds carve(double *p, int N) {
  const int n = 3 * N, m = 12 * N;
  auto take = [&](int k) { double *r = p; p += (k + 1) & ~1; return r; };
  Lds L;
  L.G = take(27 * N); L.qv = take(n); L.pred = take(9 * N); L.wbuf = take(m);
  L.xs = take(n); L.xt = take(n); L.rhs = take(n); L.tv = take(n); L.Dg = take(n);
  L.E9 = take(9 * N); L.Ec = take(n); L.Er = take(n + 3); L.cvec = take(n); L.red = take(NW);
  return L;
}

// out[3j+c] = sum_{i>=j} sum_{r in kept rows} G_{i-j}[r][c] * v[i*6 + rr]   (CCs' v); two lanes per output (the steps i of even /
// odd distance), combined over the DPP network
__device__ __forceinline__ void conv_adjoint6(double *out, const double *G, const double *v, int N) {
  for (int e2 = threadIdx.x; e2 < 2 * ((3 * N + 31) & ~31); e2 += BLK) {      // (whole wavefronts: the DPP exchange below)
    const int e = e2 >> 1, half = e2 & 1;
    double s = 0.0;
    if (e < 3 * N) {
      const int j = e / 3, c = e - 3 * j;
      for (int i = j + half; i < N; i += 2) {
        const double *g = G + (i - j) * 27 + c;
        const double *vi = v + i * 6;
        double t = 0.0;
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) t += g[SROW[rr] * 3] * vi[rr];
        s += t;
      }
    }
    s += dpp_all_f64<0xB1>(s);                             // quad_perm [1,0,3,2]: the partner lane
    if (e < 3 * N && half == 0) out[e] = s;
  }
}
// (CC U)[i][r] = sum_{j<=i} sum_c G_{i-j}[r][c] U[3j+c]
__device__ __forceinline__ double conv_forward_row(const double *G, const double *U, int i, int r) {
  double s = 0.0;
  for (int j = 0; j <= i; ++j) {
    const double *g = G + (i - j) * 27 + r * 3;
    s += g[0] * U[3 * j] + g[1] * U[3 * j + 1] + g[2] * U[3 * j + 2];
  }
  return s;
}
// y = F v for the full symmetric n x n matrix F in HBM: one wavefront per row, lanes across the columns; FOUR rows per trip,
// so that 4 x ceil(n / 64) independent loads are in flight instead of one row's (the matrix streams from L2 / HBM: 1.6 MB
// per product at N = 150 -- with one aircraft per CU this stream, not the arithmetic, is the floor of an iteration)
__device__ __forceinline__ void full_symv(double *y, const double *F, const double *v, int n) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int i0 = w; i0 < n; i0 += 4 * NW) {
    const int i1 = i0 + NW, i2 = i0 + 2 * NW, i3 = i0 + 3 * NW;
    const double *r0 = F + (size_t)i0 * n, *r1 = F + (size_t)(i1 < n ? i1 : i0) * n, *r2 = F + (size_t)(i2 < n ? i2 : i0) * n,
                 *r3 = F + (size_t)(i3 < n ? i3 : i0) * n;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int j = l; j < n; j += 64) {
      const double vj = v[j];
      s0 = fma(r0[j], vj, s0); s1 = fma(r1[j], vj, s1); s2 = fma(r2[j], vj, s2); s3 = fma(r3[j], vj, s3);
    }
    s0 = wave_reduce_dpp<true>(s0); s1 = wave_reduce_dpp<true>(s1); s2 = wave_reduce_dpp<true>(s2); s3 = wave_reduce_dpp<true>(s3);
    if (l == 0) {
      y[i0] = s0;
      if (i1 < n) y[i1] = s1;
      if (i2 < n) y[i2] = s2;
      if (i3 < n) y[i3] = s3;
    }
  }
}
// y = S v for a packed symmetric matrix in HBM (termination test: P x): row part contiguous, column part strided
__device__ __forceinline__ void packed_symv(double *y, const double *S, const double *v, int n) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int i = w; i < n; i += NW) {
    const double *row = S + tri(i, 0);
    double s = 0.0;
    for (int j = l; j <= i; j += 64) s = fma(row[j], v[j], s);
    for (int j = i + 1 + l; j < n; j += 64) s = fma(S[tri(j, i)], v[j], s);
    s = wave_reduce_dpp<true>(s);
    if (l == 0) y[i] = s;
  }
}
// In-place inverse of the packed SPD matrix S (HBM) by the symmetric sweep; c: LDS scratch [n].  Afterwards S = S^-1.
// One wavefront per row (lanes across the columns j <= i), the pivot column in LDS; returns false on a non-positive pivot.
__device__ __forceinline__ bool sweep_inverse_packed(double *S, int n, double *c) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  bool ok = true;
  for (int k = 0; k < n; ++k) {
    __syncthreads();                                      // the previous step's updates are done (block scope: also visible)
    const double piv = S[tri(k, k)];
    if (!(piv > 0.0)) ok = false;
    const double d = 1.0 / piv;
    for (int i = threadIdx.x; i < n; i += BLK) c[i] = i >= k ? S[tri(i, k)] : S[tri(k, i)];
    __syncthreads();
    for (int i = w; i < n; i += NW) {
      double *row = S + tri(i, 0);
      if (i == k) {
        for (int j = l; j < k; j += 64) row[j] = c[j] * d;
        if (l == 0) row[k] = -d;
      } else {
        const double cid = c[i] * d;
        for (int j = l; j <= i; j += 64) row[j] = j == k ? cid : row[j] - cid * c[j];
      }
    }
    __threadfence_block();
  }
  __syncthreads();
  const int np = n * (n + 1) / 2;
  for (int e = threadIdx.x; e < np; e += BLK) S[e] = -S[e];
  __threadfence_block();
  __syncthreads();
  return ok;
}

// per-aircraft HBM workspace of this solver (MpcArgs.bigws): Gram packed | K / inverse packed | inverse full
__host__ __device__ inline size_t ws_doubles(int N) {
  const size_t n = 3 * (size_t)N;
  return n * (n + 1) + n * n;
}

__global__ __launch_bounds__(BLK) void k_mpc_big(MpcArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int N = a.N, n = 3 * N, np = n * (n + 1) / 2, ms = 6 * N, m = 12 * N;
  const int l = threadIdx.x;
  const Lds L = carve(smem, N);
  double *G = L.G, *qv = L.qv, *pred = L.pred, *wbuf = L.wbuf, *xs = L.xs, *xt = L.xt, *rhs = L.rhs, *tv = L.tv, *Dg = L.Dg,
         *E9 = L.E9, *Ec = L.Ec, *Er = L.Er;
  for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
    const double *exw = a.ext + (size_t)b * mpc_ext_doubles(N);
    const double *Pg = a.Ppk + (size_t)b * np;
    double *const gram = a.bigws + (size_t)b * ws_doubles(N), *const Minv = gram + np, *const Full = Minv + np;
    __syncthreads();
    for (int e = l; e < n; e += BLK) qv[e] = exw[e];
    for (int e = l; e < 27 * N; e += BLK) G[e] = exw[n + e];
    for (int e = l; e < 9 * N; e += BLK) pred[e] = exw[n + 27 * N + e];
    __syncthreads();
    // ---------------- bounds of the kept rows (utils.py:129-152): [6N state | 3N command | 3N rate]
    double lo[TM], hi[TM], z[TM], y[TM], dy[TM], Eo[TM], eqf[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int row = l + BLK * t;
      lo[t] = 0.0; hi[t] = 0.0; z[t] = 0.0; y[t] = 0.0; dy[t] = 0.0; Eo[t] = 1.0; eqf[t] = 1.0;
      if (row < ms) {
        const int i = row / 6, rr = row - 6 * i;
        const double pm = pred[i * 9 + SROW[rr]];
        lo[t] = SLB[rr] - pm; hi[t] = SUB[rr] - pm;
      } else if (row < ms + n) {
        const int c = (row - ms) % 3;
        lo[t] = ULB[c]; hi[t] = UUB[c];
      } else if (row < m) {
        const int k = row - ms - n, c = k % 3;
        if (k < 3) {
          const double act = a.x[(13 + c) * a.ld + b];
          lo[t] = act + RLB[c] * a.dt; hi[t] = act + RUB[c] * a.dt;
        } else { lo[t] = RLB[c]; hi[t] = RUB[c]; }          // reference quirk: not multiplied by dt (utils.py:151-152)
      }
    }
    // ---------------- equilibration (scaling.c:scale_data on the original entries and the running D, E, c)
    const double sigma = a.s.sigma, alpha = a.s.alpha;
    double cs = 1.0;
    for (int e = l; e < n; e += BLK) { Dg[e] = 1.0; Ec[e] = 1.0; Er[e] = 1.0; }
    for (int e = l; e < 3; e += BLK) Er[n + e] = 0.0;
    for (int e = l; e < 9 * N; e += BLK) E9[e] = 1.0;
    __syncthreads();
    for (int pass = 0; pass < a.s.scaling; ++pass) {
      for (int e = l; e < n; e += BLK) {                   // column norms of [Pb; Ab]
        const int jb = e / 3, c = e - 3 * jb;
        double mp = 0.0, ma = 0.0;
        for (int i = 0; i < n; ++i) mp = fmax(mp, fabs(Pg[i >= e ? tri(i, e) : tri(e, i)]) * Dg[i]);
        for (int i = jb; i < N; ++i)
          for (int r = 0; r < 9; ++r) ma = fmax(ma, fabs(G[(i - jb) * 27 + r * 3 + c]) * E9[9 * i + r]);
        ma = fmax(fmax(ma, Ec[e]), fmax(Er[e], Er[e + 3]));
        tv[e] = 1.0 / sqrt(osqp_limit_scaling(Dg[e] * fmax(cs * mp, ma)));
      }
      for (int e = l; e < 9 * N; e += BLK) {               // row norms of the state block
        const int i = e / 9, r = e - 9 * i;
        double m_ = 0.0;
        for (int jb = 0; jb <= i; ++jb)
          for (int c = 0; c < 3; ++c) m_ = fmax(m_, fabs(G[(i - jb) * 27 + r * 3 + c]) * Dg[3 * jb + c]);
        wbuf[e] = 1.0 / sqrt(osqp_limit_scaling(E9[e] * m_));          // (m = 12N >= 9N)
      }
      for (int e = l; e < n; e += BLK) {
        rhs[e] = 1.0 / sqrt(osqp_limit_scaling(Ec[e] * Dg[e]));
        xt[e] = 1.0 / sqrt(osqp_limit_scaling(Er[e] * fmax(Dg[e], e >= 3 ? Dg[e - 3] : 0.0)));
      }
      __syncthreads();
      for (int e = l; e < n; e += BLK) { Dg[e] *= tv[e]; Ec[e] *= rhs[e]; Er[e] *= xt[e]; }
      for (int e = l; e < 9 * N; e += BLK) E9[e] *= wbuf[e];
      __syncthreads();
      double sm = 0.0, qn = 0.0;                          // cost scaling: mean column norm of Pb, ||qb||
      for (int e = l; e < n; e += BLK) {
        double mp = 0.0;
        for (int i = 0; i < n; ++i) mp = fmax(mp, fabs(Pg[i >= e ? tri(i, e) : tri(e, i)]) * Dg[i]);
        sm += cs * Dg[e] * mp;
        qn = fmax(qn, cs * Dg[e] * fabs(qv[e]));
      }
      sm = block_reduce<true>(sm, L.red); qn = block_reduce<false>(qn, L.red);
      cs *= 1.0 / fmax(osqp_limit_scaling(sm / n), osqp_limit_scaling(qn));
      __syncthreads();
    }
    // per-row E, scaled bounds, rho-vector factor; then the Gram weights W = E^2 (x 1e3 on equality rows); sigma D^-2
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int row = l + BLK * t;
      if (row < m) {
        Eo[t] = row < ms ? E9[9 * (row / 6) + SROW[row % 6]] : (row < ms + n ? Ec[row - ms] : Er[row - ms - n]);
        lo[t] *= Eo[t]; hi[t] *= Eo[t];
        eqf[t] = (hi[t] - lo[t] < OSQP_RHO_TOL) ? OSQP_RHO_EQ_OVER_RHO_INEQ : 1.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int row = l + BLK * t;
      if (row < ms) E9[9 * (row / 6) + SROW[row % 6]] = Eo[t] * Eo[t] * eqf[t];
      else if (row < ms + n) Ec[row - ms] = Eo[t] * Eo[t] * eqf[t];
      else if (row < m) Er[row - ms - n] = Eo[t] * Eo[t] * eqf[t];
    }
    for (int e = l; e < n; e += BLK) Dg[e] = sigma / (Dg[e] * Dg[e]);
    __syncthreads();
    const double cinv = 1.0 / cs;
    // ---------------- A'WA, packed, once (it does not depend on rho): one wavefront per row of the lower triangle
    {
      const int w = l >> 6, ll = l & 63;
      for (int ia = w; ia < n; ia += NW) {
        const int ja = ia / 3, ca = ia - 3 * ja;
        for (int ib = ll; ib <= ia; ib += 64) {
          const int jb = ib / 3, cb = ib - 3 * jb;
          double s = 0.0;
          for (int i = ja; i < N; ++i) {
            const double *ga = G + (i - ja) * 27 + ca, *gb = G + (i - jb) * 27 + cb, *wv = E9 + 9 * i;
#pragma unroll
            for (int rr = 0; rr < 6; ++rr) s += wv[SROW[rr]] * ga[SROW[rr] * 3] * gb[SROW[rr] * 3];
          }
          if (ia == ib) s += Ec[ia] + Er[ia] + Er[ia + 3];
          else if (ia == ib + 3) s -= Er[ia];
          gram[tri(ia, ib)] = s;
        }
      }
    }
    __threadfence_block();
    __syncthreads();
    double rho = a.s.rho;
    if (!(rho > 0.0)) {   // the builder's opt-in start value (no equilibration): balance the two terms of P + rho A'A
      double tp = 0.0, ta = 0.0;
      for (int e = l; e < n; e += BLK) { tp += Pg[tri(e, e)]; ta += gram[tri(e, e)]; }
      tp = block_reduce<true>(tp, L.red); ta = block_reduce<true>(ta, L.red);
      rho = fmin(fmax(RHO_AUTO_SCALE * sqrt(tp / ta), OSQP_RHO_MIN), OSQP_RHO_MAX);
    }
    auto build_minv = [&](double r) {                     // Full <- (c P + sigma D^-2 + r A'WA)^-1
      __syncthreads();
      {
        const int w = l >> 6, ll = l & 63;
        for (int ia = w; ia < n; ia += NW)
          for (int ib = ll; ib <= ia; ib += 64) {
            const int e = tri(ia, ib);
            Minv[e] = cs * Pg[e] + r * gram[e] + (ia == ib ? Dg[ia] : 0.0);
          }
      }
      __threadfence_block();
      const bool good = sweep_inverse_packed(Minv, n, L.cvec);
      {
        const int w = l >> 6, ll = l & 63;
        for (int i = w; i < n; i += NW)
          for (int j = ll; j < n; j += 64) Full[(size_t)i * n + j] = i >= j ? Minv[tri(i, j)] : Minv[tri(j, i)];
      }
      __threadfence_block();
      __syncthreads();
      return __syncthreads_and(good) != 0;
    };
    bool ok = build_minv(rho);
    for (int e = l; e < n; e += BLK) xs[e] = 0.0;
    __syncthreads();
    int it = 0;
    double rp = INFINITY, rd = INFINITY;
    bool converged = false, infeasible = false;
    bool done = !ok || a.s.max_iter <= 0;
    auto adjoint = [&](const double *wv_, int e) {        // (A' w)_e for w in the [6N | 3N | 3N] layout, tv = CCs' w_s
      return tv[e] + wv_[ms + e] + (wv_[ms + n + e] - (e + 3 < n ? wv_[ms + n + e + 3] : 0.0));
    };
    while (!done) {
      ++it;
      // w = E (rho zb - yb) -> t = A' w ; rhs = sigma D^-2 x - c q + t
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int row = l + BLK * t;
        if (row < m) wbuf[row] = Eo[t] * (rho * eqf[t] * z[t] - y[t]);
      }
      __syncthreads();
      conv_adjoint6(tv, G, wbuf, N);
      __syncthreads();
      for (int e = l; e < n; e += BLK) rhs[e] = Dg[e] * xs[e] - cs * qv[e] + adjoint(wbuf, e);
      __syncthreads();
      full_symv(xt, Full, rhs, n);                         // x~
      __syncthreads();
      // zb~ = E A x~ ; relaxation, projection, dual update
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int row = l + BLK * t;
        if (row < m) {
          double zt;
          if (row < ms) zt = conv_forward_row(G, xt, row / 6, SROW[row % 6]);
          else if (row < ms + n) zt = xt[row - ms];
          else { const int k = row - ms - n; zt = xt[k] - (k >= 3 ? xt[k - 3] : 0.0); }
          zt *= Eo[t];
          const double ro = rho * eqf[t];
          const double zr = alpha * zt + (1 - alpha) * z[t];
          const double zn = fmin(fmax(zr + y[t] / ro, lo[t]), hi[t]);
          dy[t] = ro * (zr - zn);
          y[t] = y[t] + dy[t];
          z[t] = zn;
        }
      }
      for (int e = l; e < n; e += BLK) xs[e] = alpha * xt[e] + (1 - alpha) * xs[e];
      __syncthreads();
      if (it % a.s.check_every == 0 || it >= a.s.max_iter) {
        // residuals of the UNSCALED problem (OSQP termination test) + the scaled ones for the rho estimate
        double r1 = 0.0, nAx = 0.0, nz = 0.0, r1s = 0.0, nAxs = 0.0, nzs = 0.0;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const int row = l + BLK * t;
          if (row < m) {
            double ax;
            if (row < ms) ax = conv_forward_row(G, xs, row / 6, SROW[row % 6]);
            else if (row < ms + n) ax = xs[row - ms];
            else { const int k = row - ms - n; ax = xs[k] - (k >= 3 ? xs[k - 3] : 0.0); }
            const double zu = z[t] / Eo[t];
            r1 = fmax(r1, fabs(ax - zu)); nAx = fmax(nAx, fabs(ax)); nz = fmax(nz, fabs(zu));
            r1s = fmax(r1s, fabs(Eo[t] * ax - z[t])); nAxs = fmax(nAxs, fabs(Eo[t] * ax)); nzs = fmax(nzs, fabs(z[t]));
            wbuf[row] = Eo[t] * y[t];
          }
        }
        __syncthreads();
        packed_symv(xt, Pg, xs, n);                        // P x (packed P from the workspace)
        conv_adjoint6(tv, G, wbuf, N);
        __syncthreads();
        double r2 = 0.0, nPx = 0.0, nAty = 0.0, nq = 0.0, r2s = 0.0, nPxs = 0.0, nAtys = 0.0, nqs = 0.0;
        for (int e = l; e < n; e += BLK) {
          const double aty = cinv * adjoint(wbuf, e), rr_ = xt[e] + qv[e] + aty, cD = cs * sqrt(sigma / Dg[e]);
          r2 = fmax(r2, fabs(rr_)); nPx = fmax(nPx, fabs(xt[e])); nAty = fmax(nAty, fabs(aty)); nq = fmax(nq, fabs(qv[e]));
          r2s = fmax(r2s, cD * fabs(rr_)); nPxs = fmax(nPxs, cD * fabs(xt[e])); nAtys = fmax(nAtys, cD * fabs(aty)); nqs = fmax(nqs, cD * fabs(qv[e]));
        }
        rp = block_reduce<false>(r1, L.red);
        rd = block_reduce<false>(r2, L.red);
        const double np_ = fmax(block_reduce<false>(nAx, L.red), block_reduce<false>(nz, L.red));
        const double nd_ = fmax(fmax(block_reduce<false>(nPx, L.red), block_reduce<false>(nAty, L.red)), block_reduce<false>(nq, L.red));
        __syncthreads();
        if (rp < a.s.eps_abs + a.s.eps_rel * np_ && rd < a.s.eps_abs + a.s.eps_rel * nd_) { done = true; converged = true; }
        else {
          // OSQP primal-infeasibility certificate on dy (auxil.c:is_primal_infeasible)
          double ndy = 0.0, supp = 0.0;
#pragma unroll
          for (int t = 0; t < TM; ++t) {
            const int row = l + BLK * t;
            if (row < m) {
              ndy = fmax(ndy, fabs(Eo[t] * dy[t]));
              supp += hi[t] * fmax(dy[t], 0.0) + lo[t] * fmin(dy[t], 0.0);
              wbuf[row] = Eo[t] * dy[t];
            }
          }
          ndy = block_reduce<false>(ndy, L.red);
          supp = block_reduce<true>(supp, L.red);
          __syncthreads();
          if (ndy > a.s.eps_prim_inf && supp < -a.s.eps_prim_inf * ndy) {
            conv_adjoint6(tv, G, wbuf, N);
            __syncthreads();
            double nat = 0.0;
            for (int e = l; e < n; e += BLK) nat = fmax(nat, fabs(adjoint(wbuf, e)));
            nat = block_reduce<false>(nat, L.red);
            if (nat < a.s.eps_prim_inf * ndy) { done = true; infeasible = true; }
          }
          __syncthreads();
        }
        if (done) {}
        else if (it >= a.s.max_iter) done = true;
        else if (a.s.adaptive_rho && it % a.s.rho_every == 0) {     // auxil.c:compute_rho_estimate (scaled residuals)
          const double pr = block_reduce<false>(r1s, L.red) / (fmax(block_reduce<false>(nzs, L.red), block_reduce<false>(nAxs, L.red)) + 1e-10);
          const double dr = block_reduce<false>(r2s, L.red) /
                            (fmax(fmax(block_reduce<false>(nqs, L.red), block_reduce<false>(nAtys, L.red)), block_reduce<false>(nPxs, L.red)) + 1e-10);
          const double nw = fmin(fmax(rho * sqrt(pr / (dr + 1e-10)), OSQP_RHO_MIN), OSQP_RHO_MAX);
          if (nw > OSQP_ADAPTIVE_RHO_TOLERANCE * rho || nw < rho / OSQP_ADAPTIVE_RHO_TOLERANCE) {
            rho = nw;
            if (!build_minv(rho)) { ok = false; done = true; }
          }
        }
      }
    }
    // res.x[0:3] (env.py:424); OSQP hands back NaN for a problem it certifies infeasible
    for (int e = l; e < 3; e += BLK) a.ucmd[e * a.ld + b] = infeasible ? NAN : xs[e];
    if (a.useq) for (int e = l; e < n; e += BLK) a.useq[e * a.ld + b] = infeasible ? NAN : xs[e];
    if (l == 0) {
      if (a.iters_out) a.iters_out[b] = it;
      if (a.info) {
        a.info[0 * a.ld + b] = (double)it;
        a.info[1 * a.ld + b] = rp;
        a.info[2 * a.ld + b] = rd;
        a.info[3 * a.ld + b] = rho;
      }
      if (a.status && infeasible) a.status[b] |= F16_ST_QP_INFEASIBLE;
      else if (a.status && a.s.max_iter > 0 && (!converged || !ok)) a.status[b] |= F16_ST_QP_MAXITER;
    }
    __syncthreads();
  }
}

}  // namespace big

size_t mpc_big_ws_doubles(int N) { return big::ws_doubles(N); }

int mpc_big_solve_launch(f16_ctx *ctx, const MpcArgs &a, void *stream) {
  (void)ctx;
  if (a.N < 1 || a.N > BIG_MAXN || !a.bigws || !a.ext || !a.Ppk) return set_error(F16_EINVAL, "long-horizon MPC solver: bad arguments");
  static std::mutex mu;
  static bool ready[64] = {};
  int dev = 0;
  if (int rc = hip_check(hipGetDevice(&dev), "hipGetDevice")) return rc;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (dev >= 0 && dev < 64 && !ready[dev]) {           // once per device, to the BIG_MAXN size (never per launch; not legal under capture)
      if (int rc = hip_check(hipFuncSetAttribute((const void *)big::k_mpc_big, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)(big::lds_doubles(BIG_MAXN) * sizeof(double))), "hipFuncSetAttribute(k_mpc_big)")) return rc;
      ready[dev] = true;
    }
  }
  const size_t lds = big::lds_doubles(a.N) * sizeof(double);
  const long grid = a.B < 4096 ? a.B : 4096;
  hipLaunchKernelGGL(big::k_mpc_big, dim3((unsigned)grid), dim3(big::BLK), lds, (hipStream_t)stream, a);
  return hip_check(hipGetLastError(), "f16_mpc_batch long-horizon solve launch");
}

}  // namespace f16
