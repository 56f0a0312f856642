// f16_mpc_solve.hip -- register-resident OSQP-style ADMM for the condensed MPC QP (N <= 32), gfx950.
//
// Same algorithm, settings, termination / rho-update / infeasibility rules as the generic solver in
// f16_control.hip (k_mpc) -- what changes is the mapping:
//
//   one 512-thread workgroup (8 wavefronts) per aircraft; the three linear operators of an ADMM iteration
//     stage 1   t  = CCs' w_s      (3N x 6N, block upper-triangular Toeplitz)      utils.py:163 (A' part)
//     stage 2   x~ = (P + sigma I + rho A'A)^-1 rhs        (3N x 3N dense)          OSQP linear system
//     stage 3   z~ = CCs x~        (6N x 3N, block lower-triangular Toeplitz)       utils.py:163 (A part)
//   are cut into row segments of 24 elements and each lane keeps ITS segment of each operator in registers
//   (3 x 24 fp64 = 144 VGPRs) for the whole solve; per iteration a lane does 72 FMAs against operand vectors
//   that live in LDS (16-byte reads of contiguous runs), partial sums of a row are combined through LDS
//   (stages 1, 3) or by two DPP quad exchanges (stage 2).  The register file (512 KiB/CU) is the only on-chip
//   memory that holds all three operators of an aircraft (24,840 doubles = 199 KB); LDS keeps the vectors.
//   The KKT matrix is inverted in LDS (packed lower triangle) by the symmetric sweep operator, 512 lanes wide.
//
// Per-iteration cost: 24,840 useful MACs (of 36,864 issued with padding) + 5 workgroup barriers.
#include <hip/hip_runtime.h>
#include <math.h>

#include <mutex>
#include <vector>

#include "f16_mpc.hpp"
#include "f16_smallmat.hpp"

namespace f16 {

constexpr int FT = 512;                    // lanes per aircraft
constexpr int FK = 24;                     // operator elements per lane and stage
constexpr int FN = 3 * FAST_MAXN;          // 96
constexpr int FMS = 6 * FAST_MAXN;         // 192
constexpr int FNP = FN * (FN + 1) / 2;     // 4656
constexpr int FSW = (FNP + FT - 1) / FT;   // packed elements per lane in the sweep (10)

struct FastDesc {
  short s1_row[FT], s1_i0[FT];   // stage 1: output e = 3j+c, first horizon step i0 of the lane's 4-step segment
  short s3_row[FT], s3_j0[FT];   // stage 3: state row 6i+rr, first input block j0 of the lane's 8-block segment
  short f1[FN], c1[FN];          // stage-1 partials of output e: first lane, count
  short f3[FMS], c3[FMS];        // stage-3 partials of state row r
};

static int build_desc(int N, FastDesc &d) {
  for (int t = 0; t < FT; ++t) { d.s1_row[t] = d.s3_row[t] = -1; d.s1_i0[t] = d.s3_j0[t] = 0; }
  int lane = 0;
  for (int j = 0; j < N; ++j)
    for (int c = 0; c < 3; ++c) {
      const int e = 3 * j + c, segs = (N - j + 3) / 4;
      d.f1[e] = (short)lane; d.c1[e] = (short)segs;
      for (int s = 0; s < segs; ++s, ++lane) {
        if (lane >= FT) return -1;
        d.s1_row[lane] = (short)e; d.s1_i0[lane] = (short)(j + 4 * s);
      }
    }
  lane = 0;
  for (int i = 0; i < N; ++i)
    for (int rr = 0; rr < 6; ++rr) {
      const int r = 6 * i + rr, segs = (i + 1 + 7) / 8;
      d.f3[r] = (short)lane; d.c3[r] = (short)segs;
      for (int s = 0; s < segs; ++s, ++lane) {
        if (lane >= FT) return -1;
        d.s3_row[lane] = (short)r; d.s3_j0[lane] = (short)(8 * s);
      }
    }
  return 0;
}

// 24-element dot product: registers x contiguous LDS run (16-byte reads)
__device__ __forceinline__ double dot24(const double (&W)[FK], const double *op) {
  const double2 *p = reinterpret_cast<const double2 *>(op);
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int k = 0; k < FK / 2; ++k) {
    const double2 v = p[k];
    a0 = fma(W[2 * k], v.x, a0);
    a1 = fma(W[2 * k + 1], v.y, a1);
  }
  return a0 + a1;
}

// same, both factors in LDS (stage 3: the Toeplitz operator run is read instead of being held in registers)
__device__ __forceinline__ double dot24_lds(const double *wop, const double *op) {
  const double2 *w = reinterpret_cast<const double2 *>(wop);
  const double2 *p = reinterpret_cast<const double2 *>(op);
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int k = 0; k < FK / 2; ++k) {
    const double2 u = w[k], v = p[k];
    a0 = fma(u.x, v.x, a0);
    a1 = fma(u.y, v.y, a1);
  }
  return a0 + a1;
}

__device__ __forceinline__ double quad_sum(double v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}

// workgroup-wide reductions of NV values at once (red: [8][NV] doubles of LDS)
template <int NV>
__device__ __forceinline__ void block_reduce(double (&v)[NV], const bool (&is_sum)[NV], double *red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = is_sum[i] ? wave_sum(v[i]) : wave_max(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wv * NV + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double r = red[i];
    for (int w = 1; w < FT / 64; ++w) r = is_sum[i] ? r + red[w * NV + i] : fmax(r, red[w * NV + i]);
    v[i] = r;
  }
}

// (P + sigma I + rho A'A)^-1 in LDS by the symmetric sweep operator (Gauss-Jordan without pivoting, stable for SPD):
// after pivot k the lower triangle holds the partially swept matrix, after all n pivots -inverse.
// Storage: lower triangle of a square array with leading dimension SLD (rows 16-byte aligned).  Each lane owns one
// run of SRUN consecutive columns of one row, so a pivot step is: 6 x 16-byte reads of the pivot column run, 6 x 16-byte
// reads + writes of its own run, 12 FMAs.  The pivot row and pivot column are lane-level special cases (no per-element
// selects).  One barrier per pivot: while applying pivot k the lanes that produce entries of column k+1 publish them
// (double-buffered cvec).  Out of line: its register allocation must not compete with the caller's operator registers.
constexpr int SLD = FN + 2;    // 98: rows 16-byte aligned and NOT a multiple of the 256-byte LDS bank row
constexpr int SRUN = 12;
__device__ __attribute__((noinline)) bool sweep_inverse(double *Ms, double *cvec, const double *Pg, const double *Ag, double r,
                                                        double sigma, int n, int np) {
  const int tid = threadIdx.x;
  bool ok = true;
  // lane -> (row i, first column j0): rows are cut into ceil((i+1)/SRUN) runs, enumerated row by row
  int row = -1, j0 = 0;
  {
    int acc = 0;
    for (int i = 0; i < n; ++i) {
      const int runs = (i + SRUN) / SRUN;
      if (tid >= acc && tid < acc + runs) { row = i; j0 = (tid - acc) * SRUN; }
      acc += runs;
    }
  }
  __syncthreads();
  for (int e = tid; e < np; e += FT) {                      // unpack P + r A'A into the square array (lower triangle)
    int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
    while (i * (i + 1) / 2 > e) --i;
    while ((i + 1) * (i + 2) / 2 <= e) ++i;
    const int j = e - i * (i + 1) / 2;
    Ms[i * SLD + j] = Pg[e] + r * Ag[e] + (i == j ? sigma : 0.0);
  }
  // zero the part of each row's last run that lies right of the diagonal (read by the 16-byte loads, never meaningful)
  if (row >= 0 && j0 + SRUN > row + 1)
    for (int j = row + 1 > j0 ? row + 1 : j0; j < j0 + SRUN && j < SLD; ++j) Ms[row * SLD + j] = 0.0;
  __syncthreads();
  if (tid < n) cvec[tid] = Ms[tid * SLD];                   // column 0
  if (tid >= n && tid < FN + 8) cvec[tid] = 0.0;
  if (tid < FN + 8) cvec[FN + 8 + tid] = 0.0;
  __syncthreads();
  double *mrun = Ms + (row >= 0 ? row : 0) * SLD + j0;
#ifdef F16_EXP_NOPIVOT
  n = 0;
#endif
#ifdef F16_EXP_STAMP
  unsigned long long t_work = 0, t_bar = 0, t0 = __builtin_amdgcn_s_memtime();
#endif
  for (int k = 0; k < n; ++k) {
    const double *cv = cvec + (k & 1) * (FN + 8);
    double *cn = cvec + ((k + 1) & 1) * (FN + 8);
    const double piv = cv[k];
    if (!(piv > 0.0)) ok = false;
    const double d = 1.0 / piv;
    if (row >= 0) {
      double2 cj[SRUN / 2], mv[SRUN / 2];
      const double2 *cp = reinterpret_cast<const double2 *>(cv + j0);
      double2 *mp = reinterpret_cast<double2 *>(mrun);
#pragma unroll
      for (int q = 0; q < SRUN / 2; ++q) { cj[q] = cp[q]; mv[q] = mp[q]; }
      const double ci = cv[row];
      // generic element: m - (c_i d) c_j ; pivot row (a handful of lanes): c_j d.  Entries right of the diagonal in the
      // last run of a row are scratch: they stay finite and are never read as matrix entries.
      const double cid = (row == k) ? -d : ci * d;
#pragma unroll
      for (int q = 0; q < SRUN / 2; ++q) {
        const double bx = (row == k) ? 0.0 : mv[q].x, by = (row == k) ? 0.0 : mv[q].y;
        mv[q].x = bx - cid * cj[q].x;
        mv[q].y = by - cid * cj[q].y;
      }
      const int kk = k - j0;                                 // pivot column / diagonal inside this run (register fix-up)
      if (kk >= 0 && kk < SRUN && row >= k) {
        const double fix = (row == k) ? -d : ci * d;
#pragma unroll
        for (int q = 0; q < SRUN / 2; ++q) {
          if (2 * q == kk) mv[q].x = fix;
          if (2 * q + 1 == kk) mv[q].y = fix;
        }
      }
#pragma unroll
      for (int q = 0; q < SRUN / 2; ++q) mp[q] = mv[q];
      // publish column k+1 for the next pivot, straight from registers (no LDS read-back on the critical path)
      if (row == k + 1) {
        if (j0 + SRUN <= row + 1) {
          double2 *cnp = reinterpret_cast<double2 *>(cn + j0);
#pragma unroll
          for (int q = 0; q < SRUN / 2; ++q) cnp[q] = mv[q];
        } else {
#pragma unroll
          for (int q = 0; q < SRUN / 2; ++q) {
            if (j0 + 2 * q <= row) cn[j0 + 2 * q] = mv[q].x;
            if (j0 + 2 * q + 1 <= row) cn[j0 + 2 * q + 1] = mv[q].y;
          }
        }
      } else if (row > k + 1) {
        const int k1 = k + 1 - j0;
        if (k1 >= 0 && k1 < SRUN) {
          double val = 0.0;
#pragma unroll
          for (int q = 0; q < SRUN / 2; ++q) {
            if (2 * q == k1) val = mv[q].x;
            if (2 * q + 1 == k1) val = mv[q].y;
          }
          cn[row] = val;
        }
      }
    }
#ifdef F16_EXP_STAMP
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    t_work += t1 - t0; t_bar += t2 - t1; t0 = t2;
#else
    __syncthreads();
#endif
  }
#ifdef F16_EXP_STAMP
  if (tid == 0) { cvec[0] = (double)t_work; cvec[1] = (double)t_bar; }
  __syncthreads();
#endif
  return ok;
}

__global__ __launch_bounds__(FT) void k_mpc_fast(MpcArgs a, const FastDesc *__restrict__ D) {
  __shared__ __attribute__((aligned(16))) double Mp[FN * (FN + 2)];   // square, lower triangle used (sweep_inverse)
  __shared__ __attribute__((aligned(16))) double G[27 * FAST_MAXN];
  __shared__ __attribute__((aligned(16))) double ws[FMS + FK], wc[FN], wr[FN + 4];
  __shared__ __attribute__((aligned(16))) double ys[FMS + FK], yc[FN], yr[FN + 4];
  __shared__ __attribute__((aligned(16))) double rhs[FN], xt[FN + FK], xb[FN + FK], pxv[FN];
  __shared__ double p1[FT + 8], p3[FT + 8], cvec[2 * (FN + 8)], red[8 * 8];
  // stage-3 operator: per kept state row type rr the sequence R_rr[3j'+c] = G_{N-1-j'}[S_rr][c] (0 for j' >= N); row (i,rr)
  // segment j0 is the contiguous run starting at 3(j0+N-1-i).  Second copy shifted by one double so that every run
  // start is 16-byte aligned in one of the two.
  constexpr int RL = FN + FK + 4;
  __shared__ __attribute__((aligned(16))) double Rq[2][6][RL];

  const int N = a.N, n = 3 * N, np = n * (n + 1) / 2, ms = 6 * N, m = 12 * N;
  const int tid = threadIdx.x;
  // ---- lane roles (fixed for the whole launch)
  const int r1 = D->s1_row[tid], i01 = D->s1_i0[tid];
  const int r3 = D->s3_row[tid], j03 = D->s3_j0[tid];
  const int r2 = tid >> 2, h2 = tid & 3;                      // stage 2: four lanes per row, 24 columns each
  const bool xown = tid < n, rown = tid < m;
  const int f1 = xown ? D->f1[tid] : 0, c1 = xown ? D->c1[tid] : 0;
  const bool srow = tid < ms;
  const int f3 = srow ? D->f3[tid] : 0, c3 = srow ? D->c3[tid] : 0;
  // zero the pads once (never written again)
  for (int e = tid; e < FMS + FK; e += FT) { ws[e] = 0.0; ys[e] = 0.0; }
  for (int e = tid; e < FN + FK; e += FT) { xt[e] = 0.0; xb[e] = 0.0; }
  for (int e = tid; e < FN + 4; e += FT) { wr[e] = 0.0; yr[e] = 0.0; }
  for (int e = tid; e < FN; e += FT) { rhs[e] = 0.0; wc[e] = 0.0; yc[e] = 0.0; pxv[e] = 0.0; }
  __syncthreads();

  const double sigma = a.s.sigma, alpha = a.s.alpha;

  for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
    const double *ex = a.ext + (size_t)b * mpc_ext_doubles(N);
    const double *Pg = a.Ppk + (size_t)b * np, *Ag = a.Apk + (size_t)b * np;
    const double *pred = ex + n + 27 * N;
    for (int e = tid; e < 27 * N; e += FT) G[e] = ex[n + e];
    const double qe = xown ? ex[tid] : 0.0;
    // ---- bounds of this lane's constraint row (utils.py:129-152; rows with two infinite bounds are not kept)
    double lo = 0.0, hi = 0.0, z = 0.0, y = 0.0, dy = 0.0, xs = 0.0;
    if (srow) {
      const int i = tid / 6, rr = tid - 6 * i;
      const double pm = pred[i * 9 + SROW[rr]];
      lo = SLB[rr] - pm; hi = SUB[rr] - pm;
    } else if (tid < ms + n) {
      const int c = (tid - ms) % 3;
      lo = ULB[c]; hi = UUB[c];
    } else if (rown) {
      const int k = tid - ms - n, c = k % 3;
      if (k < 3) {
        const double act = a.x[(13 + c) * a.ld + b];
        lo = act + RLB[c] * a.dt; hi = act + RUB[c] * a.dt;
      } else { lo = RLB[c]; hi = RUB[c]; }               // reference quirk: not scaled by dt (utils.py:151-152)
    }
    __syncthreads();
    // ---- operator segments into registers
    for (int e = tid; e < 6 * RL; e += FT) {
      const int rr = e / RL, t = e - rr * RL;
      constexpr int SRc[6] = {2, 3, 4, 5, 6, 8};
      const int srr = rr == 0 ? SRc[0] : rr == 1 ? SRc[1] : rr == 2 ? SRc[2] : rr == 3 ? SRc[3] : rr == 4 ? SRc[4] : SRc[5];
      auto val = [&](int u) { const int jp = u / 3, c = u - 3 * jp; return jp < N ? G[(N - 1 - jp) * 27 + srr * 3 + c] : 0.0; };
      Rq[0][rr][t] = val(t);
      Rq[1][rr][t] = val(t + 1);
    }
    const double *w3run;                                            // this lane's 24-element run of the stage-3 operator
    {
      const int i3 = r3 >= 0 ? r3 / 6 : 0, rr3 = r3 >= 0 ? r3 - 6 * i3 : 0;
      const int st = 3 * (j03 + N - 1 - i3);
      w3run = (st & 1) ? &Rq[1][rr3][st - 1] : &Rq[0][rr3][st];
    }
    double W1[FK], W2[FK];
    auto load_w2 = [&]() {                                           // -(swept) = inverse; this lane's 24 columns of row r2
#pragma unroll
      for (int k = 0; k < FK; ++k) {
        const int col = FK * h2 + k;
        W2[k] = (r2 < n && col < n) ? -(r2 >= col ? Mp[r2 * SLD + col] : Mp[col * SLD + r2]) : 0.0;
      }
      __syncthreads();
    };
    auto load_w1 = [&]() {
      // CCs'[(j,c),(i,rr)] = G_{i-j}[S_rr][c]; S = {2,3,4,5,6,8}: compile-time offsets off one base
      constexpr int SR[6] = {2, 3, 4, 5, 6, 8};
      const int j1 = r1 >= 0 ? r1 / 3 : 0, c1e = r1 >= 0 ? r1 - 3 * j1 : 0;
      const double *g1 = G + (i01 - j1) * 27 + c1e;                 // + (k/6)*27 + SR[k%6]*3
      const int lim1 = r1 >= 0 ? N - i01 : 0;                       // valid while k/6 < lim1
#pragma unroll
      for (int k = 0; k < FK; ++k) W1[k] = (k / 6 < lim1) ? g1[(k / 6) * 27 + SR[k % 6] * 3] : 0.0;
    };
    double rho = a.s.rho;
    if (!(rho > 0.0)) {   // automatic: balance the two terms of P + rho A'A (our QP is not Ruiz-scaled as OSQP's would be)
      double tr[2] = {xown ? Pg[tri(tid, tid)] : 0.0, xown ? Ag[tri(tid, tid)] : 0.0};
      const bool sums[2] = {true, true};
      block_reduce<2>(tr, sums, red);
      rho = fmin(fmax(sqrt(tr[0] / tr[1]), 1e-6), 1e6);
    }
    bool ok = true;
    ok = sweep_inverse(Mp, cvec, Pg, Ag, rho, sigma, n, np) && ok;
    load_w2();
    load_w1();

    int it = 0;
    double rp = INFINITY, rd = INFINITY;
    bool converged = false, infeasible = false;
    bool done = !ok || a.s.max_iter <= 0;
    // w = rho z - y of the start point (all zero)
    if (rown) { double *wdst = srow ? ws + tid : (tid < ms + n ? wc + (tid - ms) : wr + (tid - ms - n)); *wdst = 0.0; }
    __syncthreads();
#ifdef F16_EXP_STAMPM
    unsigned long long tS[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = __builtin_amdgcn_s_memtime();
#define MSTAMP(i) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t1 = __builtin_amdgcn_s_memtime(); tS[i] += t1 - t0; t0 = t1; }
#else
#define MSTAMP(i)
#endif
    while (!done) {
      ++it;
      // stage 1 partials: (CCs' w_s) row segments
      p1[tid] = r1 >= 0 ? dot24(W1, ws + 6 * i01) : 0.0;
      MSTAMP(0)
      __syncthreads();
      MSTAMP(1)
      if (xown) {
        double t = 0.0;
#pragma unroll
        for (int s = 0; s < 8; ++s) t += (s < c1) ? p1[f1 + s] : 0.0;
        rhs[tid] = sigma * xs - qe + (t + wc[tid] + (wr[tid] - (tid + 3 < n ? wr[tid + 3] : 0.0)));
      }
      MSTAMP(2)
      __syncthreads();
      MSTAMP(3)
      // stage 2: x~ = Minv rhs, four lanes per row
      {
        const double part = quad_sum(dot24(W2, rhs + FK * h2));
        if (h2 == 0 && r2 < n) xt[r2] = part;
      }
      MSTAMP(4)
      __syncthreads();
      MSTAMP(5)
      // stage 3 partials: (CCs x~) row segments; x relaxation
      p3[tid] = r3 >= 0 ? dot24_lds(w3run, xt + 3 * j03) : 0.0;
      if (xown) xs = alpha * xt[tid] + (1 - alpha) * xs;
      MSTAMP(6)
      __syncthreads();
      MSTAMP(7)
      if (rown) {
        double zt;
        if (srow) {
          zt = 0.0;
#pragma unroll
          for (int s = 0; s < 4; ++s) zt += (s < c3) ? p3[f3 + s] : 0.0;
        }
        else if (tid < ms + n) zt = xt[tid - ms];
        else { const int k = tid - ms - n; zt = xt[k] - (k >= 3 ? xt[k - 3] : 0.0); }
        const double zr = alpha * zt + (1 - alpha) * z;
        const double zn = fmin(fmax(zr + y / rho, lo), hi);
        dy = rho * (zr - zn);
        y = y + dy;
        z = zn;
      }
      const bool check = (it % a.s.check_every == 0) || it >= a.s.max_iter;
      if (check) {
        // ---- residuals (OSQP termination test): A x, P x, A' y
        if (rown) { double *d = srow ? ys + tid : (tid < ms + n ? yc + (tid - ms) : yr + (tid - ms - n)); *d = y; }
        if (xown) xb[tid] = xs;
        __syncthreads();
        p3[tid] = r3 >= 0 ? dot24_lds(w3run, xb + 3 * j03) : 0.0;
        p1[tid] = r1 >= 0 ? dot24(W1, ys + 6 * i01) : 0.0;
        {
          double acc = 0.0;                                    // P x from the packed workspace copy of P
          if (r2 < n) {
#pragma unroll
            for (int k = 0; k < FK; ++k) {
              const int col = FK * h2 + k;
              if (col < n) acc += (r2 >= col ? Pg[tri(r2, col)] : Pg[tri(col, r2)]) * xb[col];
            }
          }
          acc = quad_sum(acc);
          if (h2 == 0 && r2 < n) pxv[r2] = acc;
        }
        __syncthreads();
        double v[7] = {0, 0, 0, 0, 0, 0, 0};                    // r1, |Ax|, |z|, r2, |Px|, |A'y|, |q|
        if (rown) {
          double ax;
          if (srow) { ax = 0.0; for (int s = 0; s < c3; ++s) ax += p3[f3 + s]; }
          else if (tid < ms + n) ax = xb[tid - ms];
          else { const int k = tid - ms - n; ax = xb[k] - (k >= 3 ? xb[k - 3] : 0.0); }
          v[0] = fabs(ax - z); v[1] = fabs(ax); v[2] = fabs(z);
        }
        if (xown) {
          double t = 0.0;
          for (int s = 0; s < c1; ++s) t += p1[f1 + s];
          const double aty = t + yc[tid] + (yr[tid] - (tid + 3 < n ? yr[tid + 3] : 0.0));
          v[3] = fabs(pxv[tid] + qe + aty); v[4] = fabs(pxv[tid]); v[5] = fabs(aty); v[6] = fabs(qe);
        }
        const bool allmax[7] = {false, false, false, false, false, false, false};
        block_reduce<7>(v, allmax, red);
        rp = v[0]; rd = v[3];
        const double np_ = fmax(v[1], v[2]), nd_ = fmax(fmax(v[4], v[5]), v[6]);
        if (rp <= a.s.eps_abs + a.s.eps_rel * np_ && rd <= a.s.eps_abs + a.s.eps_rel * nd_) { done = true; converged = true; }
        else {
          // OSQP primal-infeasibility certificate on dy
          double u[2] = {rown ? fabs(dy) : 0.0, rown ? hi * fmax(dy, 0.0) + lo * fmin(dy, 0.0) : 0.0};
          const bool kinds[2] = {false, true};
          block_reduce<2>(u, kinds, red);
          const double ndy = u[0], supp = u[1];
          if (ndy > a.s.eps_prim_inf && supp < -a.s.eps_prim_inf * ndy) {
            if (rown) { double *d = srow ? ys + tid : (tid < ms + n ? yc + (tid - ms) : yr + (tid - ms - n)); *d = dy; }
            __syncthreads();
            p1[tid] = r1 >= 0 ? dot24(W1, ys + 6 * i01) : 0.0;
            __syncthreads();
            double w[1] = {0.0};
            if (xown) {
              double t = 0.0;
              for (int s = 0; s < c1; ++s) t += p1[f1 + s];
              w[0] = fabs(t + yc[tid] + (yr[tid] - (tid + 3 < n ? yr[tid + 3] : 0.0)));
            }
            const bool km[1] = {false};
            block_reduce<1>(w, km, red);
            if (w[0] < a.s.eps_prim_inf * ndy) { done = true; infeasible = true; }
          }
          if (!done) {
            if (it >= a.s.max_iter) done = true;
            else if (a.s.adaptive_rho && it % a.s.rho_every == 0) {
              double nw = rho * sqrt((rp / fmax(np_, 1e-10)) / fmax(rd / fmax(nd_, 1e-10), 1e-10));
              nw = fmin(fmax(nw, 1e-6), 1e6);
              if (nw > 5 * rho || nw < rho / 5) {
                rho = nw;
                ok = sweep_inverse(Mp, cvec, Pg, Ag, rho, sigma, n, np) && ok;
                load_w2();
                load_w1();
                if (!ok) done = true;
              }
            }
          }
        }
      }
      // w = rho z - y for the next iteration
      if (rown) { double *d = srow ? ws + tid : (tid < ms + n ? wc + (tid - ms) : wr + (tid - ms - n)); *d = rho * z - y; }
      MSTAMP(8)
      __syncthreads();
      MSTAMP(9)
    }
#ifdef F16_EXP_STAMPM
    if ((tid & 63) == 0 && a.useq && b == 0) for (int i = 0; i < 10; ++i) a.useq[(10 * (tid >> 6) + i) * a.ld + 1] = (double)tS[i] / it;
#endif
    // res.x[0:3] (env.py:424); OSQP hands back NaN for a problem it certifies infeasible
    if (tid < 3) a.ucmd[tid * a.ld + b] = infeasible ? NAN : xs;
    if (a.useq && xown) a.useq[tid * a.ld + b] = infeasible ? NAN : xs;
    if (tid == 0) {
      if (a.info) {
#ifdef F16_EXP_STAMP
        rp = cvec[0]; rd = cvec[1];
#endif
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

int mpc_fast_solve_launch(f16_ctx *ctx, const MpcArgs &a, void *stream) {
  if (a.N < 1 || a.N > FAST_MAXN) return set_error(F16_EINVAL, "fast MPC solver needs 1 <= N <= 32");
  static std::mutex mu;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!ctx->d_fast_desc || ctx->fast_desc_N != a.N) {
      FastDesc h;
      if (build_desc(a.N, h)) return set_error(F16_EINVAL, "internal: lane map does not fit");
      if (!ctx->d_fast_desc) {
        if (int rc = hip_check(hipMalloc(&ctx->d_fast_desc, sizeof(FastDesc)), "hipMalloc lane map")) return rc;
      } else {
        (void)hipStreamSynchronize((hipStream_t)stream);   // a previous launch may still read the old map
      }
      if (int rc = hip_check(hipMemcpy(ctx->d_fast_desc, &h, sizeof(FastDesc), hipMemcpyHostToDevice), "upload lane map")) return rc;
      ctx->fast_desc_N = a.N;
    }
  }
  const unsigned grid = (unsigned)(a.B < 512 ? a.B : 512);
  hipLaunchKernelGGL(k_mpc_fast, dim3(grid), dim3(FT), 0, (hipStream_t)stream, a, (const FastDesc *)ctx->d_fast_desc);
  return hip_check(hipGetLastError(), "f16_mpc_batch solve launch");
}

}  // namespace f16
