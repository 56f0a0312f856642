/* oracle/f16_mpc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C restatement of the reference's control chain for ONE aircraft at a time (OpenMP over aircraft in the batched
 * entry): linearise (env.py:294-342, via f16_oracle.c) -> cont2discrete zoh (env.py:50,351) -> dlqr / DARE
 * (utils.py:219-245) -> setup_OSQP (utils.py:21-167, dense, reference row order) -> the solve the reference delegates
 * to the absent `osqp` package (env.py:420-424).  Two jobs:
 *   (1) bench.py's cpu_baseline for MPC solves/s (kind "port"), single thread and all cores;
 *   (2) a second, independent checker next to oracle/mpc_oracle.py (numpy/scipy): tests/test_oracle_vs_golden.py
 *       requires both to agree (same iteration counts, x to 1e-9) on the golden QPs.
 * The ADMM follows oracle/mpc_oracle.py rule for rule:
 *   mode 0  admm_osqp_style: the same rules without scaling, rows without bounds dropped, rho0 = settings.rho or
 *           2 sqrt(tr P / tr A'A) (the builder's opt-in settings)
 *   mode 1  admm_osqp(drop_unbounded_rows = 0): OSQP's published algorithm -- Ruiz equilibration, rho vector, unscaled
 *           termination test, scaled rho estimate
 *   mode 2  the same with the unbounded rows left out of the iteration after the equilibration (what the HIP kernels do)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "f16_oracle.h"

typedef struct f16o_qp_settings {
  double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf;
  int max_iter, check_every, rho_every, adaptive_rho, scaling;
} f16o_qp_settings;

#define OSQP_INFTY 1e30
#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4
#define RHO_MIN 1e-6
#define RHO_MAX 1e6
#define RHO_TOL 1e-4
#define RHO_EQ 1e3

static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }

/* ---- small dense helpers (row-major) ------------------------------------------------------------------------------ */
static void matmul(const double *A, const double *B, double *C, int m, int k, int n) { /* C = A B */
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
      for (int p = 0; p < k; ++p) s += A[i * k + p] * B[p * n + j];
      C[i * n + j] = s;
    }
}
/* in-place inverse by Gauss-Jordan with partial pivoting; returns 0 on success */
static int invert(double *M, int n) {
  double *W = (double *)malloc(sizeof(double) * n * 2 * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 2 * n; ++j) W[i * 2 * n + j] = j < n ? M[i * n + j] : (j - n == i ? 1.0 : 0.0);
  for (int p = 0; p < n; ++p) {
    int piv = p;
    for (int r = p + 1; r < n; ++r)
      if (fabs(W[r * 2 * n + p]) > fabs(W[piv * 2 * n + p])) piv = r;
    if (W[piv * 2 * n + p] == 0.0) { free(W); return -1; }
    if (piv != p)
      for (int j = 0; j < 2 * n; ++j) { double t = W[p * 2 * n + j]; W[p * 2 * n + j] = W[piv * 2 * n + j]; W[piv * 2 * n + j] = t; }
    const double d = W[p * 2 * n + p];
    for (int j = 0; j < 2 * n; ++j) W[p * 2 * n + j] /= d;
    for (int i = 0; i < n; ++i)
      if (i != p) {
        const double f = W[i * 2 * n + p];
        if (f != 0.0)
          for (int j = 0; j < 2 * n; ++j) W[i * 2 * n + j] -= f * W[p * 2 * n + j];
      }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) M[i * n + j] = W[i * 2 * n + n + j];
  free(W);
  return 0;
}

/* scipy.signal.cont2discrete(zoh) = expm([[A,B],[0,0]] dt) top blocks (env.py:50,351): scaling-and-squaring Taylor on the
 * ns x (ns+ni) block (agrees with scipy's Pade expm to ~1e-16 on these models: tests) */
void f16o_c2d(const double *A, const double *B, int ns, int ni, double dt, double *Ad, double *Bd) {
  const int nw = ns + ni;
  double nrm = 0.0;
  for (int i = 0; i < ns; ++i) {
    double rs = 0.0;
    for (int j = 0; j < ns; ++j) rs += fabs(A[i * ns + j]);
    for (int j = 0; j < ni; ++j) rs += fabs(B[i * ni + j]);
    nrm = dmax(nrm, rs * fabs(dt));
  }
  int s = 0;
  if (nrm > 0.5) s = (int)ceil(log2(nrm / 0.5));
  if (s > 40) s = 40;
  const double sc = ldexp(dt, -s);
  double *X = (double *)malloc(sizeof(double) * (ns * ns + 3 * ns * nw));
  double *T = X + ns * ns, *Tn = T + ns * nw, *EF = Tn + ns * nw;
  for (int i = 0; i < ns; ++i) {
    for (int j = 0; j < ns; ++j) X[i * ns + j] = A[i * ns + j] * sc;
    for (int j = 0; j < nw; ++j) {
      const double v = j < ns ? A[i * ns + j] * sc : B[i * ni + (j - ns)] * sc;
      T[i * nw + j] = v;
      EF[i * nw + j] = v + (j == i ? 1.0 : 0.0);
    }
  }
  for (int k = 2; k <= 18; ++k) {
    matmul(X, T, Tn, ns, ns, nw);
    for (int e = 0; e < ns * nw; ++e) { T[e] = Tn[e] / k; EF[e] += T[e]; }
  }
  for (int q = 0; q < s; ++q) {
    for (int i = 0; i < ns; ++i)
      for (int j = 0; j < ns; ++j) X[i * ns + j] = EF[i * nw + j];
    matmul(X, EF, Tn, ns, ns, nw);
    for (int i = 0; i < ns; ++i)
      for (int j = 0; j < nw; ++j) EF[i * nw + j] = Tn[i * nw + j] + (j >= ns ? EF[i * nw + j] : 0.0);
  }
  for (int i = 0; i < ns; ++i) {
    for (int j = 0; j < ns; ++j) Ad[i * ns + j] = EF[i * nw + j];
    for (int j = 0; j < ni; ++j) Bd[i * ni + j] = EF[i * nw + ns + j];
  }
  free(X);
}

/* scipy.linalg.solve_discrete_are(A, B, Q, I) (utils.py:242) by the structure-preserving doubling algorithm:
 * A0 = A, G0 = B B', H0 = Q; W = (I + G H)^-1; A+ = A W A; G+ = G + A W G A'; H+ = H + A' H W A; H -> X.  n = 9, ni = 3. */
int f16o_dare(const double *A0, const double *B, const double *Q, int n, int ni, double *X) {
  const int nn = n * n;
  double *w = (double *)malloc(sizeof(double) * nn * 9);
  double *A = w, *G = A + nn, *H = G + nn, *W = H + nn, *T1 = W + nn, *T2 = T1 + nn, *T3 = T2 + nn, *At = T3 + nn, *T4 = At + nn;
  memcpy(A, A0, sizeof(double) * nn);
  memcpy(H, Q, sizeof(double) * nn);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
      for (int k = 0; k < ni; ++k) s += B[i * ni + k] * B[j * ni + k];
      G[i * n + j] = s;
    }
  int it = 0, rc = 0;
  for (; it < 60; ++it) {
    matmul(G, H, W, n, n, n);
    for (int i = 0; i < n; ++i) W[i * n + i] += 1.0;
    if (invert(W, n)) { rc = -1; break; }
    matmul(A, W, T1, n, n, n);                 /* A W */
    matmul(T1, G, T2, n, n, n);                /* A W G */
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) At[i * n + j] = A[j * n + i];
    matmul(T2, At, T3, n, n, n);               /* A W G A' */
    for (int e = 0; e < nn; ++e) G[e] += T3[e];
    matmul(H, W, T2, n, n, n);                 /* H W */
    matmul(T2, A, T3, n, n, n);                /* H W A */
    matmul(At, T3, T4, n, n, n);               /* A' H W A */
    double dm = 0.0, hm = 0.0;
    for (int e = 0; e < nn; ++e) { H[e] += T4[e]; dm = dmax(dm, fabs(T4[e])); hm = dmax(hm, fabs(H[e])); }
    matmul(T1, A, T2, n, n, n);                /* A W A */
    memcpy(A, T2, sizeof(double) * nn);
    if (dm <= 1e-16 * hm) { ++it; break; }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) X[i * n + j] = 0.5 * (H[i * n + j] + H[j * n + i]);
  free(w);
  return rc ? rc : (it >= 60 ? 1 : 0);
}

/* utils.py:21-167 setup_OSQP through env.py:373-416 for one aircraft: x_full[18], (Ad 9x9, Bd 9x3, Cd 9x9), horizon N.
 * Outputs in the reference's form: P [n][n], q [n], A [15N][n], l, u [15N] (+-INFINITY kept), n = 3N. */
static const int MPC_X_IDX[9] = {3, 4, 7, 8, 9, 10, 11, 17, 16};
static const double X_LB[9] = {-INFINITY, -INFINITY, -20., -30., -300., -100., -50., -INFINITY, 0.};
static const double X_UB[9] = {INFINITY, INFINITY, 90., 30., 300., 100., 50., INFINITY, 25.};
static const double U_LB[3] = {-25., -21.5, -30.}, U_UB[3] = {25., 21.5, 30.};
static const double R_LB[3] = {-60., -80., -120.}, R_UB[3] = {60., 80., 120.};

int f16o_mpc_qp(const double *x_full, const double *Ad, const double *Bd, const double *Cd, int N, double dt,
                const double *dem, double *P, double *q, double *A, double *l, double *u) {
  const int n = 3 * N, ms = 9 * N, m = 15 * N;
  double x9[9], xref[9], act[3], Q[81], Qb[81];
  for (int k = 0; k < 9; ++k) { x9[k] = x_full[MPC_X_IDX[k]]; xref[k] = x9[k]; }
  for (int k = 0; k < 3; ++k) { act[k] = x_full[13 + k]; xref[5 + k] = dem ? dem[k] : 0.0; }      /* env.py:383 (quirk 8-Q.4) */
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 9; ++j) {
      double s = 0.0;
      for (int k = 0; k < 9; ++k) s += Cd[k * 9 + i] * Cd[k * 9 + j];
      Q[i * 9 + j] = s;                                                                           /* Q = C'C, R = I */
    }
  /* K = -dlqr, Q_bar = dlyap((A+BK)', Q + K'RK): for the LQR gain that Lyapunov equation IS the DARE (utils.py:96-100) */
  const int rc = f16o_dare(Ad, Bd, Q, 9, 3, Qb);
  /* G_k = A^k B, MM x: pred_i = A^(i+1) x (utils.py:171-197) */
  double *G = (double *)malloc(sizeof(double) * (27 * N + 9 * N + (size_t)ms * n + (size_t)ms * n));
  double *pred = G + 27 * N, *CC = pred + 9 * N, *QC = CC + (size_t)ms * n;
  memcpy(G, Bd, sizeof(double) * 27);
  for (int k = 1; k < N; ++k) matmul(Ad, G + 27 * (k - 1), G + 27 * k, 9, 9, 3);
  matmul(Ad, x9, pred, 9, 9, 1);
  for (int i = 1; i < N; ++i) matmul(Ad, pred + 9 * (i - 1), pred + 9 * i, 9, 9, 1);
  memset(CC, 0, sizeof(double) * (size_t)ms * n);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j <= i; ++j)
      for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 3; ++c) CC[(size_t)(9 * i + r) * n + 3 * j + c] = G[(i - j) * 27 + r * 3 + c];
  /* QC = QQ CC ; P = 2 (CC' QC + I) ; q = -2 CC' QQ (x_ref - MM x) */
  for (int i = 0; i < N; ++i) {
    const double *Qi = i == N - 1 ? Qb : Q;
    for (int r = 0; r < 9; ++r)
      for (int col = 0; col < 3 * (i + 1); ++col) {
        double s = 0.0;
        for (int p = 0; p < 9; ++p) s += Qi[r * 9 + p] * CC[(size_t)(9 * i + p) * n + col];
        QC[(size_t)(9 * i + r) * n + col] = s;
      }
    for (int r = 0; r < 9; ++r)
      for (int col = 3 * (i + 1); col < n; ++col) QC[(size_t)(9 * i + r) * n + col] = 0.0;
  }
  for (int a = 0; a < n; ++a)
    for (int b = 0; b <= a; ++b) {
      double s = 0.0;
      for (int row = 9 * (a / 3); row < ms; ++row) s += CC[(size_t)row * n + a] * QC[(size_t)row * n + b];
      const double v = 2.0 * (s + (a == b ? 1.0 : 0.0));
      P[a * n + b] = v; P[b * n + a] = v;
    }
  for (int a = 0; a < n; ++a) {
    double s = 0.0;
    for (int i = a / 3; i < N; ++i)
      for (int r = 0; r < 9; ++r) s += (xref[r] - pred[9 * i + r]) * QC[(size_t)(9 * i + r) * n + a];
    q[a] = -2.0 * s;
  }
  memset(A, 0, sizeof(double) * (size_t)m * n);
  memcpy(A, CC, sizeof(double) * (size_t)ms * n);
  for (int k = 0; k < n; ++k) {
    A[(size_t)(ms + k) * n + k] = 1.0;
    A[(size_t)(ms + n + k) * n + k] = 1.0;
    if (k >= 3) A[(size_t)(ms + n + k) * n + k - 3] = -1.0;
  }
  for (int i = 0; i < N; ++i)
    for (int r = 0; r < 9; ++r) { l[9 * i + r] = X_LB[r] - pred[9 * i + r]; u[9 * i + r] = X_UB[r] - pred[9 * i + r]; }
  for (int k = 0; k < n; ++k) {
    l[ms + k] = U_LB[k % 3]; u[ms + k] = U_UB[k % 3];
    if (k < 3) { l[ms + n + k] = act[k] + R_LB[k] * dt; u[ms + n + k] = act[k] + R_UB[k] * dt; }
    else { l[ms + n + k] = R_LB[k % 3]; u[ms + n + k] = R_UB[k % 3]; }                        /* quirk 8-Q.5: not * dt */
  }
  free(G);
  return rc;
}

/* ---- dense Cholesky (lower, in place) + solve ----------------------------------------------------------------------- */
static int chol(double *K, int n) {
  for (int j = 0; j < n; ++j) {
    double d = K[j * n + j];
    for (int k = 0; k < j; ++k) d -= K[j * n + k] * K[j * n + k];
    if (!(d > 0.0)) return -1;
    d = sqrt(d);
    K[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = K[i * n + j];
      for (int k = 0; k < j; ++k) s -= K[i * n + k] * K[j * n + k];
      K[i * n + j] = s / d;
    }
  }
  return 0;
}
static void chol_solve(const double *L, int n, double *b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * n + k] * b[k];
    b[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k];
    b[i] = s / L[i * n + i];
  }
}
static double limit_scaling(double v) {
  v = v < MIN_SCALING ? 1.0 : v;
  return v > MAX_SCALING ? MAX_SCALING : v;
}

/* The solve.  P [n][n] symmetric, A [m][n] dense, l/u [m] (+-INFINITY allowed).  mode: see the file header.
 * x_out [n]; info[4] = iterations, r_prim, r_dual (unscaled), rho.  Returns 0 converged, 1 max_iter, 2 primal infeasible
 * (x_out = NaN), -1 factorisation failed. */
int f16o_admm(int n, int m_all, const double *P_in, const double *q_in, const double *A_in, const double *l_in,
              const double *u_in, const f16o_qp_settings *o, int mode, double *x_out, double *info) {
  double *buf = (double *)malloc(sizeof(double) * ((size_t)2 * n * n + (size_t)m_all * n + 8 * (size_t)m_all + 12 * (size_t)n));
  double *P = buf, *K = P + (size_t)n * n, *A = K + (size_t)n * n;
  double *l = A + (size_t)m_all * n, *u = l + m_all, *E = u + m_all, *z = E + m_all, *y = z + m_all, *dy = y + m_all, *rv = dy + m_all,
         *tm = rv + m_all;
  double *q = tm + m_all, *D = q + n, *x = D + n, *xt = x + n, *tn = xt + n, *Px = tn + n, *Aty = Px + n, *cn = Aty + n;
  int *ncol = (int *)malloc(sizeof(int) * m_all);
  memcpy(P, P_in, sizeof(double) * (size_t)n * n);
  memcpy(q, q_in, sizeof(double) * n);
  memcpy(A, A_in, sizeof(double) * (size_t)m_all * n);
  int m = m_all;
  for (int i = 0; i < m; ++i) {
    l[i] = dmax(l_in[i], -OSQP_INFTY);
    u[i] = dmin(u_in[i], OSQP_INFTY);
    E[i] = 1.0;
  }
  for (int j = 0; j < n; ++j) D[j] = 1.0;
  double c = 1.0;
  if (mode != 0 && o->scaling > 0) {                      /* scaling.c:scale_data */
    for (int pass = 0; pass < o->scaling; ++pass) {
      for (int j = 0; j < n; ++j) cn[j] = 0.0;
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) cn[j] = dmax(cn[j], fabs(P[i * n + j]));
      for (int i = 0; i < m; ++i) {
        double rn = 0.0;
        for (int j = 0; j < n; ++j) { const double v = fabs(A[(size_t)i * n + j]); cn[j] = dmax(cn[j], v); rn = dmax(rn, v); }
        tm[i] = 1.0 / sqrt(limit_scaling(rn));
      }
      for (int j = 0; j < n; ++j) cn[j] = 1.0 / sqrt(limit_scaling(cn[j]));
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) P[i * n + j] = cn[i] * P[i * n + j] * cn[j];
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = tm[i] * A[(size_t)i * n + j] * cn[j];
      for (int j = 0; j < n; ++j) { q[j] *= cn[j]; D[j] *= cn[j]; }
      for (int i = 0; i < m; ++i) E[i] *= tm[i];
      double mean = 0.0, qn = 0.0;
      for (int j = 0; j < n; ++j) {
        double colmax = 0.0;
        for (int i = 0; i < n; ++i) colmax = dmax(colmax, fabs(P[i * n + j]));
        mean += colmax;
        qn = dmax(qn, fabs(q[j]));
      }
      const double ct = 1.0 / dmax(limit_scaling(mean / n), limit_scaling(qn));
      for (int e = 0; e < n * n; ++e) P[e] *= ct;
      for (int j = 0; j < n; ++j) q[j] *= ct;
      c *= ct;
    }
    for (int i = 0; i < m; ++i) { l[i] *= E[i]; u[i] *= E[i]; }
  }
  /* rows without bounds: mode 0 and 2 drop them (compacting A, l, u, E), mode 1 carries them with rho_min */
  int *loose = (int *)malloc(sizeof(int) * m_all);
  {
    int k = 0;
    for (int i = 0; i < m; ++i) {
      const int lo = l[i] < -OSQP_INFTY * MIN_SCALING && u[i] > OSQP_INFTY * MIN_SCALING;
      if (lo && mode != 1) continue;
      if (k != i) { memcpy(A + (size_t)k * n, A + (size_t)i * n, sizeof(double) * n); l[k] = l[i]; u[k] = u[i]; E[k] = E[i]; }
      loose[k] = lo;
      ++k;
    }
    m = k;
  }
  for (int i = 0; i < m; ++i) {                           /* column extent of each row (block lower-triangular CC) */
    int nc = 0;
    for (int j = 0; j < n; ++j) if (A[(size_t)i * n + j] != 0.0) nc = j + 1;
    ncol[i] = nc;
  }
  double rho = o->rho;
  if (mode == 0 && !(rho > 0.0)) {                        /* the builder's start value: 2 sqrt(tr P / tr A'A) */
    double tp = 0.0, ta = 0.0;
    for (int j = 0; j < n; ++j) tp += P[j * n + j];
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < ncol[i]; ++j) ta += A[(size_t)i * n + j] * A[(size_t)i * n + j];
    rho = dmin(dmax(2.0 * sqrt(tp / ta), RHO_MIN), RHO_MAX);
  }
  const double sigma = o->sigma, alpha = o->alpha;
  int rc_f = 0;
#define SET_RV()                                                                                                      \
  for (int i = 0; i < m; ++i) rv[i] = loose[i] ? RHO_MIN : ((u[i] - l[i]) < RHO_TOL ? RHO_EQ * rho : rho)
#define FACTOR()                                                                                                      \
  do {                                                                                                                \
    for (int a = 0; a < n; ++a)                                                                                       \
      for (int b = 0; b <= a; ++b) K[a * n + b] = P[a * n + b] + (a == b ? sigma : 0.0);                              \
    for (int i = 0; i < m; ++i) {                                                                                     \
      const double *ai = A + (size_t)i * n;                                                                           \
      for (int a = 0; a < ncol[i]; ++a) {                                                                             \
        const double f = rv[i] * ai[a];                                                                               \
        if (f != 0.0)                                                                                                 \
          for (int b = 0; b <= a; ++b) K[a * n + b] += f * ai[b];                                                     \
      }                                                                                                               \
    }                                                                                                                 \
    rc_f = chol(K, n);                                                                                                \
  } while (0)
  SET_RV();
  FACTOR();
  for (int j = 0; j < n; ++j) x[j] = 0.0;
  for (int i = 0; i < m; ++i) { z[i] = 0.0; y[i] = 0.0; dy[i] = 0.0; }
  int it = 0, status = 1;
  double rp = INFINITY, rd = INFINITY;
  const double cinv = 1.0 / c;
  while (!rc_f && it < o->max_iter) {
    ++it;
    for (int j = 0; j < n; ++j) xt[j] = sigma * x[j] - q[j];
    for (int i = 0; i < m; ++i) {
      const double w = rv[i] * z[i] - y[i];
      const double *ai = A + (size_t)i * n;
      if (w != 0.0)
        for (int j = 0; j < ncol[i]; ++j) xt[j] += ai[j] * w;
    }
    chol_solve(K, n, xt);
    for (int i = 0; i < m; ++i) {
      const double *ai = A + (size_t)i * n;
      double zt = 0.0;
      for (int j = 0; j < ncol[i]; ++j) zt += ai[j] * xt[j];
      const double zr = alpha * zt + (1 - alpha) * z[i];
      const double zn = dmin(dmax(zr + y[i] / rv[i], l[i]), u[i]);
      dy[i] = rv[i] * (zr - zn);
      y[i] += dy[i];
      z[i] = zn;
    }
    for (int j = 0; j < n; ++j) x[j] = alpha * xt[j] + (1 - alpha) * x[j];
    if (it % o->check_every == 0 || it >= o->max_iter) {
      for (int j = 0; j < n; ++j) { Aty[j] = 0.0; double s = 0.0; for (int k = 0; k < n; ++k) s += P[j * n + k] * x[k]; Px[j] = s; }
      double r1 = 0, nAx = 0, nz = 0, r1s = 0, nAxs = 0, nzs = 0;
      for (int i = 0; i < m; ++i) {
        const double *ai = A + (size_t)i * n;
        double ax = 0.0;
        for (int j = 0; j < ncol[i]; ++j) { ax += ai[j] * x[j]; Aty[j] += ai[j] * y[i]; }
        const double ei = 1.0 / E[i];
        r1 = dmax(r1, fabs(ei * (ax - z[i]))); nAx = dmax(nAx, fabs(ei * ax)); nz = dmax(nz, fabs(ei * z[i]));
        r1s = dmax(r1s, fabs(ax - z[i])); nAxs = dmax(nAxs, fabs(ax)); nzs = dmax(nzs, fabs(z[i]));
      }
      double r2 = 0, nPx = 0, nAty = 0, nq = 0, r2s = 0, nPxs = 0, nAtys = 0, nqs = 0;
      for (int j = 0; j < n; ++j) {
        const double di = 1.0 / D[j], rr = Px[j] + q[j] + Aty[j];
        r2 = dmax(r2, fabs(di * rr)); nPx = dmax(nPx, fabs(di * Px[j])); nAty = dmax(nAty, fabs(di * Aty[j])); nq = dmax(nq, fabs(di * q[j]));
        r2s = dmax(r2s, fabs(rr)); nPxs = dmax(nPxs, fabs(Px[j])); nAtys = dmax(nAtys, fabs(Aty[j])); nqs = dmax(nqs, fabs(q[j]));
      }
      rp = r1; rd = cinv * r2;
      const double eps_p = o->eps_abs + o->eps_rel * dmax(nAx, nz);
      const double eps_d = o->eps_abs + o->eps_rel * cinv * dmax(dmax(nPx, nAty), nq);
      if (rp < eps_p && rd < eps_d) { status = 0; break; }
      /* primal infeasibility certificate on dy */
      double ndy = 0.0, supp = 0.0;
      for (int i = 0; i < m; ++i) {
        ndy = dmax(ndy, fabs(E[i] * dy[i]));
        supp += u[i] * dmax(dy[i], 0.0) + l[i] * dmin(dy[i], 0.0);
      }
      if (ndy > o->eps_prim_inf && supp < -o->eps_prim_inf * ndy) {
        for (int j = 0; j < n; ++j) tn[j] = 0.0;
        for (int i = 0; i < m; ++i) {
          const double *ai = A + (size_t)i * n;
          for (int j = 0; j < ncol[i]; ++j) tn[j] += ai[j] * dy[i];
        }
        double nat = 0.0;
        for (int j = 0; j < n; ++j) nat = dmax(nat, fabs(tn[j] / D[j]));
        if (nat < o->eps_prim_inf * ndy) { status = 2; break; }
      }
      if (o->adaptive_rho && it % o->rho_every == 0 && it < o->max_iter) {
        const double pr = r1s / (dmax(nzs, nAxs) + 1e-10), dr = r2s / (dmax(dmax(nqs, nAtys), nPxs) + 1e-10);
        const double nw = dmin(dmax(rho * sqrt(pr / (dr + 1e-10)), RHO_MIN), RHO_MAX);      /* compute_rho_estimate */
        if (nw > 5 * rho || nw < rho / 5) { rho = nw; SET_RV(); FACTOR(); }
      }
    }
  }
  if (rc_f) status = -1;
  for (int j = 0; j < n; ++j) x_out[j] = status == 2 ? NAN : D[j] * x[j];
  if (info) { info[0] = it; info[1] = rp; info[2] = rd; info[3] = rho; }
  free(buf); free(ncol); free(loose);
  return status;
}

void f16o_qp_default_settings(f16o_qp_settings *s, int mode) {
  s->rho = mode == 0 ? 0.0 : 0.1; s->sigma = 1e-6; s->alpha = 1.6; s->eps_abs = 1e-3; s->eps_rel = 1e-3; s->eps_prim_inf = 1e-4;
  s->max_iter = 40000; s->check_every = 25; s->rho_every = 100; s->adaptive_rho = 1; s->scaling = mode == 0 ? 0 : 10;
}

/* env.py:373-424 for B aircraft, everything on the CPU: linearise at each aircraft's own state (env.py:49-50), ZOH, QP, solve.
 * x [B][18]; dem[3] or NULL; u_cmd [B][3]; iters/status [B] may be NULL. */
void f16o_mpc_batch(const double *x, long B, int N, double dt, double xcg, int fi_flag, const double *dem,
                    const f16o_qp_settings *s, int mode, double *u_cmd, int *iters, int *status, int nthreads) {
  const int n = 3 * N, m = 15 * N;
#pragma omp parallel num_threads(nthreads > 1 ? nthreads : 1)
  {
    double *P = (double *)malloc(sizeof(double) * ((size_t)n * n + n + (size_t)m * n + 2 * m + n));
    double *q = P + (size_t)n * n, *A = q + n, *l = A + (size_t)m * n, *u = l + m, *xo = u + m;
#pragma omp for schedule(dynamic, 1)
    for (long b = 0; b < B; ++b) {
      const double *xf = x + 18 * b;
      double x9[9], u3[3], Ac[81], Bc[27], Cc[81], Dc[27], Ad[81], Bd[27], info[4];
      for (int k = 0; k < 9; ++k) x9[k] = xf[MPC_X_IDX[k]];
      for (int k = 0; k < 3; ++k) u3[k] = xf[13 + k];
      f16o_linearise_na(xf, x9, u3, 1e-5, Ac, Bc, Cc, Dc, fi_flag, xcg);
      f16o_c2d(Ac, Bc, 9, 3, dt, Ad, Bd);
      f16o_mpc_qp(xf, Ad, Bd, Cc, N, dt, dem, P, q, A, l, u);
      const int st = f16o_admm(n, m, P, q, A, l, u, s, mode, xo, info);
      for (int k = 0; k < 3; ++k) u_cmd[3 * b + k] = xo[k];
      if (iters) iters[b] = (int)info[0];
      if (status) status[b] = st;
    }
    free(P);
  }
}

/* The reference's closed MPC loop (test_env.py:480-495: cmd = _calc_MPC_action(p, q, r, N); u.values[1:] = cmd; step(u.values)) for B
 * aircraft on the CPU, the reduced model FROZEN per aircraft as env.py:49-60 freezes it (Ad [B][81], Bd [B][27], Cd [B][81] given).
 * The checker of the product's f16_rollout_mpc / dist.closed_loop_mpc_rollout, with the product's stated rules for the cases the
 * reference does not survive:
 *   - outside the envelope at the start of a step (env.py:117-124 exit()s): frozen, F16O_ST_ENVELOPE | which-state bits, no solve;
 *   - state / demands not finite: no QP to solve (OSQP would iterate on NaN to max_iter and return NaN): NaN command, 0 iterations, bit 32;
 *   - QP certified infeasible: NaN command as OSQP returns it (bit 128); hit max_iter / factorisation failed: bit 64;
 *   - hold != 0: a step without a command keeps the previous one; else the NaN goes into u (np.clip propagates it, utils.py:308-330).
 * x [B][18], u [B][4] in place; dem [B][3]; cmds [T][B][3], iters [T][B], traj [T][B][18] may be NULL; status [B] in/out (sticky). */
void f16o_mpc_closed_loop(double *x, double *u, const double *Ad, const double *Bd, const double *Cd, long B, int N, int T, double dt,
                          double xcg, int fi_flag, const double *dem, const f16o_qp_settings *s, int mode, int hold, double *cmds,
                          int *iters, int *status, double *traj, int nthreads) {
  const int n = 3 * N, m = 15 * N;
#pragma omp parallel num_threads(nthreads > 1 ? nthreads : 1)
  {
    double *P = (double *)malloc(sizeof(double) * ((size_t)n * n + n + (size_t)m * n + 2 * m + n));
    double *q = P + (size_t)n * n, *A = q + n, *l = A + (size_t)m * n, *uu = l + m, *xo = uu + m;
#pragma omp for schedule(dynamic, 1)
    for (long b = 0; b < B; ++b) {
      double *xf = x + 18 * b, *uf = u + 4 * b;
      int st = status ? status[b] : 0;
      for (int t = 0; t < T; ++t) {
        double cmd[3] = {NAN, NAN, NAN};
        int it = 0;
        if (!(st & 16)) st |= f16o_envelope_bits(xf);
        if (!(st & 16)) {
          int fin = 1;
          for (int k = 0; k < 9; ++k) fin = fin && isfinite(xf[MPC_X_IDX[k]]);
          for (int k = 0; k < 3; ++k) fin = fin && isfinite(xf[13 + k]) && isfinite(dem[3 * b + k]);
          if (fin) {
            double info[4];
            f16o_mpc_qp(xf, Ad + 81 * b, Bd + 27 * b, Cd + 81 * b, N, dt, dem + 3 * b, P, q, A, l, uu);
            const int rc = f16o_admm(n, m, P, q, A, l, uu, s, mode, xo, info);
            it = (int)info[0];
            if (rc == 2) st |= 128;
            else if (rc != 0) st |= 64;
            for (int k = 0; k < 3; ++k) cmd[k] = rc == 2 ? NAN : xo[k];
          } else {
            st |= 32;
          }
          for (int k = 0; k < 3; ++k)
            if (!(hold && isnan(cmd[k]))) uf[1 + k] = cmd[k];
          st |= f16o_step(xf, uf, dt, fi_flag, xcg);
          for (int k = 0; k < 18; ++k)
            if (!isfinite(xf[k])) st |= 32;
        }
        if (cmds) memcpy(cmds + ((size_t)t * B + b) * 3, cmd, sizeof cmd);
        if (iters) iters[(size_t)t * B + b] = it;
        if (traj) memcpy(traj + ((size_t)t * B + b) * 18, xf, 18 * sizeof(double));
      }
      if (status) status[b] = st;
    }
    free(P);
  }
}

