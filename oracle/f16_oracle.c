/* oracle/f16_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT (see f16_oracle.h).
 *
 * CPU restatement of the reference hot path in plain C99 / fp64:
 *   interpolation      C/mexndinterp.c:97-265   -> bracket(), interp_nd()
 *   hifi table fns     C/hifi_F16_AeroData.c:109-1861 (data: csrc/f16_tables_data.inc)
 *   hifi group fns     C/hifi_F16_AeroData.c:1871-1934 -> inlined in f16o_nlplant()
 *   lofi fns           C/lofi_F16_AeroData.c:12-368 -> lofi_*()
 *   Nlplant/atmos/accels C/nlplant.c:23-457, 467-490, 512-552
 *   actuators          utils.py:289-330
 *   _calc_xdot/step/_calc_xdot_na/linearise  env.py:65-130, 152-193, 294-342
 * Operation order follows the reference expression by expression so that the
 * result agrees with the reference binary to ~1e-14 relative.  Off-grid lookups are
 * undefined behaviour in the reference (mexndinterp.c:121-124); here the coordinate
 * is clamped to the grid edge and a status bit is raised.
 */
#include "f16_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../f16_mpc_oop_py_amd/csrc/f16_tables_data.inc"

/* ---------------------------------------------------------------- tables */
enum { AX_A1 = 0, AX_A2, AX_B1, AX_D1, AX_D2, N_AXES };
static const int axis_n[N_AXES] = {20, 14, 19, 5, 3};
static double axis_x[N_AXES][20];
static double *hifi[F16_N_HIFI_TABLES];
static double hifi_store[13405];
/* axes of each table, -1 = unused */
static signed char tab_axes[F16_N_HIFI_TABLES][3];
static double L_damp[9][12], L_dlda[7][12], L_dldr[7][12], L_dnda[7][12], L_dndr[7][12];
static double L_cl[7][12], L_cn[7][12], L_cx[5][12], L_cm[5][12], L_cz[12];
static int g_init = 0;
static double g_xcg = 0.25;
static int g_fix_clr = 0;
static __thread int t_status = 0;

static void fill(double *dst, const int32_t *src, int n, double scale) {
  for (int i = 0; i < n; ++i) dst[i] = (double)src[i] / scale;
}

void f16o_init(void) {
  if (g_init) return;
  fill(axis_x[AX_A1], f16_bp_alpha1, 20, F16_HIFI_SCALE);
  fill(axis_x[AX_A2], f16_bp_alpha2, 14, F16_HIFI_SCALE);
  fill(axis_x[AX_B1], f16_bp_beta1, 19, F16_HIFI_SCALE);
  fill(axis_x[AX_D1], f16_bp_dh1, 5, F16_HIFI_SCALE);
  fill(axis_x[AX_D2], f16_bp_dh2, 3, F16_HIFI_SCALE);
  double *p = hifi_store;
  for (int t = 0; t < F16_N_HIFI_TABLES; ++t) {
    hifi[t] = p;
    fill(p, f16_hifi_tables[t], f16_hifi_sizes[t], F16_HIFI_SCALE);
    p += f16_hifi_sizes[t];
    signed char a0 = -1, a1 = -1, a2 = -1;
    if (t <= F16_T_Cm) { a0 = AX_A1; a1 = AX_B1; a2 = AX_D1; }
    else if (t <= F16_T_Cl) { a0 = AX_A1; a1 = AX_B1; a2 = AX_D2; }
    else if (t <= F16_T_Cl_a20) { a0 = AX_A1; a1 = AX_B1; }
    else if (t <= F16_T_Cl_a20_lef) { a0 = AX_A2; a1 = AX_B1; }
    else if (t <= F16_T_dCm) { a0 = AX_A1; }
    else if (t <= F16_T_dCNp_lef) { a0 = AX_A2; }
    else { a0 = AX_D1; }
    tab_axes[t][0] = a0; tab_axes[t][1] = a1; tab_axes[t][2] = a2;
  }
  fill(&L_damp[0][0], f16_lofi_damp, 108, F16_LOFI_SCALE);
  fill(&L_dlda[0][0], f16_lofi_dlda, 84, F16_LOFI_SCALE);
  fill(&L_dldr[0][0], f16_lofi_dldr, 84, F16_LOFI_SCALE);
  fill(&L_dnda[0][0], f16_lofi_dnda, 84, F16_LOFI_SCALE);
  fill(&L_dndr[0][0], f16_lofi_dndr, 84, F16_LOFI_SCALE);
  fill(&L_cl[0][0], f16_lofi_cl, 84, F16_LOFI_SCALE);
  fill(&L_cn[0][0], f16_lofi_cn, 84, F16_LOFI_SCALE);
  fill(&L_cx[0][0], f16_lofi_cx, 60, F16_LOFI_SCALE);
  fill(&L_cm[0][0], f16_lofi_cm, 60, F16_LOFI_SCALE);
  fill(L_cz, f16_lofi_cz, 12, F16_LOFI_SCALE);
  g_init = 1;
}

void f16o_set_xcg(double xcg) { g_xcg = xcg; }
void f16o_set_fix_clr(int on) { g_fix_clr = on; }
int f16o_last_status(void) { return t_status; }

/* mexndinterp.c:97-143 getHyperCube for one axis.  Returns 1 when x is off-grid
 * (reference: UB) -> clamp to the nearest edge node. */
static int bracket(int ax, double *x, int *lo, int *hi) {
  const double *X = axis_x[ax];
  const int n = axis_n[ax];
  *lo = *hi = 0;
  if (*x < X[0]) { *x = X[0]; return 1; }
  if (*x > X[n - 1]) { *x = X[n - 1]; *lo = *hi = n - 1; return 1; }
  for (int j = 0; j < n - 1; ++j) {
    if (*x == X[j]) { *lo = *hi = j; break; }
    if (*x == X[j + 1]) { *lo = *hi = j + 1; break; }
    if (*x > X[j] && *x < X[j + 1]) { *lo = j; *hi = j + 1; break; }
  }
  return 0;
}

/* mexndinterp.c:161-265: gather the 2^n cell vertices (bit j of the vertex number
 * selects lo/hi on axis j; axis 0 is the fastest-varying table index) and collapse
 * axis 0 first with  lambda*f2 + (1-lambda)*f1, or f1 on a degenerate axis. */
static double interp_nd(int tid, const double *v_in) {
  static const int offgrid_bit[N_AXES] = {F16O_ST_ALPHA1, F16O_ST_ALPHA2, F16O_ST_BETA, F16O_ST_EL, F16O_ST_EL};
  const signed char *ax = tab_axes[tid];
  int nd = ax[2] >= 0 ? 3 : (ax[1] >= 0 ? 2 : 1);
  int lo[3], hi[3];
  double v[3], x0[3], x1[3];
  for (int d = 0; d < nd; ++d) {
    v[d] = v_in[d];
    if (bracket(ax[d], &v[d], &lo[d], &hi[d])) t_status |= offgrid_bit[(int)ax[d]];
    x0[d] = axis_x[(int)ax[d]][lo[d]];
    x1[d] = axis_x[(int)ax[d]][hi[d]];
  }
  double T[8];
  const double *Y = hifi[tid];
  for (int i = 0; i < (1 << nd); ++i) {
    int lin = 0, stride = 1;
    for (int d = 0; d < nd; ++d) {
      lin += stride * (((i >> d) & 1) ? hi[d] : lo[d]);
      stride *= axis_n[(int)ax[d]];
    }
    T[i] = Y[lin];
  }
  for (int d = 0; d < nd; ++d) {
    int m = 1 << (nd - d - 1);
    for (int i = 0; i < m; ++i) {
      double f1 = T[2 * i], f2 = T[2 * i + 1];
      if (x0[d] != x1[d]) {
        double lambda = (v[d] - x0[d]) / (x1[d] - x0[d]);
        T[i] = lambda * f2 + (1 - lambda) * f1;
      } else {
        T[i] = f1;
      }
    }
  }
  return T[0];
}

double f16o_table(int tid, double alpha, double beta, double el) {
  f16o_init();
  /* Reference defect reproduced on purpose: _CLr never loads CL1320_ALPHA1_606.dat -- its fscanf
   * loop is the body of `if(fp==NULL)` (C/hifi_F16_AeroData.c:964-972), so the table is
   * uninitialised malloc memory; observed in this container as denormals (~1e-310), i.e. the
   * reference's roll-due-to-yaw-rate derivative is numerically zero.  Restated as exactly 0. */
  if (tid == F16_T_CLr && !g_fix_clr) return 0.0;
  double v[3];
  const signed char *ax = tab_axes[tid];
  if (ax[0] == AX_D1) { v[0] = el; }
  else { v[0] = alpha; v[1] = beta; v[2] = el; }
  return interp_nd(tid, v);
}
#define TAB3(t, a, b, e) f16o_table(F16_T_##t, a, b, e)
#define TAB2(t, a, b) f16o_table(F16_T_##t, a, b, 0.0)
#define TAB1(t, a) f16o_table(F16_T_##t, a, 0.0, 0.0)

/* ------------------------------------------------------------------ lofi */
static int sgn(double v) { return (v > 0) - (v < 0); }
static int fixi(double v) { return (int)trunc(v); }

/* shared alpha index arithmetic, lofi_F16_AeroData.c:31-45 */
static void lofi_alpha(double alpha, int *k, int *L, double *da) {
  double s = .2 * alpha;
  int kk = fixi(s);
  if (kk <= -2) kk = -1; else if (kk >= 9) kk = 8;
  *da = s - kk;
  int LL = kk + fixi(1.1 * sgn(*da));
  *k = kk + 3; *L = LL + 3;
}

static void lofi_damping(double alpha, double *c) {           /* :12-56 */
  int k, L; double da;
  lofi_alpha(alpha, &k, &L, &da);
  for (int i = 0; i < 9; ++i)
    c[i] = L_damp[i][k - 1] + fabs(da) * (L_damp[i][L - 1] - L_damp[i][k - 1]);
}

static double lofi_bilin(double T[][12], int m, int n, int k, int L, double da, double db) {
  double t = T[m - 1][k - 1], u = T[n - 1][k - 1];
  double v = t + fabs(da) * (T[m - 1][L - 1] - t);
  double w = u + fabs(da) * (T[n - 1][L - 1] - u);
  return v + (w - v) * db;
}

static void lofi_dmomdcon(double alpha, double beta, double *c) { /* :59-183 */
  int k, L; double da;
  lofi_alpha(alpha, &k, &L, &da);
  double s = 0.2 * fabs(beta);
  int m = fixi(s);
  if (m >= 7) m = 6;
  double db = s - m;
  int n = m + 1;
  m = m + 1; n = n + 1;
  if (n > 7) { n = 7; t_status |= F16O_ST_BETA; }  /* reference reads past the 7-row table at |beta|>=30 */
  c[0] = lofi_bilin(L_dlda, m, n, k, L, da, db);
  c[1] = lofi_bilin(L_dldr, m, n, k, L, da, db);
  c[2] = lofi_bilin(L_dnda, m, n, k, L, da, db);
  c[3] = lofi_bilin(L_dndr, m, n, k, L, da, db);
}

static void lofi_clcn(double alpha, double beta, double *c) {   /* :185-262 */
  int k, L; double da;
  lofi_alpha(alpha, &k, &L, &da);
  double s = .2 * fabs(beta);
  int m = fixi(s);
  if (m == 0) m = 1; else if (m >= 6) m = 5;
  double db = s - m;
  int n = m + fixi(1.1 * sgn(db));
  m = m + 1; n = n + 1;
  c[0] = lofi_bilin(L_cl, m, n, k, L, da, fabs(db)) * sgn(beta);
  c[1] = lofi_bilin(L_cn, m, n, k, L, da, fabs(db)) * sgn(beta);
}

static void lofi_cxcm(double alpha, double dele, double *c) {   /* :265-336 */
  int k, L; double da;
  lofi_alpha(alpha, &k, &L, &da);
  double s = dele / 12.0;
  int m = fixi(s);
  if (m <= -2) m = -1; else if (m >= 2) m = 1;
  double de = s - m;
  int n = m + fixi(1.1 * sgn(de));
  m = m + 3; n = n + 3;
  c[0] = lofi_bilin(L_cx, m, n, k, L, da, fabs(de));
  c[1] = lofi_bilin(L_cm, m, n, k, L, da, fabs(de));
}

static void lofi_cz(double alpha, double beta, double dele, double *c) { /* :339-368 */
  int k, L; double da;
  lofi_alpha(alpha, &k, &L, &da);
  double s = L_cz[k - 1] + fabs(da) * (L_cz[L - 1] - L_cz[k - 1]);
  c[0] = s * (1 - pow((beta / 57.3), 2)) - .19 * (dele) / 25;
}

void f16o_lofi(int which, double alpha, double beta, double el, double *out) {
  f16o_init();
  switch (which) {
    case 0: lofi_damping(alpha, out); break;
    case 1: lofi_dmomdcon(alpha, beta, out); break;
    case 2: lofi_clcn(alpha, beta, out); break;
    case 3: lofi_cxcm(alpha, el, out); break;
    default: lofi_cz(alpha, beta, el, out); break;
  }
}

/* ------------------------------------------------ atmos / accels / plant */
void atmos(double alt, double vt, double *coeff) {             /* nlplant.c:467-490 */
  double rho0 = 2.377e-3;
  double tfac = 1 - .703e-5 * (alt);
  double temp = 519.0 * tfac;
  if (alt >= 35000.0) temp = 390;
  double rho = rho0 * pow(tfac, 4.14);
  double mach = (vt) / sqrt(1.4 * 1716.3 * temp);
  double qbar = .5 * rho * pow(vt, 2);
  double ps = 1715.0 * rho * temp;
  if (ps == 0) ps = 1715;
  coeff[0] = mach; coeff[1] = qbar; coeff[2] = ps;
}

void accels(double *state, double *xdot, double *y) {          /* nlplant.c:512-552 */
  const double grav = 32.174;
  double sina = sin(state[7]), cosa = cos(state[7]);
  double sinb = sin(state[8]), cosb = cos(state[8]);
  double vel_u = state[6] * cosb * cosa;
  double vel_v = state[6] * sinb;
  double vel_w = state[6] * cosb * sina;
  double u_dot = cosb * cosa * xdot[6] - state[6] * sinb * cosa * xdot[8] - state[6] * cosb * sina * xdot[7];
  double v_dot = sinb * xdot[6] + state[6] * cosb * xdot[8];
  double w_dot = cosb * sina * xdot[6] - state[6] * sinb * sina * xdot[8] + state[6] * cosb * cosa * xdot[7];
  y[0] = 1.0 / grav * (u_dot + state[10] * vel_w - state[11] * vel_v) + sin(state[4]);
  y[1] = 1.0 / grav * (v_dot + state[11] * vel_u - state[9] * vel_w) - cos(state[4]) * sin(state[3]);
  y[2] = -1.0 / grav * (w_dot + state[9] * vel_v - state[10] * vel_u) + cos(state[4]) * cos(state[3]);
}

void f16o_nlplant(const double *xu, double *xdot, int fi_flag, double xcg) { /* nlplant.c:23-457 */
  f16o_init();
  t_status = 0;
  const double g = 32.17, m = 636.94, B = 30.0, S = 300.0, cbar = 11.32, xcgr = 0.35;
  const double Heng = 0.0;
  const double pi = acos(-1);
  const double Jy = 55814.0, Jxz = 982.0, Jz = 63100.0, Jx = 9496.0;
  const double r2d = 180.0 / pi;

  double alt = xu[2], phi = xu[3], theta = xu[4], psi = xu[5];
  double vt = xu[6];
  double alpha = xu[7] * r2d, beta = xu[8] * r2d;
  double P = xu[9], Q = xu[10], R = xu[11];
  double sa = sin(xu[7]), ca = cos(xu[7]);
  double sb = sin(xu[8]), cb = cos(xu[8]);
  double st = sin(theta), ct = cos(theta), tt = tan(theta);
  double sphi = sin(phi), cphi = cos(phi), spsi = sin(psi), cpsi = cos(psi);
  if (vt <= 0.01) vt = 0.01;

  double T = xu[12], el = xu[13], ail = xu[14], rud = xu[15], lef = xu[16];
  double dail = ail / 21.5;
  double drud = rud / 30.0;
  double dlef = (1 - lef / 25.0);

  double at[3];
  atmos(alt, vt, at);
  double mach = at[0], qbar = at[1], ps = at[2];

  double U = vt * ca * cb, V = vt * sb, W = vt * sa * cb;
  xdot[0] = U * (ct * cpsi) + V * (sphi * cpsi * st - cphi * spsi) + W * (cphi * st * cpsi + sphi * spsi);
  xdot[1] = U * (ct * spsi) + V * (sphi * spsi * st + cphi * cpsi) + W * (cphi * st * spsi - sphi * cpsi);
  xdot[2] = U * st - V * (sphi * ct) - W * (cphi * ct);
  xdot[3] = P + tt * (Q * sphi + R * cphi);
  xdot[4] = Q * cphi - R * sphi;
  xdot[5] = (Q * sphi + R * cphi) / ct;

  double Cx, Cz, Cm, Cy, Cn, Cl;
  double Cxq, Cyr, Cyp, Czq, Clr, Clp, Cmq, Cnr, Cnp;
  double delta_Cx_lef = 0, delta_Cz_lef = 0, delta_Cm_lef = 0, delta_Cy_lef = 0, delta_Cn_lef = 0, delta_Cl_lef = 0;
  double delta_Cxq_lef = 0, delta_Cyr_lef = 0, delta_Cyp_lef = 0, delta_Czq_lef = 0, delta_Clr_lef = 0,
         delta_Clp_lef = 0, delta_Cmq_lef = 0, delta_Cnr_lef = 0, delta_Cnp_lef = 0;
  double delta_Cy_r30 = 0, delta_Cn_r30, delta_Cl_r30;
  double delta_Cy_a20 = 0, delta_Cy_a20_lef = 0, delta_Cn_a20, delta_Cn_a20_lef = 0, delta_Cl_a20, delta_Cl_a20_lef = 0;
  double delta_Cnbeta = 0, delta_Clbeta = 0, delta_Cm = 0, eta_el = 1.0, delta_Cm_ds = 0;

  if (fi_flag == 1) {
    /* hifi_C :1871 */
    Cx = TAB3(Cx, alpha, beta, el); Cz = TAB3(Cz, alpha, beta, el); Cm = TAB3(Cm, alpha, beta, el);
    Cy = TAB2(Cy, alpha, beta);
    Cn = TAB3(Cn, alpha, beta, el); Cl = TAB3(Cl, alpha, beta, el);
    /* hifi_damping :1880 */
    Cxq = TAB1(CXq, alpha); Cyr = TAB1(CYr, alpha); Cyp = TAB1(CYp, alpha);
    Czq = TAB1(CZq, alpha); Clr = TAB1(CLr, alpha); Clp = TAB1(CLp, alpha);
    Cmq = TAB1(CMq, alpha); Cnr = TAB1(CNr, alpha); Cnp = TAB1(CNp, alpha);
    /* hifi_C_lef :1892 */
    delta_Cx_lef = TAB2(Cx_lef, alpha, beta) - TAB3(Cx, alpha, beta, 0);
    delta_Cz_lef = TAB2(Cz_lef, alpha, beta) - TAB3(Cz, alpha, beta, 0);
    delta_Cm_lef = TAB2(Cm_lef, alpha, beta) - TAB3(Cm, alpha, beta, 0);
    delta_Cy_lef = TAB2(Cy_lef, alpha, beta) - TAB2(Cy, alpha, beta);
    delta_Cn_lef = TAB2(Cn_lef, alpha, beta) - TAB3(Cn, alpha, beta, 0);
    delta_Cl_lef = TAB2(Cl_lef, alpha, beta) - TAB3(Cl, alpha, beta, 0);
    /* hifi_damping_lef :1901 */
    delta_Cxq_lef = TAB1(dCXq_lef, alpha); delta_Cyr_lef = TAB1(dCYr_lef, alpha); delta_Cyp_lef = TAB1(dCYp_lef, alpha);
    delta_Czq_lef = TAB1(dCZq_lef, alpha); delta_Clr_lef = TAB1(dCLr_lef, alpha); delta_Clp_lef = TAB1(dCLp_lef, alpha);
    delta_Cmq_lef = TAB1(dCMq_lef, alpha); delta_Cnr_lef = TAB1(dCNr_lef, alpha); delta_Cnp_lef = TAB1(dCNp_lef, alpha);
    /* hifi_rudder :1913 */
    delta_Cy_r30 = TAB2(Cy_r30, alpha, beta) - TAB2(Cy, alpha, beta);
    delta_Cn_r30 = TAB2(Cn_r30, alpha, beta) - TAB3(Cn, alpha, beta, 0);
    delta_Cl_r30 = TAB2(Cl_r30, alpha, beta) - TAB3(Cl, alpha, beta, 0);
    /* hifi_ailerons :1919 */
    delta_Cy_a20 = TAB2(Cy_a20, alpha, beta) - TAB2(Cy, alpha, beta);
    delta_Cy_a20_lef = TAB2(Cy_a20_lef, alpha, beta) - TAB2(Cy_lef, alpha, beta) - delta_Cy_a20;
    delta_Cn_a20 = TAB2(Cn_a20, alpha, beta) - TAB3(Cn, alpha, beta, 0);
    delta_Cn_a20_lef = TAB2(Cn_a20_lef, alpha, beta) - TAB2(Cn_lef, alpha, beta) - delta_Cn_a20;
    delta_Cl_a20 = TAB2(Cl_a20, alpha, beta) - TAB3(Cl, alpha, beta, 0);
    delta_Cl_a20_lef = TAB2(Cl_a20_lef, alpha, beta) - TAB2(Cl_lef, alpha, beta) - delta_Cl_a20;
    /* hifi_other_coeffs :1928 */
    delta_Cnbeta = TAB1(dCNbeta, alpha); delta_Clbeta = TAB1(dCLbeta, alpha); delta_Cm = TAB1(dCm, alpha);
    eta_el = f16o_table(F16_T_eta_el, 0, 0, el);
    delta_Cm_ds = 0;
  } else {
    /* nlplant.c:245-323 */
    double c[9];
    dlef = 0.0;
    lofi_damping(alpha, c);
    Cxq = c[0]; Cyr = c[1]; Cyp = c[2]; Czq = c[3]; Clr = c[4]; Clp = c[5]; Cmq = c[6]; Cnr = c[7]; Cnp = c[8];
    lofi_dmomdcon(alpha, beta, c);
    delta_Cl_a20 = c[0]; delta_Cl_r30 = c[1]; delta_Cn_a20 = c[2]; delta_Cn_r30 = c[3];
    lofi_clcn(alpha, beta, c);
    Cl = c[0]; Cn = c[1];
    lofi_cxcm(alpha, el, c);
    Cx = c[0]; Cm = c[1];
    Cy = -.02 * beta + .021 * dail + .086 * drud;
    lofi_cz(alpha, beta, el, c);
    Cz = c[0];
  }

  /* totals, NASA TP-1538 p37-40 as written at nlplant.c:333-377 (quirk: dZdQ uses delta_Cz_lef) */
  double dXdQ = (cbar / (2 * vt)) * (Cxq + delta_Cxq_lef * dlef);
  double Cx_tot = Cx + delta_Cx_lef * dlef + dXdQ * Q;
  double dZdQ = (cbar / (2 * vt)) * (Czq + delta_Cz_lef * dlef);
  (void)delta_Czq_lef;
  double Cz_tot = Cz + delta_Cz_lef * dlef + dZdQ * Q;
  double dMdQ = (cbar / (2 * vt)) * (Cmq + delta_Cmq_lef * dlef);
  double Cm_tot = Cm * eta_el + Cz_tot * (xcgr - xcg) + delta_Cm_lef * dlef + dMdQ * Q + delta_Cm + delta_Cm_ds;
  double dYdail = delta_Cy_a20 + delta_Cy_a20_lef * dlef;
  double dYdR = (B / (2 * vt)) * (Cyr + delta_Cyr_lef * dlef);
  double dYdP = (B / (2 * vt)) * (Cyp + delta_Cyp_lef * dlef);
  double Cy_tot = Cy + delta_Cy_lef * dlef + dYdail * dail + delta_Cy_r30 * drud + dYdR * R + dYdP * P;
  double dNdail = delta_Cn_a20 + delta_Cn_a20_lef * dlef;
  double dNdR = (B / (2 * vt)) * (Cnr + delta_Cnr_lef * dlef);
  double dNdP = (B / (2 * vt)) * (Cnp + delta_Cnp_lef * dlef);
  double Cn_tot = Cn + delta_Cn_lef * dlef - Cy_tot * (xcgr - xcg) * (cbar / B) + dNdail * dail + delta_Cn_r30 * drud
                  + dNdR * R + dNdP * P + delta_Cnbeta * beta;
  double dLdail = delta_Cl_a20 + delta_Cl_a20_lef * dlef;
  double dLdR = (B / (2 * vt)) * (Clr + delta_Clr_lef * dlef);
  double dLdP = (B / (2 * vt)) * (Clp + delta_Clp_lef * dlef);
  double Cl_tot = Cl + delta_Cl_lef * dlef + dLdail * dail + delta_Cl_r30 * drud + dLdR * R + dLdP * P + delta_Clbeta * beta;

  double Udot = R * V - Q * W - g * st + qbar * S * Cx_tot / m + T / m;
  double Vdot = P * W - R * U + g * ct * sphi + qbar * S * Cy_tot / m;
  double Wdot = Q * U - P * V + g * ct * cphi + qbar * S * Cz_tot / m;
  xdot[6] = (U * Udot + V * Vdot + W * Wdot) / vt;
  xdot[7] = (U * Wdot - W * Udot) / (U * U + W * W);
  xdot[8] = (Vdot * vt - V * xdot[6]) / (vt * vt * cb);

  double L_tot = Cl_tot * qbar * S * B;
  double M_tot = Cm_tot * qbar * S * cbar;
  double N_tot = Cn_tot * qbar * S * B;
  double denom = Jx * Jz - Jxz * Jxz;
  xdot[9] = (Jz * L_tot + Jxz * N_tot - (Jz * (Jz - Jy) + Jxz * Jxz) * Q * R + Jxz * (Jx - Jy + Jz) * P * Q + Jxz * Q * Heng) / denom;
  xdot[10] = (M_tot + (Jz - Jx) * P * R - Jxz * (P * P - R * R) - R * Heng) / Jy;
  xdot[11] = (Jx * N_tot + Jxz * L_tot + (Jx * (Jx - Jy) + Jxz * Jxz) * P * Q - Jxz * (Jx - Jy + Jz) * Q * R + Jx * Q * Heng) / denom;

  double y[3];
  accels((double *)xu, xdot, y);
  xdot[12] = y[0]; xdot[13] = y[1]; xdot[14] = y[2];
  xdot[15] = mach; xdot[16] = qbar; xdot[17] = ps;
}

void Nlplant(double *xu, double *xdot, int fidelity) { f16o_nlplant(xu, xdot, fidelity, g_xcg); }

/* ------------------------------------------------------------- actuators */
/* np.clip (utils.py:303-330): NaN in -> NaN out (fmin / fmax alone would return the bound: a NaN command -- what OSQP hands
 * back for an infeasible QP, env.py:420-424 -- would become a hard-over at full rate).  Pinned by fixture G3b. */
static double clipd(double a, double lo, double hi) { return isnan(a) ? a : fmin(fmax(a, lo), hi); }
static const double PI_NP = 3.141592653589793;  /* numpy.pi */

/* utils.py:289-306 -> (lf1_dot, lf2_dot) */
static void upd_lef(double h, double V, double alpha, double lf1, double lf2, double *lf1_dot, double *lf2_dot) {
  double coeff[3];
  atmos(h, V, coeff);
  double atmos_out = coeff[1] / coeff[2] * 9.05;
  double alpha_deg = alpha * 180 / PI_NP;
  double LF_err = alpha_deg - (lf1 + (2 * alpha_deg));
  double LF_out = (lf1 + (2 * alpha_deg)) * 1.38;
  double lef_cmd = LF_out + 1.45 - atmos_out;
  lef_cmd = clipd(lef_cmd, 0., 25);
  double lef_err = clipd((1 / 0.136) * (lef_cmd - lf2), -25, 25);
  *lf1_dot = LF_err * 7.25;
  *lf2_dot = lef_err;
}

void f16o_calc_xdot(const double *x, const double *u, double *xdot, int fi_flag, double xcg) { /* env.py:65-103 */
  double t0 = clipd(clipd(u[0], 1000, 19000) - x[12], -10000, 10000);       /* utils.py:308-312 */
  double t1 = clipd(20.2 * (clipd(u[1], -25, 25) - x[13]), -60, 60);        /* :314-318 */
  double t2 = clipd(20.2 * (clipd(u[2], -21.5, 21.5) - x[14]), -80, 80);    /* :320-324 */
  double t3 = clipd(20.2 * (clipd(u[3], -30., 30) - x[15]), -120, 120);     /* :326-330 */
  double lf1_dot, lf2_dot;
  upd_lef(x[2], x[6], x[7], x[17], x[16], &lf1_dot, &lf2_dot);
  f16o_nlplant(x, xdot, fi_flag, xcg);
  xdot[12] = t0; xdot[13] = t1; xdot[14] = t2; xdot[15] = t3;
  xdot[16] = lf2_dot; xdot[17] = lf1_dot;
}

static const int MPC_X_IDX[9] = {3, 4, 7, 8, 9, 10, 11, 17, 16};  /* parameters.py:135,161 */
static const int OBS_X_IDX[10] = {2, 3, 4, 7, 8, 9, 10, 11, 16, 17}; /* parameters.py:134,160 */

void f16o_calc_xdot_na(const double *x_full, const double *x9, const double *u3, double *xdot9, int fi_flag, double xcg) {
  /* env.py:152-193 */
  double sv[18], xd[18], svd[18];
  memcpy(sv, x_full, sizeof sv);
  for (int i = 0; i < 9; ++i) sv[MPC_X_IDX[i]] = x9[i];
  for (int i = 0; i < 3; ++i) sv[13 + i] = u3[i];
  double lf1_dot, lf2_dot;
  upd_lef(sv[2], sv[6], sv[7], sv[17], sv[16], &lf1_dot, &lf2_dot);
  f16o_nlplant(sv, xd, fi_flag, xcg);
  for (int i = 0; i < 12; ++i) svd[i] = xd[i];
  svd[12] = svd[13] = svd[14] = svd[15] = 0.0;
  svd[16] = lf1_dot;   /* quirk: swapped w.r.t. _calc_xdot (env.py:184,189) */
  svd[17] = lf2_dot;
  for (int i = 0; i < 9; ++i) xdot9[i] = svd[MPC_X_IDX[i]];
}

static const double X_LB[18] = {-INFINITY, -INFINITY, 0, -INFINITY, -INFINITY, -INFINITY, 0, -20., -30., -300, -100, -50,
                                1000, -25, -21.5, -30., 0., -INFINITY};
static const double X_UB[18] = {INFINITY, INFINITY, 100000, INFINITY, INFINITY, INFINITY, 900, 90, 30, 300, 100, 50,
                                19000, 25, 21.5, 30, 25, INFINITY};

int f16o_envelope_bits(const double *x) { /* env.py:117-124: 0 inside, else F16O_ST_ENVELOPE | bit 8 + i per state outside */
  int out = 0;
  for (int i = 0; i < 18; ++i)
    if (x[i] < X_LB[i] || x[i] > X_UB[i]) out |= F16O_ST_ENVELOPE | (1 << (8 + i));      /* bit 8 + i: state i was outside (env.py:121-123 prints it) */
  return out;
}

int f16o_step(double *x, const double *u, double dt, int fi_flag, double xcg) { /* env.py:105-130 */
  const int out = f16o_envelope_bits(x);
  if (out) return out;
  double xd[18];
  f16o_calc_xdot(x, u, xd, fi_flag, xcg);
  int st = t_status;
  for (int i = 0; i < 18; ++i) x[i] += xd[i] * dt;
  return st;
}

void f16o_linearise_na(const double *x_full, const double *x9, const double *u3, double eps,
                       double *A, double *B, double *C, double *D, int fi_flag, double xcg) { /* env.py:294-342 */
  double f0[9], f1[9], xp[9], up[3];
  for (int i = 0; i < 9; ++i) {
    memcpy(xp, x9, sizeof xp);
    xp[i] = x9[i] + eps;
    f16o_calc_xdot_na(x_full, xp, u3, f1, fi_flag, xcg);
    f16o_calc_xdot_na(x_full, x9, u3, f0, fi_flag, xcg);
    for (int r = 0; r < 9; ++r) {
      A[r * 9 + i] = (f1[r] - f0[r]) / eps;
      C[r * 9 + i] = (xp[r] - x9[r]) / eps;
    }
  }
  for (int i = 0; i < 3; ++i) {
    memcpy(up, u3, sizeof up);
    up[i] = u3[i] + eps;
    f16o_calc_xdot_na(x_full, x9, up, f1, fi_flag, xcg);
    f16o_calc_xdot_na(x_full, x9, u3, f0, fi_flag, xcg);
    for (int r = 0; r < 9; ++r) {
      B[r * 3 + i] = (f1[r] - f0[r]) / eps;
      D[r * 3 + i] = (x9[r] - x9[r]) / eps;
    }
  }
}

void f16o_linearise_full(const double *x, const double *u, double eps,
                         double *A, double *B, double *C, double *D, int fi_flag, double xcg) {
  double f0[18], f1[18], xp[18], up[4];
  for (int i = 0; i < 18; ++i) {
    memcpy(xp, x, sizeof xp);
    xp[i] = x[i] + eps;
    f16o_calc_xdot(xp, u, f1, fi_flag, xcg);
    f16o_calc_xdot(x, u, f0, fi_flag, xcg);
    for (int r = 0; r < 18; ++r) A[r * 18 + i] = (f1[r] - f0[r]) / eps;
    for (int r = 0; r < 10; ++r) C[r * 18 + i] = (xp[OBS_X_IDX[r]] - x[OBS_X_IDX[r]]) / eps;
  }
  for (int i = 0; i < 4; ++i) {
    memcpy(up, u, sizeof up);
    up[i] = u[i] + eps;
    f16o_calc_xdot(x, up, f1, fi_flag, xcg);
    f16o_calc_xdot(x, u, f0, fi_flag, xcg);
    for (int r = 0; r < 18; ++r) B[r * 4 + i] = (f1[r] - f0[r]) / eps;
    for (int r = 0; r < 10; ++r) D[r * 4 + i] = 0.0;
  }
}

/* ---------------------------------------------------------------- batched */
void f16o_xdot_batch(const double *x, const double *u, double *xdot, long B, int fi_flag, double xcg, int nthreads) {
  f16o_init();
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)
#endif
  for (long b = 0; b < B; ++b) f16o_calc_xdot(x + 18 * b, u + 4 * b, xdot + 18 * b, fi_flag, xcg);
}

void f16o_rollout(double *x, const double *u, long B, int T, double dt, int fi_flag, double xcg,
                  double *traj, int *status, int nthreads) {
  f16o_init();
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)
#endif
  for (long b = 0; b < B; ++b) {
    int st = status ? status[b] : 0;
    for (int t = 0; t < T; ++t) {
      if (!(st & F16O_ST_ENVELOPE)) st |= f16o_step(x + 18 * b, u + 4 * b, dt, fi_flag, xcg);
      if (traj) memcpy(traj + ((long)t * B + b) * 18, x + 18 * b, 18 * sizeof(double));
    }
    if (status) status[b] = st;
  }
}

/* The reference's nonlinear LQR loop (test_env_mk2.py:70-85): per step u = _calc_LQR_action(p, q, r, K, x._get_mpc_x(),
 * u.initial_condition[1:]) (env.py:360-371: x_ref = copy of x9 with [4:7] = demands; u = -K (x_ref - x9) + u0), u.values[1:] = u,
 * step(u.values) (env.py:105-130).  x [B][18] in place, u0 [B][4], K [B][27] (3 x 9 row-major, the reference's K = -dlqr),
 * dem [B][3]; traj (may be NULL) [T][B][18]; u_out (may be NULL) [B][4] = u.values after the loop. */
void f16o_rollout_lqr(double *x, const double *u0, const double *K, const double *dem, long B, int T, double dt, int fi_flag,
                      double xcg, double *traj, double *u_out, int *status, int nthreads) {
  f16o_init();
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)
#endif
  for (long b = 0; b < B; ++b) {
    int st = status ? status[b] : 0;
    double u[4];
    memcpy(u, u0 + 4 * b, sizeof u);
    for (int t = 0; t < T; ++t) {
      if (!(st & F16O_ST_ENVELOPE)) {
        double x9[9], xr[9];
        for (int i = 0; i < 9; ++i) x9[i] = xr[i] = x[18 * b + MPC_X_IDX[i]];
        xr[4] = dem[3 * b]; xr[5] = dem[3 * b + 1]; xr[6] = dem[3 * b + 2];
        double un[4] = {u[0], 0.0, 0.0, 0.0};
        for (int i = 0; i < 3; ++i) {
          double s = 0.0;
          for (int j = 0; j < 9; ++j) s += -K[27 * b + 9 * i + j] * (xr[j] - x9[j]);
          un[1 + i] = s + u0[4 * b + 1 + i];
        }
        const int s1 = f16o_step(x + 18 * b, un, dt, fi_flag, xcg);
        st |= s1;
        /* an aircraft found outside its envelope takes no step (the reference exit()s inside step, env.py:121-124): u_out keeps
         * the action of the last step it TOOK (u0 if none) -- the rule of every device kernel */
        if (!(s1 & F16O_ST_ENVELOPE)) memcpy(u, un, sizeof u);
      }
      if (traj) memcpy(traj + ((long)t * B + b) * 18, x + 18 * b, 18 * sizeof(double));
    }
    if (u_out) memcpy(u_out + 4 * b, u, sizeof u);
    if (status) status[b] = st;
  }
}
