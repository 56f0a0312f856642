/* oracle/f16_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement (plain C, fp64, no fast-math, no FMA contraction) of the
 * reference's F-16 plant + actuator + Euler path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product
 * (f16_mpc_oop_py_amd + libf16hip.so) never links, imports or calls it.
 *
 * Parity pinning: checked in tests/test_oracle_vs_golden.py against fixtures that
 * were generated in the build container from the reference's own prebuilt
 * C/nlplant_xcg25.so / nlplant_xcg35.so and its Python (tools/make_golden.py).
 *
 * State x[18] = {npos,epos,h,phi,theta,psi,V,alpha,beta,p,q,r,T,dh,da,dr,lf2,lf1}
 * (parameters.py:116), input u[4] = {T,dh,da,dr} (parameters.py:117).
 */
#ifndef F16_ORACLE_H
#define F16_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

/* status bits (sticky per aircraft in the batched calls) */
#define F16O_ST_ALPHA1 1  /* alpha outside ALPHA1 grid [-20,90] deg           */
#define F16O_ST_ALPHA2 2  /* alpha outside ALPHA2 grid [-20,45] deg (lef tabs) */
#define F16O_ST_BETA 4    /* |beta| > 30 deg                                  */
#define F16O_ST_EL 8      /* |el| > 25 deg                                    */
#define F16O_ST_ENVELOPE 16 /* env.py:117-124 box check failed -> frozen       */

void f16o_init(void);
void f16o_set_xcg(double xcg);   /* used by the drop-in Nlplant symbol (default 0.25) */
void f16o_set_fix_clr(int on);   /* 1: use the real CLr table instead of the reference's unloaded one */
int f16o_last_status(void);      /* grid-clamp bits raised by the last single call in this thread */

/* reference-shaped drop-in symbols (C/nlplant.c:23, :467, :512) */
void Nlplant(double *xu, double *xdot, int fidelity);
void atmos(double alt, double vt, double *coeff);
void accels(double *state, double *xdot, double *y);

/* one of the 43 hifi table functions (C/hifi_F16_AeroData.c:109-1861); tid = enum f16_table_id */
double f16o_table(int tid, double alpha, double beta, double el);
/* lofi group functions, C/lofi_F16_AeroData.c:12-368: which = 0 damping(9) 1 dmomdcon(4) 2 clcn(2) 3 cxcm(2) 4 cz(1) */
void f16o_lofi(int which, double alpha, double beta, double el, double *out);

void f16o_nlplant(const double *xu, double *xdot, int fi_flag, double xcg);
/* env.py:65-103 */
void f16o_calc_xdot(const double *x, const double *u, double *xdot, int fi_flag, double xcg);
/* env.py:152-193: x_full/u_full are self.x.values / self.u.values; x9,u3 the MPC vectors */
void f16o_calc_xdot_na(const double *x_full, const double *x9, const double *u3, double *xdot9, int fi_flag, double xcg);
int f16o_envelope_bits(const double *x);   /* env.py:117-124 alone */
/* env.py:105-130; returns status bits (envelope bit => state left untouched) */
int f16o_step(double *x, const double *u, double dt, int fi_flag, double xcg);
/* env.py:294-342 with _calc_xdot_na/_get_obs_na: A[9*9] B[9*3] C[9*9] D[9*3], row-major */
void f16o_linearise_na(const double *x_full, const double *x9, const double *u3, double eps,
                       double *A, double *B, double *C, double *D, int fi_flag, double xcg);
/* env.py:294-342 default (18-state): A[18*18] B[18*4] C[10*18] D[10*4] */
void f16o_linearise_full(const double *x, const double *u, double eps,
                         double *A, double *B, double *C, double *D, int fi_flag, double xcg);

/* batched helpers (row-major [B][18] / [B][4]); nthreads<=1 => serial. traj may be NULL,
 * else [T][B][18] filled after every step. status[B] in/out (sticky). */
void f16o_xdot_batch(const double *x, const double *u, double *xdot, long B, int fi_flag, double xcg, int nthreads);
void f16o_rollout(double *x, const double *u, long B, int T, double dt, int fi_flag, double xcg,
                  double *traj, int *status, int nthreads);

/* test_env_mk2.py:70-85 (nonlinear LQR loop) for B aircraft: env.py:360-371 action + env.py:105-130 step, T times */
void f16o_rollout_lqr(double *x, const double *u0, const double *K, const double *dem, long B, int T, double dt, int fi_flag,
                      double xcg, double *traj, double *u_out, int *status, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
