/* include/f16_hip.h -- C-ABI of libf16hip.so: the MI355X (gfx950) replacement for the reference's
 * ctypes boundary  CDLL("C/nlplant_xcg{25,35}.so")  (parameters.py:108-114; call sites env.py:100,
 * env.py:187, utils.py:291).
 *
 * Two groups of entry points:
 *  (1) the reference's own two symbols, same signatures, host pointers, synchronous -- so an
 *      unmodified reference-style caller keeps working when its CDLL handle points here;
 *  (2) batched entry points over DEVICE pointers for thousands of independent aircraft per
 *      launch.  They replace the Python loops of env.py:105-130 (step), :65-103 (_calc_xdot),
 *      :152-193 (_calc_xdot_na), :294-342 (linearise), :344-371 (LQR), :373-424 (MPC).
 *
 * Conventions for group (2)
 *  - every pointer is a device pointer unless named h_*; nothing is retained after the call
 *  - batched vectors are state-major ("SoA"): element k of aircraft b lives at  p[k*ld + b],
 *    ld >= B (leading dimension, in doubles).  State order x[18] and input order u[4] are the
 *    reference's (parameters.py:116-117).
 *  - all arithmetic fp64; xcg is a run-time argument (the reference bakes it in at compile time,
 *    C/nlplant.c:34, and ships two binaries); fi_flag 1 = hifi Nguyen, 0 = lofi Stevens-Lewis
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous
 *  - return value: 0 on success, negative F16_E* on a host-side error (bad argument, HIP error);
 *    per-aircraft conditions are reported in the int32 status[] words (sticky OR of F16_ST_* bits).
 *    The reference has no error returns: it printf()s and runs into UB off-grid
 *    (C/mexndinterp.c:121-124) and exit()s on an envelope violation (env.py:117-124).
 */
#ifndef F16_HIP_H
#define F16_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct f16_ctx f16_ctx;

/* host-side error codes */
#define F16_OK 0
#define F16_EINVAL (-1)   /* bad argument (NULL pointer, ld < B, unsupported size) */
#define F16_EHIP (-2)     /* a HIP runtime call failed; see f16_last_error() */
#define F16_ENOGPU (-3)   /* no gfx950 device visible */

/* per-aircraft status bits */
#define F16_ST_ALPHA1 1     /* alpha left the ALPHA1 grid [-20,90] deg: lookup clamped        */
#define F16_ST_ALPHA2 2     /* alpha left the ALPHA2 grid [-20,45] deg (lef tables): clamped  */
#define F16_ST_BETA 4       /* |beta| > 30 deg: clamped                                       */
#define F16_ST_EL 8         /* |el| > 25 deg: clamped                                         */
#define F16_ST_ENVELOPE 16  /* env.py:117-124 box check failed: aircraft frozen from then on  */
#define F16_ST_ENV_STATE(k) (1 << (8 + (k))) /* ... and WHICH of the 18 states were outside their box at that step (set with
                                                F16_ST_ENVELOPE; the reference prints the offending state before it exits)       */
#define F16_ST_NONFINITE 32 /* a state became NaN/Inf                                         */
#define F16_ST_QP_MAXITER 64 /* ADMM hit max_iter before meeting the OSQP termination test    */
#define F16_ST_QP_INFEASIBLE 128 /* OSQP primal-infeasibility certificate met: command = NaN  */
#define F16_ST_LOOP_STALL (1 << 26) /* f16_rollout_mpc: a wavefront gave up waiting (tens of seconds) for the previous step of this
                                       aircraft and went on regardless -- a guard that lets the grid drain; never observed          */

/* behaviour flags */
#define F16_FLAG_FIX_CLR 1u      /* use the real CLr table (reference never loads it: hifi_F16_AeroData.c:964-972) */
#define F16_FLAG_NO_ENVELOPE 2u  /* skip the env.py:117-124 box check in step/rollout                               */
#define F16_FLAG_ONE_LANE 4u     /* f16_rollout / f16_rollout_lqr: always the one-lane-per-aircraft kernel (64-lane workgroups), whatever
                                    the batch size -- results then do not depend on B bit for bit (the launch rules otherwise pick a
                                    four-lanes-per-aircraft or four-wavefront kernel for small batches: same terms, the six coefficient
                                    totals summed in another order, ulp-level differences).  f16_rollout_mpc steps with exactly this code. */
#define F16_FLAG_HOLD_COMMAND 8u /* closed MPC loops (f16_rollout_mpc; dist.closed_loop_mpc_rollout(hold_command=True)): a step whose QP
                                    is infeasible (or whose state is not finite) keeps the PREVIOUS surface commands instead of the NaN
                                    the reference would write into u.values (env.py:420-424 -> test_env.py:490-493)                  */

/* ---- lifetime --------------------------------------------------------------------------- */
/* Builds the fp64 table image (int/1e5, IEEE division) and uploads it to `device`. */
int f16_create(f16_ctx **out, int device);
void f16_destroy(f16_ctx *ctx);
const char *f16_last_error(void);
/* bytes of LDS image / number of doubles, for tests and the roofline bookkeeping */
size_t f16_table_image_doubles(void);
/* copy the device table image back to host (tests): n = f16_table_image_doubles() */
int f16_debug_read_tables(f16_ctx *ctx, double *h_out);
/* the same tables as the scaled-integer image the large-batch rollout reads (value = k / 1e5; breakpoints as doubles in
 * front, csrc/f16_tables.h namespace i32): n_ints = f16_table_image_i32_ints() */
size_t f16_table_image_i32_ints(void);
int f16_debug_read_tables_i32(f16_ctx *ctx, int32_t *h_out);

/* Tests: ONE of the reference's 43 hifi table functions (C/hifi_F16_AeroData.c:109-1861: `_Cx(alpha,beta,el)`,
 * `_CXq(alpha)`, ... each a lazy file read + interpn(), C/mexndinterp.c:97-265) evaluated on the device by the bracket /
 * interpolation helpers the dynamics kernels use, at n query points given on the HOST (degrees, as the reference's
 * functions take them; beta / el ignored by tables without that axis).  tid = enum f16_table_id
 * (csrc/f16_tables_data.inc: 0 Cx, 1 Cz, 2 Cm, 3 Cn, 4 Cl, 5 Cy, 6-8 r30, 9-11 a20, 12-17 lef, 18-20 a20_lef, 21-29 damping,
 * 30-32 brett, 33-41 lef damping, 42 eta_el).  h_status (may be NULL) receives the F16_ST_* grid-clamp bits. */
int f16_debug_table_lookup(f16_ctx *ctx, int tid, const double *h_alpha, const double *h_beta, const double *h_el,
                           int n, double *h_out, int32_t *h_status);

/* ---- (1) drop-in symbols of the reference .so ---------------------------------------------- */
/* replaces C/nlplant.c:23  void Nlplant(double *xu, double *xdot, int fidelity)
 * host pointers; reads xu[0..16], writes xdot[0..17]; runs ONE aircraft on the GPU. */
void Nlplant(double *xu, double *xdot, int fidelity);
/* replaces C/nlplant.c:467 void atmos(double alt, double vt, double *coeff) -> coeff[0..2]=mach,qbar,ps */
void atmos(double alt, double vt, double *coeff);
/* replaces the choice between nlplant_xcg25.so / nlplant_xcg35.so (parameters.py:108-111) for the two
 * symbols above; default 0.25.  flags as F16_FLAG_*. */
void f16_dropin_config(double xcg, unsigned flags);

/* ---- (2) batched dynamics --------------------------------------------------------------- */
/* env.py:65-103 _calc_xdot for B aircraft: xdot[18][ld] = f(x[18][ld], u[4][ld]). status may be NULL. */
int f16_xdot_batch(f16_ctx *ctx, const double *x, const double *u, double *xdot, int32_t *status,
                   long B, long ld, double xcg, int fi_flag, unsigned flags, void *stream);
/* C/nlplant.c:23-457 Nlplant itself (no actuator models, outputs 12..17 = nx,ny,nz,mach,qbar,ps). */
int f16_nlplant_batch(f16_ctx *ctx, const double *xu, double *xdot, int32_t *status,
                      long B, long ld, double xcg, int fi_flag, unsigned flags, void *stream);
/* env.py:105-130 step, in place: envelope check, x += xdot*dt.  nsteps Euler steps per launch with the
 * state held in registers; traj (may be NULL) receives the state after every `traj_every`-th step as
 * [nsteps/traj_every][18][ld].
 * Split launches: rollout(n) followed by rollout(m) equals rollout(n + m) BIT FOR BIT for B <= 16,384 (the kernels of that
 * range evaluate every sine / cosine from scratch) and, for larger batches, whenever n is a multiple of 32: the one-lane
 * kernels of the large-batch range carry the five sin / cos pairs of a step to the next one by the exact increment of their
 * angles and re-evaluate them exactly at steps 0, 32, 64, ... of the LAUNCH (default build; -DF16_NO_INC_TRIG switches it off), so
 * a split at another step re-evaluates at other steps: the two results then differ by the rounding of the carried pairs, <= 1e-12
 * relative over 100 steps (tests/test_gpu_dynamics.py::test_split_launches_*).  For the same reason an aircraft's last bits
 * depend on which kernel its batch size selects; F16_FLAG_ONE_LANE pins the kernel (not the carried pairs). */
int f16_rollout(f16_ctx *ctx, double *x, const double *u, double *traj, int32_t *status,
                long B, long ld, int nsteps, int traj_every, double dt, double xcg, int fi_flag,
                unsigned flags, void *stream);
/* The reference's closed loop under its LQR controller (the only controller its drivers actually run: flight_sim.py:139,181;
 * nonlinear loop test_env_mk2.py:70-85) as ONE launch: per step the action of env.py:360-371
 *     u[1:4] = -K (x_ref - x9) + u0[1:4],   x9 = x[mpc idx] (parameters.py:135), x_ref = x9 with x_ref[4:7] = (p, q, r)_dem
 * from the state at the start of the step, the thrust command u0[0] held (test_env_mk2.py:79 overwrites u.values[1:] only),
 * then env.py:105-130 step -- inside the same kernels f16_rollout launches, the state in registers for all nsteps.
 * K[27][ld]: the gain as the reference holds it (`_calc_LQR_gain` returns K = -dlqr; 3 x 9 row-major: f16_lqr_batch's output);
 * dem[3][ld]; u0[4][ld] = u.initial_condition.  x in place; traj as in f16_rollout; u_out (may be NULL) [4][ld] receives the
 * action of the last step (what self.u.values holds after the loop).  (x_ref - x9 is exactly zero outside the three rate
 * entries, so only K[:, 4:7] enters the product.) */
int f16_rollout_lqr(f16_ctx *ctx, double *x, const double *u0, const double *K, const double *dem, double *traj,
                    double *u_out, int32_t *status, long B, long ld, int nsteps, int traj_every, double dt, double xcg,
                    int fi_flag, unsigned flags, void *stream);
/* The reference's LINEAR-model closed loops as ONE launch (9-state reduced model, 3 inputs; one lane per aircraft, its matrices in
 * registers):   per step   u = -K (x_ref - x) + u0,   x = Ad x + Bd u
 *   test_env_mk2.py:46-62 `LQR(linear=True)` (what main.py:35 runs): the frozen model ssr.Ad / ssr.Bd under env.py:360-371
 *     `_calc_LQR_action` -- K as the reference holds it (K = -dlqr, f16_lqr_batch's output), x_ref = the CURRENT state with
 *     x_ref[4:7] = (p, q, r)_dem: track_mask = 0x70 and x_ref[4..6][ld] = the demands, u0 = u.initial_condition[1:];
 *   test_env.py:501-576 `test_LQR_lin`: u = -K' (x - x_ref) with K' = dlqr and a FIXED reference: track_mask = 0x1FF, K = -K', u0 NULL.
 * x9[9][ld] in place (MPC-state order, parameters.py:135); Ad[81][ld], Bd[27][ld], K[27][ld] row-major per aircraft; x_ref[9][ld]
 * (entries outside track_mask are not read); u0[3][ld] or NULL (= 0).  traj_x (may be NULL) [nsteps / traj_every][9][ld] and traj_u
 * (may be NULL) [..][3][ld] receive the state AFTER and the action OF every traj_every-th step (x_storage / u_storage of the
 * reference's loops).  No envelope, no saturation: the reference's loops have none (on the reference's own model the first loop
 * grows like exp(14 t) -- SURVEY.md 8-Q.3 -- and so does this one, fixture G13). */
int f16_rollout_lqr_linear(f16_ctx *ctx, double *x9, const double *Ad, const double *Bd, const double *K, const double *x_ref,
                           const double *u0, double *traj_x, double *traj_u, long B, long ld, int nsteps, int traj_every,
                           unsigned track_mask, void *stream);
/* env.py:152-193 _calc_xdot_na: x9[9][ld], u3[3][ld] scattered over x_full[18][ld] -> xdot9[9][ld] */
int f16_xdot_na_batch(f16_ctx *ctx, const double *x_full, const double *x9, const double *u3, double *xdot9,
                      int32_t *status, long B, long ld, double xcg, int fi_flag, unsigned flags, void *stream);

/* ---- (2) batched trim ---------------------------------------------------- */
/* env.py:198-292 F16.trim(h_t, v_t) for B flight conditions h[B], v[B] (device): the reference's Nelder-Mead
 * (scipy defaults, tol 1e-10, maxiter 5e4; initial guess env.py:265-271 unless h_x0[5] on the HOST is given)
 * -> x_trim[18][ld]; cost[B], iters[B], nfev[B], status[B] may be NULL. */
int f16_trim_batch(f16_ctx *ctx, const double *h, const double *v, double *x_trim, double *cost, int32_t *iters,
                   int32_t *nfev, int32_t *status, long B, long ld, double xcg, int fi_flag, unsigned flags,
                   int maxiter, const double *h_x0, void *stream);

/* ---- (2) batched control chain ---------------------------------------------------------- */
/* env.py:294-342 with _calc_xdot_na/_get_obs_na at each aircraft's own point: x9 = x[mpc idx] of x[18][ld],
 * u3 = u[1..3] of u[4][ld] (self.u._get_mpc_u(), env.py:348): Ac[81][ld] Bc[27][ld] Cc[81][ld]
 * (row-major element index = r*ncols+c); eps = 1e-5 in the reference (env.py:319). */
int f16_linearise_batch(f16_ctx *ctx, const double *x, const double *u, double *Ac, double *Bc, double *Cc,
                        int32_t *status, long B, long ld, double eps, double xcg, int fi_flag,
                        unsigned flags, void *stream);
/* scipy.signal.cont2discrete(zoh) (env.py:50,351): Ad[81][ld], Bd[27][ld] = expm([[A,B],[0,0]] dt) blocks */
int f16_c2d_batch(f16_ctx *ctx, const double *Ac, const double *Bc, double *Ad, double *Bd,
                  long B, long ld, double dt, void *stream);
/* env.py:45-46: the 18-state model.  linearise with the default _calc_xdot/get_obs: Ac[324][ld] (18x18), Bc[72][ld]
 * (18x4), Cc[180][ld] (10x18, observed states parameters.py:134); then cont2discrete(zoh): Ad[324][ld], Bd[72][ld]. */
int f16_linearise_full_batch(f16_ctx *ctx, const double *x, const double *u, double *Ac, double *Bc, double *Cc,
                             int32_t *status, long B, long ld, double eps, double xcg, int fi_flag,
                             unsigned flags, void *stream);
int f16_c2d_full_batch(f16_ctx *ctx, const double *Ac, const double *Bc, double *Ad, double *Bd,
                       long B, long ld, double dt, void *stream);
/* utils.py:219-245 dlqr with Q = Cd'Cd, R = I3 (env.py:353-356): K[27][ld] = -dlqr (3x9 row-major),
 * Pare[81][ld] = DARE solution (may be NULL). */
int f16_lqr_batch(f16_ctx *ctx, const double *Ad, const double *Bd, const double *Cd, double *K, double *Pare,
                  int32_t *status, long B, long ld, void *stream);
/* env.py:373-424 _calc_MPC_action for B aircraft: per aircraft (Ad,Bd,Cd) + current state x[18][ld]
 * + demands dem[3][ld] (p,q,r; written to x_ref[5:8] exactly as the reference does) -> first move
 * u_cmd[3][ld].  Dense condensed QP of horizon hzn (utils.py:21-167) solved as the reference's call solves it
 * (env.py:420-422: osqp.OSQP().setup(P, q, A, l, u, max_iter=40000, polish=False), every other setting at its default):
 * OSQP's published ADMM with Ruiz equilibration (`scaling` passes, D / E / c), rho = 0.1, sigma 1e-6, alpha 1.6,
 * termination on the UNSCALED residuals (eps 1e-3) every `check_every` iterations, rho re-estimated from the SCALED
 * residuals every `rho_every` iterations (OSQP's own interval is wall-clock based; 100 is its no-timer constant
 * ADAPTIVE_RHO_FIXED; must be a multiple of `check_every`, else F16_EINVAL: the estimate is formed from the residuals of a
 * termination test) and applied when it moves by more than 5x, primal-infeasibility certificate (-> NaN command +
 * F16_ST_QP_INFEASIBLE, as OSQP returns).  Rows of A with two infinite bounds (phi, theta, lf1) take part in the
 * equilibration and are then left out of the iteration (OSQP carries them with rho_min = 1e-6; they never bind).
 * Opt-in alternative (the builder's rule, faster on this family of QPs): scaling = 0, rho = 0 -> no equilibration and
 * the start value rho = 2 sqrt(tr P / tr A'A); scaling = 0, rho > 0 -> no equilibration, fixed start value.
 * Horizons: 1 <= hzn <= 150.  hzn <= 30 (equilibrated solves) runs one wavefront per aircraft, hzn <= 32 the 512-lane
 * register-resident solver, larger horizons (the reference's own sweep goes to 150, env.py:426-436) one 512-lane workgroup
 * per aircraft with the KKT inverse in the HBM workspace (the stream of its symmetric half, 0.93 MB per iteration at hzn = 150,
 * sets the pace; f16_mpc_hzn_sweep below solves all horizons of a sweep in one launch).
 * u_seq (may be NULL) gets the full [3*hzn][ld] sequence, info (may be NULL) gets [4][ld] = iterations, r_prim, r_dual
 * (unscaled), rho.
 * Nothing the results depend on is retained between calls: the QP workspace is allocated and freed per call, stream-ordered
 * on `stream`; calls on different streams of one context do not share buffers.  Under stream capture the call returns
 * F16_EINVAL (graph replays of the stream-ordered allocation were measured unreliable on ROCm 7.2): capture
 * f16_mpc_plan_solve instead, whose workspace lives with the plan.
 * Scheduling only: workgroups are dispatched longest-first by the iteration counts of the previous call of the same
 * batch size on the same stream (results do not depend on it; F16_MPC_DISPATCH_ORDER=0 keeps the caller's order). */
/* Weights, reference and bounds of the QP as ARGUMENTS -- what utils.py:21 `setup_OSQP(x_ref, A, B, Q, R, hzn, dt, x, act_states,
 * x_lb, x_ub, u_lb, u_ub, udot_lb, udot_ub)` and utils.py:219 `dlqr(A, B, Q, R)` take; env.py:373-424 fills them with
 * constants (Q = Cd'Cd, R = I, x_ref = x with x_ref[5:8] = demands, the boxes of parameters.py:59-129), and the entry points without
 * the `_w` suffix do the same.  The `_w` entry points take a HOST pointer h_w to this struct (NULL = env.py's constants, bit for
 * bit the plain entry point) and, where a reference enters, a DEVICE pointer x_ref[9][ld] (NULL = env.py:380-383 from `dem`; with
 * x_ref given `dem` may be NULL).  Weights and bounds are uniform over the batch; x_ref is per aircraft.  MPC-state order:
 * phi, theta, alpha, beta, p, q, r, lf1, lf2 (parameters.py:135); +-INFINITY = no bound.
 * Solvers: the six state rows the reference bounds (alpha, beta, p, q, r, lf2) stay the bounded ones -- their VALUES are free
 * (one side may be infinite), but a finite bound on phi / theta / lf1 or no bound at all on one of the six is F16_EINVAL;
 * f16_mpc_qp_debug_w (the QP alone, dense, in the reference's form) takes any pattern. */
typedef struct f16_mpc_weights {
  int q_from_cd;                 /* 1: Q = Cd'Cd per aircraft (env.py:389) and Q[] is ignored */
  double Q[81];                  /* 9 x 9 row-major, symmetric positive semidefinite (the author's own alternative: env.py:391-401) */
  double R[9];                   /* 3 x 3 row-major, symmetric positive definite (env.py:403-407) */
  double x_lb[9], x_ub[9];       /* parameters.py _vec_mpc_x_lb / _ub */
  double u_lb[3], u_ub[3];       /* _vec_mpc_u_lb / _ub */
  double udot_lb[3], udot_ub[3]; /* _vec_mpc_udot_lb / _ub */
} f16_mpc_weights;
void f16_mpc_default_weights(f16_mpc_weights *w);      /* env.py's constants */
int f16_lqr_batch_w(f16_ctx *ctx, const double *Ad, const double *Bd, const double *Cd, const f16_mpc_weights *h_w, double *K,
                    double *Pare, int32_t *status, long B, long ld, void *stream);        /* utils.py:219: Q, R of h_w */

typedef struct f16_qp_settings {
  double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf;
  int max_iter, check_every, rho_every, adaptive_rho;
  int scaling;     /* Ruiz equilibration passes (OSQP default 10; 0 = none) */
} f16_qp_settings;
void f16_qp_default_settings(f16_qp_settings *s);
int f16_mpc_batch(f16_ctx *ctx, const double *Ad, const double *Bd, const double *Cd, const double *x,
                  const double *dem, double *u_cmd, double *u_seq, double *info, int32_t *status,
                  long B, long ld, int hzn, double dt, const f16_qp_settings *s, void *stream);
int f16_mpc_batch_w(f16_ctx *ctx, const double *Ad, const double *Bd, const double *Cd, const double *x, const double *dem,
                    const double *x_ref, const f16_mpc_weights *h_w, double *u_cmd, double *u_seq, double *info, int32_t *status,
                    long B, long ld, int hzn, double dt, const f16_qp_settings *s, void *stream);
/* env.py:426-436 `_calc_constr_checking_hzn`: the first move of calc_MPC_action for the SAME states, demands and model at
 * every horizon hzn_lo..hzn_hi (1 <= hzn_lo <= hzn_hi <= 150) as one call.  u_cmd [hzn_hi - hzn_lo + 1][3][ld], info (may be
 * NULL) [..][4][ld], status (may be NULL, OR-ed into) [..][ld]; slice k = hzn - hzn_lo holds exactly what f16_mpc_batch returns
 * for that horizon (bit-identical: same kernels).  Horizons <= 32 run one after the other; the longer ones are built per
 * horizon and solved by ONE launch over every (horizon, aircraft) pair, longest horizon first, so that the few solves that
 * need tens of thousands of iterations do not hold a launch of their own.  Workspace: stream-ordered, in groups of horizons
 * of at most F16_SWEEP_WS_GB (default 32) GB (4.45 MB per aircraft at N = 150: P, the QP extras and the solver's 3.59 MB of operands).  Not capturable.
 * Scheduling only: a repeated sweep of the same horizons on the same stream takes its pairs costliest-first by the iteration
 * counts of the previous one (results do not depend on it; F16_MPC_DISPATCH_ORDER=0 switches it off). */
int f16_mpc_hzn_sweep(f16_ctx *ctx, const double *Ad, const double *Bd, const double *Cd, const double *x,
                      const double *dem, double *u_cmd, double *info, int32_t *status, long B, long ld, int hzn_lo,
                      int hzn_hi, double dt, const f16_qp_settings *s, void *stream);
/* Prepared plans.  The reference freezes (Ad,Bd,Cd) at construction (env.py:49-60) yet rebuilds the whole QP on every
 * _calc_MPC_action call (utils.py:21-167 inside env.py:373-424).  A plan computes the model-only part once -- DARE,
 * terminal weight, prediction blocks, P; with scaling = 0 also the start value of rho and the inverse of the KKT matrix
 * (73.7 KB per aircraft) -- and f16_mpc_plan_solve does what is left per call: the state-dependent vectors, the
 * iterations and, with OSQP's defaults, equilibration + factorisation (OSQP's scaling looks at q, i.e. at the state of the
 * call).  Results are bit-identical to f16_mpc_batch with the same settings.  hzn <= 40 (33..40: the model part is kept, every solve runs the
 * long-horizon solver; no warm start there).  (Ad,Bd,Cd) are read during
 * f16_mpc_plan_create only. */
typedef struct f16_mpc_plan f16_mpc_plan;
/* The reference's closed MPC loop (test_env.py:480-495; BASELINE config 5) as ONE launch on a prepared plan (hzn <= 30, OSQP's
 * default settings: scaling > 0; every solve starts cold as the reference's does, unless f16_mpc_plan_warm_start switched the plan's
 * opt-in warm start on: then step t starts from the solution of step t - 1):  per step  cmd = _calc_MPC_action(p, q, r, hzn);  u.values[1:] = cmd;  step(u.values).
 * x[18][ld] in place; u[4][ld] = u.values in place (thrust command held, u[1:4] receives every step's command and ends up
 * holding the last one, as the reference's u.values does); dem[3][ld].  traj (may be NULL) [nsteps / traj_every][18][ld]: the state
 * after every traj_every-th step; cmd_traj (may be NULL) [nsteps][3][ld]: what calc_MPC_action returned at each step; iters_traj
 * (may be NULL) [nsteps][ld] int32: its ADMM iterations; status[ld] (may be NULL): sticky OR of the step's and the solves' bits.
 * Work items are (step, aircraft) pairs drawn from one ticket counter by one wavefront per SIMD; a wavefront builds the state-
 * dependent vectors of the QP, solves it, writes the command and takes the Euler step of its pair, so that no step waits for another
 * aircraft's solve (the host loop joins the batch after every solve).  Results: bit-identical to the host loop
 * (f16_mpc_plan_solve + f16_rollout(..., nsteps = 1, flags | F16_FLAG_ONE_LANE) per step) for every aircraft that stays inside
 * its envelope.  Per-aircraft conditions:
 *   - QP certified infeasible: the command is NaN, as OSQP returns it (F16_ST_QP_INFEASIBLE): the actuator models propagate it
 *     (np.clip, utils.py:308-330), the surface states turn NaN (F16_ST_NONFINITE) and every later solve of that aircraft is skipped
 *     (NaN command, zero iterations) -- the reference's own loop is left with NaN states from there.  F16_FLAG_HOLD_COMMAND keeps
 *     the previous command instead and the aircraft flies on.
 *   - outside the envelope at the start of a step (env.py:117-124: the reference exit()s): frozen, flagged
 *     (F16_ST_ENVELOPE | F16_ST_ENV_STATE(k)) and NOT solved for any more: cmd_traj holds NaN, iters_traj 0, u keeps its value.
 * Not capturable on its first call on a plan (allocates the ticket / progress counters).  A plan serves one call at a time (its
 * workspace and these counters are per plan): order calls on one plan through one stream.  nsteps x B < 2^32 per call. */
int f16_rollout_mpc(f16_mpc_plan *plan, double *x, double *u, const double *dem, double *traj, double *cmd_traj,
                    int32_t *iters_traj, int32_t *status, int nsteps, int traj_every, double xcg, int fi_flag, unsigned flags,
                    void *stream);
int f16_mpc_plan_create(f16_ctx *ctx, f16_mpc_plan **plan, const double *Ad, const double *Bd, const double *Cd,
                        long B, long ld, int hzn, double dt, const f16_qp_settings *s, void *stream);
int f16_mpc_plan_solve(f16_mpc_plan *plan, const double *x, const double *dem, double *u_cmd, double *u_seq,
                       double *info, int32_t *status, void *stream);
int f16_mpc_plan_create_w(f16_ctx *ctx, f16_mpc_plan **plan, const double *Ad, const double *Bd, const double *Cd,
                          const f16_mpc_weights *h_w, long B, long ld, int hzn, double dt, const f16_qp_settings *s, void *stream);
int f16_mpc_plan_solve_w(f16_mpc_plan *plan, const double *x, const double *dem, const double *x_ref, double *u_cmd,
                         double *u_seq, double *info, int32_t *status, void *stream);
/* Optional: start each solve of the plan from the previous solve's x, z, y (what OSQP does by default inside ONE
 * solver object; the reference builds a new object per call, i.e. always starts cold -- so this is off by default and
 * results then differ from the cold start within the termination tolerance).  Switching it on or off forgets the
 * stored solution; a solve that did not converge is not reused. */
int f16_mpc_plan_warm_start(f16_mpc_plan *plan, int on);
void f16_mpc_plan_destroy(f16_mpc_plan *plan);      /* waits for the plan's own stream only */

/* utils.py:21-167 setup_OSQP alone for aircraft b (tests): h_P[n*n] h_q[n] h_A[(m rows)*n] h_l h_u on the
 * host, n = 3*hzn, rows = 15*hzn, reference row order. */
int f16_mpc_qp_debug(f16_ctx *ctx, const double *Ad, const double *Bd, const double *Cd, const double *x,
                     const double *dem, long b, long ld, int hzn, double dt,
                     double *h_P, double *h_q, double *h_A, double *h_l, double *h_u);

int f16_mpc_qp_debug_w(f16_ctx *ctx, const double *Ad, const double *Bd, const double *Cd, const double *x, const double *dem,
                       const double *x_ref, const f16_mpc_weights *h_w, long b, long ld, int hzn, double dt,
                       double *h_P, double *h_q, double *h_A, double *h_l, double *h_u);

/* Tests: inverse of B packed (lower triangle, row-major) SPD n x n matrices, n <= 96, on the device through the
 * KKT-inverse routine of the MPC solver (blocked sweep on the fp64 matrix cores, v_mfma_f64_16x16x4_f64).
 * packed [B][n(n+1)/2], out [B][n*n] (device pointers); NaN where a pivot block was not positive definite. */
int f16_debug_spd_inverse(f16_ctx *ctx, const double *packed, double *out, int n, long B, void *stream);

#ifdef __cplusplus
}
#endif
#endif
