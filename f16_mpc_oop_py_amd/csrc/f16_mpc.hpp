// f16_mpc.hpp -- shared definitions of the MPC kernels (f16_control.hip: QP build + generic ADMM;
// f16_mpc_solve.hip: register-resident ADMM for N <= 32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/f16_hip.h"
#include "f16_ctx.h"

namespace f16 {

// The builder's opt-in start value of rho (f16_qp_settings.rho <= 0, scaling = 0): RHO_AUTO_SCALE * sqrt(tr P / tr A'A).
// The square root balances the two terms of P + rho A'A of the UNscaled QP; the factor is tuned on the config-4 workload:
// mean ADMM iterations 50 -> 38 (checked every 25), worst case 125 -> 75, against factor 1 (the test-side restatement
// uses the same rule).
constexpr double RHO_AUTO_SCALE = 2.0;
// OSQP constants (osqp/include/constants.h as recalled in SURVEY.md Appendix C; the test-side restatement uses the same)
constexpr double OSQP_MIN_SCALING = 1e-4, OSQP_MAX_SCALING = 1e4, OSQP_RHO_MIN = 1e-6, OSQP_RHO_MAX = 1e6;
constexpr double OSQP_RHO_TOL = 1e-4, OSQP_RHO_EQ_OVER_RHO_INEQ = 1e3, OSQP_ADAPTIVE_RHO_TOLERANCE = 5.0;
__host__ __device__ inline double osqp_limit_scaling(double v) {
  v = v < OSQP_MIN_SCALING ? 1.0 : v;
  return v > OSQP_MAX_SCALING ? OSQP_MAX_SCALING : v;
}
constexpr int MAXN = 40;                         // horizon limit of the LDS-resident solver
constexpr int MAXT = (12 * MAXN + 63) / 64;      // constraint rows per lane
constexpr int BIG_MAXN = 150;                    // horizon limit of the slow path (row values and KKT inverse in HBM): the
                                                 // reference's own sweep range, env.py:426-436
static __constant__ int SROW[6] = {2, 3, 4, 5, 6, 8};   // MPC states that carry bounds: alpha,beta,p,q,r,lf2 (parameters.py:59-95,135)

// Weights and bounds of the QP -- the arguments utils.py:21 `setup_OSQP(x_ref, A, B, Q, R, hzn, dt, x, act_states, x_lb, x_ub,
// u_lb, u_ub, udot_lb, udot_ub)` takes, which env.py:373-424 fills with constants.  By value in the kernel arguments (uniform over
// the batch); mpc_default_prob = env.py's constants: Q = Cd'Cd per aircraft (env.py:389), R = I (env.py:405-407), the bounds of
// parameters.py:59-129.  Infinite bounds arrive as +-1e30 (OSQP's own infinity); the rows phi, theta, lf1 carry none.
struct MpcProb {
  int custom_q, custom_r;   // 0: Q = Cd'Cd formed per aircraft / R = I (the default path, bit for bit what it always was)
  double Q[81], R[9], Rinv[9];
  double slb[6], sub[6];    // kept state rows (SROW order)
  double ulb[3], uub[3], rlb[3], rub[3];
};
inline void mpc_default_prob(MpcProb *p) {
  *p = MpcProb{};
  const double slb[6] = {-20., -30., -300., -100., -50., 0.}, sub[6] = {90., 30., 300., 100., 50., 25.};
  const double ulb[3] = {-25., -21.5, -30.}, uub[3] = {25., 21.5, 30.}, rlb[3] = {-60., -80., -120.}, rub[3] = {60., 80., 120.};
  for (int i = 0; i < 6; ++i) { p->slb[i] = slb[i]; p->sub[i] = sub[i]; }
  for (int i = 0; i < 3; ++i) { p->ulb[i] = ulb[i]; p->uub[i] = uub[i]; p->rlb[i] = rlb[i]; p->rub[i] = rub[i]; p->R[4 * i] = 1.0; p->Rinv[4 * i] = 1.0; }
}
int mpc_fill_prob(MpcProb *p, const f16_mpc_weights *w);      // host: validation + OSQP's infinity; F16_OK / F16_EINVAL

struct MpcArgs {
  const double *Ad, *Bd, *Cd, *x, *dem;
  const double *xref;         // [9][ld] reference of the tracking cost (utils.py:21 `x_ref`), or null: x with x[5:8] = dem (env.py:380-383)
  MpcProb pb;
  double *ucmd, *useq, *info;
  int32_t *status;
  double *Ppk;                // workspace [B][np] packed P (A'A is never stored: the solvers form the weighted Gram themselves)
  double *ext;                // workspace [B][mpc_ext_doubles(N)]: q | G | pred | A | Q | Qbar | rho, ok | scaling (setup kernel -> solver / debug)
  double *tiles;              // prepared plans without equilibration: [B][MPC_TILE_DOUBLES] the re-laid KKT inverse
  double *bigws;              // horizons beyond MAXN: [B][mpc_big_doubles(N)] seven per-row vectors | packed KKT inverse
  double *gramws;             // [B][MPC_TILE_DOUBLES] A'WA as matrix-core tiles, written by a solve's first factorisation and
                              // re-read by its rho updates (the Gram product does not depend on rho); may be null (recomputed)
  double *pblk;               // wavefront solver: [B][WAVE_PBLK_DOUBLES] P as the symmetric block image (termination test) | the lanes'
                              // Toeplitz operands [36][64] double2; plans: the
                              // `tiles` block (unused with equilibration)
  int wave_ruiz;              // wavefront solver: equilibrate in the solver itself (else: k_mpc_fast mode 3 went before it)
  unsigned wave_stride;       // wavefront solver without a dispatch order: workgroup w solves aircraft (w * wave_stride) % B, a
                              // stride coprime to B (0: the identity)
  unsigned *wave_queue;       // wavefront solver: work-queue counter (zeroed on the stream before the launch): the grid is one
                              // workgroup per SIMD and every workgroup takes the next aircraft of the dispatch order when it is free;
                              // null: one workgroup per aircraft (the hardware's own distribution)
  int mode;                   // 0 one-shot; 1 prepare (build + factor, keep everything, no iterations); 2 solve from a plan;
                              // 3 equilibration only: D | E | c -> gramws + WAVE_SCAL_OFF (for the wavefront solver)
  double *warm;               // plans with warm start: [B][MPC_WARM_DOUBLES] x, z, y of the previous solve (per lane)
  int warm_load;              // start from them (else from zero, as the reference's fresh OSQP object does)
  const int32_t *order;       // plans: workgroup -> aircraft map (longest solve of the previous call first), or null
  int32_t *iters_out;         // plans: iteration count of this solve per aircraft (input of the next call's order)
  long B, ld;
  int N;
  double dt;
  f16_qp_settings s;
};


// per-aircraft extras written by the setup kernel: q[n] | G[27N] | pred[9N] | A[81] Q[81] Qbar[81] | rho, ok | nonfinite, pad
__host__ __device__ inline size_t mpc_ext_doubles(int N) { return (size_t)3 * N + 27 * N + 9 * N + 243 + 4; }
__host__ __device__ inline size_t mpc_ext_model(int N) { return (size_t)3 * N + 27 * N + 9 * N; }      // offset of A | Q | Qbar | rho
__host__ __device__ inline size_t mpc_ext_flag(int N) { return mpc_ext_model(N) + 245; }              // 1.0: the state of this call is not finite
// A call whose state (x9, the actuator positions, the reference) is not finite has no QP to solve: OSQP would iterate on NaN up to
// max_iter (40,000 iterations, env.py:421) and hand back NaN.  Same answer here without the iterations: the build kernel raises the
// flag, every solver asks for it first and writes NaN commands, zero iterations and F16_ST_NONFINITE.  (A NaN state is what the
// reference's loop is left with one step after an infeasible QP: NaN command -> NaN actuator state, utils.py:308-330.)
// (Both take the few fields they need BY VALUE: handing the kernel-argument struct itself to a function -- even an inlined one with a
//  loop over its pointers -- made the compiler keep a 1.3 KB copy of it in scratch in k_mpc_fast, and every later field access a scratch
//  load: +26 % on the solves without equilibration until it was noticed in the bench record.)
__device__ __forceinline__ bool mpc_job_nonfinite(const double *ext, int N, long b) {
  return ext && ext[(size_t)b * mpc_ext_doubles(N) + mpc_ext_flag(N)] != 0.0;
}
__device__ __forceinline__ void mpc_write_nonfinite(double *ucmd, double *useq, double *info, int32_t *iters_out, int32_t *status, long ld,
                                                    int N, double rho, long b, int tid, int nthreads) {
  const double nan_ = __builtin_nan("");
  for (int e = tid; e < 3 * N; e += nthreads) {
    if (e < 3) ucmd[e * ld + b] = nan_;
    if (useq) useq[e * ld + b] = nan_;
  }
  if (tid == 0) {
    if (iters_out) iters_out[b] = 0;
    if (info) { info[0 * ld + b] = 0.0; info[1 * ld + b] = nan_; info[2 * ld + b] = nan_; info[3 * ld + b] = rho; }
    if (status) status[b] |= F16_ST_NONFINITE;
  }
}
__host__ __device__ inline size_t mpc_big_doubles(int N) {      // (see k_mpc<false, true>)
  const size_t rows = (size_t)64 * ((12 * N + 63) / 64), n = (size_t)3 * N;
  return 7 * rows + n * (n + 1) / 2;
}
constexpr int MPC_WARM_DOUBLES = 3 * 512;
constexpr int MPC_TILE_DOUBLES = 6 * 6 * 4 * 64;   // six tile rows x six tiles x four accumulator registers x 64 lanes

// f16_mpc_wave.hip: one wavefront per aircraft (N <= 30, equilibrated solves); its per-aircraft workspace is the `gramws`
// block: A'WA tiles in front, D | E | c of the equilibration (written by k_mpc_fast in mode 3) at WAVE_SCAL_OFF
constexpr int WAVE_MAXN = 30;
constexpr int WAVE_SCAL_OFF = 7936;
constexpr int WAVE_PBLK_DOUBLES = 2 * 36 * 64 * 2;      // P block image | per-lane image of the Toeplitz operands
bool mpc_wave_enabled(const MpcArgs &a);
int mpc_wave_solve_launch(f16_ctx *ctx, const MpcArgs &a, void *stream);
// the closed-loop rollout (f16_rollout_mpc): per-call arguments beside the plan's MpcArgs
struct RolloutMpcCall {
  double *x, *u;
  const double *dem;
  double *traj, *cmd_traj;
  int32_t *iters_traj, *status;
  void *sync;                 // [8 bytes ticket counter | B x int32 progress], zeroed by the launch
  double *warm;               // the plan's warm-start buffer (opt-in), or null: every solve starts cold, as the reference's does
  int warm_load;              // step 0 starts from what the plan's previous call left (later steps always start from the step before)
  int T, every;
  double xcg;
  int fi;
  unsigned flags;
};
int mpc_wave_rollout_launch(f16_ctx *ctx, const MpcArgs &a, const RolloutMpcCall &c, void *stream);

// f16_mpc_big.hip: one 512-lane workgroup per aircraft, 33 <= N <= 150 (operands in the HBM workspace `bigws`)
size_t mpc_big_ws_doubles(int N);
int mpc_big_opt_in();      // once per device: the kernel's dynamic-LDS limit (not legal under stream capture)
int mpc_big_solve_launch(f16_ctx *ctx, const MpcArgs &a, void *stream);
// the same states at every horizon lo..hi in one launch (longest first); per horizon [B][mpc_big_sweep_job_doubles(N)] behind base
size_t mpc_big_sweep_job_doubles(int N);
int mpc_big_sweep_launch(f16_ctx *ctx, const MpcArgs &a, int lo, int hi, double *base, double *ucmd, double *info, int32_t *status,
                         unsigned int *next, int32_t *iters, const int32_t *order, void *stream);
// next: the work-queue counter, zero at the launch; iters (may be null): [pairs] iteration count of pair (hi - N) * B + aircraft;
// order (may be null): the pairs in the order the queue hands them out (mpc_big_sweep_order_launch: costliest first)
int mpc_big_sweep_order_launch(const int32_t *iters, int32_t *order, long pairs, long B, int hi, void *stream);

// f16_mpc_solve.hip
constexpr int FAST_MAXN = 32;
int mpc_fast_solve_launch(f16_ctx *ctx, const MpcArgs &a, void *stream);
int mpc_plan_order_launch(const int32_t *iters, int32_t *order, long B, int check_every, void *stream);
int mpc_first_order_launch(const MpcArgs &a, int32_t *scratch, int32_t *order, void *stream);      // first call: longest-first by ||q||_inf

}  // namespace f16
