// f16_dynamics.hip -- batched F-16 dynamics kernels for gfx950 + their C-ABI launchers.
//
//   k_xdot      env.py:65-103 _calc_xdot for B aircraft                (f16_xdot_batch)
//   k_nlplant   C/nlplant.c:23-457 incl. accels outputs                (f16_nlplant_batch, drop-in Nlplant)
//   k_rollout   env.py:105-130 step x nsteps, state in registers       (f16_rollout)
//   k_xdot_na   env.py:152-193                                         (f16_xdot_na_batch)
//   k_rollout_exact       the same step x nsteps through the out-of-line euler_step_exact: one kernel for every batch size
//                         (F16_FLAG_ONE_LANE; the step the closed MPC loop f16_rollout_mpc takes)
//   k_rollout_lqr_linear  test_env_mk2.py:46-62 / test_env.py:501-576: the linear-model LQR loops (f16_rollout_lqr_linear)
//
// Mapping: one lane = one aircraft; state-major [k][ld] arrays so a wave's 64 lanes read 512 contiguous
// bytes per state component.  Every workgroup first copies the 112,928-byte fp64 table image from
// global memory (L2-resident after the first block) into LDS with 16-byte loads; all 168 table-vertex
// fetches of an evaluation are then LDS reads.  The kernels are bound by fp64 VALU issue and, at the
// reference's batch of 4096 (= 64 wavefronts), by single-wave latency; algorithmic HBM traffic is
// 320 B per aircraft-step for the state-resident step and 144 B per stored trajectory sample.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/f16_hip.h"
#include "f16_ctx.h"
#include "f16_plant.hpp"
#include "f16_plant_quad.hpp"

namespace f16 {

struct DynArgs {
  const double *tab;    // hifi image, global
  const int *tab32;     // hifi image as scaled integers (f16_tables.h, namespace i32), global
  const double *lofi;   // lofi image, global
  const double *x;      // [18][ld] (or x_full for xdot_na)
  const double *u;      // [4][ld]  (x9 for xdot_na)
  const double *u3;     // xdot_na only
  double *out;          // xdot / x (rollout: in place) / xdot9
  double *traj;
  int32_t *status;
  long B, ld;
  int nsteps, traj_every;
  double dt, xcg;
  int fi;
  unsigned flags;
  // closed loop under the LQR law (f16_rollout_lqr; env.py:360-371 inside the loop of test_env_mk2.py:70-85)
  const double *K;      // [27][ld] the reference's K = -dlqr (3 x 9 row-major), or null
  const double *dem;    // [3][ld] p, q, r demands
  double *u_out;        // [4][ld] the action of the last step (self.u.values after the loop), may be null
};

// env.py:360-371 `_calc_LQR_action`: u = -K (x_ref - x) + u0 with x_ref = x except x_ref[4:7] = (p, q, r)_dem -- x_ref - x is
// exactly zero outside the three rate entries, so row i of the product is K[i][4..6] . (dem - (p, q, r)); the thrust command is
// not an LQR output (test_env_mk2.py:76-79 overwrites u.values[1:] only).  kr = K[i][4..6] of the lane's row.
F16_DEV double lqr_action(double kr0, double kr1, double kr2, double e0, double e1, double e2, double u0) {
  return -(kr0 * e0 + kr1 * e1 + kr2 * e2) + u0;
}

// Cooperative copy of the table image into LDS (16 B per lane per load).
__device__ __forceinline__ void stage_tables(double *lds, const double *__restrict__ g) {
  const double2 *src = reinterpret_cast<const double2 *>(g);
  double2 *dst = reinterpret_cast<double2 *>(lds);
  for (int i = threadIdx.x; i < TABLE_IMAGE_DOUBLES / 2; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
}

template <int BLOCK, int FI>
__global__ __launch_bounds__(BLOCK) void k_xdot(DynArgs a) {
  __shared__ __attribute__((aligned(16))) double tab[TABLE_IMAGE_DOUBLES];
  if (a.fi == 1) stage_tables(tab, a.tab);
  for (long b = (long)blockIdx.x * BLOCK + threadIdx.x; b < a.B; b += (long)gridDim.x * BLOCK) {
    double x[18], u[4], xd[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) x[k] = a.x[k * a.ld + b];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = a.u[k * a.ld + b];
    int st = 0;
    calc_xdot<FI>((const double *)tab, a.lofi, x, u, xd, a.xcg, a.fi, a.flags, st);
#pragma unroll
    for (int k = 0; k < 18; ++k) a.out[k * a.ld + b] = xd[k];
    if (a.status) a.status[b] |= st;
  }
}

// Nlplant proper.  TAB_LDS=false reads the image straight from global/L2: used for tiny batches (the
// single-aircraft drop-in call), where staging 113 KB would cost more than the ~170 gathers it saves.
template <int BLOCK, bool TAB_LDS>
__global__ __launch_bounds__(BLOCK) void k_nlplant(DynArgs a) {
  __shared__ __attribute__((aligned(16))) double tab[TAB_LDS ? TABLE_IMAGE_DOUBLES : 2];
  if (TAB_LDS && a.fi == 1) stage_tables(tab, a.tab);
  const double *T = TAB_LDS ? (const double *)tab : a.tab;
  for (long b = (long)blockIdx.x * BLOCK + threadIdx.x; b < a.B; b += (long)gridDim.x * BLOCK) {
    double x[17], xd[18], qbar, ps;
#pragma unroll
    for (int k = 0; k < 17; ++k) x[k] = a.x[k * a.ld + b];
    int st = 0;
    plant<true>(T, a.lofi, x, xd, a.xcg, a.fi, a.flags, st, qbar, ps);
#pragma unroll
    for (int k = 0; k < 18; ++k) a.out[k * a.ld + b] = xd[k];
    if (a.status) a.status[b] |= st;
  }
}

// One lane = one aircraft, the whole rollout in registers.  TP = the table image the lookups read (plant header: fp64
// values or scaled integers).
// INCT (default numerics only; the strict build and every single-evaluation kernel keep the full sincos): the five sin / cos pairs
// of a step are carried from step to step (trig_advance: rotated by the exact increment of their angles, re-evaluated exactly every
// 32nd step) instead of evaluated from scratch -- ~155 of the ~1,555 wave-instructions of a hifi step.  tgs: lane-indexed LDS slots.
#if defined(F16_FAST_TRIG) && !defined(F16_NO_INC_TRIG)
constexpr bool INC_TRIG = true;
#else
constexpr bool INC_TRIG = false;
#endif
template <int BLOCK, int FI, typename TP, bool LQR = false, bool INCT = false>
__device__ __forceinline__ void rollout_lanes(const DynArgs &a, TP T, double (*us)[BLOCK], double (*kq)[BLOCK] = nullptr,
                                              double (*tgs)[BLOCK] = nullptr) {
  for (long b = (long)blockIdx.x * BLOCK + threadIdx.x; b < a.B; b += (long)gridDim.x * BLOCK) {
    double x[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) x[k] = a.out[k * a.ld + b];
#pragma unroll
    for (int k = 0; k < 4; ++k) us[k][threadIdx.x] = a.u[k * a.ld + b];
    double ul[3] = {us[1][threadIdx.x], us[2][threadIdx.x], us[3][threadIdx.x]};   // (u_out of an aircraft that never steps: u0)
    if (LQR) {   // K[i][4..6] and the demands: lane-indexed LDS slots like the inputs (constant over the rollout, used once per step)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) kq[3 * i + j][threadIdx.x] = a.K[(9 * i + 4 + j) * a.ld + b];
#pragma unroll
      for (int j = 0; j < 3; ++j) kq[9 + j][threadIdx.x] = a.dem[j * a.ld + b];
    }
    int st = a.status ? a.status[b] : 0;
    double *tr = a.traj ? a.traj + b : nullptr;
    int until_store = a.traj_every;
    bool stale = true;                                   // (INCT) the first step evaluates the five pairs exactly
    for (int t = 0; t < a.nsteps; ++t) {
      // env.py:117-124: the reference exit()s; here the aircraft is frozen and flagged
      if (!(a.flags & FLAG_NO_ENVELOPE) && outside_envelope(x)) st |= ST_ENVELOPE;
      if (!(st & ST_ENVELOPE)) {
        double xd[18], u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = us[k][threadIdx.x];
        if (LQR) {
          const double e0 = kq[9][threadIdx.x] - x[9], e1 = kq[10][threadIdx.x] - x[10], e2 = kq[11][threadIdx.x] - x[11];
#pragma unroll
          for (int i = 0; i < 3; ++i)
            ul[i] = u[1 + i] = lqr_action(kq[3 * i][threadIdx.x], kq[3 * i + 1][threadIdx.x], kq[3 * i + 2][threadIdx.x], e0, e1, e2, u[1 + i]);
        }
        if (INCT) {
          const TrigSlots ts{&tgs[0][threadIdx.x], BLOCK};
          if (stale || (t & 31) == 0) { Trig5 g; trig_exact(x, g); trig_store(ts, g); }
          calc_xdot<FI, TP, true>(T, a.lofi, x, u, xd, a.xcg, a.fi, a.flags, st, &ts);
          const double xo5[5] = {x[7], x[8], x[4], x[3], x[5]};
#pragma unroll
          for (int k = 0; k < 18; ++k) x[k] += xd[k] * a.dt;   // env.py:126
          stale = trig_advance(xo5, x, ts);
        } else {
        calc_xdot<FI>(T, a.lofi, x, u, xd, a.xcg, a.fi, a.flags, st);
#pragma unroll
        for (int k = 0; k < 18; ++k) x[k] += xd[k] * a.dt;   // env.py:126
        }
      }
      if (tr && --until_store == 0) {
        until_store = a.traj_every;
#pragma unroll
        for (int k = 0; k < 18; ++k) __builtin_nontemporal_store(x[k], tr + k * a.ld);
        tr += 18 * a.ld;
      }
    }
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 18; ++k) finite = finite && isfinite(x[k]);
    if (!finite) st |= ST_NONFINITE;
    // WHICH states were outside: a frozen aircraft keeps the state it was frozen with, so the bits are formed here, once
    if (st & ST_ENVELOPE) st |= envelope_state_bits(x);
#pragma unroll
    for (int k = 0; k < 18; ++k) a.out[k * a.ld + b] = x[k];
    if (a.status) a.status[b] = st;
    if (LQR && a.u_out) {
      a.u_out[b] = us[0][threadIdx.x];
#pragma unroll
      for (int i = 0; i < 3; ++i) a.u_out[(1 + i) * a.ld + b] = ul[i];
    }
  }
}

// F16_FLAG_ONE_LANE: one lane per aircraft, 64-lane workgroups, every step through the out-of-line euler_step_exact (table image read
// from global memory: no LDS staging) -- ONE instruction sequence for every batch size, the one the closed MPC loop (f16_rollout_mpc)
// steps with.  Same rules as rollout_lanes (envelope freeze, status bits, trajectory samples, u_out); a reproducibility path, not a
// fast one.
template <bool LQR>
__global__ __launch_bounds__(64) void k_rollout_exact(DynArgs a) {
  for (long b = (long)blockIdx.x * 64 + threadIdx.x; b < a.B; b += (long)gridDim.x * 64) {
    double x[18], u[4];
#pragma unroll
    for (int k = 0; k < 18; ++k) x[k] = a.out[k * a.ld + b];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = a.u[k * a.ld + b];
    double kq[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, u0[3] = {u[1], u[2], u[3]};
    if (LQR) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) kq[3 * i + j] = a.K[(9 * i + 4 + j) * a.ld + b];
#pragma unroll
      for (int j = 0; j < 3; ++j) kq[9 + j] = a.dem[j * a.ld + b];
    }
    int st = a.status ? a.status[b] : 0;
    double *tr = a.traj ? a.traj + b : nullptr;
    int until_store = a.traj_every;
    for (int t = 0; t < a.nsteps; ++t) {
      if (!(a.flags & FLAG_NO_ENVELOPE) && outside_envelope(x)) st |= ST_ENVELOPE;      // env.py:117-124
      if (!(st & ST_ENVELOPE)) {
        if (LQR) {
          const double e0 = kq[9] - x[9], e1 = kq[10] - x[10], e2 = kq[11] - x[11];
#pragma unroll
          for (int i = 0; i < 3; ++i) u[1 + i] = lqr_action(kq[3 * i], kq[3 * i + 1], kq[3 * i + 2], e0, e1, e2, u0[i]);
        }
        euler_step_exact(a.tab, a.lofi, x, u, a.dt, a.xcg, a.fi, a.flags, &st);
      }
      if (tr && --until_store == 0) {
        until_store = a.traj_every;
#pragma unroll
        for (int k = 0; k < 18; ++k) __builtin_nontemporal_store(x[k], tr + k * a.ld);
        tr += 18 * a.ld;
      }
    }
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 18; ++k) finite = finite && isfinite(x[k]);
    if (!finite) st |= ST_NONFINITE;
    if (st & ST_ENVELOPE) st |= envelope_state_bits(x);
#pragma unroll
    for (int k = 0; k < 18; ++k) a.out[k * a.ld + b] = x[k];
    if (a.status) a.status[b] = st;
    if (LQR && a.u_out) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a.u_out[i * a.ld + b] = u[i];
    }
  }
}

template <int BLOCK, int FI, bool LQR = false>
__global__ __launch_bounds__(BLOCK) void k_rollout(DynArgs a) {
  __shared__ __attribute__((aligned(16))) double tab[TABLE_IMAGE_DOUBLES];
  // the four inputs of a lane are constant over the rollout and used once per step: kept in lane-indexed (conflict-free) LDS
  // slots rather than in eight registers that the 512-lane instantiation (256 registers per lane) spilled and reloaded per step
  __shared__ double us[4][BLOCK];
  __shared__ double kq[LQR ? 12 : 1][LQR ? BLOCK : 1];
  constexpr bool INCT = INC_TRIG && BLOCK <= 256 && !(LQR && BLOCK == 256);      // (the ten slots must fit beside the fp64 table image and, closed loop, the gain slots)
  __shared__ double tgs[INCT ? 10 : 1][INCT ? BLOCK : 1];
  if (a.fi == 1) stage_tables(tab, a.tab);
  rollout_lanes<BLOCK, FI, const double *, LQR, INCT>(a, (const double *)tab, us, reinterpret_cast<double (*)[BLOCK]>(kq),
                                                      reinterpret_cast<double (*)[BLOCK]>(tgs));
}

// The same rollout on the scaled-integer table image (hifi, default numerics; large batches: the LDS pipe -- 1.5 KB of
// table vertices per aircraft-step as doubles, more than half of its cycles bank-conflict replays of the per-lane gathers --
// is one of the two ceilings of k_rollout there).
template <int BLOCK, bool LQR = false>
__global__ __launch_bounds__(BLOCK) void k_rollout_i(DynArgs a) {
  __shared__ __attribute__((aligned(16))) int tab[i32::IMAGE_INTS];
  __shared__ double us[4][BLOCK];
  __shared__ double kq[LQR ? 12 : 1][LQR ? BLOCK : 1];
  constexpr bool INCT = INC_TRIG && !LQR;              // (closed loop: the gain slots take the room)
  __shared__ double tgs[INCT ? 10 : 1][INCT ? BLOCK : 1];
  {
    const int4 *src = reinterpret_cast<const int4 *>(a.tab32);
    int4 *dst = reinterpret_cast<int4 *>(tab);
    for (int i = threadIdx.x; i < i32::IMAGE_INTS / 4; i += BLOCK) dst[i] = src[i];
    __syncthreads();
  }
  rollout_lanes<BLOCK, 1, TabI32, LQR, INCT>(a, TabI32{tab}, us, reinterpret_cast<double (*)[BLOCK]>(kq), reinterpret_cast<double (*)[BLOCK]>(tgs));
}

// Four-wavefront rollout (latency regime, hifi): one workgroup = 64 aircraft on the four SIMDs of a CU; the state is
// split by owner (every wave integrates and range-checks what it owns) and the table lookups -- the LDS-latency-bound
// part of a step -- are spread over all four waves as four partial coefficient triples (aero_part<1..4>):
//   wave 0  x[0..8]   trigonometry, navigation + kinematic eqs, longitudinal damping tables | force equations, Euler
//   wave 1  --        longitudinal 3-D/2-D tables; stores the PREVIOUS step's x[0..8] sample that wave 0 published at
//                     step start (takes 9 stores per sample off wave 0's critical path)
//   wave 2  x[9..11]  lateral-directional 3-D/2-D tables                                    | moment equations, Euler
//   wave 3  x[12..17] atmosphere, actuator + flap models (Euler), lateral damping tables
// Two barriers per step; everything that crosses goes through lane-indexed (conflict-free) LDS arrays.
// Same terms as k_rollout (C/nlplant.c:333-377); the six totals are summed in a different order (ulp level).
#ifdef F16_EXP_STAMP4W
#define STAMP(acc) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1; }
#else
#define STAMP(acc)
#endif
template <bool LQR>
__global__ __launch_bounds__(256) void k_rollout_4w(DynArgs a) {
  __shared__ __attribute__((aligned(16))) double tab[TABLE_IMAGE_DOUBLES];
  __shared__ double xs[17][64], xt[14][64];  // xs: x[0..16] published at step start ; xt: 4 x 3 partial totals, qbar, ps
  __shared__ int xenv[3][64], xst[4][64];
  {
    const double2 *src = reinterpret_cast<const double2 *>(a.tab);
    double2 *dst = reinterpret_cast<double2 *>(tab);
    for (int i = threadIdx.x; i < TABLE_IMAGE_DOUBLES / 2; i += 256) dst[i] = src[i];
    __syncthreads();
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool envchk = !(a.flags & FLAG_NO_ENVELOPE);
  for (long b0 = (long)blockIdx.x * 64; b0 < a.B; b0 += (long)gridDim.x * 64) {
    const bool valid = b0 + lane < a.B;
    const long b = valid ? b0 + lane : a.B - 1;           // ragged tail: shadow the last aircraft, never stored
    double x[18], u[4];
#pragma unroll
    for (int k = 0; k < 18; ++k) x[k] = a.out[k * a.ld + b];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = a.u[k * a.ld + b];
    double kq[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ul[3] = {u[1], u[2], u[3]};   // LQR (wave 3): K[i][4..6], demands; last action (u0 if it never steps)
    if (LQR && wave == 3) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) kq[3 * i + j] = a.K[(9 * i + 4 + j) * a.ld + b];
#pragma unroll
      for (int j = 0; j < 3; ++j) kq[9 + j] = a.dem[j * a.ld + b];
    }
    int st = a.status ? a.status[b] : 0;
    double *tr = a.traj ? a.traj + b : nullptr;           // next sample to be written by THIS wave
    int until_store = a.traj_every;
#ifdef F16_EXP_STAMP4W
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, t0 = __builtin_amdgcn_s_memtime();
#endif
    for (int t = 0; t <= a.nsteps; ++t) {
      const bool last = t == a.nsteps;                    // extra trip: only flushes the final x[0..8] sample
      // ---- step start: envelope test on the owned states (env.py:117-124), publish them
      if (wave == 0) {
        xenv[0][lane] = envchk && (x[2] < 0 || x[2] > 100000 || x[6] < 0 || x[6] > 900 || x[7] < -20. || x[7] > 90 ||
                                   x[8] < -30. || x[8] > 30);
#pragma unroll
        for (int k = 0; k < 9; ++k) xs[k][lane] = x[k];
      } else if (wave == 2) {
        xenv[1][lane] = envchk && (x[9] < -300 || x[9] > 300 || x[10] < -100 || x[10] > 100 || x[11] < -50 || x[11] > 50);
#pragma unroll
        for (int k = 9; k < 12; ++k) xs[k][lane] = x[k];
      } else if (wave == 3) {
        xenv[2][lane] = envchk && (x[12] < 1000 || x[12] > 19000 || x[13] < -25 || x[13] > 25 || x[14] < -21.5 ||
                                   x[14] > 21.5 || x[15] < -30. || x[15] > 30 || x[16] < 0. || x[16] > 25);
#pragma unroll
        for (int k = 12; k < 17; ++k) xs[k][lane] = x[k];
      }
      STAMP(tD)
      __syncthreads();
      STAMP(tA)
      if (wave == 1 && tr && t > 0 && --until_store == 0) {   // sample of the step that just finished
        until_store = a.traj_every;
        if (valid) {
#pragma unroll
          for (int k = 0; k < 9; ++k) __builtin_nontemporal_store(xs[k][lane], tr + k * a.ld);
        }
        tr += 18 * a.ld;
      }
      if (last) break;
      if (xenv[0][lane] | xenv[1][lane] | xenv[2][lane]) st |= ST_ENVELOPE;
      const bool live = !(st & ST_ENVELOPE);
      // every wave sees the full published state (its own part is identical to its registers)
      double xa[17];
#pragma unroll
      for (int k = 0; k < 17; ++k) xa[k] = xs[k][lane];
      Pre p;
      double xd[18], o[3];
      int sa = 0;
      if (wave == 0) {
        plant_pre<false>(xa, p, xd);
        aero_part<3>((const double *)tab, xa, a.flags, o, sa);       // longitudinal damping derivatives (1-D tables)
      } else if (wave == 1) {
        aero_part<1>((const double *)tab, xa, a.flags, o, sa);       // longitudinal 3-D / 2-D tables
      } else if (wave == 2) {
        aero_part<2>((const double *)tab, xa, a.flags, o, sa);       // lateral-directional 3-D / 2-D tables
      } else {
        double vt = xa[6];
        if (vt <= 0.01) vt = 0.01;
        double mach, qbar, ps;
        atmos_dev(xa[2], vt, mach, qbar, ps);
        xt[12][lane] = qbar; xt[13][lane] = ps;
        aero_part<4>((const double *)tab, xa, a.flags, o, sa);       // lateral damping derivatives (1-D tables)
        if (live) {
          double xq[18];
#pragma unroll
          for (int k = 0; k < 17; ++k) xq[k] = xa[k];
          xq[17] = x[17];
          if (LQR) {
            const double e0 = kq[9] - xa[9], e1 = kq[10] - xa[10], e2 = kq[11] - xa[11];
            double uc[4] = {u[0], 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 3; ++i) ul[i] = uc[1 + i] = lqr_action(kq[3 * i], kq[3 * i + 1], kq[3 * i + 2], e0, e1, e2, u[1 + i]);
            actuators_dev(xq, uc, qbar, ps, xd);
          } else
          actuators_dev(xq, u, qbar, ps, xd);
#pragma unroll
          for (int k = 12; k < 18; ++k) x[k] += xd[k] * a.dt;   // env.py:126 on the actuator / flap states
        }
        if (tr && --until_store == 0) {
          until_store = a.traj_every;
          if (valid) {
#pragma unroll
            for (int k = 12; k < 18; ++k) __builtin_nontemporal_store(x[k], tr + k * a.ld);
          }
          tr += 18 * a.ld;
        }
      }
      // partial totals: wave 1 -> 0..2 (long. static), wave 0 -> 3..5 (long. damping), wave 2 -> 6..8, wave 3 -> 9..11
      const int slot = wave == 1 ? 0 : (wave == 0 ? 3 : (wave == 2 ? 6 : 9));
      xt[slot][lane] = o[0]; xt[slot + 1][lane] = o[1]; xt[slot + 2][lane] = o[2];
      xst[wave][lane] = sa;
      STAMP(tB)
      __syncthreads();
      STAMP(tC)
      // ---- second half: force equations on wave 0 || moment equations on wave 2
      if (wave == 0 || wave == 2) {
        double ls[3], ldm[3], ts[3], td[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { ls[k] = xt[k][lane]; ldm[k] = xt[3 + k][lane]; ts[k] = xt[6 + k][lane]; td[k] = xt[9 + k][lane]; }
        Totals tt;
        compose_totals(ls, ldm, ts, td, a.xcg, tt);
        if (wave == 0) {
          if (live) {
            p.qbar = xt[12][lane]; p.ps = xt[13][lane];
            st |= xst[0][lane] | xst[1][lane] | xst[2][lane] | xst[3][lane];
            plant_forces(xa, p, tt.Cx, tt.Cy, tt.Cz, xd);
#pragma unroll
            for (int k = 0; k < 9; ++k) x[k] += xd[k] * a.dt;   // env.py:126
          }
        } else {
          if (live) {
            plant_moments(x[9], x[10], x[11], xt[12][lane], tt.Cl, tt.Cm, tt.Cn, xd);
#pragma unroll
            for (int k = 9; k < 12; ++k) x[k] += xd[k] * a.dt;
          }
          if (tr && --until_store == 0) {
            until_store = a.traj_every;
            if (valid) {
#pragma unroll
              for (int k = 9; k < 12; ++k) __builtin_nontemporal_store(x[k], tr + k * a.ld);
            }
            tr += 18 * a.ld;
          }
        }
      }
    }
#ifdef F16_EXP_STAMP4W
    if (lane == 0 && blockIdx.x == 0 && a.traj) {   // diagnostic build: cycles per segment, per wave, into trajectory row 2
      double *d = a.traj + 18 * a.ld * 2 + wave * 4;
      d[0] = (double)tD; d[1] = (double)tA; d[2] = (double)tB; d[3] = (double)tC;
    }
#endif
    if (valid && wave != 1) {
      bool finite = true;
      const int k0 = wave == 0 ? 0 : (wave == 2 ? 9 : 12), k1 = wave == 0 ? 9 : (wave == 2 ? 12 : 18);
#pragma unroll
      for (int k = 0; k < 18; ++k)
        if (k >= k0 && k < k1) { finite = finite && isfinite(x[k]); a.out[k * a.ld + b] = x[k]; }
      // (which states were outside: a frozen aircraft keeps the state it was frozen with; every owner reports its own states)
      if (st & ST_ENVELOPE) st |= envelope_state_bits(x) & (((1 << k1) - (1 << k0)) << 8);
      if (a.status) atomicOr(&a.status[b], st | (finite ? 0 : ST_NONFINITE));
      if (LQR && wave == 3 && a.u_out) {
        a.u_out[b] = u[0];
#pragma unroll
        for (int i = 0; i < 3; ++i) a.u_out[(1 + i) * a.ld + b] = ul[i];
      }
    }
    __syncthreads();
  }
}

// Quad rollout (B <= 4096, hifi): FOUR LANES per aircraft, 16 aircraft per workgroup, so the reference batch of 4096
// fills all 256 CUs (the 4-wave kernel above occupies 64) and every role below runs sub-lane-parallel wherever the
// arithmetic is uniform (f16_plant_quad.hpp): a step costs a CU far fewer wave-instructions.  Lane l = 4 a + s.
//   wave 0  --        Cx, Cz, Cm totals on sub-lanes 0,1,2 (3-D / 2-D / 1-D longitudinal tables)
//   wave 1  x[9..11]  Cy, Cn, Cl totals on sub-lanes 0,1,2                                   | moment equations, Euler
//   wave 2  x[0..8]   sin/cos of phi, theta, psi, alpha on sub-lanes 0..3, beta; kinematic + navigation equations
//                                                                                            | force equations, Euler
//   wave 3  x[12..17] atmosphere; the four actuators on sub-lanes 0..3; flap model (Euler)
// Owned states are replicated over the sub-lanes of their wave; two barriers per step as in k_rollout_4w.
// GROUPS = 2 (4096 < B <= 8192): two independent 16-aircraft groups per workgroup share the LDS table image, one role
// wave of each on every SIMD -- the two dependency chains interleave.
template <int GROUPS, bool LQR = false>
__global__ __launch_bounds__(256 * GROUPS) void k_rollout_q(DynArgs a) {
  constexpr int NA = 16 * GROUPS;
  __shared__ __attribute__((aligned(16))) double tab[TABLE_IMAGE_DOUBLES];
  __shared__ double xs[18][NA], xt[11][NA];                // published state; Cx Cz Cm | Cy Cn Cl static | qbar ps | Cy Cn Cl damping
  __shared__ int xenv[3][NA], xst[2][NA];
  {
    const double2 *src = reinterpret_cast<const double2 *>(a.tab);
    double2 *dst = reinterpret_cast<double2 *>(tab);
    for (int i = threadIdx.x; i < TABLE_IMAGE_DOUBLES / 2; i += 256 * GROUPS) dst[i] = src[i];
    __syncthreads();
  }
  const int wave = (threadIdx.x >> 6) & 3, grp = threadIdx.x >> 8, lane = threadIdx.x & 63, ac = 16 * grp + (lane >> 2), s = lane & 3;
  const bool envchk = !(a.flags & FLAG_NO_ENVELOPE);
  for (long b0 = (long)blockIdx.x * NA; b0 < a.B; b0 += (long)gridDim.x * NA) {
    const bool valid = b0 + ac < a.B;
    const long b = valid ? b0 + ac : a.B - 1;              // ragged tail: shadow the last aircraft, never stored
    double x[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) x[k] = a.out[k * a.ld + b];
    const double ucmd = a.u[s * a.ld + b];                 // wave 3: sub-lane s drives actuator s
    // LQR (wave 3, sub-lanes 1..3): row s - 1 of K, columns 4..6, and the demands; the action of the last step
    double kr0 = 0, kr1 = 0, kr2 = 0, dm0 = 0, dm1 = 0, dm2 = 0, ulast = ucmd;
    if (LQR && wave == 3) {
      const long row = 9 * (s > 0 ? s - 1 : 0) + 4;
      kr0 = a.K[row * a.ld + b]; kr1 = a.K[(row + 1) * a.ld + b]; kr2 = a.K[(row + 2) * a.ld + b];
      dm0 = a.dem[b]; dm1 = a.dem[a.ld + b]; dm2 = a.dem[2 * a.ld + b];
    }
    double xact = s == 0 ? x[12] : (s == 1 ? x[13] : (s == 2 ? x[14] : x[15]));   // wave 3: its actuator state
    int st = a.status ? a.status[b] : 0;
    double *tr = a.traj ? a.traj + b : nullptr;            // next sample (written by wave 2 from the published state)
    int until_store = a.traj_every;
#ifdef F16_EXP_STAMPQ
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, t0 = __builtin_amdgcn_s_memtime();
#define QSTAMP(acc) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1; }
#else
#define QSTAMP(acc)
#endif
    for (int t = 0; t <= a.nsteps; ++t) {                  // the extra trip only publishes + stores the final sample
      // ---- step start: envelope test on the owned states (env.py:117-124), publish them
      if (wave == 2) {
        xenv[0][ac] = envchk && (x[2] < 0 || x[2] > 100000 || x[6] < 0 || x[6] > 900 || x[7] < -20. || x[7] > 90 ||
                                 x[8] < -30. || x[8] > 30);
        if (s == 0) {
#pragma unroll
          for (int k = 0; k < 9; ++k) xs[k][ac] = x[k];
        }
      } else if (wave == 1) {
        xenv[1][ac] = envchk && (x[9] < -300 || x[9] > 300 || x[10] < -100 || x[10] > 100 || x[11] < -50 || x[11] > 50);
        if (s < 3) xs[9 + s][ac] = s == 0 ? x[9] : (s == 1 ? x[10] : x[11]);
      } else if (wave == 3) {
        const double lim = s == 1 ? 25.0 : (s == 2 ? 21.5 : 30.0);
        const bool bad = s == 0 ? (xact < 1000 || xact > 19000) : (xact < -lim || xact > lim);
        const bool badl = x[16] < 0. || x[16] > 25;
        // any sub-lane out of range flags the aircraft: combine over the quad through LDS writes of `true` only
        if (s == 0) xenv[2][ac] = 0;
        xs[12 + s][ac] = xact;
        if (s < 2) xs[16 + s][ac] = s == 0 ? x[16] : x[17];
        __builtin_amdgcn_wave_barrier();
        if (envchk && (bad || badl)) xenv[2][ac] = 1;
      }
      QSTAMP(tD)
      __syncthreads();
      QSTAMP(tA)
      // trajectory sample of the step that just finished: all 18 states straight from the published copy, by the wave
      // with the most slack in the first half (wave 2 since the actuator wave became the longest; sub-lane s stores states
      // s, s+4, s+8, ...)
      if (wave == 2 && tr && t > 0 && --until_store == 0) {
        until_store = a.traj_every;
        if (valid) {
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const int kk = s + 4 * j;
            if (kk < 18) __builtin_nontemporal_store(xs[kk][ac], tr + kk * a.ld);
          }
        }
        tr += 18 * a.ld;
      }
      if (t == a.nsteps) break;
      if (xenv[0][ac] | xenv[1][ac] | xenv[2][ac]) st |= ST_ENVELOPE;
      const bool live = !(st & ST_ENVELOPE);
      double xa[17];
#pragma unroll
      for (int k = 0; k < 17; ++k) xa[k] = xs[k][ac];
      // first-half results wave 2 keeps in registers for the second half
      double xd[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      double U = 0, V = 0, W = 0, s_t = 0, c_t = 0, s_phi = 0, c_phi = 0, cb = 0, vtc = 0, r1 = 0, r2 = 0, r3 = 0;
      double fu0 = 0, fv0 = 0, fw0 = 0, mo0 = 0, mo1 = 0, mo2 = 0;
      if (wave == 0) {
        int sa_ = 0;
        double latd;
        const double tot = quad_long((const double *)tab, xa, s, a.xcg, a.flags, latd, sa_);
        if (s < 3) { xt[s][ac] = tot; xt[8 + s][ac] = latd; }
        xst[0][ac] = sa_;
      } else if (wave == 1) {
        int sa_ = 0;
        const double tot = quad_lat((const double *)tab, xa, s, sa_);
        if (s < 3) xt[3 + s][ac] = tot;
        xst[1][ac] = sa_;
        {   // the rate-product terms of the moment equations (C/nlplant.c:413-436, Heng = 0) do not need the totals: first half
          const double Jy = 55814.0, Jxz = 982.0, Jz = 63100.0, Jx = 9496.0;
          const double rden = 1.0 / (9496.0 * 63100.0 - 982.0 * 982.0);
          const double P = x[9], Q = x[10], R = x[11];
          mo0 = (Jxz * (Jx - Jy + Jz) * P * Q - (Jz * (Jz - Jy) + Jxz * Jxz) * Q * R) * rden;
          mo1 = F16_DIVC((Jz - Jx) * P * R - Jxz * (P * P - R * R), Jy);
          mo2 = ((Jx * (Jx - Jy) + Jxz * Jxz) * P * Q - Jxz * (Jx - Jy + Jz) * Q * R) * rden;
        }
      } else if (wave == 2) {
        // sin / cos of phi, theta, psi, alpha: one angle per sub-lane, then shared across the quad
        const double ang = s == 0 ? xa[3] : (s == 1 ? xa[4] : (s == 2 ? xa[5] : xa[7]));
        double sn, cs, sb;
        F16_SINCOS(ang, &sn, &cs);
        F16_SINCOS(xa[8], &sb, &cb);
        s_phi = quad_bcast<0>(sn); c_phi = quad_bcast<0>(cs);
        s_t = quad_bcast<1>(sn); c_t = quad_bcast<1>(cs);
        const double s_psi = quad_bcast<2>(sn), c_psi = quad_bcast<2>(cs);
        const double sal = quad_bcast<3>(sn), cal = quad_bcast<3>(cs);
        vtc = xa[6];
        if (vtc <= 0.01) vtc = 0.01;
        U = vtc * cal * cb; V = vtc * sb; W = vtc * sal * cb;                       // C/nlplant.c:148-150
        const double P = xa[9], Q = xa[10], R = xa[11];
#ifdef F16_FAST_DIV
        const double rct = f16_rcp(c_t);
        const double tt = s_t * rct;
#elif defined(F16_FAST_TAN)
        const double rct = 0.0, tt = s_t / c_t;
#else
        const double rct = 0.0, tt = tan(xa[4]);
#endif
        xd[0] = U * (c_t * c_psi) + V * (s_phi * c_psi * s_t - c_phi * s_psi) + W * (c_phi * s_t * c_psi + s_phi * s_psi);
        xd[1] = U * (c_t * s_psi) + V * (s_phi * s_psi * s_t + c_phi * c_psi) + W * (c_phi * s_t * s_psi - s_phi * c_psi);
        xd[2] = U * s_t - V * (s_phi * c_t) - W * (c_phi * c_t);
        xd[3] = P + tt * (Q * s_phi + R * c_phi);                                     // :169-176
        xd[4] = Q * c_phi - R * s_phi;
#ifdef F16_FAST_DIV
        xd[5] = (Q * s_phi + R * c_phi) * rct;
        r1 = f16_rcp(vtc); r2 = f16_rcp(U * U + W * W); r3 = f16_rcp(vtc * vtc * cb);   // divisors of :393-405, hoisted
#else
        xd[5] = (Q * s_phi + R * c_phi) / c_t;
        (void)rct;
#endif
        // this wave has slack in the first half and sets the pace of the second: what the force equations and the Euler
        // update do not need the coefficient totals for is done here (same expressions, :383-387 regrouped)
        {
          const double g = 32.17, m = 636.94;
          fu0 = R * V - Q * W - g * s_t + F16_DIVC(xa[12], m);
          fv0 = P * W - R * U + g * c_t * s_phi;
          fw0 = Q * U - P * V + g * c_t * c_phi;
          if (live) {
#pragma unroll
            for (int k = 0; k < 6; ++k) x[k] += xd[k] * a.dt;   // env.py:126 (navigation / kinematic states)
          }
        }
      } else {
        double vt = xa[6];
        if (vt <= 0.01) vt = 0.01;
        double mach, qbar, ps;
        atmos_dev(xa[2], vt, mach, qbar, ps);
        if (s == 0) { xt[6][ac] = qbar; xt[7][ac] = ps; }
        if (live) {
          // utils.py:308-330: thrust on sub-lane 0, elevator / aileron / rudder on 1..3 (same form, own limits)
          const double lim = s == 1 ? 25.0 : (s == 2 ? 21.5 : 30.0), rate = s == 1 ? 60.0 : (s == 2 ? 80.0 : 120.0);
          double uc = ucmd;
          if (LQR) {
            const double ua = lqr_action(kr0, kr1, kr2, dm0 - xa[9], dm1 - xa[10], dm2 - xa[11], ucmd);
            uc = s > 0 ? ua : ucmd;
            ulast = uc;
          }
          const double dth = actuator_rate(ucmd, 1000, 19000, 1.0, xact, 10000);
          const double dsf = actuator_rate(uc, -lim, lim, 20.2, xact, rate);
          double lf1_dot, lf2_dot;
          upd_lef_dev(xa[2], xa[6], xa[7], x[17], x[16], qbar, ps, lf1_dot, lf2_dot);
          xact += (s == 0 ? dth : dsf) * a.dt;             // env.py:126 on the actuator / flap states
          x[16] += lf2_dot * a.dt;
          x[17] += lf1_dot * a.dt;
        }
      }
      QSTAMP(tB)
      __syncthreads();
      QSTAMP(tC)
      // ---- second half: force equations on wave 2 || moment equations on wave 1
      if (wave == 2) {
        if (live) {
          st |= xst[0][ac] | xst[1][ac];
          const double Cx = xt[0][ac], Cz = xt[1][ac], Cy = xt[3][ac] + xt[8][ac], qbar = xt[6][ac];
          const double m = 636.94, S = 300.0;
          const double qsm = F16_DIVC(qbar * S, m);
          const double Udot = fu0 + qsm * Cx;                                                          // :383-387
          const double Vdot = fv0 + qsm * Cy;
          const double Wdot = fw0 + qsm * Cz;
#ifdef F16_FAST_DIV
          xd[6] = (U * Udot + V * Vdot + W * Wdot) * r1;                                                 // :393-405
          xd[7] = (U * Wdot - W * Udot) * r2;
          xd[8] = (Vdot * vtc - V * xd[6]) * r3;
#else
          xd[6] = (U * Udot + V * Vdot + W * Wdot) / vtc;
          xd[7] = (U * Wdot - W * Udot) / (U * U + W * W);
          xd[8] = (Vdot * vtc - V * xd[6]) / (vtc * vtc * cb);
#endif
#pragma unroll
          for (int k = 6; k < 9; ++k) x[k] += xd[k] * a.dt;   // env.py:126 (x[0..5] were advanced in the first half)
        }
      } else if (wave == 1) {
        if (live) {
          const double Cy = xt[3][ac] + xt[8][ac];
          const double Cn = xt[4][ac] + xt[9][ac] - Cy * (0.35 - a.xcg) * (11.32 / 30.0);   // C/nlplant.c:367
          const double Cl = xt[5][ac] + xt[10][ac];
          const double Jy = 55814.0, Jxz = 982.0, Jz = 63100.0, Jx = 9496.0, S = 300.0;
          const double rden = 1.0 / (9496.0 * 63100.0 - 982.0 * 982.0);
          const double qs = xt[6][ac] * S;
          const double L_tot = Cl * qs * 30.0, M_tot = xt[2][ac] * qs * 11.32, N_tot = Cn * qs * 30.0;   // :413-415
          x[9] += (mo0 + (Jz * L_tot + Jxz * N_tot) * rden) * a.dt;                                  // :417-436 + env.py:126
          x[10] += (mo1 + F16_DIVC(M_tot, Jy)) * a.dt;
          x[11] += (mo2 + (Jx * N_tot + Jxz * L_tot) * rden) * a.dt;
        }
      }
    }
#ifdef F16_EXP_STAMPQ
    if (lane == 0 && blockIdx.x == 0 && a.traj) {   // diagnostic build: cycles per segment, per wave, into trajectory row 2
      double *d = a.traj + 18 * a.ld * 2 + wave * 4;
      d[0] = (double)tD; d[1] = (double)tA; d[2] = (double)tB; d[3] = (double)tC;
    }
#endif
    // ---- write the final state back (owners), flags
    if (valid) {
      if (wave == 2 && s == 0) {
        bool finite = true;
#pragma unroll
        for (int k = 0; k < 9; ++k) { finite = finite && isfinite(x[k]); a.out[k * a.ld + b] = x[k]; }
        // (which states were outside: a frozen aircraft keeps the state it was frozen with; every owner reports its own states)
        if (st & ST_ENVELOPE) st |= envelope_state_bits(x) & (0x1FF << 8);
        if (a.status) atomicOr(&a.status[b], st | (finite ? 0 : ST_NONFINITE));
      } else if (wave == 1 && s == 0) {
        bool finite = true;
#pragma unroll
        for (int k = 9; k < 12; ++k) { finite = finite && isfinite(x[k]); a.out[k * a.ld + b] = x[k]; }
        if (st & ST_ENVELOPE) st |= envelope_state_bits(x) & (0x7 << 17);
        if (a.status) atomicOr(&a.status[b], st | (finite ? 0 : ST_NONFINITE));
      } else if (wave == 3) {
        a.out[(12 + s) * a.ld + b] = xact;
        bool finite = isfinite(xact);
        if (s < 2) { const double v = s == 0 ? x[16] : x[17]; finite = finite && isfinite(v); a.out[(16 + s) * a.ld + b] = v; }
        if (st & ST_ENVELOPE) {
          const double lim = s == 1 ? 25.0 : (s == 2 ? 21.5 : 30.0);
          if (s == 0 ? (xact < 1000 || xact > 19000) : (xact < -lim || xact > lim)) st |= 1 << (8 + 12 + s);
          if (s == 0 && (x[16] < 0. || x[16] > 25)) st |= 1 << (8 + 16);
        }
        if (a.status) atomicOr(&a.status[b], st | (finite ? 0 : ST_NONFINITE));
        if (LQR && a.u_out) a.u_out[s * a.ld + b] = ulast;
      }
    }
    __syncthreads();
  }
}

// The reference's LINEAR-model closed loops (test_env_mk2.py:46-62 `LQR(linear=True)` -- what main.py:35 runs -- and
// test_env.py:501-576 `test_LQR_lin`): per step  u = -K (x_ref - x) + u0,  x = Ad x + Bd u  on the reduced model (9 states, 3 inputs).
// One lane = one aircraft; its three matrices (135 doubles) are loaded ONCE and stay in registers (one wave per SIMD, 512 registers
// per lane), so a step is 135 fused multiply-adds and, with every sample stored, 96 bytes of HBM writes: 2.8 FLOP per byte, under the
// fp64 ridge -- the stored rollout is HBM-write-bound.  track: bit j set = entry j of the reference is given (x_ref[j]); clear =
// the reference follows the current state (env.py:362-367: x_ref = copy(x); x_ref[4:7] = demands).
struct LinArgsL {
  double *x;            // [9][ld] in place
  const double *Ad, *Bd, *K, *xref, *u0;
  double *trx, *tru;    // [T / every][9][ld], [T / every][3][ld]; may be null
  long B, ld;
  int nsteps, every;
  unsigned track;
};
__global__ __launch_bounds__(64, 1) void k_rollout_lqr_linear(LinArgsL a) {
  const long b = (long)blockIdx.x * 64 + threadIdx.x;
  if (b >= a.B) return;
  double A[81], Bm[27], K[27], x[9], xr[9], u0[3];
#pragma unroll
  for (int e = 0; e < 81; ++e) A[e] = a.Ad[e * a.ld + b];
#pragma unroll
  for (int e = 0; e < 27; ++e) { Bm[e] = a.Bd[e * a.ld + b]; K[e] = a.K[e * a.ld + b]; }
#pragma unroll
  for (int j = 0; j < 9; ++j) { x[j] = a.x[j * a.ld + b]; xr[j] = a.xref[j * a.ld + b]; }
#pragma unroll
  for (int c = 0; c < 3; ++c) u0[c] = a.u0 ? a.u0[c * a.ld + b] : 0.0;
  double *px = a.trx ? a.trx + b : nullptr, *pu = a.tru ? a.tru + b : nullptr;
  int until = a.every;
  for (int t = 0; t < a.nsteps; ++t) {
    double e[9], u[3], xn[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) e[j] = (((a.track >> j) & 1u) ? xr[j] : x[j]) - x[j];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 9; ++j) s += K[9 * c + j] * e[j];
      u[c] = -s + u0[c];                                     // env.py:371 / test_env.py:555
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      double s = 0.0, v = 0.0;
#pragma unroll
      for (int j = 0; j < 9; ++j) s += A[9 * r + j] * x[j];
#pragma unroll
      for (int c = 0; c < 3; ++c) v += Bm[3 * r + c] * u[c];
      xn[r] = s + v;                                         // test_env_mk2.py:58 / test_env.py:556
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) x[r] = xn[r];
    if (--until == 0) {
      until = a.every;
      if (px) {
#pragma unroll
        for (int r = 0; r < 9; ++r) __builtin_nontemporal_store(x[r], px + r * a.ld);
        px += 9 * a.ld;
      }
      if (pu) {
#pragma unroll
        for (int c = 0; c < 3; ++c) __builtin_nontemporal_store(u[c], pu + c * a.ld);
        pu += 3 * a.ld;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 9; ++r) a.x[r * a.ld + b] = x[r];
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_xdot_na(DynArgs a) {
  __shared__ __attribute__((aligned(16))) double tab[TABLE_IMAGE_DOUBLES];
  if (a.fi == 1) stage_tables(tab, a.tab);
  for (long b = (long)blockIdx.x * BLOCK + threadIdx.x; b < a.B; b += (long)gridDim.x * BLOCK) {
    double sv[18], xd9[9];
#pragma unroll
    for (int k = 0; k < 18; ++k) sv[k] = a.x[k * a.ld + b];
    // env.py:172-177: scatter x9 -> mpc_x_idx [3,4,7,8,9,10,11,17,16], u3 -> [13,14,15]
    sv[3] = a.u[0 * a.ld + b]; sv[4] = a.u[1 * a.ld + b]; sv[7] = a.u[2 * a.ld + b]; sv[8] = a.u[3 * a.ld + b];
    sv[9] = a.u[4 * a.ld + b]; sv[10] = a.u[5 * a.ld + b]; sv[11] = a.u[6 * a.ld + b];
    sv[17] = a.u[7 * a.ld + b]; sv[16] = a.u[8 * a.ld + b];
    sv[13] = a.u3[0 * a.ld + b]; sv[14] = a.u3[1 * a.ld + b]; sv[15] = a.u3[2 * a.ld + b];
    int st = 0;
    calc_xdot_na((const double *)tab, a.lofi, sv, xd9, a.xcg, a.fi, a.flags, st);
#pragma unroll
    for (int k = 0; k < 9; ++k) a.out[k * a.ld + b] = xd9[k];
    if (a.status) a.status[b] |= st;
  }
}

// ------------------------------------------------------------------------------------ launchers
// One lane per aircraft; with the table image in LDS a CU holds one workgroup, so the workgroup size decides how many
// waves share a SIMD: fill all 256 CUs first (64 / 128 / 256 lanes), then two waves per SIMD (512 lanes) from 131,072
// aircraft on -- the phased lookups (aero_totals_phased) keep a hifi step inside 256 registers there; measured with
// every step stored: 15.8 (256 lanes) vs 17.1 (512) G steps/s at B = 262,144, 17.9 vs 21.5 at B = 1,048,576.
struct Geometry { int block, grid; };
static Geometry geometry(long B, int fi) {
  Geometry g;
  (void)fi;
  if (B <= 64L * 256) g.block = 64;          // <= 256 one-wave workgroups: one per CU
  else if (B <= 128L * 256) g.block = 128;   // fill all 256 CUs before stacking waves on a CU
  else if (B < 512L * 256) g.block = 256;    // one wave per SIMD
  else g.block = 512;                        // two waves per SIMD
  static const int force = [] { const char *e = getenv("F16_DYN_BLOCK"); return e ? atoi(e) : 0; }();   // tuning knob
  if (force == 64 || force == 128 || force == 256 || force == 512) g.block = force;
  long blocks = (B + g.block - 1) / g.block;
  g.grid = (int)(blocks < 256 ? blocks : 256);
  return g;
}

#define LAUNCH_BY_BLOCK(KERN, g, stream, args)                                       \
  do {                                                                               \
    if ((g).block == 64) hipLaunchKernelGGL(KERN<64>, dim3((g).grid), dim3(64), 0, stream, args);        \
    else if ((g).block == 128) hipLaunchKernelGGL(KERN<128>, dim3((g).grid), dim3(128), 0, stream, args); \
    else if ((g).block == 256) hipLaunchKernelGGL(KERN<256>, dim3((g).grid), dim3(256), 0, stream, args); \
    else hipLaunchKernelGGL(KERN<512>, dim3((g).grid), dim3(512), 0, stream, args);                       \
  } while (0)
// same; the lofi model gets an instantiation with the fidelity fixed at compile time (3.94 vs 4.6 ms per 1000 steps at
// B=4096).  For hifi the run-time-flag kernel measured FASTER than a compile-time one (4.59 vs 4.65 ms; 5.31 vs 4.85 G
// steps/s at B=262144: the scheduler does worse on the merged basic block), so hifi keeps the run-time path.
#define LAUNCH_BY_BLOCK_FI(KERN, g, stream, args)                                                              \
  do {                                                                                                         \
    const int fi_ = (args).fi;                                                                                  \
    if ((g).block == 64) {                                                                                     \
      if (fi_ == 0) hipLaunchKernelGGL((KERN<64, 0>), dim3((g).grid), dim3(64), 0, stream, args);             \
      else hipLaunchKernelGGL((KERN<64, -1>), dim3((g).grid), dim3(64), 0, stream, args);                      \
    } else if ((g).block == 128) {                                                                             \
      if (fi_ == 0) hipLaunchKernelGGL((KERN<128, 0>), dim3((g).grid), dim3(128), 0, stream, args);           \
      else hipLaunchKernelGGL((KERN<128, -1>), dim3((g).grid), dim3(128), 0, stream, args);                    \
    } else if ((g).block == 256) {                                                                             \
      if (fi_ == 0) hipLaunchKernelGGL((KERN<256, 0>), dim3((g).grid), dim3(256), 0, stream, args);           \
      else hipLaunchKernelGGL((KERN<256, -1>), dim3((g).grid), dim3(256), 0, stream, args);                    \
    } else {                                                                                                   \
      if (fi_ == 0) hipLaunchKernelGGL((KERN<512, 0>), dim3((g).grid), dim3(512), 0, stream, args);           \
      else hipLaunchKernelGGL((KERN<512, -1>), dim3((g).grid), dim3(512), 0, stream, args);                    \
    }                                                                                                          \
  } while (0)

static int check_common(f16_ctx *ctx, const void *p0, const void *p1, long B, long ld) {
  if (!ctx || !p0 || !p1 || B < 0 || ld < B) return set_error(F16_EINVAL, "bad argument (NULL pointer, B < 0 or ld < B)");
  return F16_OK;
}

}  // namespace f16

using namespace f16;

extern "C" int f16_xdot_batch(f16_ctx *ctx, const double *x, const double *u, double *xdot, int32_t *status, long B,
                              long ld, double xcg, int fi_flag, unsigned flags, void *stream) {
  if (int rc = check_common(ctx, x, xdot, B, ld)) return rc;
  if (!u) return set_error(F16_EINVAL, "u is NULL");
  if (B == 0) return F16_OK;
  DynArgs a{};
  a.tab = ctx->d_tab; a.lofi = ctx->d_lofi; a.x = x; a.u = u; a.out = xdot; a.status = status;
  a.B = B; a.ld = ld; a.xcg = xcg; a.fi = fi_flag; a.flags = flags;
  Geometry g = geometry(B, fi_flag);
  LAUNCH_BY_BLOCK_FI(k_xdot, g, (hipStream_t)stream, a);
  return hip_check(hipGetLastError(), "f16_xdot_batch launch");
}

extern "C" int f16_nlplant_batch(f16_ctx *ctx, const double *xu, double *xdot, int32_t *status, long B, long ld,
                                 double xcg, int fi_flag, unsigned flags, void *stream) {
  if (int rc = check_common(ctx, xu, xdot, B, ld)) return rc;
  if (B == 0) return F16_OK;
  DynArgs a{};
  a.tab = ctx->d_tab; a.lofi = ctx->d_lofi; a.x = xu; a.out = xdot; a.status = status;
  a.B = B; a.ld = ld; a.xcg = xcg; a.fi = fi_flag; a.flags = flags;
  if (B <= 256) {
    hipLaunchKernelGGL((k_nlplant<64, false>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
  } else {
    Geometry g = geometry(B, fi_flag);
    if (g.block == 64) hipLaunchKernelGGL((k_nlplant<64, true>), dim3(g.grid), dim3(64), 0, (hipStream_t)stream, a);
    else if (g.block == 128) hipLaunchKernelGGL((k_nlplant<128, true>), dim3(g.grid), dim3(128), 0, (hipStream_t)stream, a);
    else if (g.block == 256) hipLaunchKernelGGL((k_nlplant<256, true>), dim3(g.grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_nlplant<512, true>), dim3(g.grid), dim3(512), 0, (hipStream_t)stream, a);
  }
  return hip_check(hipGetLastError(), "f16_nlplant_batch launch");
}

// LQR = true: the closed loop of f16_rollout_lqr (same launch rules; the one-lane kernels keep K[i][4..6] and the demands in
// lane-indexed LDS slots, which a 512-lane workgroup has room for only beside the integer table image)
template <bool LQR>
static int rollout_dispatch(f16_ctx *ctx, DynArgs &a, void *stream) {
  const long B = a.B;
  const int fi_flag = a.fi;
  static const long max4w = [] { const char *e = getenv("F16_ROLLOUT_4W_MAXB"); return e ? atol(e) : 64L * 256; }();
  static const long maxq = [] { const char *e = getenv("F16_ROLLOUT_QUAD_MAXB"); return e ? atol(e) : 16L * 256; }();
  if (a.flags & F16_FLAG_ONE_LANE) {
    // results independent of the batch size: ONE kernel for every B, its step an out-of-line function (k_rollout_exact)
    const long blocks = (B + 63) / 64;
    hipLaunchKernelGGL(k_rollout_exact<LQR>, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(64), 0, (hipStream_t)stream, a);
    return hip_check(hipGetLastError(), "f16_rollout launch");
  }
  if (fi_flag == 1 && B <= maxq) {
    // at most 16 aircraft per CU: four lanes per aircraft, one 16-aircraft workgroup per CU
    hipLaunchKernelGGL((k_rollout_q<1, LQR>), dim3((unsigned)((B + 15) / 16)), dim3(256), 0, (hipStream_t)stream, a);
    return hip_check(hipGetLastError(), "f16_rollout launch");
  }
  if (fi_flag == 1 && B <= 2 * maxq) {
    // at most 32 per CU: two 16-aircraft groups per workgroup
    hipLaunchKernelGGL((k_rollout_q<2, LQR>), dim3((unsigned)((B + 31) / 32)), dim3(512), 0, (hipStream_t)stream, a);
    return hip_check(hipGetLastError(), "f16_rollout launch");
  }
  // (three groups per workgroup leave 168 registers per lane: the roles spill, 3.96 ms against the 4-wave kernel's 2.13)
  if (fi_flag == 1 && B <= max4w) {
    // latency regime: four wavefronts per 64 aircraft, one workgroup per CU
    hipLaunchKernelGGL(k_rollout_4w<LQR>, dim3((unsigned)((B + 63) / 64)), dim3(256), 0, (hipStream_t)stream, a);
    return hip_check(hipGetLastError(), "f16_rollout launch");
  }
  Geometry g = geometry(B, fi_flag);
#ifdef F16_FAST_DIV
  // throughput regime, default numerics: lookups on the scaled-integer image
  static const int use_i32 = [] { const char *e = getenv("F16_ROLLOUT_I32"); return e ? atoi(e) : 1; }();
  if (fi_flag == 1 && use_i32 && g.block == 512) {
    a.tab32 = ctx->d_tab32;
    hipLaunchKernelGGL((k_rollout_i<512, LQR>), dim3(g.grid), dim3(512), 0, (hipStream_t)stream, a);
    return hip_check(hipGetLastError(), "f16_rollout launch");
  }
#endif
  if (LQR) {
    if (g.block == 512) g.block = 256;       // (fp64 image + 16 slots of 512 lanes would not fit the 160 KB of a CU)
    const hipStream_t st = (hipStream_t)stream;
    if (g.block == 64) {
      if (fi_flag == 0) hipLaunchKernelGGL((k_rollout<64, 0, LQR>), dim3(g.grid), dim3(64), 0, st, a);
      else hipLaunchKernelGGL((k_rollout<64, -1, LQR>), dim3(g.grid), dim3(64), 0, st, a);
    } else if (g.block == 128) {
      if (fi_flag == 0) hipLaunchKernelGGL((k_rollout<128, 0, LQR>), dim3(g.grid), dim3(128), 0, st, a);
      else hipLaunchKernelGGL((k_rollout<128, -1, LQR>), dim3(g.grid), dim3(128), 0, st, a);
    } else {
      if (fi_flag == 0) hipLaunchKernelGGL((k_rollout<256, 0, LQR>), dim3(g.grid), dim3(256), 0, st, a);
      else hipLaunchKernelGGL((k_rollout<256, -1, LQR>), dim3(g.grid), dim3(256), 0, st, a);
    }
    return hip_check(hipGetLastError(), "f16_rollout launch");
  }
  LAUNCH_BY_BLOCK_FI(k_rollout, g, (hipStream_t)stream, a);
  return hip_check(hipGetLastError(), "f16_rollout launch");
}

extern "C" int f16_rollout(f16_ctx *ctx, double *x, const double *u, double *traj, int32_t *status, long B, long ld,
                           int nsteps, int traj_every, double dt, double xcg, int fi_flag, unsigned flags,
                           void *stream) {
  if (int rc = check_common(ctx, x, u, B, ld)) return rc;
  if (nsteps < 0 || (traj && (traj_every < 1 || nsteps % traj_every != 0)))
    return set_error(F16_EINVAL, "nsteps must be >= 0 and a multiple of traj_every >= 1 when traj is given");
  if (B == 0 || nsteps == 0) return F16_OK;
  DynArgs a{};
  a.tab = ctx->d_tab; a.lofi = ctx->d_lofi; a.u = u; a.out = x; a.traj = traj; a.status = status;
  a.B = B; a.ld = ld; a.nsteps = nsteps; a.traj_every = traj ? traj_every : nsteps + 1;
  a.dt = dt; a.xcg = xcg; a.fi = fi_flag; a.flags = flags;
  return rollout_dispatch<false>(ctx, a, stream);
}

extern "C" int f16_rollout_lqr(f16_ctx *ctx, double *x, const double *u0, const double *K, const double *dem, double *traj,
                               double *u_out, int32_t *status, long B, long ld, int nsteps, int traj_every, double dt,
                               double xcg, int fi_flag, unsigned flags, void *stream) {
  if (int rc = check_common(ctx, x, u0, B, ld)) return rc;
  if (!K || !dem) return set_error(F16_EINVAL, "K / dem is NULL");
  if (nsteps < 0 || (traj && (traj_every < 1 || nsteps % traj_every != 0)))
    return set_error(F16_EINVAL, "nsteps must be >= 0 and a multiple of traj_every >= 1 when traj is given");
  if (B == 0 || nsteps == 0) return F16_OK;
  DynArgs a{};
  a.tab = ctx->d_tab; a.lofi = ctx->d_lofi; a.u = u0; a.out = x; a.traj = traj; a.status = status;
  a.B = B; a.ld = ld; a.nsteps = nsteps; a.traj_every = traj ? traj_every : nsteps + 1;
  a.dt = dt; a.xcg = xcg; a.fi = fi_flag; a.flags = flags;
  a.K = K; a.dem = dem; a.u_out = u_out;
  return rollout_dispatch<true>(ctx, a, stream);
}

extern "C" int f16_rollout_lqr_linear(f16_ctx *ctx, double *x9, const double *Ad, const double *Bd, const double *K, const double *x_ref,
                                      const double *u0, double *traj_x, double *traj_u, long B, long ld, int nsteps, int traj_every,
                                      unsigned track_mask, void *stream) {
  if (int rc = check_common(ctx, x9, Ad, B, ld)) return rc;
  if (!Bd || !K || !x_ref) return set_error(F16_EINVAL, "Bd / K / x_ref is NULL");
  if (nsteps < 0 || ((traj_x || traj_u) && (traj_every < 1 || nsteps % traj_every != 0)))
    return set_error(F16_EINVAL, "nsteps must be >= 0 and a multiple of traj_every >= 1 when a trajectory is given");
  if (B == 0 || nsteps == 0) return F16_OK;
  LinArgsL a{};
  a.x = x9; a.Ad = Ad; a.Bd = Bd; a.K = K; a.xref = x_ref; a.u0 = u0; a.trx = traj_x; a.tru = traj_u;
  a.B = B; a.ld = ld; a.nsteps = nsteps; a.every = (traj_x || traj_u) ? traj_every : nsteps + 1; a.track = track_mask & 0x1FFu;
  hipLaunchKernelGGL(k_rollout_lqr_linear, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
  return hip_check(hipGetLastError(), "f16_rollout_lqr_linear launch");
}

extern "C" int f16_xdot_na_batch(f16_ctx *ctx, const double *x_full, const double *x9, const double *u3, double *xdot9,
                                 int32_t *status, long B, long ld, double xcg, int fi_flag, unsigned flags,
                                 void *stream) {
  if (int rc = check_common(ctx, x_full, xdot9, B, ld)) return rc;
  if (!x9 || !u3) return set_error(F16_EINVAL, "x9/u3 is NULL");
  if (B == 0) return F16_OK;
  DynArgs a{};
  a.tab = ctx->d_tab; a.lofi = ctx->d_lofi; a.x = x_full; a.u = x9; a.u3 = u3; a.out = xdot9; a.status = status;
  a.B = B; a.ld = ld; a.xcg = xcg; a.fi = fi_flag; a.flags = flags;
  Geometry g = geometry(B, fi_flag);
  LAUNCH_BY_BLOCK(k_xdot_na, g, (hipStream_t)stream, a);
  return hip_check(hipGetLastError(), "f16_xdot_na_batch launch");
}

#ifdef F16_EXP_STAMPQ2
extern "C" int f16_debug_qstamps(unsigned long long *h_out) {     // diagnostic build only
  return hip_check(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(f16::g_qstamp), 8 * sizeof(unsigned long long)), "read stamps");
}
#endif
