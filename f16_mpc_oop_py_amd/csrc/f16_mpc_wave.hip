// f16_mpc_wave.hip -- ONE WAVEFRONT PER AIRCRAFT: the OSQP solve of the condensed MPC QP (env.py:420-424) for N <= 30, gfx950.
//
// Same algorithm, same coordinates and the same arithmetic rules as f16_mpc_solve.hip (x, z, y = yb / E unscaled; linear system
// (c P + sigma D^-2 + rho A'WA) x~ = sigma D^-2 x - c q + A' W (rho z - y); termination on the unscaled residuals; rho estimate on
// the scaled ones) -- what changes is the mapping.  There: one 512-lane workgroup per aircraft, 54 FMAs per lane and iteration
// under ~270 instructions of reductions, loads and three workgroup barriers.  Here: a workgroup IS one wavefront, four of them
// resident per CU (one per SIMD, 40 KB of LDS each), no barrier anywhere:
//   * the KKT inverse lives in LDS as a SYMMETRIC block image (15 x 15 blocks of 6 x 6, every unordered pair of block rows once,
//     34.6 KB): lane (r, s) holds blocks (r, r - (2s+1)) and (r, r - (2s+2)) (mod 15; s = 3: the diagonal block) and uses each
//     twice -- y_r += B x_c and y_c += B' x_r, the second fetched by its owner with ds_bpermute (tools/wave_tables.py emulates
//     and checks the layout);
//   * the two block-Toeplitz operators (utils.py:171-197: CC[i,j] = A^(i-j) B; stage 1 = CCs' w, stage 3 = CCs x~) share 72
//     register-resident doubles per lane: lane (o, t) of octet o (steps 4o..4o+3) holds the kept rows of G_4t..G_4t+3; partial
//     sums meet inside the octet by recursive halving on the DPP network (row split free by a per-lane row order of G);
//   * the KKT matrix is assembled and inverted on the fp64 matrix cores by the SAME blocked symmetric sweep as
//     f16_mpc_solve.hip (four pivots per step, one v_mfma_f64_16x16x4_f64 per tile and step), but with all 21 lower-triangular
//     tiles in this one wavefront's accumulators; the pivot panel goes through LDS without a barrier;
//   * Ruiz equilibration: k_mpc_fast in its scale-only mode (f16_mpc_solve.hip) leaves D, E, c in the workspace.
// Launch: grid = B workgroups of 64 lanes, __launch_bounds__(64, 1) (512 registers), 40,960 B of static LDS.
//
// Two kernels share the solve (solve_aircraft):
//   k_mpc_wave      one calc_MPC_action per aircraft (f16_mpc_batch / f16_mpc_plan_solve), aircraft from a work queue
//   k_rollout_mpc   the reference's closed MPC loop test_env.py:480-495 as ONE launch (f16_rollout_mpc, round 5): (step, aircraft)
//                   pairs from a ticket counter, per pair the state-dependent QP vectors, the solve, the command, one Euler step
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "f16_mpc.hpp"
#include "f16_mpc_state.hpp"
#include "f16_plant.hpp"
#include "f16_smallmat.hpp"
#include "f16_wave_tables.inc"

namespace f16 {
namespace wave {

typedef double d4_t __attribute__((ext_vector_type(4)));

constexpr int WN = WAVE_MAXN;              // 30
constexpr int NB = 15, NL = 60;            // block rows; lanes that hold blocks
constexpr int NT = 6;                      // 16 x 16 tiles per side of the padded 96 x 96 tile image
constexpr int NTILES = NT * (NT + 1) / 2;  // 21 lower-triangular tiles
constexpr int FN = 16 * NT;                // 96

// ---- LDS map (doubles).  [0, KI_SIZE) is the block image of the KKT inverse during the iterations and scratch of the
// factorisation before that; the vectors of the iteration sit behind it.
constexpr int KI_SIZE = 36 * NL * 2;                  // 4320
constexpr int WS_OFF = KI_SIZE, WS_SIZE = 444;         // state-row vector: six kept rows per step, steps 0..33 (equilibration: E of all nine rows)
// ... in the iterations six doubles of padding follow every five steps (ws_idx): the lanes of a stage-1 block (s, I) read 16-byte
// slots at 30 I + 2 s doubles, which put (0,0), (1,1), (2,2) on the same four banks (tools/wave_lds_conflicts.py: 27 extra LDS cycles
// on the nine reads of an iteration, 9 with the padding)
constexpr int WS_PAD = 6;
__host__ __device__ constexpr int ws_idx(int step) { return 6 * step + WS_PAD * (step / 5); }
static_assert(ws_idx(33) + 6 <= WS_SIZE, "state-row vector with its block padding");
constexpr int WC_OFF = WS_OFF + WS_SIZE, WC_SIZE = 92;                 // command rows (also: rhs of the linear system)
constexpr int WR_OFF = WC_OFF + WC_SIZE, WR_SIZE = 96;                 // rate rows (read up to k + 3)
constexpr int LDS_DOUBLES = 5120;                      // 40,960 B: four wavefront-workgroups per CU
constexpr int XT_PAD = 21, XT_OFF = WR_OFF + WR_SIZE, XT_SIZE = LDS_DOUBLES - XT_OFF;    // 168.  Equilibration / KKT diagonal: three
                                                       // per step behind 7 zero steps (XT_PAD); iterations: FOUR per step behind XPADS zero steps
constexpr int XPADS = 7;
static_assert(6 * 34 <= WS_SIZE && 4 * (XPADS + WN + 5) <= XT_SIZE && XT_PAD + 3 * WN + 5 <= XT_SIZE, "LDS map");
// During the iterations only the B blocks of the KKT inverse stay in LDS (chunks 18..35 of the block image, where the scatter of
// the factorisation leaves them); the A blocks live in registers (reloaded from the workspace at every entry of the hot loop), and
// their half of the image holds the partial sums of the two Toeplitz stages:
constexpr int KB_OFF = 18 * NL * 2;                    // 2160
constexpr int P3_OFF = 0, P3_REC = 30, P3_SIZE = 22 * P3_REC;             // stage 3: record (block) = [5 steps][6 kept rows]; record 21: the idle lane's
constexpr int P1_OFF = P3_OFF + P3_SIZE, P1_REC = 22, P1_SIZE = 64 * P1_REC;   // stage 1: record (lane) = [5 steps][3 sums + 1 pad] + 2: a record
                                                       // stride of 44 banks puts the 16 lanes of an LDS pass on 16 different groups of four banks (40: on 8)
constexpr int ZP_OFF = P1_OFF + P1_SIZE;               // eight zeros: what an absent partial sum reads
constexpr int MVX_OFF = 0, MVX_REC = 14;               // mat-vec: the transposed parts of a lane (2 x 6 doubles + 2: 28 banks), in the records' space
static_assert(ZP_OFF + 8 <= KB_OFF && (P1_OFF & 1) == 0 && (ZP_OFF & 1) == 0 && MVX_OFF + 64 * MVX_REC <= ZP_OFF, "partial-sum records");
// scratch of the factorisation phases inside [0, KI_SIZE)
constexpr int GL_OFF = 0, GL_SIZE = 27 * (WN + 1) + 1;                 // all nine rows of every G_k + a zero block
constexpr int WG_OFF = GL_OFF + GL_SIZE, WG_SIZE = 6 * WN + 8;         // Gram weights of the kept state rows (zero padded)
constexpr int WCV_OFF = WG_OFF + WG_SIZE, WRV_OFF = WCV_OFF + 96;      // Gram weights of command / rate rows (rate: + 4)
constexpr int PAN_SIZE = FN * 4 + 2;                    // a panel: C[col][0..3] = M[pivot row][col] (FN x 4)
constexpr int PAN_OFF = WRV_OFF + 100;
static_assert(PAN_OFF + 2 * PAN_SIZE <= KI_SIZE, "factorisation scratch");
static_assert((PAN_OFF & 1) == 0 && (PAN_SIZE & 1) == 0 && (XT_OFF & 1) == 0 && (WS_OFF & 1) == 0 && (WC_OFF & 1) == 0, "16-byte alignment");

// ---- per-aircraft workspace in HBM (the `gramws` block of MpcArgs, MPC_TILE_DOUBLES = 9216 doubles)
constexpr int GW_TILES = 0;                            // [21][4][64] A'WA as lower-triangular tiles
constexpr int GW_SCAL = WAVE_SCAL_OFF;                 // D[96] | E state [192] | E command [96] | E rate [96] | c   (k_mpc_fast mode 3)
constexpr int GW_QUEUE = 484;                          // (behind D | E | c of the scaling block: the work-queue counter of a launch, in aircraft 0's block)
static_assert(NTILES * 256 <= WAVE_SCAL_OFF && WAVE_SCAL_OFF + 488 <= MPC_TILE_DOUBLES, "workspace map");

__shared__ __attribute__((aligned(16))) double s_w[LDS_DOUBLES];

__device__ __forceinline__ int tile_idx(int w, int J) { return w * (w + 1) / 2 + J; }

// pointers into the workspace are GLOBAL memory: said so, the loads are global_load (one counter) instead of flat_load
typedef const double __attribute__((address_space(1))) *gptr_t;
__device__ __forceinline__ gptr_t as_global(const double *p) { return (gptr_t)p; }

template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HMIRROR = 0x141;
__device__ __forceinline__ double bperm(double v, int src_lane) {
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
#define WAVE_LDS_PHASE() __builtin_amdgcn_sched_barrier(0x7)      /* memory operations stay put, ALU may float */
// producer lanes -> consumer lanes of the SAME wavefront through LDS: the LDS instructions of one wave execute in order, so a
// later read sees an earlier write without any wait; all that is needed is that the COMPILER keeps the accesses in order
// (a wavefront-scope fence: no s_waitcnt, unlike the workgroup-scope one, which drained both counters ~6 times per iteration)
#ifdef F16_WAVE_SYNC_WORKGROUP
__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
#else
__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#endif

struct Role {
  int l, o, t, estar, par, h, b1, b2, istep;   // octet layout: lane (o, t) owns step istep = 4 o + t / 2
  int r, s, cA, cB;                            // block layout: quad r = block row, s = lane % 4
  bool act;                                    // istep < N
};
__device__ __forceinline__ Role role(int N) {
  Role R;
  R.l = threadIdx.x; R.o = R.l >> 3; R.t = R.l & 7;
  R.b2 = (R.t >> 2) & 1; R.b1 = (R.t >> 1) & 1; R.par = R.t & 1;
  R.estar = R.t >> 1; R.h = R.par ^ R.b2;
  R.istep = 4 * R.o + R.estar; R.act = R.istep < N;
  R.r = R.l >> 2; R.s = R.l & 3;
  const int rr = R.r < NB ? R.r : NB - 1;
  R.cA = (rr + NB - (2 * R.s + 1)) % NB;
  R.cB = R.s < 3 ? (rr + NB - (2 * R.s + 2)) % NB : rr;
  return R;
}

// ---- per-lane constants of a solve: the lane's three owned variables (both lanes of a pair hold them), its three state
// rows (step istep, kept rows 3h..3h+2) and its three command (par = 0) or rate (par = 1) rows of the same step.
struct LaneConst {
  double sg[3], cq[3];                         // sigma D^-2, c q of the owned variables
  double loA[3], hiA[3], WA[3], loB[3], hiB[3], WB[3];   // bounds and row weight W = E^2
  int eqA, eqB;                                // bit c: the row carries 1e3 rho (equality row after scaling)
  double cs, cinv;
};
struct SolveState {
  double x[3], zA[3], yA[3], zB[3], yB[3], dyA[3], dyB[3];
  double rho, rp, rd;
  int it, to_check, done, converged, infeasible;
};
struct IterSettings { double alpha, eps_abs, eps_rel, eps_prim_inf; int max_iter, check_every, rho_every, adaptive_rho; };

// ----------------------------------------------------------------------------------------------------------------
// The two block-Toeplitz stages (utils.py:171-197: CC[i,j] = A^(i-j) B, never formed; stage 3 = CCs x~, stage 1 = CCs' w) with
// the causal zeros folded away (round 4).  The N x N triangle of (step, lag) pairs is cut into 5 x 5 blocks: six step blocks I,
// six lag blocks T, and only the 21 blocks with T <= I hold products that are not structurally zero.  Lane (s, I, T) -- 63 of
// the 64 lanes, l = 21 s + I (I + 1) / 2 + T -- owns block (I, T) for the kept state rows 2s, 2s + 1: THIRTY register-resident
// doubles Gd[u][r][c] = G_(5T+u)[kept row 2s + r][c] (round 3: 72 per lane, half of whose products multiplied zeros), used both
// ways: stage 3 gives the lane's two rows of steps 5I..5I+4 from the variable steps 5(I-T)-4 .. 5(I-T)+4, stage 1 gives the
// variables of steps 5(I-T)..5(I-T)+4 from its two rows of steps 5I..5I+8 -- 150 products each.  The partial sums of a block
// row / block diagonal do not sit in a DPP-friendly lane pattern, so they meet in LDS: every lane stores its partial sums in a
// record of its own, and the lane that owns a row (a variable) adds the records that hold a part of it, in a fixed order
// (absent records read a block of zeros).
constexpr int TB = 5;
constexpr int GIMG_DOUBLES = 15 * 64 * 2;         // the lanes' Toeplitz operands in the workspace; the A blocks of the KKT inverse ([18][64] double2) follow
static_assert(GIMG_DOUBLES + 18 * 64 * 2 <= WAVE_PBLK_DOUBLES / 2, "workspace map");
struct TJob { int s, I, T, D, rec3; };
__device__ __forceinline__ TJob tjob() {
  const int l = threadIdx.x;
  TJob J;
  const bool live = l < 63;
  const int ll = live ? l : 0;
  J.s = ll / 21;
  const int j = ll - 21 * J.s;
  J.I = j >= 15 ? 5 : (j >= 10 ? 4 : (j >= 6 ? 3 : (j >= 3 ? 2 : (j >= 1 ? 1 : 0))));
  J.T = j - J.I * (J.I + 1) / 2;
  J.D = J.I - J.T;
  J.rec3 = live ? j : 21;                                   // (lane 63: zero operands, a record nobody reads)
  return J;
}
// from G_k in the workspace ([k][9 rows][3], build kernel); zero for lags beyond the horizon and on the idle lane
__device__ __forceinline__ void load_G(double (&Gd)[TB][2][3], const double *Gg, const TJob &J, int N) {
  // (unconditional loads from clamped addresses, the masks applied afterwards: a load under a lane condition becomes a branch)
  const bool live = threadIdx.x < 63;
#pragma unroll
  for (int u = 0; u < TB; ++u) {
    const int d = TB * J.T + u;
    const gptr_t gp = as_global(Gg) + (d < N ? d : 0) * 27;
    const double m = (d < N && live) ? 1.0 : 0.0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int kept = 2 * J.s + r;
      const int row = kept < 5 ? kept + 2 : 8;                        // MPC state row (SROW)
#pragma unroll
      for (int c = 0; c < 3; ++c) Gd[u][r][c] = gp[row * 3 + c] * m;
    }
  }
}
// The same 30 doubles as a per-lane image in the workspace ([15][64] double2, lane-major: fully coalesced 16-byte loads) -- the hot
// loop and the termination test reload them at every call.
__device__ __forceinline__ void load_G_image(double (&Gd)[TB][2][3], const double *gimg, int l) {
  typedef double dbl2_t __attribute__((ext_vector_type(2)));
  typedef const dbl2_t __attribute__((address_space(1))) *g2ptr_t;
  const g2ptr_t gp = (g2ptr_t)gimg + l;
#pragma unroll
  for (int m = 0; m < 15; ++m) {
    const dbl2_t a = gp[m * 64];
    const int e0 = 2 * m, e1 = 2 * m + 1;
    Gd[e0 / 6][(e0 % 6) / 3][e0 % 3] = a.x; Gd[e1 / 6][(e1 % 6) / 3][e1 % 3] = a.y;
  }
}
__device__ __forceinline__ void store_G_image(const double (&Gd)[TB][2][3], double *gimg, int l) {
  double2 *gp = reinterpret_cast<double2 *>(gimg) + l;
#pragma unroll
  for (int m = 0; m < 15; ++m) {
    const int e0 = 2 * m, e1 = 2 * m + 1;
    gp[m * 64] = make_double2(Gd[e0 / 6][(e0 % 6) / 3][e0 % 3], Gd[e1 / 6][(e1 % 6) / 3][e1 % 3]);
  }
}

// stage 3, block (I, T): partial sums of (CCs v)_i, i = 5I + e, rows 2s, 2s + 1, over the lags 5T..5T+4, from the variable vector
// in LDS (four doubles per step behind XPADS zero steps) -> record rec3: [e][6 kept rows]
__device__ __forceinline__ void stage3_partials(const double (&Gd)[TB][2][3], const TJob &J) {
  const double *xp = s_w + XT_OFF + 4 * (TB * J.D - 4 + XPADS);
  double f[9][3];
#pragma unroll
  for (int m = 0; m < 9; ++m) {
    const double2 a = *reinterpret_cast<const double2 *>(xp + 4 * m);
    f[m][0] = a.x; f[m][1] = a.y; f[m][2] = xp[4 * m + 2];
  }
  WAVE_LDS_PHASE();
  double acc[TB][2];
#pragma unroll
  for (int e = 0; e < TB; ++e) { acc[e][0] = 0.0; acc[e][1] = 0.0; }
#pragma unroll
  for (int u = 0; u < TB; ++u)
#pragma unroll
    for (int e = 0; e < TB; ++e)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[e][r] = fma(Gd[u][r][c], f[e - u + 4][c], acc[e][r]);
  double2 *out = reinterpret_cast<double2 *>(s_w + P3_OFF + J.rec3 * P3_REC + 2 * J.s);
#pragma unroll
  for (int e = 0; e < TB; ++e) out[3 * e] = make_double2(acc[e][0], acc[e][1]);
}
// ... and the totals of the lane's three kept rows 3h..3h+2 of step istep: a3[T] = LDS index of those rows in the record of
// block (I, T), or of the zero block
__device__ __forceinline__ void stage3_totals(const int (&a3)[6], double (&out)[3]) {
  double v[6][3];
#pragma unroll
  for (int T = 0; T < 6; ++T)
#pragma unroll
    for (int c = 0; c < 3; ++c) v[T][c] = s_w[a3[T] + c];
  WAVE_LDS_PHASE();
#pragma unroll
  for (int c = 0; c < 3; ++c) out[c] = ((v[0][c] + v[1][c]) + (v[2][c] + v[3][c])) + (v[4][c] + v[5][c]);
}
// stage 1, block (I, T): partial sums of (CCs' v)_j, j = 5(I - T) + e, all three variables, from rows 2s, 2s + 1 of the state-row
// vector in LDS (six per step), steps 5I..5I+8 -> record of the lane: [e][3 + one pad]
__device__ __forceinline__ void stage1_partials(const double (&Gd)[TB][2][3], const TJob &J) {
  const double2 *wp = reinterpret_cast<const double2 *>(s_w + WS_OFF + ws_idx(TB * J.I) + 2 * J.s);
  double wv[9][2];
#pragma unroll
  for (int m = 0; m < 9; ++m) { const double2 a = wp[ws_idx(m) / 2];      /* step 5 I + m: ws_idx(5 I + m) = ws_idx(5 I) + ws_idx(m) */ wv[m][0] = a.x; wv[m][1] = a.y; }
  WAVE_LDS_PHASE();
  double acc[TB][3];
#pragma unroll
  for (int e = 0; e < TB; ++e)
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[e][c] = 0.0;
#pragma unroll
  for (int u = 0; u < TB; ++u)
#pragma unroll
    for (int e = 0; e < TB; ++e)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[e][c] = fma(Gd[u][r][c], wv[e + u][r], acc[e][c]);
  double *out = s_w + P1_OFF + threadIdx.x * P1_REC;
#pragma unroll
  for (int e = 0; e < TB; ++e) {
    *reinterpret_cast<double2 *>(out + 4 * e) = make_double2(acc[e][0], acc[e][1]);
    out[4 * e + 2] = acc[e][2];
  }
}
// ... and the totals of step istep in BOTH lanes of its pair: a block diagonal has up to 18 records (six lag blocks x three row
// pairs); the even lane adds nine of them, the odd lane the other nine (a1[k]: LDS index of the step's three sums in record k of
// this lane's half, or of the zero block), then the two halves meet over DPP
__device__ __forceinline__ void stage1_totals(const int (&a1)[9], double (&out)[3]) {
  double v[9][3];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const double2 a = *reinterpret_cast<const double2 *>(s_w + a1[k]);
    v[k][0] = a.x; v[k][1] = a.y; v[k][2] = s_w[a1[k] + 2];
  }
  WAVE_LDS_PHASE();
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double t = (((v[0][c] + v[1][c]) + (v[2][c] + v[3][c])) + ((v[4][c] + v[5][c]) + (v[6][c] + v[7][c]))) + v[8][c];
    out[c] = t + dpp<DPP_XOR1>(t);
  }
}
// LDS addresses of the records a lane consumes: its step istep = 5 I + e, kept rows 3h..3h+2 (stage 3); its pair's half of the
// block diagonal D = istep / 5 (stage 1): lag blocks 3 par .. 3 par + 2, row pairs 0..2
struct Consume { int a3[6], a1[9]; };
__device__ __forceinline__ Consume consume_addresses(const Role &R) {
  Consume Q;
  const bool in = R.istep < 6 * TB;
  const int i = in ? R.istep : 0, I = i / TB, e = i - TB * I;
#pragma unroll
  for (int T = 0; T < 6; ++T) Q.a3[T] = (in && T <= I) ? P3_OFF + (I * (I + 1) / 2 + T) * P3_REC + 6 * e + 3 * R.h : ZP_OFF;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int T = 3 * R.par + k / 3, sp = k % 3, II = I + T;              // (I = the diagonal D of the variable's step here)
    Q.a1[k] = (in && II <= 5) ? P1_OFF + (21 * sp + II * (II + 1) / 2 + T) * P1_REC + 4 * e : ZP_OFF;
  }
  return Q;
}

// ---- y = M v for a symmetric block image: lane (r, s) holds blocks A = (r, cA) and Bk = (r, cB) and uses each twice -- y_r += B x_c
// and y_c += B' x_r, the second fetched by its owner with ds_bpermute.  All four lanes of quad r receive y[6r..6r+5].
template <bool VIA_LDS = false>
__device__ __forceinline__ void sym_matvec_core(const double (&A)[6][6], const double (&Bk)[6][6], const double (&xr)[6],
                                                const double (&xa)[6], const double (&xb)[6], const Role &R, double (&y)[6]) {
  const int rr = R.r < NB ? R.r : NB - 1;
  double yd[6], ytA[6], ytB[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) { s0 = fma(A[i][j], xa[j], s0); s1 = fma(Bk[i][j], xb[j], s1); }
    yd[i] = s0 + s1;
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { s0 = fma(A[i][j], xr[i], s0); s1 = fma(Bk[i][j], xr[i], s1); }
    ytA[j] = s0; ytB[j] = s1;
  }
  // the transposed products belong to block rows cA / cB: their quads fetch them (lane (c, s) from lane (c + k, s))
  const int srcA = 4 * ((rr + 2 * R.s + 1) % NB) + R.s, srcB = 4 * ((rr + 2 * R.s + 2) % NB) + R.s;
  if (VIA_LDS) {
    // ... through LDS records of 14 doubles per lane (six 16-byte writes, six 16-byte reads; the partial-sum records of the Toeplitz
    // stages are dead while the mat-vec runs) instead of 24 ds_bpermute_b32: a bpermute costs the wave ~22 cycles under the
    // contention of four waves per CU (tools/micro/w2_iteration.hip: 520 cycles of an iteration for the 24, 370 for this form)
    double *o = s_w + MVX_OFF + R.l * MVX_REC;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      *reinterpret_cast<double2 *>(o + 2 * m) = make_double2(ytA[2 * m], ytA[2 * m + 1]);
      *reinterpret_cast<double2 *>(o + 6 + 2 * m) = make_double2(ytB[2 * m], ytB[2 * m + 1]);
    }
    wave_lds_sync();
    const double *pa = s_w + MVX_OFF + srcA * MVX_REC, *pb = s_w + MVX_OFF + srcB * MVX_REC + 6;
    double qa[6], qb[6];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const double2 a = *reinterpret_cast<const double2 *>(pa + 2 * m), b = *reinterpret_cast<const double2 *>(pb + 2 * m);
      qa[2 * m] = a.x; qa[2 * m + 1] = a.y; qb[2 * m] = b.x; qb[2 * m + 1] = b.y;
    }
    WAVE_LDS_PHASE();
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double sm = yd[j] + qa[j] + (R.s < 3 ? qb[j] : 0.0);
      sm += dpp<DPP_XOR1>(sm);
      sm += dpp<DPP_XOR2>(sm);
      y[j] = sm;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const double pa = bperm(ytA[j], srcA), pb = bperm(ytB[j], srcB);
    double sm = yd[j] + pa + (R.s < 3 ? pb : 0.0);
    sm += dpp<DPP_XOR1>(sm);
    sm += dpp<DPP_XOR2>(sm);
    y[j] = sm;
  }
}
// the A blocks of the KKT inverse: LDS image (chunks 0..17, as the factorisation's scatter leaves them) -> workspace
// ([18][64] double2, lane-major), once per factorisation; workspace -> registers at every entry of the hot loop
__device__ __forceinline__ void store_KA_image(double *kimg, int l) {
  const int lb = l < NL ? l : NL - 1;
  const double2 *ki = reinterpret_cast<const double2 *>(s_w) + lb;
  double2 *dst = reinterpret_cast<double2 *>(kimg) + l;
#pragma unroll
  for (int m = 0; m < 18; ++m) dst[m * 64] = ki[m * NL];
}
__device__ __forceinline__ void load_KA_image(double (&A)[6][6], const double *kimg, int l) {
  typedef double dbl2_t __attribute__((ext_vector_type(2)));
  typedef const dbl2_t __attribute__((address_space(1))) *g2ptr_t;
  const g2ptr_t gp = (g2ptr_t)kimg + l;
#pragma unroll
  for (int m = 0; m < 18; ++m) {
    const dbl2_t a = gp[m * 64];
    A[(2 * m) / 6][(2 * m) % 6] = a.x; A[(2 * m + 1) / 6][(2 * m + 1) % 6] = a.y;
  }
}
// x~ = K^-1 rhs in the hot loop: A blocks in registers, B blocks from LDS (chunks 18..35 of the image), rhs in natural order (LDS)
__device__ __forceinline__ void kkt_matvec(const double (&A)[6][6], const double *v, const Role &R, double (&y)[6]) {
  // (the four idle lanes read the chunks of lanes 36..39 -- lanes of their own 16-lane ds_read_b128 group, i.e. a broadcast; reading
  //  lane 59's, as before round 5, put them on the banks of lane 43: one extra LDS cycle on each of the 18 reads)
  const int lb = R.l < NL ? R.l : R.l - 24;
  const int rr = R.r < NB ? R.r : NB - 1;
  double Bk[6][6];
  const double2 *ki = reinterpret_cast<const double2 *>(s_w) + lb;
#pragma unroll
  for (int m = 0; m < 18; ++m) {
    const double2 b = ki[(18 + m) * NL];
    Bk[(2 * m) / 6][(2 * m) % 6] = b.x; Bk[(2 * m + 1) / 6][(2 * m + 1) % 6] = b.y;
  }
  double xr[6], xa[6], xb[6];
  const double2 *pr = reinterpret_cast<const double2 *>(v + 6 * rr), *pa = reinterpret_cast<const double2 *>(v + 6 * R.cA),
                *pb = reinterpret_cast<const double2 *>(v + 6 * R.cB);
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const double2 a = pr[m], b = pa[m], c = pb[m];
    xr[2 * m] = a.x; xr[2 * m + 1] = a.y; xa[2 * m] = b.x; xa[2 * m + 1] = b.y; xb[2 * m] = c.x; xb[2 * m + 1] = c.y;
  }
  WAVE_LDS_PHASE();
  sym_matvec_core<true>(A, Bk, xr, xa, xb, R, y);
}
// P x in the termination test: both blocks from the block image of P in the workspace (36 coalesced 16-byte loads), x from the
// iteration's variable vector in LDS (four per step behind XPADS zero steps)
__device__ __forceinline__ void p_matvec(const double *Pg, const Role &R, double (&y)[6]) {
  typedef double dbl2_t __attribute__((ext_vector_type(2)));
  typedef const dbl2_t __attribute__((address_space(1))) *g2ptr_t;
  const g2ptr_t pb = (g2ptr_t)Pg + R.l;
  double A[6][6], Bk[6][6];
#pragma unroll
  for (int m = 0; m < 18; ++m) {
    const dbl2_t a = pb[m * 64], b = pb[(18 + m) * 64];
    A[(2 * m) / 6][(2 * m) % 6] = a.x; A[(2 * m + 1) / 6][(2 * m + 1) % 6] = a.y;
    Bk[(2 * m) / 6][(2 * m) % 6] = b.x; Bk[(2 * m + 1) / 6][(2 * m + 1) % 6] = b.y;
  }
  const int rr = R.r < NB ? R.r : NB - 1;
  const double *x4 = s_w + XT_OFF + 4 * XPADS;
  double xr[6], xa[6], xb[6];
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    const int o = 4 * (m / 3) + m % 3;                      // block row = two steps of four doubles
    xr[m] = x4[8 * rr + o]; xa[m] = x4[8 * R.cA + o]; xb[m] = x4[8 * R.cB + o];
  }
  WAVE_LDS_PHASE();
  sym_matvec_core(A, Bk, xr, xa, xb, R, y);
}

// ----------------------------------------------------------------------------------------------------------------
// A'WA of the kept state rows as lower-triangular matrix-core tiles (the Gram product of f16_mpc_solve.hip: k-step kk covers rows
// 4kk..4kk+3 of the 6N x 3N block-Toeplitz matrix, gathered from the nine-row G image in LDS; causality skips the k-steps
// before gram_kk0), one tile row at a time in this wavefront, + the command / rate rows as a diagonal / third-off-diagonal
// fix-up; parked in the workspace (it does not depend on rho).
__host__ __device__ constexpr int gram_kk0(int T) { return (6 * ((16 * T) / 3)) >> 2; }
template <int W>
__device__ __forceinline__ void gram_row(double *gw, int N, int lc, int lq, int lane) {
  const int n = 3 * N;
  const double *Gl = s_w + GL_OFF, *Wg = s_w + WG_OFF, *Wcv = s_w + WCV_OFF, *Wrv = s_w + WRV_OFF;
  d4_t acc[W + 1];
#pragma unroll
  for (int J = 0; J <= W; ++J) acc[J] = d4_t{0.0, 0.0, 0.0, 0.0};
  int offT[W + 1], jT[W + 1];
#pragma unroll
  for (int T = 0; T <= W; ++T) {
    const int col = 16 * T + lc;
    jT[T] = col < n ? col / 3 : 1 << 20;
    offT[T] = (col - 3 * (col / 3)) - 27 * (col / 3);
  }
  const int nk = (6 * N + 3) >> 2;
  for (int kk = gram_kk0(W); kk < nk; ++kk) {
    const int rw = 4 * kk + lq, i = rw / 6, rr = rw - 6 * i;
    const int base = 27 * i + 3 * (rr < 5 ? rr + 2 : 8);
    const double wgt = Wg[rw];                           // 0 beyond row 6N
    const int ia = base + offT[W];
    const double ga = Gl[ia < 0 ? 0 : ia];
    const double a_op = i >= jT[W] ? ga : 0.0;
#pragma unroll
    for (int J = 0; J <= W; ++J) {
      const int ib = base + offT[J];
      const double gb = Gl[ib < 0 ? 0 : ib];
      const double b_op = i >= jT[J] ? gb * wgt : 0.0;
      acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op, b_op, acc[J], 0, 0, 0);
    }
  }
#pragma unroll
  for (int J = 0; J <= W; ++J)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int i = 16 * W + 4 * qq + lq, j = 16 * J + lc;
      double v = acc[J][qq];
      if (i < n && j < n) {
        if (i == j) v += Wcv[i] + Wrv[i] + Wrv[i + 3];
        else if (i == j + 3) v -= Wrv[i];
        else if (j == i + 3) v -= Wrv[j];
      }
      gw[(tile_idx(W, J) * 4 + qq) * 64 + lane] = v;
    }
}
__device__ __noinline__ void gram_tiles(double *gw, int N) {
  const int l = threadIdx.x, lc = l & 15, lq = l >> 4;
  gram_row<0>(gw, N, lc, lq, l);
  gram_row<1>(gw, N, lc, lq, l);
  gram_row<2>(gw, N, lc, lq, l);
  gram_row<3>(gw, N, lc, lq, l);
  gram_row<4>(gw, N, lc, lq, l);
  gram_row<5>(gw, N, lc, lq, l);
}

// ---- the blocked symmetric sweep (f16_mpc_solve.hip: inverse_step) on all 21 lower-triangular tiles in ONE wavefront.
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
}
__device__ __forceinline__ bool inv4_spd(const double (&d)[4][4], double (&o)[4][4]) {
  const double a = d[0][0], b = d[1][0], c = d[1][1];
  const double detA = a * c - b * b;
  const double ia = rcp_nr(detA);
  const double A00 = c * ia, A10 = -b * ia, A11 = a * ia;
  const double B00 = d[2][0], B01 = d[2][1], B10 = d[3][0], B11 = d[3][1];
  const double T00 = B00 * A00 + B01 * A10, T01 = B00 * A10 + B01 * A11;
  const double T10 = B10 * A00 + B11 * A10, T11 = B10 * A10 + B11 * A11;
  const double S00 = d[2][2] - (T00 * B00 + T01 * B01);
  const double S10 = d[3][2] - (T10 * B00 + T11 * B01);
  const double S11 = d[3][3] - (T10 * B10 + T11 * B11);
  const double detS = S00 * S11 - S10 * S10;
  const double is = rcp_nr(detS);
  const double I00 = S11 * is, I10 = -S10 * is, I11 = S00 * is;
  const double L00 = -(I00 * T00 + I10 * T10), L01 = -(I00 * T01 + I10 * T11);
  const double L10 = -(I10 * T00 + I11 * T10), L11 = -(I10 * T01 + I11 * T11);
  o[2][2] = I00; o[3][2] = o[2][3] = I10; o[3][3] = I11;
  o[2][0] = o[0][2] = L00; o[2][1] = o[1][2] = L01; o[3][0] = o[0][3] = L10; o[3][1] = o[1][3] = L11;
  o[0][0] = A00 - (T00 * L00 + T10 * L10);
  o[1][0] = o[0][1] = A10 - (T01 * L00 + T11 * L10);
  o[1][1] = A11 - (T01 * L01 + T11 * L11);
  return a > 0.0 && detA > 0.0 && S00 > 0.0 && detS > 0.0;
}
__device__ __forceinline__ double sel4(double v0, double v1, double v2, double v3, int k) {
  const double lo = (k & 1) ? v1 : v0, hi = (k & 1) ? v3 : v2;
  return (k & 2) ? hi : lo;
}
// pan: C[col][0..3] = M[pivot row][col] (FN x 4).  Publish pivots 16 Kt + 4 KQ .. + 3 from the tiles; the pivot block's inverse: P.
struct PivInv { double2 a0, a1; double dpiv; };      // row l / 16 of the pivot block's inverse, and its element (l / 16, l % 4)
template <int Kt, int KQ>
__device__ __forceinline__ void publish_panel(const d4_t (&acc)[NTILES], double *pan, int lc, int lq, PivInv &P, bool &ok) {
  // the part of the pivot rows left of and inside the diagonal tile: tile row Kt, register KQ
#pragma unroll
  for (int J = 0; J <= Kt; ++J) pan[(16 * J + lc) * 4 + lq] = acc[tile_idx(Kt, J)][KQ];
  // the part right of the diagonal tile = the pivot COLUMNS of the tile rows below (symmetry): lanes whose column is a pivot
  if ((lc >> 2) == KQ) {
#pragma unroll
    for (int w = Kt + 1; w < NT; ++w)
#pragma unroll
      for (int q = 0; q < 4; ++q) pan[(16 * w + 4 * q + lq) * 4 + (lc & 3)] = acc[tile_idx(w, Kt)][q];
  }
  wave_lds_sync();
  double D[4][4], Di[4][4];
  const double2 *src = reinterpret_cast<const double2 *>(pan + (16 * Kt + 4 * KQ) * 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double2 u = src[2 * i], v = src[2 * i + 1];
    D[i][0] = u.x; D[i][1] = u.y; D[i][2] = v.x; D[i][3] = v.y;
  }
  // every lane holds the inverse: what it needs of it (row l / 16, one element of that row) comes by selects -- round 3 wrote it to the
  // panel and read it back behind a second sync (~250 cycles of a pivot step)
  const bool good = inv4_spd(D, Di);
  ok = ok && good;
  P.a0 = make_double2(sel4(Di[0][0], Di[1][0], Di[2][0], Di[3][0], lq), sel4(Di[0][1], Di[1][1], Di[2][1], Di[3][1], lq));
  P.a1 = make_double2(sel4(Di[0][2], Di[1][2], Di[2][2], Di[3][2], lq), sel4(Di[0][3], Di[1][3], Di[2][3], Di[3][3], lq));
  P.dpiv = sel4(P.a0.x, P.a0.y, P.a1.x, P.a1.y, lc & 3);
}
template <int Kt, int KQ>
__device__ __forceinline__ void sweep_step(d4_t (&acc)[NTILES], const double *cb, int lc, int lq, double ndel, const PivInv &P) {
  const bool pl = (lc >> 2) == KQ;                       // this lane's column (within a tile) is a pivot column
  double a_op[NT], b_op[NT];
  {
    const double2 a0 = P.a0, a1 = P.a1;                  // row l/16 of D^-1
    const double dpiv = P.dpiv;
#pragma unroll
    for (int w = 0; w < NT; ++w) {
      const double2 *src = reinterpret_cast<const double2 *>(cb + (16 * w + lc) * 4);
      const double2 c0 = src[0], c1 = src[1];
      a_op[w] = -(c0.x * a0.x + c0.y * a0.y + c1.x * a1.x + c1.y * a1.y);
      b_op[w] = cb[(16 * w + lc) * 4 + lq];
    }
    if (pl) { a_op[Kt] = dpiv; b_op[Kt] = ndel; }
  }
  // pivot columns (tile column Kt, tile rows Kt..) and pivot rows (tile row Kt, register KQ) start from zero
#pragma unroll
  for (int w = Kt; w < NT; ++w)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[tile_idx(w, Kt)][q] = pl ? 0.0 : acc[tile_idx(w, Kt)][q];
#pragma unroll
  for (int J = 0; J <= Kt; ++J) acc[tile_idx(Kt, J)][KQ] = 0.0;
#pragma unroll
  for (int w = 0; w < NT; ++w)
#pragma unroll
    for (int J = 0; J <= w; ++J)
      acc[tile_idx(w, J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[w], b_op[J], acc[tile_idx(w, J)], 0, 0, 0);
}
template <int Kt>
__device__ __forceinline__ void sweep_tile_row(d4_t (&acc)[NTILES], double *c0, double *c1, int lc, int lq, double ndel, PivInv &P, bool &ok) {
  // entry: the panel of pivots (Kt, 0) is in c0, the pivot block's inverse in P
  sweep_step<Kt, 0>(acc, c0, lc, lq, ndel, P);
  publish_panel<Kt, 1>(acc, c1, lc, lq, P, ok);
  sweep_step<Kt, 1>(acc, c1, lc, lq, ndel, P);
  publish_panel<Kt, 2>(acc, c0, lc, lq, P, ok);
  sweep_step<Kt, 2>(acc, c0, lc, lq, ndel, P);
  publish_panel<Kt, 3>(acc, c1, lc, lq, P, ok);
  sweep_step<Kt, 3>(acc, c1, lc, lq, ndel, P);
  if (Kt + 1 < NT) publish_panel<(Kt + 1 < NT ? Kt + 1 : 0), 0>(acc, c0, lc, lq, P, ok);
}

// (Round 4, measured and not kept: the six products that feed the next pivot's rows first, the other fifteen pinned one at a time
// between the pieces of the 4 x 4 inverse -- 90.6 k against 92.2 k cycles per sweep, because there is no shadow to work in:
// v_mfma_f64_16x16x4_f64 holds the wave for 64 cycles and every fp64 vector FMA issued between two products ADDS its ~7 cycles
// (tools/micro/mfma_f64_rate.hip: 64.0 / 92.5 / 112.5 / 132.5 cycles per product with 0 / 4 / 8 / 12 FMAs behind it) -- the fp64
// matrix instruction runs on the same fp64 lanes as the vector FMA.  A variant that computes the next pivot block ahead of the
// products, D' - C' D^-1 C'^T, to save the second sync of a step was slower for the same reason: 109 k.)
#ifdef F16_EXP_STAMPW
__device__ int g_stamp_wg = -1;                  // the workgroup that solves job 0 of the launch (while it does)
#define STAMP_WG() (blockIdx.x == g_stamp_wg)
__device__ unsigned long long g_wstamp[16];      // diagnostic build: cycles per phase of the iterations of workgroup 0, + counts
__device__ unsigned long long g_tstamp[8];       // phases of the termination test
#define TSTAMP(i) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); if (STAMP_WG() && threadIdx.x == 0) g_tstamp[i] += t1_ - tt0_; tt0_ = t1_; }
#define WSTAMP(i) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); tS[i] += t1_ - t0_; t0_ = t1_; }
#else
#define WSTAMP(i)
#define TSTAMP(i)
#endif
// One KKT factorisation: K = c P + sigma D^-2 + rho A'WA as lower-triangular tiles (identity on the padding), the sweep, and
// the scatter of the inverse into the symmetric block image in LDS (table KSCAT).  sg2v: sigma D^-2 per variable (LDS).
// (okp: an out-parameter in the caller's frame on purpose -- a call that passes no pointer into the caller's frame is marked as a
// tail call, which switches the "no callee-saved registers" treatment of internal functions off: 248 scratch stores + loads per call.)
__device__ __noinline__ void factorise(const double *Pg, const double *gw, const double *sg2v, double *kimg, int N, double cs, double rho, int *okp) {
  const int n = 3 * N, l = threadIdx.x, lc = l & 15, lq = l >> 4;
#ifdef F16_EXP_STAMPW
  unsigned long long tt0_ = __builtin_amdgcn_s_memtime();
#endif
  d4_t acc[NTILES];
  {
    // TWO batches of loads (unconditional, clamped addresses) for the whole matrix, then the arithmetic: a load under a lane
    // condition becomes a branch around it and every element then waits out its own memory round trip; a batch per tile row
    // (round 3) was six round trips to the Infinity Cache in sequence
    const gptr_t Pgg = as_global(Pg), gwg = as_global(gw);
    double dg[NT][4];
#pragma unroll
    for (int w = 0; w < NT; ++w)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) { const int i = 16 * w + 4 * qq + lq; dg[w][qq] = sg2v[i < n ? i : 0]; }
#pragma unroll
    for (int w = 0; w < NT; ++w)
#pragma unroll
      for (int J = 0; J <= w; ++J)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) acc[tile_idx(w, J)][qq] = gwg[(tile_idx(w, J) * 4 + qq) * 64 + l];
    __builtin_amdgcn_sched_barrier(0);
    double pv[NTILES][4];
#pragma unroll
    for (int w = 0; w < NT; ++w)
#pragma unroll
      for (int J = 0; J <= w; ++J)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int i = 16 * w + 4 * qq + lq, j = 16 * J + lc;
          const bool in = i < n && j < n;
          const int hi_ = i >= j ? i : j, lo_ = i >= j ? j : i;
          pv[tile_idx(w, J)][qq] = Pgg[in ? hi_ * (hi_ + 1) / 2 + lo_ : 0];
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int w = 0; w < NT; ++w)
#pragma unroll
      for (int J = 0; J <= w; ++J)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int i = 16 * w + 4 * qq + lq, j = 16 * J + lc;
          const bool in = i < n && j < n;
          const double kin = cs * pv[tile_idx(w, J)][qq] + rho * acc[tile_idx(w, J)][qq] + (i == j ? dg[w][qq] : 0.0);
          acc[tile_idx(w, J)][qq] = in ? kin : (i == j ? 1.0 : 0.0);
        }
  }
  double *c0 = s_w + PAN_OFF, *c1 = s_w + PAN_OFF + PAN_SIZE;
  bool ok = true;
  const double ndel = (lq == (lc & 3)) ? -1.0 : 0.0;
  wave_lds_sync();
  TSTAMP(5)
  PivInv P;
  publish_panel<0, 0>(acc, c0, lc, lq, P, ok);
  sweep_tile_row<0>(acc, c0, c1, lc, lq, ndel, P, ok);
  sweep_tile_row<1>(acc, c0, c1, lc, lq, ndel, P, ok);
  sweep_tile_row<2>(acc, c0, c1, lc, lq, ndel, P, ok);
  sweep_tile_row<3>(acc, c0, c1, lc, lq, ndel, P, ok);
  sweep_tile_row<4>(acc, c0, c1, lc, lq, ndel, P, ok);
  sweep_tile_row<5>(acc, c0, c1, lc, lq, ndel, P, ok);
  wave_lds_sync();
  TSTAMP(6)
  // acc = MINUS the inverse: scatter into the block image (the panels are dead); the table entries of all tiles as one batch
  {
    unsigned dd[NTILES][4];
#pragma unroll
    for (int t = 0; t < NTILES; ++t)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) dd[t][qq] = F16_WAVE_KSCAT[(t * 4 + qq) * 64 + l];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NTILES; ++t)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const unsigned d0 = dd[t][qq] & 0xFFFFu, d1 = dd[t][qq] >> 16;
        const double v = -acc[t][qq];
        if (d0 != 0xFFFFu) s_w[d0] = v;
        if (d1 != 0xFFFFu) s_w[d1] = v;
      }
  }
  wave_lds_sync();
  store_KA_image(kimg, l);                                  // the A blocks: to the workspace (registers of the hot loop); their half
  wave_lds_sync();                                          // of the LDS image is the iterations' partial-sum space from here on
  TSTAMP(7)
  *okp = __ballot(!ok) == 0 ? 1 : 0;
}

// ----------------------------------------------------------------------------------------------------------------
// The iterations (f16_mpc_solve.hip: admm_iterate, rule for rule) as THREE out-of-line pieces with register allocations of
// their own: run_iterations (the hot loop: nothing but what an iteration needs is live in it), terminate_test (every
// check_every iterations) and start_point (after a factorisation).  The state passes through the SolveState in memory; the
// 72 Toeplitz doubles are (re)loaded by whoever needs them (L2-resident).  The last iteration before a test is peeled,
// because only it must keep the dual step dy (primal-infeasibility certificate).  ANYEQ = false: no row of this aircraft
// carries the 1e3 rho of an equality row (the common case): rho and its reciprocal are wave-uniform scalars.
__device__ __forceinline__ double uniform_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

template <bool ANYEQ>
struct RowOps {                                  // per-row rho and 1 / rho of a lane's six rows
  double rho1, rhoE, rinv1, rinvE;
  int eqA, eqB;
  __device__ __forceinline__ RowOps(double rho, const LaneConst &C) {
    rho1 = uniform_f64(rho); rhoE = rho1 * OSQP_RHO_EQ_OVER_RHO_INEQ;
    rinv1 = uniform_f64(1.0 / rho1); rinvE = uniform_f64(1.0 / rhoE);
    eqA = C.eqA; eqB = C.eqB;
  }
  __device__ __forceinline__ double roA(int c) const { return ANYEQ ? (((eqA >> c) & 1) ? rhoE : rho1) : rho1; }
  __device__ __forceinline__ double riA(int c) const { return ANYEQ ? (((eqA >> c) & 1) ? rinvE : rinv1) : rinv1; }
  __device__ __forceinline__ double roB(int c) const { return ANYEQ ? (((eqB >> c) & 1) ? rhoE : rho1) : rho1; }
  __device__ __forceinline__ double riB(int c) const { return ANYEQ ? (((eqB >> c) & 1) ? rinvE : rinv1) : rinv1; }
};
struct RowSlots {                                // where a lane's rows live in the LDS vectors
  double *wsn, *wB;
  bool act;
  __device__ __forceinline__ RowSlots(const Role &R) {
    wsn = s_w + WS_OFF + ws_idx(R.istep) + 3 * R.h;
    wB = s_w + (R.par ? WR_OFF : WC_OFF) + 3 * R.istep;
    act = R.act;
  }
  __device__ __forceinline__ void put(const double (&vA)[3], const double (&vB)[3]) const {      // row vectors -> LDS
    if (act) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { wsn[c] = vA[c]; wB[c] = vB[c]; }
    }
  }
};
// the lane's variable step (and the one before it) of the x vector in LDS: four doubles per step behind XPADS zero steps
__device__ __forceinline__ double *x4_step(int step) { return s_w + XT_OFF + 4 * (step + XPADS); }

// Iterations from the state in *st (w of that point is in LDS) up to the next termination test that needs MORE than the
// primal side, block of check_every iterations after block.  After every block the primal half of OSQP's test runs here,
// on the operands this function holds anyway (A x of the new point by one more stage 3, the primal residual and its scale,
// the two scalars of the primal-infeasibility certificate): the test cannot pass while the primal residual is above its
// bound, so on a block end that is no rho-update iteration, not the last iteration and no certificate candidate the
// decision is "go on" whatever P x and A' y are -- same decisions as evaluating everything, and the P image (37 KB from the
// Infinity Cache), the second Toeplitz pass and the reload of this function's operands are paid on a third of the block ends
// (config-4 batch: 5.6 of 18.3).  Leaves the state, the dual step of the last iteration, w of the new point, st->it / to_check.
template <bool ANYEQ>
__device__ __noinline__ void run_iterations(SolveState *st, const LaneConst *lcp, const double *Gg, const double *kimg, int N, IterSettings oa) {
  IterSettings o;                                           // (arguments of a device function arrive in vector registers)
  o.alpha = uniform_f64(oa.alpha); o.eps_abs = uniform_f64(oa.eps_abs); o.eps_rel = uniform_f64(oa.eps_rel);
  o.eps_prim_inf = uniform_f64(oa.eps_prim_inf);
  o.max_iter = __builtin_amdgcn_readfirstlane(oa.max_iter); o.check_every = __builtin_amdgcn_readfirstlane(oa.check_every);
  o.rho_every = __builtin_amdgcn_readfirstlane(oa.rho_every); o.adaptive_rho = __builtin_amdgcn_readfirstlane(oa.adaptive_rho);
  const double alpha = o.alpha;
  const Role R = role(N);
  const TJob J = tjob();
  const Consume Q = consume_addresses(R);
  const LaneConst C = *lcp;
  double Gd[TB][2][3], KA[6][6];
  load_G_image(Gd, Gg, R.l);                                // (Gg: the per-lane image)
  load_KA_image(KA, kimg, R.l);
  double x[3], zA[3], yA[3], zB[3], yB[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { x[c] = st->x[c]; zA[c] = st->zA[c]; yA[c] = st->yA[c]; zB[c] = st->zB[c]; yB[c] = st->yB[c]; }
  const RowOps<ANYEQ> ro(st->rho, C);
  const RowSlots slots(R);
  double *const wc = s_w + WC_OFF, *const wr = s_w + WR_OFF;
  double *const rhs = wc;                                   // (the owner of k reads wc[k] before it writes rhs[k])
  const int kx = 3 * R.istep, kxa = R.act ? kx : 0;         // first owned variable / command / rate row
  const bool wrx = R.act && R.par == 0;                     // one lane of the pair writes what both own
  const double *const xk4 = x4_step(R.act ? R.istep : 0);
  double dyA[3] = {0.0, 0.0, 0.0}, dyB[3] = {0.0, 0.0, 0.0};
  if (R.l < 8) s_w[ZP_OFF + R.l] = 0.0;                     // (the factorisation's scratch: the zero block is ours from here)
  wave_lds_sync();
#ifdef F16_EXP_STAMPW
  unsigned long long tS[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = __builtin_amdgcn_s_memtime();
#endif
  auto iteration = [&](auto keep) {
    constexpr bool KEEP = decltype(keep)::value;
    WSTAMP(7)
    // A: rhs = sigma D^-2 x - c q + A' W (rho z - y)
    stage1_partials(Gd, J);
    wave_lds_sync();
    {
      double t1[3];
      stage1_totals(Q.a1, t1);
      double wce[3], wre[3], wrn[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { wce[c] = wc[kxa + c]; wre[c] = wr[kxa + c]; wrn[c] = wr[kxa + c + 3]; }
      WAVE_LDS_PHASE();
      if (wrx) {
#pragma unroll
        for (int c = 0; c < 3; ++c) rhs[kx + c] = C.sg[c] * x[c] - C.cq[c] + (t1[c] + wce[c] + (wre[c] - wrn[c]));
      }
    }
    WSTAMP(0)
    wave_lds_sync();
    WSTAMP(1)
    // B: x~ = K^-1 rhs
    {
      double y6[6];
      kkt_matvec(KA, rhs, R, y6);
      WSTAMP(2)
      if (R.s == 0 && R.r < NB) {                           // (x~ has a buffer of its own: nothing to wait for)
        double *const o = x4_step(2 * R.r);
        *reinterpret_cast<double2 *>(o) = make_double2(y6[0], y6[1]); o[2] = y6[2];
        *reinterpret_cast<double2 *>(o + 4) = make_double2(y6[3], y6[4]); o[6] = y6[5];
      }
    }
    wave_lds_sync();
    WSTAMP(3)
    // C: z~ = A x~, relaxation, projection, dual update (unscaled z, y = yb / E); w of the new point
    stage3_partials(Gd, J);
    wave_lds_sync();
    {
      double z3[3], xk[3], xkm[3];
      stage3_totals(Q.a3, z3);
      WSTAMP(4)
      {
        const double2 a = *reinterpret_cast<const double2 *>(xk4), b = *reinterpret_cast<const double2 *>(xk4 - 4);
        xk[0] = a.x; xk[1] = a.y; xk[2] = xk4[2]; xkm[0] = b.x; xkm[1] = b.y; xkm[2] = xk4[-2];
      }
      WAVE_LDS_PHASE();
      double wA[3], wBv[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        x[c] = alpha * xk[c] + (1 - alpha) * x[c];
        {
          const double zr = alpha * z3[c] + (1 - alpha) * zA[c];
          const double zn = fmin(fmax(fma(yA[c], ro.riA(c), zr), C.loA[c]), C.hiA[c]);
          const double d = ro.roA(c) * (zr - zn);
          if (KEEP) dyA[c] = d;
          yA[c] = yA[c] + d; zA[c] = zn;
          wA[c] = C.WA[c] * (ro.roA(c) * zA[c] - yA[c]);
        }
        {
          const double zt = R.par ? xk[c] - xkm[c] : xk[c];
          const double zr = alpha * zt + (1 - alpha) * zB[c];
          const double zn = fmin(fmax(fma(yB[c], ro.riB(c), zr), C.loB[c]), C.hiB[c]);
          const double d = ro.roB(c) * (zr - zn);
          if (KEEP) dyB[c] = d;
          yB[c] = yB[c] + d; zB[c] = zn;
          wBv[c] = C.WB[c] * (ro.roB(c) * zB[c] - yB[c]);
        }
      }
      WSTAMP(5)
      slots.put(wA, wBv);                                   // w = W (rho z - y) of the new point
    }
    wave_lds_sync();
    WSTAMP(6)
  };
  int it = __builtin_amdgcn_readfirstlane(st->it), to_check = __builtin_amdgcn_readfirstlane(st->to_check);     // (scalar loop counters)
  double rp = INFINITY;
  for (;;) {
    const int left = o.max_iter - it;
    const int nrun = left < to_check ? left : to_check;       // iterations up to the next test (>= 1)
    for (int k = nrun; k > 1; --k) iteration(std::false_type{});     // (a single loop body that always keeps the dual step: 1 % slower)
    iteration(std::true_type{});
    it += nrun; to_check -= nrun;
    if (to_check == 0) to_check = o.check_every;
    // primal half of the test (terminate_test: same expressions on the same operands)
    if (wrx) {
      double *const xw = x4_step(R.istep);
      *reinterpret_cast<double2 *>(xw) = make_double2(x[0], x[1]); xw[2] = x[2];
    }
    wave_lds_sync();
    stage3_partials(Gd, J);
    wave_lds_sync();
    double ax3[3], v0 = 0.0, v12 = 0.0, v7 = 0.0, v8 = 0.0;
    stage3_totals(Q.a3, ax3);
    if (R.act) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double xkk = xk4[c], xkm = xk4[c - 4];
        const double axB = R.par ? xkk - xkm : xkk;
        v0 = fmax(v0, fmax(fabs(ax3[c] - zA[c]), fabs(axB - zB[c])));
        v12 = fmax(v12, fmax(fmax(fabs(ax3[c]), fabs(axB)), fmax(fabs(zA[c]), fabs(zB[c]))));
        v7 = fmax(v7, fmax(C.WA[c] * fabs(dyA[c]), C.WB[c] * fabs(dyB[c])));
        v8 += C.WA[c] * (C.hiA[c] * fmax(dyA[c], 0.0) + C.loA[c] * fmin(dyA[c], 0.0)) +
              C.WB[c] * (C.hiB[c] * fmax(dyB[c], 0.0) + C.loB[c] * fmin(dyB[c], 0.0));
      }
    }
    v0 = wave_reduce_dpp<false>(v0); v12 = wave_reduce_dpp<false>(v12);
    v7 = wave_reduce_dpp<false>(v7); v8 = wave_reduce_dpp<true>(v8);
    rp = v0;
    const bool prim_ok = rp < o.eps_abs + o.eps_rel * v12;
    const bool cert = v7 > o.eps_prim_inf && v8 < -o.eps_prim_inf * v7;
    const bool full = prim_ok || cert || it >= o.max_iter || (o.adaptive_rho && it % o.rho_every == 0);
    WSTAMP(7)
    if (__builtin_amdgcn_readfirstlane((int)full)) break;   // (wave-uniform: the reductions leave the same value on every lane)
    wave_lds_sync();                                          // (the x~ buffer and the stage-3 records are rewritten by the next iteration)
  }
  st->it = it; st->to_check = to_check; st->rp = rp;
#pragma unroll
  for (int c = 0; c < 3; ++c) { st->x[c] = x[c]; st->zA[c] = zA[c]; st->yA[c] = yA[c]; st->zB[c] = zB[c]; st->yB[c] = yB[c];
                                st->dyA[c] = dyA[c]; st->dyB[c] = dyB[c]; }
#ifdef F16_EXP_STAMPW
  if (STAMP_WG() && R.l == 0)
    for (int i = 0; i < 8; ++i) g_wstamp[i] += tS[i];
#endif
}

// After a factorisation: zero what carries zero padding, then w = W (rho z - y) of the current point.
template <bool ANYEQ>
__device__ __noinline__ void start_point(const SolveState *st, const LaneConst *lcp, int N) {
  const Role R = role(N);
  const LaneConst C = *lcp;
  const RowOps<ANYEQ> ro(st->rho, C);
  const RowSlots slots(R);
  for (int i = R.l; i < WS_SIZE + WC_SIZE + WR_SIZE + XT_SIZE; i += 64) s_w[WS_OFF + i] = 0.0;
  wave_lds_sync();
  double wA[3], wBv[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { wA[c] = C.WA[c] * (ro.roA(c) * st->zA[c] - st->yA[c]); wBv[c] = C.WB[c] * (ro.roB(c) * st->zB[c] - st->yB[c]); }
  slots.put(wA, wBv);
  wave_lds_sync();
}

// The termination test of the point in *st (OSQP, on the UNSCALED problem: A x, P x, A' W y / c), the primal-infeasibility
// certificate and, every rho_every iterations, the rho estimate on the scaled residuals.  Sets done / converged / infeasible,
// returns 1 when rho left the 5x band (st->rho then holds the new value and the caller re-factorises); leaves w of the
// current point in LDS for the next iteration otherwise.
template <bool ANYEQ>
__device__ __noinline__ int terminate_test(SolveState *st, const LaneConst *lcp, const double *Pg, const double *Gg, const double *qv,
                                           const double *Dv, int N, IterSettings o) {
  const Role R = role(N);
  const TJob J = tjob();
  const Consume Q = consume_addresses(R);
  const LaneConst C = *lcp;
  double x[3], zA[3], yA[3], zB[3], yB[3], dyA[3], dyB[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { x[c] = st->x[c]; zA[c] = st->zA[c]; yA[c] = st->yA[c]; zB[c] = st->zB[c]; yB[c] = st->yB[c];
                                dyA[c] = st->dyA[c]; dyB[c] = st->dyB[c]; }
  double rho = uniform_f64(st->rho);
  const RowOps<ANYEQ> ro(rho, C);
  const RowSlots slots(R);
  const int it = st->it;
  double *const wc = s_w + WC_OFF, *const wr = s_w + WR_OFF;
  const int kx = 3 * R.istep, kxa = R.act ? kx : 0;
  const bool wrx = R.act && R.par == 0;
  double *const xk4 = x4_step(R.act ? R.istep : 0);
  bool done = false, converged = false, infeasible = false, refactor = false;
#ifdef F16_EXP_STAMPW
  unsigned long long tt0_ = __builtin_amdgcn_s_memtime();
#endif
  if (R.l < 8) s_w[ZP_OFF + R.l] = 0.0;
  // x behind the zero pad (operand of P x and A x); P x first, while the Toeplitz doubles are not live yet: its 72 matrix
  // elements want the registers
  if (wrx) {
    *reinterpret_cast<double2 *>(xk4) = make_double2(x[0], x[1]); xk4[2] = x[2];
  }
  wave_lds_sync();
  // the lanes' Toeplitz operands: their loads go out together with the P image's (one round trip to the Infinity Cache instead of two)
  double Gd[TB][2][3];
  load_G_image(Gd, Gg, R.l);                                // (Gg: the per-lane image)
  double px[3];
  {
    double p6[6];
    p_matvec(Pg, R, p6);
    wave_lds_sync();                                          // (the command-row buffer still holds w: nobody reads it any more)
    if (R.s == 0 && R.r < NB) {
#pragma unroll
      for (int j = 0; j < 6; ++j) wc[6 * R.r + j] = p6[j];
    }
    wave_lds_sync();
#pragma unroll
    for (int c = 0; c < 3; ++c) px[c] = wc[kxa + c];
  }
  TSTAMP(0)
  wave_lds_sync();
  {
    double eA[3], eB[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { eA[c] = C.WA[c] * yA[c]; eB[c] = C.WB[c] * yB[c]; }
    slots.put(eA, eB);
  }
  wave_lds_sync();
  double qu[3], cD[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { qu[c] = R.act ? qv[kx + c] : 0.0; cD[c] = C.cs * (R.act ? Dv[kx + c] : 1.0); }
  TSTAMP(1)
  double ax3[3], aty3[3], axB[3];
  stage3_partials(Gd, J);
  stage1_partials(Gd, J);
  wave_lds_sync();
  stage3_totals(Q.a3, ax3);
  stage1_totals(Q.a1, aty3);
  TSTAMP(2)
  double aty[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int k = kxa + c;
    aty[c] = C.cinv * (aty3[c] + wc[k] + (wr[k] - wr[k + 3]));
    const double xkk = xk4[c], xkm = xk4[c - 4];
    axB[c] = R.par ? xkk - xkm : xkk;
  }
  double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};                // r1, |Ax|, |z|, r2, |Px|, |A'y|, |q|, |E dyb|, support(dyb)
  if (R.act) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      v[0] = fmax(v[0], fmax(fabs(ax3[c] - zA[c]), fabs(axB[c] - zB[c])));
      v[1] = fmax(v[1], fmax(fabs(ax3[c]), fabs(axB[c])));
      v[2] = fmax(v[2], fmax(fabs(zA[c]), fabs(zB[c])));
      v[7] = fmax(v[7], fmax(C.WA[c] * fabs(dyA[c]), C.WB[c] * fabs(dyB[c])));
      v[8] += C.WA[c] * (C.hiA[c] * fmax(dyA[c], 0.0) + C.loA[c] * fmin(dyA[c], 0.0)) +
              C.WB[c] * (C.hiB[c] * fmax(dyB[c], 0.0) + C.loB[c] * fmin(dyB[c], 0.0));
      v[3] = fmax(v[3], fabs(px[c] + qu[c] + aty[c]));
      v[4] = fmax(v[4], fabs(px[c])); v[5] = fmax(v[5], fabs(aty[c])); v[6] = fmax(v[6], fabs(qu[c]));
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = wave_reduce_dpp<false>(v[i]);
  v[8] = wave_reduce_dpp<true>(v[8]);
  TSTAMP(3)
  const double rp = v[0], rd = v[3];
  const double np_ = fmax(v[1], v[2]), nd_ = fmax(fmax(v[4], v[5]), v[6]);
  if (rp < o.eps_abs + o.eps_rel * np_ && rd < o.eps_abs + o.eps_rel * nd_) { done = true; converged = true; }
  else {
    // OSQP primal-infeasibility certificate on dy (auxil.c:is_primal_infeasible)
    const double ndy = v[7], supp = v[8];
    if (ndy > o.eps_prim_inf && supp < -o.eps_prim_inf * ndy) {
      {
        double eA[3], eB[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { eA[c] = C.WA[c] * dyA[c]; eB[c] = C.WB[c] * dyB[c]; }
        slots.put(eA, eB);
      }
      wave_lds_sync();
      double t3[3];
      stage1_partials(Gd, J);
      wave_lds_sync();
      stage1_totals(Q.a1, t3);
      double wmax = 0.0;
      if (R.act) {
#pragma unroll
        for (int c = 0; c < 3; ++c) wmax = fmax(wmax, fabs(t3[c] + wc[kx + c] + (wr[kx + c] - wr[kx + c + 3])));
      }
      wmax = wave_reduce_dpp<false>(wmax);
      if (wmax < o.eps_prim_inf * ndy) { done = true; infeasible = true; }
    }
    if (!done) {
      if (it >= o.max_iter) done = true;
      else if (o.adaptive_rho && it % o.rho_every == 0) {
        // auxil.c:compute_rho_estimate on the SCALED residuals
        double sv[7] = {0, 0, 0, 0, 0, 0, 0};
        if (R.act) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const double ErA = sqrt(C.WA[c]), ErB = sqrt(C.WB[c]);
            sv[0] = fmax(sv[0], fmax(ErA * fabs(ax3[c] - zA[c]), ErB * fabs(axB[c] - zB[c])));
            sv[1] = fmax(sv[1], fmax(ErA * fabs(ax3[c]), ErB * fabs(axB[c])));
            sv[2] = fmax(sv[2], fmax(ErA * fabs(zA[c]), ErB * fabs(zB[c])));
            sv[3] = fmax(sv[3], cD[c] * fabs(px[c] + qu[c] + aty[c]));
            sv[4] = fmax(sv[4], cD[c] * fabs(px[c])); sv[5] = fmax(sv[5], cD[c] * fabs(aty[c]));
            sv[6] = fmax(sv[6], cD[c] * fabs(qu[c]));
          }
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) sv[i] = wave_reduce_dpp<false>(sv[i]);
        const double pr = sv[0] / (fmax(sv[2], sv[1]) + 1e-10), dr = sv[3] / (fmax(fmax(sv[6], sv[5]), sv[4]) + 1e-10);
        const double nw = fmin(fmax(rho * sqrt(pr / (dr + 1e-10)), OSQP_RHO_MIN), OSQP_RHO_MAX);
        if (nw > OSQP_ADAPTIVE_RHO_TOLERANCE * rho || nw < rho / OSQP_ADAPTIVE_RHO_TOLERANCE) { rho = nw; refactor = true; }
      }
    }
  }
  wave_lds_sync();
  if (!done && !refactor) {                                   // w of the current point for the next iteration
    double wA[3], wBv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { wA[c] = C.WA[c] * (ro.roA(c) * zA[c] - yA[c]); wBv[c] = C.WB[c] * (ro.roB(c) * zB[c] - yB[c]); }
    slots.put(wA, wBv);
    wave_lds_sync();
  }
  TSTAMP(4)
  st->rho = rho; st->rp = rp; st->rd = rd;
  st->done = done; st->converged = converged; st->infeasible = infeasible;
  return refactor ? 1 : 0;
}

// factorise / iterate / test until done (st->done) for one ANYEQ flavour; returns whether every factorisation succeeded
template <bool ANYEQ>
__device__ __forceinline__ bool solve_loop(SolveState *st, const LaneConst *lcp, const double *Pg, const double *pb, const double *gw, const double *Gg,
                                           const double *qv, const double *Dv, int N, double cs, IterSettings o, const Role &R) {
  double *const xt = s_w + XT_OFF;
  const int kx = 3 * R.istep;
  bool ok = true;
  while (!st->done) {
    if (R.act && R.par == 0) {                               // sigma D^-2 for the KKT diagonal (natural order, behind the pad)
#pragma unroll
      for (int c = 0; c < 3; ++c) xt[XT_PAD + kx + c] = lcp->sg[c];
    }
    wave_lds_sync();
#ifdef F16_EXP_STAMPW
    unsigned long long tq0 = __builtin_amdgcn_s_memtime();
#define WSTAMPK(i, j) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long tq1 = __builtin_amdgcn_s_memtime(); if (STAMP_WG() && R.l == 0) { g_wstamp[i] += tq1 - tq0; g_wstamp[j] += 1; } tq0 = tq1; }
#else
#define WSTAMPK(i, j)
#endif
    int fok = 0;
    factorise(Pg, gw, xt + XT_PAD, const_cast<double *>(Gg) + GIMG_DOUBLES, N, cs, st->rho, &fok);
    ok = fok != 0 && ok;
    WSTAMPK(10, 12)
    if (!ok) break;
    start_point<ANYEQ>(st, lcp, N);
    bool refactor = false;
    while (!st->done && !refactor) {
      WSTAMPK(14, 15)
      run_iterations<ANYEQ>(st, lcp, Gg, Gg + GIMG_DOUBLES, N, o);      // (to the next block end whose test needs the dual side too)
      WSTAMPK(13, 15)
      refactor = terminate_test<ANYEQ>(st, lcp, pb, Gg, qv, Dv, N, o) != 0;
      WSTAMPK(9, 11)
    }
  }
  return ok;
}

// ----------------------------------------------------------------------------------------------------------------
// Ruiz equilibration (OSQP scaling.c:scale_data; f16_mpc_solve.hip: ruiz_equilibrate, rule for rule) in this one wavefront.
// Norms of the scaled matrices are formed from the ORIGINAL entries and the running D, E, c:
//   rows / columns of Pb = c D P D   the block image of P with (max, x) in place of (+, x): direct and transposed use of each block
//   columns of Ab = E A D            the stage-1 pattern on |G_d| (all NINE state rows: rows without bounds are rows of A as OSQP
//   rows of Ab                       sees it) and E;  the stage-3 pattern on |G_d| and D
// Ownership as in the iterations (lane (o, t): step istep, variables 3 istep + c, kept rows 3h + c, command or rate rows), plus
// the rows without bounds of that step: phi, theta on the even lane of a pair, lf1 on the odd one.
// LDS: the nine-row G image at GL_OFF (prologue); D behind 7 zero steps in the x~ buffer; E of the state rows, 10 doubles per
// step, in the state-row buffer; E of command / rate rows in theirs; row norms of P behind the G image.
constexpr int NPM_OFF = 840, E9_REC = 10;
struct RuizOut { double De[3], EA[3], EB[3], cs; };

template <bool IS_MAX>
__device__ __forceinline__ double octet_allreduce(double v) {
  auto op = [](double a, double b) { return IS_MAX ? fmax(a, b) : a + b; };
  v = op(v, dpp<DPP_XOR1>(v));
  v = op(v, dpp<DPP_XOR2>(v));
  v = op(v, dpp<DPP_HMIRROR>(v));
  return v;
}

// the lane's two blocks of |P| (block image in the workspace): loaded ONCE per equilibration -- every pass re-read them before (11 x 36
// loads of 16 bytes, each batch a round trip to the Infinity Cache)
__device__ __forceinline__ void p_abs_blocks(const double *pb, const Role &R, double (&A)[6][6], double (&Bk)[6][6]) {
  typedef double dbl2_t __attribute__((ext_vector_type(2)));
  typedef const dbl2_t __attribute__((address_space(1))) *g2ptr_t;
  const g2ptr_t pp = (g2ptr_t)pb + R.l;
#pragma unroll
  for (int m = 0; m < 18; ++m) {
    const dbl2_t a = pp[m * 64], b = pp[(18 + m) * 64];
    A[(2 * m) / 6][(2 * m) % 6] = fabs(a.x); A[(2 * m + 1) / 6][(2 * m + 1) % 6] = fabs(a.y);
    Bk[(2 * m) / 6][(2 * m) % 6] = fabs(b.x); Bk[(2 * m + 1) / 6][(2 * m + 1) % 6] = fabs(b.y);
  }
}
// npm[k] = D_k max_j |P_kj| D_j for every variable (natural order, LDS) from the lane's blocks of |P| and D in LDS
__device__ __forceinline__ void p_row_norms(const double (&A)[6][6], const double (&Bk)[6][6], const double *Dn, double *npm, const Role &R) {
  const int rr = R.r < NB ? R.r : NB - 1;
  double dr[6], da[6], db[6];
#pragma unroll
  for (int m = 0; m < 6; ++m) { dr[m] = Dn[6 * rr + m]; da[m] = Dn[6 * R.cA + m]; db[m] = Dn[6 * R.cB + m]; }
  double md[6], mtA[6], mtB[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double m0 = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) m0 = fmax(m0, fmax(A[i][j] * da[j], Bk[i][j] * db[j]));
    md[i] = m0;
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double m0 = 0.0, m1 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { m0 = fmax(m0, A[i][j] * dr[i]); m1 = fmax(m1, Bk[i][j] * dr[i]); }
    mtA[j] = m0; mtB[j] = m1;
  }
  const int srcA = 4 * ((rr + 2 * R.s + 1) % NB) + R.s, srcB = 4 * ((rr + 2 * R.s + 2) % NB) + R.s;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const double pa = bperm(mtA[j], srcA), pb_ = bperm(mtB[j], srcB);
    double m = fmax(md[j], fmax(pa, R.s < 3 ? pb_ : 0.0));
    m = fmax(m, dpp<DPP_XOR1>(m));
    m = fmax(m, dpp<DPP_XOR2>(m));
    if (R.s == 0 && R.r < NB) npm[6 * R.r + j] = dr[j] * m;
  }
}

__device__ __noinline__ void ruiz_wave(const double *pb, const double *qv, int N, int passes, RuizOut *out) {
  const Role R = role(N);
  const int n = 3 * N, kx = 3 * R.istep, kxa = R.act ? kx : 0;
  const bool wrx = R.act && R.par == 0;
  const double *Gl = s_w + GL_OFF;
  double *const Dz = s_w + XT_OFF, *const Dn = Dz + XT_PAD;        // D: zero-padded / natural order views of one buffer
  double *const E9 = s_w + WS_OFF, *const Ec = s_w + WC_OFF, *const Er = s_w + WR_OFF, *const npm = s_w + NPM_OFF;
  double De[3] = {1.0, 1.0, 1.0}, EA[3] = {1.0, 1.0, 1.0}, EB[3] = {1.0, 1.0, 1.0}, EU[2] = {1.0, 1.0}, cs = 1.0;
  double qa[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) qa[c] = fabs(as_global(qv)[kxa + c]);
  // MPC state rows of the lane's slots: kept rows 3h + c; without bounds: phi, theta (even lane), lf1 (odd lane)
  int rowA[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { const int kk = 3 * R.h + c; rowA[c] = kk < 5 ? kk + 2 : 8; }
  const int rowU0 = R.par ? 7 : 0, rowU1 = R.par ? 7 : 1;
  // D = E = 1 inside the problem, 0 outside (the buffers are zero)
  if (R.act) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { E9[E9_REC * R.istep + rowA[c]] = 1.0; (R.par ? Er : Ec)[kx + c] = 1.0; }
    E9[E9_REC * R.istep + rowU0] = 1.0; E9[E9_REC * R.istep + rowU1] = 1.0;
    if (R.par == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) Dn[kx + c] = 1.0;
    }
  }
  wave_lds_sync();
  double PA[6][6], PB[6][6];
  p_abs_blocks(pb, R, PA, PB);
  p_row_norms(PA, PB, Dn, npm, R);
  wave_lds_sync();
  const int base1 = 4 * (R.o + R.t), wb = base1 < N ? base1 : N;          // first operand step of the stage-1 pattern
  const int s0 = 4 * (R.o - R.t) - 3, sb = s0 > -7 ? s0 : -7;              // ... of the stage-3 pattern
  for (int pass = 0; pass < passes; ++pass) {
    // column norms of the state block of Ab before the D of the column (col[e][c], step 4o + e) and row norms before the E of
    // the row (row[e][r], all nine rows), this lane's four lags; then the maxima over the octet
    double col[4][3], row[4][9];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int c = 0; c < 3; ++c) col[e][c] = 0.0;
#pragma unroll
      for (int r9 = 0; r9 < 9; ++r9) row[e][r9] = 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int d = 4 * R.t + u;
      const double *gp = Gl + 27 * (d < N ? d : N);                         // (block N is zero)
      double ag[9][3];
#pragma unroll
      for (int r9 = 0; r9 < 9; ++r9)
#pragma unroll
        for (int c = 0; c < 3; ++c) ag[r9][c] = fabs(gp[3 * r9 + c]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double *ep = E9 + E9_REC * (wb + e + u);                      // E of step 4(o + t) + e + u
        const double *dp = Dz + 3 * (sb + 7 + e - u + 3);                   // D of step 4(o - t) + e - u
        double ev[9], dv[3];
#pragma unroll
        for (int r9 = 0; r9 < 9; ++r9) ev[r9] = ep[r9];
#pragma unroll
        for (int c = 0; c < 3; ++c) dv[c] = dp[c];
#pragma unroll
        for (int r9 = 0; r9 < 9; ++r9)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            col[e][c] = fmax(col[e][c], ag[r9][c] * ev[r9]);
            row[e][r9] = fmax(row[e][r9], ag[r9][c] * dv[c]);
          }
      }
    }
    double colS[3] = {0.0, 0.0, 0.0}, rowS[9];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { const double v = octet_allreduce<true>(col[e][c]); colS[c] = R.estar == e ? v : colS[c]; }
#pragma unroll
      for (int r9 = 0; r9 < 9; ++r9) { const double v = octet_allreduce<true>(row[e][r9]); rowS[r9] = (R.estar == e || e == 0) ? v : rowS[r9]; }
    }
    auto pick9 = [&](int r) { double v = rowS[0];
#pragma unroll
      for (int r9 = 1; r9 < 9; ++r9) v = r == r9 ? rowS[r9] : v;
      return v; };
    double Dt[3], EtA[3], EtB[3], EtU[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int k = kxa + c;
      const double colA = De[c] * fmax(fmax(colS[c], Ec[k]), fmax(Er[k], Er[k + 3]));
      Dt[c] = 1.0 / sqrt(osqp_limit_scaling(fmax(cs * npm[k], colA)));
      EtA[c] = 1.0 / sqrt(osqp_limit_scaling(EA[c] * pick9(rowA[c])));
      const double dk = Dz[XT_PAD + k], dkm = Dz[XT_PAD + k - 3];
      EtB[c] = 1.0 / sqrt(osqp_limit_scaling(EB[c] * (R.par ? fmax(dk, dkm) : dk)));
    }
    EtU[0] = 1.0 / sqrt(osqp_limit_scaling(EU[0] * pick9(rowU0)));
    EtU[1] = 1.0 / sqrt(osqp_limit_scaling(EU[1] * pick9(rowU1)));
    wave_lds_sync();                                        // every lane has read the old D and E
#pragma unroll
    for (int c = 0; c < 3; ++c) { De[c] *= Dt[c]; EA[c] *= EtA[c]; EB[c] *= EtB[c]; }
    EU[0] *= EtU[0]; EU[1] *= EtU[1];
    if (R.act) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { E9[E9_REC * R.istep + rowA[c]] = EA[c]; (R.par ? Er : Ec)[kx + c] = EB[c]; }
      E9[E9_REC * R.istep + rowU0] = EU[0];
      if (R.par == 0) {
        E9[E9_REC * R.istep + rowU1] = EU[1];
#pragma unroll
        for (int c = 0; c < 3; ++c) Dn[kx + c] = De[c];
      }
    }
    wave_lds_sync();
    p_row_norms(PA, PB, Dn, npm, R);                        // with the new D: cost scaling now, column norms of the next pass
    wave_lds_sync();
    double sm = 0.0, qn = 0.0;
    if (wrx) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { sm += cs * npm[kx + c]; qn = fmax(qn, cs * De[c] * qa[c]); }
    }
    sm = wave_reduce_dpp<true>(sm);
    qn = wave_reduce_dpp<false>(qn);
    cs *= 1.0 / fmax(osqp_limit_scaling(sm / n), osqp_limit_scaling(qn));
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) { out->De[c] = De[c]; out->EA[c] = EA[c]; out->EB[c] = EB[c]; }
  out->cs = cs;
  // leave the buffers as the prologue expects them: zero
  wave_lds_sync();
  for (int i = R.l; i < WS_SIZE + WC_SIZE + WR_SIZE + XT_SIZE; i += 64) s_w[WS_OFF + i] = 0.0;
  for (int i = R.l; i < 96; i += 64) s_w[NPM_OFF + i] = 0.0;
  wave_lds_sync();
}

// P (packed lower triangle, workspace of the build kernel) -> its symmetric block image [36][64] double2 (lane-major chunks, as the
// LDS image of the KKT inverse): 72 gathered elements per lane, once per solve; zero outside the n x n matrix.
__device__ __noinline__ void p_block_image(const double *Pg, double *pb, int n) {
  const int l = threadIdx.x;
  const gptr_t Pgg = as_global(Pg);
  unsigned ij[72];
#pragma unroll
  for (int e = 0; e < 72; ++e) ij[e] = F16_WAVE_PGATH[e * 64 + l];
  __builtin_amdgcn_sched_barrier(0);
  double pv[72];
#pragma unroll
  for (int e = 0; e < 72; ++e) {
    const int i = (int)(ij[e] >> 8), j = (int)(ij[e] & 255);
    pv[e] = Pgg[(i < n && j < n) ? i * (i + 1) / 2 + j : 0];          // (i >= j)
  }
  __builtin_amdgcn_sched_barrier(0);
  double2 *dst = reinterpret_cast<double2 *>(pb) + l;
#pragma unroll
  for (int m = 0; m < 36; ++m) {
    const int i0 = (int)(ij[2 * m] >> 8), j0 = (int)(ij[2 * m] & 255), i1 = (int)(ij[2 * m + 1] >> 8), j1 = (int)(ij[2 * m + 1] & 255);
    dst[m * 64] = make_double2((i0 < n && j0 < n) ? pv[2 * m] : 0.0, (i1 < n && j1 < n) ? pv[2 * m + 1] : 0.0);
  }
}

// ----------------------------------------------------------------------------------------------------------------
// The solve of one aircraft per wavefront-workgroup (grid = B).  D, E, c of the equilibration come from the workspace
// (k_mpc_fast, scale-only mode), P / q / G_k / pred from the build kernel (k_mpc<true>).
// Dispatch (round 4): the hardware hands workgroups to the XCDs, and inside an XCD to its four shader engines, ROUND-ROBIN -- 32
// static partitions of 32 SIMDs that each work through their own 128 aircraft (tools/gpu_wave_timeline.py: every shader engine
// solves exactly B / 32 aircraft, 996 of 1024 SIMDs exactly four), so the longest-first order is greedy only inside a partition and
// the launch ends when the unluckiest partition does: 7.96 ms where one greedy queue over all 1024 SIMDs needs 7.53 (simulated from
// the measured durations of the same launch).  With `wave_queue` the grid is one workgroup per SIMD and the queue is ours: a
// workgroup takes the next aircraft of the dispatch order from an atomic counter whenever it is free, until the counter passes B
// (an exit every workgroup reaches, whatever the number of resident ones).
// ... of aircraft b by this wavefront: prologue, equilibration, factorise / iterate / test until done.  Leaves the solution in
// `st` (lane 0 holds the first move st.x[0..2]) and, with a warm-start buffer, the solution for the next call; returns whether every
// factorisation succeeded.  Called by k_mpc_wave (one calc_MPC_action per aircraft) and by k_rollout_mpc (one per aircraft and step).
__device__ __forceinline__ bool solve_aircraft(const MpcArgs &a, long b, long job, const Role &R, SolveState &st, int warm_load
#ifdef F16_EXP_STAMPW
                                               , unsigned long long &tK0
#endif
                                               ) {
  const int N = a.N, n = 3 * N;
  const int l = R.l;
  (void)job;
  double *const exw = a.ext + (size_t)b * mpc_ext_doubles(N);
  const double *Pg = a.Ppk + (size_t)b * (n * (n + 1) / 2);
  const double *Gg = exw + n, *pred = exw + n + 27 * N;
  double *const gw = a.gramws + (size_t)b * MPC_TILE_DOUBLES;
  const double *scal = gw + GW_SCAL;
  for (int i = l; i < LDS_DOUBLES; i += 64) s_w[i] = 0.0;
  wave_lds_sync();
  for (int i = l; i < 27 * N; i += 64) s_w[GL_OFF + i] = Gg[i];          // (block N stays zero: what rows beyond 6N gather)
  double *const pb = a.pblk + (size_t)b * WAVE_PBLK_DOUBLES;
  p_block_image(Pg, pb, n);                                  // P as the symmetric block image (equilibration, termination test)
  double *const gimg = pb + WAVE_PBLK_DOUBLES / 2;            // the lanes' Toeplitz operands, lane-major
  {
    double Gd[TB][2][3];
    load_G(Gd, Gg, tjob(), N);
    store_G_image(Gd, gimg, l);
  }
  wave_lds_sync();
  // ---- equilibration: in this wavefront (default), or D | E | c left in the workspace by k_mpc_fast (mode 3, F16_WAVE_RUIZ=0)
  const int kx = 3 * R.istep;
  double *const scalw = gw + GW_SCAL;
  RuizOut rz;
  if (a.wave_ruiz) {
    ruiz_wave(pb, exw, N, a.s.scaling, &rz);
    if (R.act && R.par == 0) {                               // D of the owned variables: the rho estimate reads it back
#pragma unroll
      for (int c = 0; c < 3; ++c) scalw[kx + c] = rz.De[c];
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int k = R.act ? kx + c : 0;
      rz.De[c] = R.act ? scal[k] : 1.0;
      rz.EA[c] = R.act ? scal[96 + 6 * R.istep + 3 * R.h + c] : 0.0;
      rz.EB[c] = R.act ? scal[(R.par ? 384 : 288) + k] : 0.0;
    }
    rz.cs = scal[480];
  }
  // ---- lane constants: owned variables, state rows (step istep, kept rows 3h..3h+2), command / rate rows of the same step
  LaneConst C;
  const double cs = rz.cs;
  C.cs = cs; C.cinv = 1.0 / cs;
  C.eqA = 0; C.eqB = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int k = R.act ? kx + c : 0;
    const double De = R.act ? rz.De[c] : 1.0, qe = R.act ? exw[k] : 0.0;
    C.sg[c] = a.s.sigma / (De * De); C.cq[c] = cs * qe;
    {   // state row: kept index kk = 3h + c (utils.py:129-133; rows with two infinite bounds are not kept)
      const int kk = 3 * R.h + c;
      const double Eo = R.act ? rz.EA[c] : 0.0;
      const double pm = R.act ? pred[R.istep * 9 + SROW[kk]] : 0.0;
      const double lo = R.act ? a.pb.slb[kk] - pm : 0.0, hi = R.act ? a.pb.sub[kk] - pm : 0.0;
      const bool eq = R.act && Eo * (hi - lo) < OSQP_RHO_TOL;
      C.loA[c] = lo; C.hiA[c] = hi; C.WA[c] = Eo * Eo; C.eqA |= eq ? (1 << c) : 0;
      if (R.act) s_w[WG_OFF + 6 * R.istep + kk] = Eo * Eo * (eq ? OSQP_RHO_EQ_OVER_RHO_INEQ : 1.0);
    }
    {   // command row (par = 0, utils.py:139-140) or rate row (par = 1, utils.py:148-152) of variable k
      const double Eo = R.act ? rz.EB[c] : 0.0;
      double lo, hi;
      if (R.par == 0) { lo = a.pb.ulb[c]; hi = a.pb.uub[c]; }
      else if (R.istep == 0) {
        const double act = a.x ? a.x[(13 + c) * a.ld + b] : 0.0;
        lo = act + a.pb.rlb[c] * a.dt; hi = act + a.pb.rub[c] * a.dt;
      } else { lo = a.pb.rlb[c]; hi = a.pb.rub[c]; }                 // reference quirk: not scaled by dt (utils.py:151-152)
      if (!R.act) { lo = 0.0; hi = 0.0; }
      const bool eq = R.act && Eo * (hi - lo) < OSQP_RHO_TOL;
      C.loB[c] = lo; C.hiB[c] = hi; C.WB[c] = Eo * Eo; C.eqB |= eq ? (1 << c) : 0;
      if (R.act) s_w[(R.par ? WRV_OFF : WCV_OFF) + k] = Eo * Eo * (eq ? OSQP_RHO_EQ_OVER_RHO_INEQ : 1.0);
    }
  }
  wave_lds_sync();
  gram_tiles(gw, N);                                         // A'WA -> workspace (every lane reads back what it wrote itself)
#pragma unroll
  for (int c = 0; c < 3; ++c) { st.x[c] = 0.0; st.zA[c] = 0.0; st.yA[c] = 0.0; st.zB[c] = 0.0; st.yB[c] = 0.0; st.dyA[c] = 0.0; st.dyB[c] = 0.0; }
  st.rp = INFINITY; st.rd = INFINITY; st.it = 0; st.to_check = a.s.check_every > 0 ? a.s.check_every : 1;
  st.done = 0; st.converged = 0; st.infeasible = 0; st.rho = a.s.rho;
  double *const wm = a.warm ? a.warm + (size_t)b * MPC_WARM_DOUBLES + l : nullptr;      // [15][64]: x, zA, yA, zB, yB (unscaled)
  if (wm && warm_load) {
    bool fin = true;
    double t15[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) { t15[k] = wm[k * 64]; fin = fin && isfinite(t15[k]); }
    if (fin) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        st.x[c] = t15[c]; st.zA[c] = t15[3 + c]; st.zB[c] = t15[9 + c];
        st.yA[c] = C.WA[c] > 0.0 ? cs * t15[6 + c] / C.WA[c] : 0.0;
        st.yB[c] = C.WB[c] > 0.0 ? cs * t15[12 + c] / C.WB[c] : 0.0;
      }
    }
  }
#ifdef F16_EXP_STAMPW
  if (job == 0 && l < 16) g_wstamp[l] = 0;
  if (job == 0 && l < 8) g_tstamp[l] = 0;
  tK0 = __builtin_amdgcn_s_memtime();
#endif
  IterSettings o;
  o.alpha = a.s.alpha; o.eps_abs = a.s.eps_abs; o.eps_rel = a.s.eps_rel; o.eps_prim_inf = a.s.eps_prim_inf;
  o.max_iter = a.s.max_iter; o.check_every = a.s.check_every; o.rho_every = a.s.rho_every; o.adaptive_rho = a.s.adaptive_rho;
  const bool anyeq = __ballot((C.eqA | C.eqB) != 0) != 0;      // (wave-uniform)
  const bool ok = anyeq ? solve_loop<true>(&st, &C, Pg, pb, gw, gimg, exw, scal, N, cs, o, R)
                        : solve_loop<false>(&st, &C, Pg, pb, gw, gimg, exw, scal, N, cs, o, R);
  if (wm) {                                                  // keep the solution for the next warm start
    const bool good = st.converged != 0 && st.infeasible == 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      wm[c * 64] = good ? st.x[c] : NAN; wm[(3 + c) * 64] = good ? st.zA[c] : NAN; wm[(9 + c) * 64] = good ? st.zB[c] : NAN;
      wm[(6 + c) * 64] = good ? C.WA[c] * st.yA[c] / cs : NAN; wm[(12 + c) * 64] = good ? C.WB[c] * st.yB[c] / cs : NAN;
    }
  }
  return ok;
}

__global__ __launch_bounds__(64, 1) void k_mpc_wave(MpcArgs a) {
  const int N = a.N;
  const Role R = role(N);
  const int l = R.l;
  for (long job = blockIdx.x;; ) {
  if (a.wave_queue) {
    unsigned q = 0;
    if (l == 0) q = atomicAdd(a.wave_queue, 1u);
    job = (long)(unsigned)__builtin_amdgcn_readfirstlane((int)q);
    if (job >= a.B) break;
    wave_lds_sync();                                         // (the previous aircraft's LDS traffic is over on every lane)
  }
#ifdef F16_EXP_STAMPW
  const unsigned long long wc0_ = wall_clock64();
  if (job == 0 && l == 0) g_stamp_wg = (int)blockIdx.x;
  unsigned long long tK0 = 0;
#endif
  // No order from a previous call (first call on this stream / batch size): NOT the caller's order either -- workgroup ids go to
  // the XCDs round-robin, and how hard an aircraft is tends to follow its index (the config-4 workload: every aircraft = 0, 3, 6
  // mod 8 needs twice the iterations), so the identity gives three XCDs twice the work of the others.  A fixed stride coprime to
  // B spreads any such pattern; the results do not depend on the map.
  const long b = a.order ? (long)__builtin_amdgcn_readfirstlane(a.order[job])
                         : (a.wave_stride ? (long)(((unsigned long long)job * a.wave_stride) % (unsigned long long)a.B) : job);
  if (mpc_job_nonfinite(a.ext, a.N, b)) {                             // (wave-uniform) no QP to solve: NaN command, zero iterations
    mpc_write_nonfinite(a.ucmd, a.useq, a.info, a.iters_out, a.status, a.ld, a.N, a.s.rho, b, l, 64);
    if (!a.wave_queue) break;
    continue;
  }
  SolveState st;
#ifdef F16_EXP_STAMPW
  const bool ok = solve_aircraft(a, b, job, R, st, a.warm_load, tK0);
#else
  const bool ok = solve_aircraft(a, b, job, R, st, a.warm_load);
#endif
  const int kx = 3 * R.istep;
  const bool converged = st.converged != 0, infeasible = st.infeasible != 0;
  // res.x[0:3] (env.py:424); OSQP hands back NaN for a problem it certifies infeasible
  if (R.act && R.par == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (R.istep == 0) a.ucmd[c * a.ld + b] = infeasible ? NAN : st.x[c];
      if (a.useq) a.useq[(kx + c) * a.ld + b] = infeasible ? NAN : st.x[c];
    }
  }
#ifdef F16_EXP_STAMPW
  if (job == 0 && l == 0 && a.useq) {      // diagnostic build: the stamps replace rows 60.. of this aircraft's u_seq column
    for (int i = 0; i < 8; ++i) a.useq[(60 + i) * a.ld + b] = (double)g_wstamp[i];
    a.useq[68 * a.ld + b] = (double)st.it;
    for (int i = 9; i < 16; ++i) a.useq[(60 + 12 + i - 9) * a.ld + b] = (double)g_wstamp[i];
    for (int i = 0; i < 8; ++i) a.useq[(80 + i) * a.ld + b] = (double)g_tstamp[i];
    a.useq[71 * a.ld + b] = (double)(__builtin_amdgcn_s_memtime() - tK0);
  }
#endif
  if (l == 0) {
    if (a.iters_out) a.iters_out[b] = st.it;
    if (a.info) {
      a.info[0 * a.ld + b] = (double)st.it;
      a.info[1 * a.ld + b] = st.rp;
      a.info[2 * a.ld + b] = st.rd;
      a.info[3 * a.ld + b] = st.rho;
#ifdef F16_EXP_STAMPW
      a.info[1 * a.ld + b] = (double)wc0_;                 // diagnostic build: start / end of this solve on the 100 MHz clock
      a.info[2 * a.ld + b] = (double)wall_clock64();
      a.info[3 * a.ld + b] = (double)((__builtin_amdgcn_s_getreg(63508) & 15) * 65536 + (__builtin_amdgcn_s_getreg(63492) & 0xFFFF));    // XCC_ID | HW_ID: which SIMD
#endif
    }
    if (a.status && infeasible) a.status[b] |= F16_ST_QP_INFEASIBLE;
    else if (a.status && (!converged || !ok)) a.status[b] |= F16_ST_QP_MAXITER;
  }
#ifdef F16_EXP_STAMPW
  if (job == 0 && l == 0) g_stamp_wg = -1;
#endif
  if (!a.wave_queue) break;
  }
}

// ----------------------------------------------------------------------------------------------------------------
// The closed-loop MPC rollout as ONE launch (BASELINE config 5; the reference's loop test_env.py:480-495:
//     cmd = _calc_MPC_action(p, q, r, hzn);  u.values[1:] = cmd;  step(u.values)
// per step, the model frozen as env.py:49-60 freezes it).  Aircraft never interact, so nothing in the loop needs a step to END for
// every aircraft before the next one starts -- the host loop (dist.closed_loop_mpc_rollout: six launches per step) joins the whole
// batch after every solve and a step lasts as long as its slowest aircraft (mean 458 iterations, longest 4,503 in the recorded
// config-5 run).  Here the grid is one wavefront-workgroup per SIMD, as in k_mpc_wave's queue mode, and the work items are the
// (step, aircraft) pairs in step-major order, drawn from ONE ticket counter: ticket k = step k / B of aircraft perm(k % B).  A
// wavefront does everything of its pair: the state-dependent vectors of the QP (f16_mpc_state.hpp, the build kernel's own code),
// the solve (solve_aircraft above), the command into u.values, one Euler step of the plant (env.py:105-130; the table image is read
// from global memory -- it stays in L2 -- because the solver's 40 KB of LDS per wavefront leave no room for it), the trajectory
// sample.  Step t of an aircraft needs step t - 1 of the SAME aircraft only: the wavefront that draws (t, b) waits for
// progress[b] == t, a per-aircraft counter the finisher of (t - 1, b) bumps behind an agent-scope release; the taker runs an
// agent-scope acquire before it reads x / u / status (MI355X_MICROARCH.md, "inter-workgroup visibility": plain stores -> vmcnt(0)
// -> barrier -> release fence -> vmcnt(0) -> relaxed flag store | relaxed poll -> acquire fence -> vmcnt(0) -> barrier -> plain
// loads).  No deadlock: tickets are handed out in order, so the pair a wavefront waits for was drawn earlier, i.e. by a wavefront
// that is resident and itself waits only for an even earlier ticket; every wavefront leaves when the counter passes T x B.
// With B >> 1024 the predecessor ticket is B tickets old -- finished long ago unless it is a many-times-the-mean straggler.
struct RollMpcArgs {
  MpcArgs m;                     // the plan's arguments: workspace, settings, weights (m.x = x: the solver reads the actuator states)
  double *x;                     // [18][ld] x.values, in place
  double *u;                     // [4][ld] u.values: thrust command held, u[1:4] <- the command of every step
  const double *dem;             // [3][ld] (p, q, r) demands
  double *traj;                  // [T / every][18][ld] or null
  double *cmd_traj;              // [T][3][ld] or null: what calc_MPC_action returned at each step (NaN: infeasible / not finite / no solve)
  int32_t *iters_traj;           // [T][ld] or null
  int32_t *status;               // [ld], sticky
  unsigned *queue;               // ticket counter, zero at the launch
  int32_t *progress;             // [B] steps completed per aircraft, zero at the launch
  const double *tab, *lofi;      // table images (global)
  int T, every;
  double xcg;
  int fi;
  unsigned flags, stride, total; // total = T x B tickets (< 2^32: checked by the launcher)
};

__device__ __forceinline__ double bcast0_f64(double v) {       // lane 0's value on every lane
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ bool uniform_flag(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }   // (a scalar branch for what every lane agrees on)

constexpr int RU_A = 0, RU_Q = 81, RU_QB = 162, RU_G = 256, RU_X9 = RU_G + 27 * WN + 6, RU_XREF = RU_X9 + 10, RU_PRED = RU_XREF + 10,
              RU_WBUF = RU_PRED + 9 * WN + 2, RU_QV = RU_WBUF + 9 * WN + 2;
static_assert(RU_QV + 3 * WN <= LDS_DOUBLES, "state-vector scratch of the rollout kernel");

// The two ends of a (step, aircraft) pair as out-of-line functions with register allocations of their own (as the pieces of the
// solve are): the kernel's top level then holds little more than k_mpc_wave's does.  (The first version had everything inline in the
// ticket loop -- ~9,000 instructions around eleven calls, three whole-wave spill registers of scalar state parked in accumulator
// registers -- and did not survive the compiler: ticket 0's body ran for ever, or the wave stopped behind its first release, depending
// on unrelated edits; phase markers read from another stream while the kernel ran showed where: tools/gpu_fused_marks.py.)
struct PairIO {
  double *x, *u;
  const double *dem;
  double *traj, *cmd_traj;
  int32_t *iters_traj, *status;
  const double *tab, *lofi;
  double *exw;                   // this aircraft's extras block of the plan's workspace: q | G | pred | A Q Qbar
  long ld, b;
  int t, N, every, fi;
  unsigned flags;
  double xcg, dt;
  double cmd[3];                 // what calc_MPC_action returned (NaN: infeasible / not finite / no solve)
  int iters, stw, stall;
};
enum { PAIR_SOLVE = 0, PAIR_FROZEN = 1, PAIR_NONFINITE = 2 };

// Start of a pair: the aircraft at the start of the step; env.py:117-124 first (the reference exit()s there: frozen, flagged, not solved
// for any more -- the host loop goes on solving for it: same states, wasted iterations); then the state-dependent vectors of the QP
// into the aircraft's workspace block (what the build kernel does per call: f16_mpc_state.hpp).
__device__ __noinline__ int pair_prepare(PairIO *io) {
  const int l = threadIdx.x;
  const long ld = io->ld, b = io->b;
  const int N = __builtin_amdgcn_readfirstlane(io->N), n = 3 * N;
  const double *xg = io->x, *dg = io->dem;
  double x[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) x[i] = xg[i * ld + b];
  int stw = io->status ? io->status[b] : 0;
  if (!(stw & ST_ENVELOPE) && !(io->flags & FLAG_NO_ENVELOPE) && outside_envelope(x)) stw |= ST_ENVELOPE | envelope_state_bits(x);
  int code = PAIR_SOLVE;
  if (uniform_flag((stw & ST_ENVELOPE) != 0)) code = PAIR_FROZEN;
  else {
    bool fin = true;
    const int MX[9] = {3, 4, 7, 8, 9, 10, 11, 17, 16};
#pragma unroll
    for (int i = 0; i < 9; ++i) fin = fin && isfinite(x[MX[i]]);
    fin = fin && isfinite(x[13]) && isfinite(x[14]) && isfinite(x[15]);
    double dm[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { dm[c] = dg[c * ld + b]; fin = fin && isfinite(dm[c]); }
    if (!uniform_flag(fin)) { code = PAIR_NONFINITE; stw |= ST_NONFINITE; }      // no QP to solve (f16_mpc.hpp: mpc_job_nonfinite)
    else {
      double *const exw = io->exw;
      const double *exm = exw + mpc_ext_model(N);
      wave_lds_sync();
      for (int e = l; e < 81; e += 64) { s_w[RU_A + e] = exm[e]; s_w[RU_Q + e] = exm[81 + e]; s_w[RU_QB + e] = exm[162 + e]; }
      for (int e = l; e < 27 * N; e += 64) s_w[RU_G + e] = exw[n + e];
      if (l < 9) {
        double v = x[0];
#pragma unroll
        for (int i = 0; i < 18; ++i) v = MX[l] == i ? x[i] : v;
        s_w[RU_X9 + l] = v;
        s_w[RU_XREF + l] = (l >= 5 && l < 8) ? (l == 5 ? dm[0] : (l == 6 ? dm[1] : dm[2])) : v;          // env.py:380-383
      }
      __syncthreads();
      mpc_state_vectors(s_w + RU_A, s_w + RU_Q, s_w + RU_QB, s_w + RU_G, s_w + RU_X9, s_w + RU_XREF, s_w + RU_PRED, s_w + RU_WBUF,
                        s_w + RU_QV, N);
      for (int e = l; e < n; e += 64) exw[e] = s_w[RU_QV + e];
      for (int e = l; e < 9 * N; e += 64) exw[n + 27 * N + e] = s_w[RU_PRED + e];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  io->stw = stw;                   // (private memory: every lane keeps its own copy of *io, all with the same values)
  return code;
}

// End of a pair: u.values[1:] = cmd (test_env.py:490-493; F16_FLAG_HOLD_COMMAND: a step without a command keeps the previous one),
// step(u.values) (env.py:126: euler_step_exact, the out-of-line step of the F16_FLAG_ONE_LANE rollout kernel; the table image from
// global memory), then the stores of the pair: state, command, flags, samples.
__device__ __noinline__ void pair_finish(const PairIO *io, int code) {
  const int l = threadIdx.x;
  const long ld = io->ld, b = io->b;
  const int t = __builtin_amdgcn_readfirstlane(io->t);
  const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane((int)io->flags);
  double *xg = io->x, *ug = io->u;
  double x[18], u[4];
#pragma unroll
  for (int i = 0; i < 18; ++i) x[i] = xg[i * ld + b];
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = ug[i * ld + b];
  int stw = io->stw | io->stall;
  double cmd[3] = {io->cmd[0], io->cmd[1], io->cmd[2]};
  const bool live = __builtin_amdgcn_readfirstlane(code) != PAIR_FROZEN;
  if (live) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (!((flags & F16_FLAG_HOLD_COMMAND) && cmd[c] != cmd[c])) u[1 + c] = cmd[c];
    euler_step_exact(io->tab, io->lofi, x, u, io->dt, io->xcg, __builtin_amdgcn_readfirstlane(io->fi), flags, &stw);
    bool finx = true;
#pragma unroll
    for (int i = 0; i < 18; ++i) finx = finx && isfinite(x[i]);
    if (!finx) stw |= ST_NONFINITE;
  }
  if (l == 0) {
    if (live) {
#pragma unroll
      for (int i = 0; i < 18; ++i) xg[i * ld + b] = x[i];
#pragma unroll
      for (int c = 0; c < 3; ++c) ug[(1 + c) * ld + b] = u[1 + c];
    }
    if (io->status) io->status[b] = stw;
    const int every = io->every;
    if (io->traj && (t + 1) % every == 0) {
      double *tr = io->traj + (size_t)((t + 1) / every - 1) * 18 * ld + b;
#pragma unroll
      for (int i = 0; i < 18; ++i) __builtin_nontemporal_store(x[i], tr + i * ld);
    }
    if (io->cmd_traj) {
#pragma unroll
      for (int c = 0; c < 3; ++c) io->cmd_traj[((size_t)t * 3 + c) * ld + b] = cmd[c];
    }
    if (io->iters_traj) io->iters_traj[(size_t)t * ld + b] = io->iters;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Bring-up aid (-DF16_DBG_MARK; tools/gpu_fused_marks.py): phase markers as system-scope stores into the first words of cmd_traj, readable
// by a copy on ANOTHER stream while the kernel runs -- how a kernel that never returns is located.  Compiled out of the product.
#ifdef F16_DBG_PAIRSTAMP   // measurement build: cycles per phase of a pair, summed over every pair of the launch -> eight words BEHIND the command record
#define PSTAMP(i) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && ra.cmd_traj) atomicAdd(reinterpret_cast<unsigned long long *>(ra.cmd_traj + (size_t)ra.T * 3 * a.ld) + (i), t1_ - tp0_); tp0_ = t1_; }
#else
#define PSTAMP(i)
#endif
#ifdef F16_DBG_MARK
#define DBGM(i, v) { if (threadIdx.x == 0 && ra.cmd_traj) __hip_atomic_store(reinterpret_cast<long long *>(ra.cmd_traj) + 32 + (i), (long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
#else
#define DBGM(i, v)
#endif
__global__ __launch_bounds__(64, 1) void k_rollout_mpc(RollMpcArgs ra) {
  const MpcArgs &a = ra.m;
  const int N = a.N;
  const Role R = role(N);
  const int l = R.l;
  const unsigned Bu = (unsigned)a.B;
#ifdef F16_DBG_PAIRSTAMP
  unsigned long long tp0_ = __builtin_amdgcn_s_memtime();
#endif
  for (;;) {
    unsigned k = 0;
    if (l == 0) k = atomicAdd(ra.queue, 1u);
    k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
    PSTAMP(0)
    DBGM(0, 1000 + k)
    if (k >= ra.total) break;
    const int t = __builtin_amdgcn_readfirstlane((int)(k / Bu));
    const unsigned j = k - (unsigned)t * Bu;
    const long b = (long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ra.stride ? (unsigned)(((unsigned long long)j * ra.stride) % Bu) : j));
    // ---- wait for step t - 1 of this aircraft, then acquire what its wavefront published.  (Every lane polls the same word -- one
    // broadcast load -- and every lane publishes below: no lane-0-only region on either side of the loop's back edge.)
    // The wait is bounded (2^24 polls, tens of seconds -- the longest legitimate wait is one 40,000-iteration solve, ~0.1 s): a wave
    // that never sees its predecessor flags the aircraft (F16_ST_LOOP_STALL) and goes on, so that the grid drains whatever happens.
    int stall = 0;
    if (t > 0) {
      int polls = 0;
      while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&ra.progress[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < t) {
        __builtin_amdgcn_s_sleep(16);
        if (++polls > (1 << 24)) { stall = F16_ST_LOOP_STALL; break; }
      }
    }
    PSTAMP(1)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PSTAMP(2)
    PairIO io;
    io.x = ra.x; io.u = ra.u; io.dem = ra.dem; io.traj = ra.traj; io.cmd_traj = ra.cmd_traj; io.iters_traj = ra.iters_traj;
    io.status = ra.status; io.tab = ra.tab; io.lofi = ra.lofi; io.exw = a.ext + (size_t)b * mpc_ext_doubles(N);
    io.ld = a.ld; io.b = b; io.t = t; io.N = N; io.every = ra.every; io.fi = ra.fi; io.flags = ra.flags; io.xcg = ra.xcg; io.dt = a.dt;
    io.cmd[0] = NAN; io.cmd[1] = NAN; io.cmd[2] = NAN; io.iters = 0; io.stw = 0; io.stall = stall;
    DBGM(1, 2000 + t)
    const int code = __builtin_amdgcn_readfirstlane(pair_prepare(&io));
    DBGM(2, 3000 + code)
    PSTAMP(3)
    if (code == PAIR_SOLVE) {
      SolveState st;
#ifdef F16_DBG_SKIP_SOLVE      // (measurement build: what a pair costs WITHOUT its solve -- tools/gpu_config5_only.py under F16HIP_SO)
      st.infeasible = 0; st.converged = 1; st.it = 0; st.x[0] = -0.5; st.x[1] = 0.0; st.x[2] = 0.0;
      const bool ok = true;
#elif defined(F16_EXP_STAMPW)
      unsigned long long tK0 = 0;
      const bool ok = solve_aircraft(a, b, (long)k + 1, R, st, (t > 0 || a.warm_load) ? 1 : 0, tK0);
#else
      // (warm start, opt-in: step 0 starts from the plan's previous call if there was one, every later step from the step before --
      //  the buffer is handed from wavefront to wavefront with the state, behind the same release / acquire)
      const bool ok = solve_aircraft(a, b, (long)k + 1, R, st, (t > 0 || a.warm_load) ? 1 : 0);
#endif
      const bool infeasible = __builtin_amdgcn_readfirstlane(st.infeasible) != 0;
      const bool converged = __builtin_amdgcn_readfirstlane(st.converged) != 0;
      const double c0 = bcast0_f64(st.x[0]), c1 = bcast0_f64(st.x[1]), c2 = bcast0_f64(st.x[2]);      // res.x[0:3], env.py:424 (lane 0 owns step 0)
      io.iters = __builtin_amdgcn_readfirstlane(st.it);
      io.cmd[0] = infeasible ? NAN : c0; io.cmd[1] = infeasible ? NAN : c1; io.cmd[2] = infeasible ? NAN : c2;
      if (infeasible) io.stw |= F16_ST_QP_INFEASIBLE;         // OSQP hands back NaN for a problem it certifies infeasible
      else if (!converged || !uniform_flag(ok)) io.stw |= F16_ST_QP_MAXITER;
      wave_lds_sync();
    }
    DBGM(3, 4000 + io.iters)
    PSTAMP(4)
    pair_finish(&io, code);
    PSTAMP(5)
    DBGM(4, 5000)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&ra.progress[b], t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (64 lanes, one word, one value)
    PSTAMP(6)
    DBGM(5, 6000 + t)
  }
  DBGM(6, 7000)
}

}  // namespace wave

bool mpc_wave_enabled(const MpcArgs &a) {
  static const bool off = [] { const char *e = getenv("F16_MPC_WAVE"); return e && e[0] == '0'; }();
  return !off && a.N >= 1 && a.N <= WAVE_MAXN && a.s.scaling > 0 && a.s.max_iter > 0 && a.gramws != nullptr && a.pblk != nullptr;
}

int mpc_wave_solve_launch(f16_ctx *ctx, const MpcArgs &a, void *stream) {
  (void)ctx;
  if (a.N < 1 || a.N > WAVE_MAXN || !a.gramws) return set_error(F16_EINVAL, "wavefront MPC solver needs 1 <= N <= 30 and a workspace");
  if (a.B > 0x7fffffffL) return set_error(F16_EINVAL, "batch too large for one launch");
  MpcArgs w = a;
  static const bool spread = [] { const char *e = getenv("F16_MPC_SPREAD"); return !(e && e[0] == '0'); }();
  w.wave_stride = 0;
  if (!a.order && spread && a.B > 16) {
    static const unsigned primes[] = {2053, 2063, 2069, 2081, 2083, 2087, 2089, 2099};      // (B has at most a few of these factors)
    for (unsigned p : primes)
      if (a.B % p != 0) { w.wave_stride = p; break; }
  }
  // one workgroup per SIMD and a work queue of our own (see k_mpc_wave); the counter is the spare slot behind the scaling block of
  // aircraft 0's workspace.  F16_MPC_WAVE_QUEUE=0: one workgroup per aircraft, distributed by the hardware.
  static const bool queue = [] { const char *e = getenv("F16_MPC_WAVE_QUEUE"); return !(e && e[0] == '0'); }();
  unsigned grid = (unsigned)a.B;
  w.wave_queue = nullptr;
  if (queue) {
    static const int cus = [] {                              // (one GPU model per process: asked once)
      int dev = 0, n = 0;
      return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) ? n : 0;
    }();
    if (cus > 0 && a.B > 4L * cus) {
      w.wave_queue = reinterpret_cast<unsigned *>(a.gramws + wave::GW_SCAL + wave::GW_QUEUE);
      int rc = hip_check(hipMemsetAsync(w.wave_queue, 0, sizeof(unsigned), (hipStream_t)stream), "f16_mpc_batch work-queue counter");
      if (rc) return rc;
      grid = 4u * (unsigned)cus;
    }
  }
  hipLaunchKernelGGL(wave::k_mpc_wave, dim3(grid), dim3(64), 0, (hipStream_t)stream, w);
  return hip_check(hipGetLastError(), "f16_mpc_batch wavefront solve launch");
}

// f16_rollout_mpc: the closed loop as one launch (k_rollout_mpc).  `sync` = the plan's [8 bytes ticket counter | B x int32 progress].
int mpc_wave_rollout_launch(f16_ctx *ctx, const MpcArgs &a, const RolloutMpcCall &c, void *stream) {
  if (a.N < 1 || a.N > WAVE_MAXN || !a.gramws || !a.pblk || !a.ext || !a.Ppk) return set_error(F16_EINVAL, "closed-loop MPC rollout needs a plan with 1 <= hzn <= 30");
  if ((unsigned long long)a.B * (unsigned long long)c.T >= 0xffffffffULL) return set_error(F16_EINVAL, "nsteps x B must stay below 2^32 per call");
  wave::RollMpcArgs r{};
  r.m = a;
  r.m.x = c.x; r.m.dem = c.dem; r.m.xref = nullptr; r.m.ucmd = nullptr; r.m.useq = nullptr; r.m.info = nullptr; r.m.status = nullptr;
  r.m.mode = 0; r.m.wave_ruiz = 1; r.m.order = nullptr; r.m.iters_out = nullptr; r.m.warm = c.warm; r.m.warm_load = c.warm_load; r.m.wave_queue = nullptr;
  r.x = c.x; r.u = c.u; r.dem = c.dem; r.traj = c.traj; r.cmd_traj = c.cmd_traj; r.iters_traj = c.iters_traj; r.status = c.status;
  r.queue = reinterpret_cast<unsigned *>(c.sync);
  r.progress = reinterpret_cast<int32_t *>(c.sync) + 2;
  r.tab = ctx->d_tab; r.lofi = ctx->d_lofi;
  r.T = c.T; r.every = c.every; r.xcg = c.xcg; r.fi = c.fi; r.flags = c.flags;
  r.total = (unsigned)((unsigned long long)a.B * (unsigned long long)c.T);
  r.stride = 0;
  static const bool spread = [] { const char *e = getenv("F16_MPC_SPREAD"); return !(e && e[0] == '0'); }();
  if (spread && a.B > 16) {
    static const unsigned primes[] = {2053, 2063, 2069, 2081, 2083, 2087, 2089, 2099};
    for (unsigned p : primes)
      if (a.B % p != 0) { r.stride = p; break; }
  }
  if (int rc = hip_check(hipMemsetAsync(c.sync, 0, 8 + (size_t)a.B * sizeof(int32_t), (hipStream_t)stream), "f16_rollout_mpc counters")) return rc;
  static const int cus = [] {
    int dev = 0, n = 0;
    return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) ? n : 0;
  }();
  // one wavefront-workgroup per SIMD (all resident: 40 KB of LDS each, four per CU); never more than aircraft -- the surplus could
  // only wait
  const long slots = cus > 0 ? 4L * cus : 1024;
  const unsigned grid = (unsigned)(a.B < slots ? a.B : slots);
  hipLaunchKernelGGL(wave::k_rollout_mpc, dim3(grid), dim3(64), 0, (hipStream_t)stream, r);
  return hip_check(hipGetLastError(), "f16_rollout_mpc launch");
}

}  // namespace f16
