// f16_mpc_solve.hip -- register-resident OSQP ADMM for the condensed MPC QP (N <= 32), gfx950.
//
// The solve the reference delegates to `osqp` (env.py:420-424), by OSQP's published algorithm: Ruiz equilibration
// (D, E, c), rho vector, over-relaxed ADMM, termination on unscaled residuals, rho re-estimated from the scaled ones
// (the tests compare with a rule-for-rule numpy restatement of it).  Same rules as the generic
// solver in f16_control.hip (k_mpc) -- what changes is the mapping: one 512-lane workgroup (8 wavefronts) per aircraft
// and every operator of an ADMM iteration lives in registers for the whole solve.
// Coordinates.  OSQP iterates on xb = D^-1 x, zb = E z, yb with Pb = c D P D, qb = c D q, Ab = E A D.  Here the
// variable is kept UNscaled (x = D xb): the linear system becomes (c P + sigma D^-2 + rho A' W A) x~ = sigma D^-2 x - c q
// + A' E (rho zb - yb) with the row weights W = E^2 (x 1e3 on equality rows) -- the same iterates, but the Toeplitz operators
// A, A' stay the unscaled block-Toeplitz ones.  The constraint variables are kept unscaled as well (z = zb / E, y = yb / E):
// projection bounds are the original l, u, and E enters ONLY as the row weight W = E^2 in w = W (rho z - y) and in A'WA;
// D only through sigma D^-2; c through c q and the dual residual:
//     stage 1   t  = CCs' w_s      (3N x 6N, block upper-triangular Toeplitz)      utils.py:163 (A' part)
//     stage 2   x~ = (P + sigma I + rho A'A)^-1 rhs        (3N x 3N dense)          OSQP linear system
//     stage 3   z~ = CCs x~        (6N x 3N, block lower-triangular Toeplitz)       utils.py:163 (A part)
//   * the kernel is a thin driver over three out-of-line phases that share namespace-scope LDS: ruiz_equilibrate (once),
//     then kkt_factorise / admm_iterate alternately -- each phase has a register allocation of its own;
//   * the KKT matrix c P + sigma D^-2 + rho A'WA is assembled (Gram product) and inverted ON THE fp64 MATRIX CORES
//     (v_mfma_f64_16x16x4_f64, blocked symmetric sweep, four pivots per step, ONE product per tile and step) with the
//     matrix resident in the MFMA accumulators; the inverse is then re-laid through LDS so that stage 2 is 18 FMAs per
//     lane on all eight waves -- it never touches HBM;
//   * stages 1 and 3 are the same block-Toeplitz operator used both ways: lane q of the 16-lane DPP row of horizon
//     step i holds the two 6x3 blocks G_{2q}, G_{2q+1} (36 fp64), reads 12 / 6 contiguous operand doubles from LDS;
//   * partial sums of a row are combined inside the DPP row by recursive halving (mirror / half-mirror / quad
//     exchanges), so an iteration has three workgroup barriers and ~120 KB of LDS data return (the first version: five
//     barriers, 410 KB).
#include <hip/hip_runtime.h>
#include <math.h>

#include <mutex>

#include "f16_mpc.hpp"
#include "f16_smallmat.hpp"

namespace f16 {

constexpr int FT = 512;                    // lanes per aircraft
constexpr int FN = 3 * FAST_MAXN;          // 96

// workgroup-wide reductions of NV <= 16 values at once (red: [8][NV] doubles of LDS); max only of non-negative values.
// Wave totals on the DPP network (uniform), one LDS slot per (wave, value); then lane i < NV of EVERY wave combines column
// i over the eight waves (8 reads per lane) and the NV results become wave-uniform again through v_readlane.  (The first
// version had every lane read all 8 x NV partials: 36 wide reads per lane, 8 clocks each on the one LDS pipe of the CU, for
// the nine values of the termination test -- more LDS time than three ADMM iterations.)
template <int NV>
__device__ __forceinline__ void block_reduce(double (&v)[NV], const bool (&is_sum)[NV], double *red) {
  static_assert(NV <= 16, "one lane per value");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = is_sum[i] ? wave_reduce_dpp<true>(v[i]) : wave_reduce_dpp<false>(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wv * NV + i] = v[i];
  }
  __syncthreads();
  unsigned summask = 0;
#pragma unroll
  for (int i = 0; i < NV; ++i) summask |= is_sum[i] ? (1u << i) : 0u;
  const int col = lane < NV ? lane : 0;
  double x[FT / 64];
#pragma unroll
  for (int w = 0; w < FT / 64; ++w) x[w] = red[w * NV + col];
  double rs = x[0], rm = x[0];
#pragma unroll
  for (int w = 1; w < FT / 64; ++w) { rs += x[w]; rm = fmax(rm, x[w]); }
  const double r = ((summask >> col) & 1u) ? rs : rm;
  const int rlo = __double2loint(r), rhi = __double2hiint(r);
#pragma unroll
  for (int i = 0; i < NV; ++i)
    v[i] = __hiloint2double(__builtin_amdgcn_readlane(rhi, i), __builtin_amdgcn_readlane(rlo, i));
}

// ---------------------------------------------------------------------------------------------------------------
// (P + sigma I + rho A'A)^-1 by the symmetric sweep operator (Gauss-Jordan without pivoting, stable for SPD: after
// sweeping all pivots the array holds MINUS the inverse), in blocked form on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64): the matrix lives in MFMA
// accumulators -- wave w owns tile row w (rows 16w..16w+15, all six 16x16 tiles: 24 fp64 per lane) -- and a block step
// sweeps FOUR pivots at once:
//     M_rest <- M_rest - (C D^-1) C'      one 16x16x4 MFMA per tile        (C = the 4 pivot columns, D = 4x4 pivot block)
//     M[:,K] <- C D^-1,   M[K,K] <- -D^-1                                   (lane-level fix-ups after the MFMAs)
// Per block step: the owner of the pivot rows publishes them to LDS (the panel, by symmetry = the pivot columns), one
// barrier, every lane inverts D in registers (2x2 Schur form), builds its A/B operands from the panel, six MFMAs.
// 4*ceil(n/16) block steps instead of n pivots; padding rows/cols (n..16*nt) carry an identity block.
// Register layout of v_mfma_f64_16x16x4_f64 (probed on the hardware, pinned by tests/test_gpu_control.py::
// test_mfma_inverse): accumulator register q of lane l holds D[4q + l/16][l%16]; A operand: lane l -> A[l%16][l/16];
// B operand: lane l -> B[l/16][l%16].  The four pivot rows of a block step are therefore ONE accumulator register
// (q = p%4) across the whole wave.
typedef double d4_t __attribute__((ext_vector_type(4)));
#ifdef F16_EXP_STAMPM
__device__ double g_inv_stamp[16];      // diagnostic build: per-wave work / barrier-wait cycles of the factorisation
__device__ double g_it_stamp[8 * 6];    // per wave: cycles in phase A, barrier, B, barrier, C (+ test), barrier, summed over iterations
__device__ double g_f_stamp[8];         // last factorisation of workgroup 0: Gram, assembly, sweep, re-layout; equilibration; iterations stamped
#endif
constexpr int NT = FN / 16;   // 6 tile rows / columns

// 1/x to <= 1 ulp without the division sequence (v_rcp_f64 + two Newton steps); x is a positive, normal pivot minor
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
}

// inverse of a 4x4 SPD block given by its lower triangle; false if a leading minor is not positive
__device__ __forceinline__ bool inv4_spd(const double (&d)[4][4], double (&o)[4][4]) {
  const double a = d[0][0], b = d[1][0], c = d[1][1];
  const double detA = a * c - b * b;
  const double ia = rcp_nr(detA);
  const double A00 = c * ia, A10 = -b * ia, A11 = a * ia;                  // A^-1
  const double B00 = d[2][0], B01 = d[2][1], B10 = d[3][0], B11 = d[3][1];  // rows 2,3 x cols 0,1
  const double T00 = B00 * A00 + B01 * A10, T01 = B00 * A10 + B01 * A11;    // T = B A^-1
  const double T10 = B10 * A00 + B11 * A10, T11 = B10 * A10 + B11 * A11;
  const double S00 = d[2][2] - (T00 * B00 + T01 * B01);                     // S = E - T B'
  const double S10 = d[3][2] - (T10 * B00 + T11 * B01);
  const double S11 = d[3][3] - (T10 * B10 + T11 * B11);
  const double detS = S00 * S11 - S10 * S10;
  const double is = rcp_nr(detS);
  const double I00 = S11 * is, I10 = -S10 * is, I11 = S00 * is;            // S^-1
  const double L00 = -(I00 * T00 + I10 * T10), L01 = -(I00 * T01 + I10 * T11);   // -S^-1 T
  const double L10 = -(I10 * T00 + I11 * T10), L11 = -(I10 * T01 + I11 * T11);
  o[2][2] = I00; o[3][2] = o[2][3] = I10; o[3][3] = I11;
  o[2][0] = o[0][2] = L00; o[2][1] = o[1][2] = L01; o[3][0] = o[0][3] = L10; o[3][1] = o[1][3] = L11;
  o[0][0] = A00 - (T00 * L00 + T10 * L10);                                    // A^-1 + T' S^-1 T
  o[1][0] = o[0][1] = A10 - (T01 * L00 + T11 * L10);
  o[1][1] = A11 - (T01 * L01 + T11 * L11);
  return a > 0.0 && detA > 0.0 && S00 > 0.0 && detS > 0.0;
}

// lane-dependent pick of one of four values.  Scalars BY VALUE on purpose: with an array reference the optimiser turns
// the selects into a dynamically indexed load before inlining, and the array then lives in scratch memory.
__device__ __forceinline__ double sel4(double v0, double v1, double v2, double v3, int k) {
  const double lo = (k & 1) ? v1 : v0, hi = (k & 1) ? v3 : v2;
  return (k & 2) ? hi : lo;
}

// Panel buffer of one block step: C[col][0..3] = M[pivot row][col] (FN x 4), then D^-1 (4 x 4, row-major) and a flag.
constexpr int PAN_DI = FN * 4, PAN_OK = FN * 4 + 16, PAN_SIZE = FN * 4 + 18;

// The wave that published pivot rows [k0, k0+4) into pan also inverts the 4x4 pivot block, once for everybody: reads the
// block back (its own LDS writes, same wave), inverts it in registers, lanes 0..3 store one row each.
__device__ __forceinline__ void publish_dinv(double *pan, int k0, int l) {
  double D[4][4], Di[4][4];
  const double2 *src = reinterpret_cast<const double2 *>(pan + k0 * 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double2 u = src[2 * i], v = src[2 * i + 1];
    D[i][0] = u.x; D[i][1] = u.y; D[i][2] = v.x; D[i][3] = v.y;
  }
  const bool good = inv4_spd(D, Di);
  if (l < 4) {
    double2 *dst = reinterpret_cast<double2 *>(pan + PAN_DI + 4 * l);
    dst[0] = make_double2(sel4(Di[0][0], Di[1][0], Di[2][0], Di[3][0], l), sel4(Di[0][1], Di[1][1], Di[2][1], Di[3][1], l));
    dst[1] = make_double2(sel4(Di[0][2], Di[1][2], Di[2][2], Di[3][2], l), sel4(Di[0][3], Di[1][3], Di[2][3], Di[3][3], l));
    if (l == 0) pan[PAN_OK] = good ? 1.0 : 0.0;
  }
}

// One block step of the sweep: pivots 16 Kt + 4 KQ .. +3 (accumulator register KQ of tile row Kt).  cb: this step's
// panel (pivot rows, D^-1); cbn: where the NEXT step's panel is published.
#define F16_LDS_PHASE() __builtin_amdgcn_sched_barrier(0x7)   /* LDS / memory ops stay put, ALU may float */

// Which tiles of the NTT x NTT tile matrix a wavefront holds (wave-uniform): tile row tr, tile columns [j0, j1).  Up to
// four tile rows: wave w owns row w.  Six tile rows (N = 22..32) on eight waves / four SIMDs: waves 0-3 own rows 0-3,
// rows 4 and 5 are split in halves over waves 4-7, so that every SIMD carries one and a half tile rows (with six
// whole-row waves, SIMDs 0 and 1 carried two rows each, SIMDs 2 and 3 one, and waves 6, 7 idled through every
// factorisation).
struct TileOwn { int tr, j0, j1; };
template <int NTT>
__device__ __forceinline__ TileOwn tile_own() {
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  TileOwn o;
#ifdef F16_MPC_SPLIT_ROWS
  if (NTT == 6 && w >= 4) { o.tr = 4 + ((w - 4) >> 1); o.j0 = 3 * ((w - 4) & 1); o.j1 = o.j0 + 3; }
  else
#endif
  { o.tr = w; o.j0 = 0; o.j1 = w < NTT ? NTT : 0; }
  return o;
}
#define F16_OWNS(J) ((J) >= own.j0 && (J) < own.j1)

template <int NTT, int KQ>
__device__ __forceinline__ void inverse_step(d4_t (&acc)[NT], const double *cb, double *cbn, int Kt, const TileOwn own, int lc,
                                             int lq, double ndel, bool &ok) {
  const int w = own.tr;                                  // tile row of this wave
  const bool pl = (lc >> 2) == KQ;                       // this lane's column (within a tile) is a pivot column
  const bool owner = w == Kt;                            // this wave holds (its columns of) the pivot rows (wave-uniform)
  // ---- every LDS read of the step first, and as few as possible: the panel reads of all waves go through the one LDS
  // pipe of the CU right after the barrier (64 lanes x 16 B = 8 clocks per instruction whatever the addresses), and with
  // the lane-level fix-ups of the first versions (28-40 wide reads per lane) that was ~1 k clocks of a ~3 k-clock step.
  double2 a0, a1, ca0, ca1;
  double b_op[NTT], okf, dpiv;
  {
    const double2 *ra = reinterpret_cast<const double2 *>(cb + PAN_DI + 4 * lq);       // row l/16 of D^-1 (= its column)
    a0 = ra[0]; a1 = ra[1];
    okf = cb[PAN_OK];
    dpiv = cb[PAN_DI + 4 * lq + (lc & 3)];
    const double2 *src = reinterpret_cast<const double2 *>(cb + (16 * w + lc) * 4);
    ca0 = src[0]; ca1 = src[1];
#pragma unroll
    for (int J = 0; J < NTT; ++J) b_op[J] = F16_OWNS(J) ? cb[(16 * J + lc) * 4 + lq] : 0.0;
  }
  F16_LDS_PHASE();
  ok = ok && okf > 0.5;
  // ---- ONE matrix-core product per tile does the whole block step.  With X = C D^-1 (C = the pivot columns = the panel):
  //   A operand  -X[i][k] on ordinary rows, +D^-1[r][k] on the four pivot rows (owner wave)
  //   B operand  C[j][k] on ordinary columns, -delta(k, kk) on the four pivot columns
  //   accumulator zeroed beforehand on pivot rows and pivot columns
  // gives  M[i][j] -= X[i] . C[j]  |  M[i][kk] = X[i][kk]  |  M[r][j] = (D^-1 C')[r][j] = X[j][r]  |  M[r][kk] = -D^-1[r][kk]
  // -- the rank-4 update, the new pivot columns, the new pivot rows and the pivot block, with no lane-level fix-up reads.
  double a_op = -(ca0.x * a0.x + ca0.y * a0.y + ca1.x * a1.x + ca1.y * a1.y);
  if (owner && pl) a_op = dpiv;
#pragma unroll
  for (int J = 0; J < NTT; ++J)
    if (J == Kt && F16_OWNS(J)) {
      b_op[J] = pl ? ndel : b_op[J];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[J][q] = pl ? 0.0 : acc[J][q];
    }
  if (owner) {
#pragma unroll
    for (int J = 0; J < NTT; ++J) acc[J][KQ] = 0.0;
  }
#pragma unroll
  for (int J = 0; J < NTT; ++J)
    if (F16_OWNS(J)) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op, b_op[J], acc[J], 0, 0, 0);
  // ---- publish the next pivot rows (every wave of that tile row its own columns) and the inverse of their pivot block
  // (the wave that holds the diagonal tile: it reads back its own writes)
  if (KQ < 3) {
    if (owner) {
#pragma unroll
      for (int J = 0; J < NTT; ++J)
        if (F16_OWNS(J)) cbn[(16 * J + lc) * 4 + lq] = acc[J][(KQ + 1) & 3];
      if (F16_OWNS(Kt)) publish_dinv(cbn, 16 * Kt + 4 * (KQ + 1), 16 * lq + lc);
    }
  } else if (w == Kt + 1 && Kt + 1 < NTT) {
#pragma unroll
    for (int J = 0; J < NTT; ++J)
      if (F16_OWNS(J)) cbn[(16 * J + lc) * 4 + lq] = acc[J][0];
    if (F16_OWNS(Kt + 1)) publish_dinv(cbn, 16 * (Kt + 1), 16 * lq + lc);
  }
}

// In-register blocked sweep over NTT x NTT tiles.  On return acc holds MINUS the inverse of P + sigma I + r A'A
// (identity block on the padding rows n..16 NTT); tile (w,J), register q, lane l <-> element (16w + 4q + l/16, 16J + l%16).
// Cs: LDS scratch [2][PAN_SIZE] doubles (double-buffered panel).  Return value is uniform over the workgroup.
// Out of line on purpose: the block step wants ~200 registers of its own; as a call, the caller's live state is parked
// once per inverse instead of being spilled and reloaded inside every block step.
#ifndef F16_INV_ATTR
#define F16_INV_ATTR __forceinline__
#endif
template <int NTT>
__device__ F16_INV_ATTR bool mfma_inverse(d4_t *acc_out, double *Cs, const d4_t *acc_in) {
  const int tid = threadIdx.x, l = tid & 63;
  const TileOwn own = tile_own<NTT>();
  int lc = l & 15, lq = l >> 4;
  // opaque to the optimiser: otherwise the tile index arithmetic below (loop-invariant for the caller's factorisation
  // loop) is hoisted to the top of the kernel, does not fit in registers there and comes back through scratch memory
  asm volatile("" : "+v"(lc), "+v"(lq));
  bool ok = true;
  const double ndel = (lq == (lc & 3)) ? -1.0 : 0.0;     // B operand on a pivot column: -delta(k, kk)
  d4_t acc[NT];
#pragma unroll
  for (int J = 0; J < NTT; ++J) acc[J] = acc_in[J];
  double *c0 = Cs, *c1 = Cs + PAN_SIZE;
  __syncthreads();                        // previous users of Cs are done
  if (own.tr == 0 && own.j1 > 0) {
#pragma unroll
    for (int J = 0; J < NTT; ++J) c0[(16 * J + lc) * 4 + lq] = acc[J][0];
    publish_dinv(c0, 0, l);
  }
#ifdef F16_EXP_STAMPM
  unsigned long long tw = 0, tb = 0, ts0 = __builtin_amdgcn_s_memtime();
#define ISTAMP(acc_) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t1_ = __builtin_amdgcn_s_memtime(); acc_ += t1_ - ts0; ts0 = t1_; }
#else
#define ISTAMP(acc_)
#endif
#pragma unroll
  for (int Kt = 0; Kt < NTT; ++Kt) {
    __syncthreads();
    ISTAMP(tb)
    if (own.j1 > 0) inverse_step<NTT, 0>(acc, c0, c1, Kt, own, lc, lq, ndel, ok);
    ISTAMP(tw)
    __syncthreads();
    ISTAMP(tb)
    if (own.j1 > 0) inverse_step<NTT, 1>(acc, c1, c0, Kt, own, lc, lq, ndel, ok);
    ISTAMP(tw)
    __syncthreads();
    ISTAMP(tb)
    if (own.j1 > 0) inverse_step<NTT, 2>(acc, c0, c1, Kt, own, lc, lq, ndel, ok);
    ISTAMP(tw)
    __syncthreads();
    ISTAMP(tb)
    if (own.j1 > 0) inverse_step<NTT, 3>(acc, c1, c0, Kt, own, lc, lq, ndel, ok);
    ISTAMP(tw)
  }
#ifdef F16_EXP_STAMPM
  if (blockIdx.x == 0 && l == 0) { g_inv_stamp[2 * (tid >> 6)] = (double)tw; g_inv_stamp[2 * (tid >> 6) + 1] = (double)tb; }
#endif
#pragma unroll
  for (int J = 0; J < NTT; ++J) acc_out[J] = acc[J];
  return __syncthreads_and(ok) != 0;      // uniform over the workgroup (only the tile-row waves looked at pivots)
}

// tiles of a packed (lower triangle, row-major) symmetric matrix: element (16w + 4q + l/16, 16J + l%16), identity padding
template <int NTT>
__device__ __forceinline__ void load_packed_tiles(d4_t (&acc)[NT], const double *Pg, int n, int w, int lc, int lq, int j0 = 0,
                                                  int j1 = NTT) {
#pragma unroll
  for (int J = 0; J < NTT; ++J) {
    if (J < j0 || J >= j1) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 16 * w + 4 * q + lq, j = 16 * J + lc;
      const bool in = i < n && j < n;
      const int e = in ? (i >= j ? tri(i, j) : tri(j, i)) : 0;
      const double pv = Pg[e];
      acc[J][q] = in ? pv : (i == j ? 1.0 : 0.0);
    }
  }
}

// diagnostic / test entry: inverse of B packed SPD matrices through mfma_inverse
__global__ __launch_bounds__(FT) void k_dbg_inverse(const double *pk, double *out, int n, long B) {
  __shared__ __attribute__((aligned(16))) double cv[2 * FN * 4 + 40];
  const int np = n * (n + 1) / 2, nt = (n + 15) >> 4;
  const int lc = threadIdx.x & 15, lq = (threadIdx.x & 63) >> 4;
  const TileOwn own = tile_own<NT>();
  const int w = own.tr;
  for (long b = blockIdx.x; b < B; b += gridDim.x) {
    const double *Pg = pk + (size_t)b * np;
    double *o = out + (size_t)b * n * n;
    d4_t acc[NT], m0[NT];
#pragma unroll
    for (int J = 0; J < NT; ++J) m0[J] = d4_t{0.0, 0.0, 0.0, 0.0};
    load_packed_tiles<NT>(m0, Pg, n, w, lc, lq, own.j0, own.j1);
    const bool ok = mfma_inverse<NT>(acc, cv, m0);
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 16 * w + 4 * q + lq, j = 16 * J + lc;
        if (F16_OWNS(J) && w < nt && J < nt && i < n && j < n) o[i * n + j] = ok ? -acc[J][q] : NAN;
      }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Cross-lane sums over the 16 lanes of a DPP row by recursive halving: at each of the first steps a lane keeps half of
// its values and hands the other half to its mirror partner, so V values cost about V + log steps instead of 4 V.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HMIRROR = 0x141, DPP_MIRROR = 0x140;

// h,g,e = bits 3,2,1 of the lane's position in its row.  Totals land in: lanes 0-3 v0, 4-7 v1, 8-11 v2 (12-15: 0)
__device__ __forceinline__ double reduce3(double v0, double v1, double v2, bool h, bool g) {
  const double s0 = h ? v0 : v2, s1 = h ? v1 : 0.0;
  double k0 = h ? v2 : v0, k1 = h ? 0.0 : v1;
  k0 += dpp_f64<DPP_MIRROR>(s0);
  k1 += dpp_f64<DPP_MIRROR>(s1);
  const double s = g ? k0 : k1;
  double k = g ? k1 : k0;
  k += dpp_f64<DPP_HMIRROR>(s);
  k += dpp_f64<DPP_XOR2>(k);
  k += dpp_f64<DPP_XOR1>(k);
  return k;
}
// The same with the three inputs in SLOT ORDER -- a = the value this lane's half of the row keeps (h ? v2 : v0), b = v1,
// c = the value it hands to its mirror partner (h ? v0 : v2): the caller stores its operator rows in that order per lane
// (stage 2: the rows of the re-laid KKT inverse), which removes the four selects of the mirror step.  v1 is summed on both
// halves (the upper half's copy ends in lanes 12-15, which own nothing).
__device__ __forceinline__ double reduce3_slots(double a, double b, double c, bool g) {
  const double k0 = a + dpp_f64<DPP_MIRROR>(c);
  const double k1 = b + dpp_f64<DPP_MIRROR>(b);
  const double s = g ? k0 : k1;
  double k = g ? k1 : k0;
  k += dpp_f64<DPP_HMIRROR>(s);
  k += dpp_f64<DPP_XOR2>(k);
  k += dpp_f64<DPP_XOR1>(k);
  return k;
}
// reduce4: quad k of the row (lanes 4k..4k+3) ends up with the total of v[k] -- mirror step keeps two of the four values
// and hands two over, half-mirror step keeps one, then two butterfly steps.  Written naively that is twelve selects per
// call; instead the caller keeps its four values in slot order (slot_of below): slot 0 = the value this lane's quad ends
// up owning, slot 1 = the one its half-mirror partner owns, slots 2, 3 = what the mirror partner keeps in its slots 0, 1
// -- a lane-dependent permutation the caller applies ONCE to the operator it multiplies with (stage 2: the accumulator
// registers of the KKT inverse).
__device__ __forceinline__ int slot_of(int slot, bool h, bool g) {
  const int hh = h ? 1 : 0, gg = g ? 1 : 0;
  return slot == 0 ? 2 * hh + gg : (slot == 1 ? 2 * hh + (1 - gg) : (slot == 2 ? 2 * (1 - hh) + (1 - gg) : 2 * (1 - hh) + gg));
}
__device__ __forceinline__ double reduce4_slots(d4_t p) {
  const double k0 = p[0] + dpp_f64<DPP_MIRROR>(p[2]);
  const double k1 = p[1] + dpp_f64<DPP_MIRROR>(p[3]);
  double k = k0 + dpp_f64<DPP_HMIRROR>(k1);
  k += dpp_f64<DPP_XOR2>(k);
  k += dpp_f64<DPP_XOR1>(k);
  return k;
}
// totals land in: lanes 0,1 v0 | 2,3 v1 | 4-7 v2 | 8,9 v3 | 10,11 v4 | 12-15 v5.  The six inputs come in slot order:
// p0..p2 = the three values this lane's half of the row keeps (h ? v3,v4,v5 : v0,v1,v2), p3..p5 = the three it hands to
// its mirror partner -- the caller's operator (the rows of the Toeplitz blocks) is stored in that order per lane, which
// removes the twelve selects of the first step.
__device__ __forceinline__ double reduce6_slots(double p0, double p1, double p2, double p3, double p4, double p5, bool g,
                                                bool e) {
  const double k0 = p0 + dpp_f64<DPP_MIRROR>(p3);
  const double k1 = p1 + dpp_f64<DPP_MIRROR>(p4);
  const double k2 = p2 + dpp_f64<DPP_MIRROR>(p5);
  const double t0 = g ? k0 : k2, t1 = g ? k1 : 0.0;
  double n0 = g ? k2 : k0, n1 = g ? 0.0 : k1;
  n0 += dpp_f64<DPP_HMIRROR>(t0);
  n1 += dpp_f64<DPP_HMIRROR>(t1);
  const bool lo2 = !g && !e, hi2 = !g && e;
  const double u = lo2 ? n1 : n0;
  double r = hi2 ? n1 : n0;
  r += dpp_f64<DPP_XOR2>(u);
  r += dpp_f64<DPP_XOR1>(r);
  return r;
}

// ---- the two Toeplitz operators of an iteration share ONE register-resident pair of 6x3 blocks per lane:
// lane q of DPP row `blk` holds Gs_d = G_d[S][:] for d = 2q, 2q+1 (zero for d >= N), S = kept state rows.
//   stage 1 (CCs' w)_j   = sum_d Gs_d' w_{j+d}     -> row blk = j, reads the 12 contiguous doubles w_{j+2q}, w_{j+2q+1}
//   stage 3 (CCs x~)_i   = sum_d Gs_d  x~_{i-d}    -> row blk = i, reads the 6 contiguous doubles x~_{i-2q-1}, x~_{i-2q}
// Out-of-range blocks read zeros (the vectors are zero-padded in LDS), so there is no per-lane masking.
// (operand loads and arithmetic are separate calls: the caller issues every LDS read of a phase first -- left alone the
//  compiler emits read, wait, use in source order and a phase pays three or four LDS round trips instead of one)
// (slot order of the kept state rows, see reduce6_slots: lanes of the upper half of a row hold rows 3,4,5,0,1,2.  The
//  state-row vectors in LDS carry NINE doubles per horizon step -- rows 0..5, then 0..2 again -- so that either order
//  is six contiguous doubles: wp points at 9 * step + (h ? 3 : 0).)
constexpr int WROW = 9;
__device__ __forceinline__ void stage1_load(const double *wp, double (&wv)[12]) {
#pragma unroll
  for (int k = 0; k < 6; ++k) { wv[k] = wp[k]; wv[6 + k] = wp[WROW + k]; }
}
__device__ __forceinline__ void stage1_fma(const double (&Gd)[2][6][3], const double (&wv)[12], double (&o)[3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) { s0 = fma(Gd[0][rr][c], wv[rr], s0); s1 = fma(Gd[1][rr][c], wv[6 + rr], s1); }
    o[c] = s0 + s1;
  }
}
__device__ __forceinline__ void stage3_load(const double *xp, double (&xv)[6]) {
#pragma unroll
  for (int k = 0; k < 6; ++k) xv[k] = xp[k];          // x~_{i-2q-1} (3), x~_{i-2q} (3)
}
__device__ __forceinline__ void stage3_fma(const double (&Gd)[2][6][3], const double (&xv)[6], double (&o)[6]) {
#pragma unroll
  for (int rr = 0; rr < 6; ++rr) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) { s0 = fma(Gd[0][rr][c], xv[3 + c], s0); s1 = fma(Gd[1][rr][c], xv[c], s1); }
    o[rr] = s0 + s1;
  }
}

#define MPC_PHASE() __builtin_amdgcn_sched_barrier(0x7)      /* LDS / memory ops stay put, ALU may float */
constexpr int WSP = WROW * 64;         // zero-padded state-row vectors (stage 1 reads up to block 31 + 31 + 1)
constexpr int XOFF = 3 * 32;           // zeros in front of x~ (stage 3 reads down to block -31)
constexpr int XTP = XOFF + FN + 8;

// max over the 16 lanes of a DPP row, the result in every lane (values >= 0)
__device__ __forceinline__ double row_allmax(double v) {
  v = fmax(v, dpp_f64<DPP_XOR1>(v));
  v = fmax(v, dpp_f64<DPP_XOR2>(v));
  v = fmax(v, dpp_f64<DPP_HMIRROR>(v));
  v = fmax(v, dpp_f64<DPP_MIRROR>(v));
  return v;
}

constexpr int ES9 = 9 * 64;            // E of all nine state rows per horizon step (equilibration), zero padded like wsP
constexpr int WGN = 6 * FAST_MAXN + 8; // Gram weights of the kept state rows, zero padded to whole k-steps

// ---------------------------------------------------------------------------------------------------------------
// LDS of the solver kernel (143 KB of the 160 KB of a CU: one workgroup per CU in any case -- 512 lanes x 256 registers
// fill its register file).  Namespace scope, so that the phases of a solve can be SEPARATE FUNCTIONS (real calls):
// equilibrate once, then factorise / iterate alternately.  Each phase then gets a register allocation of its own; as one
// inlined kernel the allocator kept values of the factorisation path alive across the iteration loop and spilled the
// loop's operators (the N = 30 instantiation reloaded 16 of its 36 Toeplitz operands from scratch in every iteration).
// The iteration's zero-padded vectors sit in ONE block: during a factorisation the same memory carries one tile row of
// the inverse on its way to the per-lane layout (Mst, 16 x 96 doubles).
__shared__ __attribute__((aligned(16))) double s_itv[2 * WSP + 2 * XTP];
static_assert(2 * WSP + 2 * XTP >= 16 * FN, "the relayout buffer must fit in the iteration vectors");
#define wsP (s_itv)
#define ysP (s_itv + WSP)
#define xtP (s_itv + 2 * WSP)
#define xcP (s_itv + 2 * WSP + XTP)
#define Mst (s_itv)
__shared__ __attribute__((aligned(16))) double s_Cs[2 * FN * 4 + 40];      // pivot panels of the sweep
__shared__ double s_rhs[FN], s_wc[FN], s_wr[FN + 4], s_yc[FN], s_yr[FN + 4], s_red[8 * 9];
__shared__ double s_sg2[FN], s_cq[FN], s_q[FN], s_cD[FN], s_cinv;          // per variable: sigma D^-2, c q, q, c D; 1 / c
__shared__ double s_Dv[XTP], s_Es9[ES9], s_Ecv[FN], s_Erv[FN + 4], s_Wg[WGN], s_Wcv[FN], s_Wrv[FN + 4], s_nPm[FN];
__shared__ __attribute__((aligned(16))) double s_Gl[27 * (FAST_MAXN + 1)]; // all nine rows of every G_k (equilibration, Gram) + a zero block
__shared__ double s_px[18 * FT];       // this lane's 18 entries of P (termination test): lane q of row blk holds P[3 blk + c][6 q + cc]
__shared__ double s_lc[4 * FT];        // per constraint row: lower bound | upper bound | weight W = E^2 | rho-vector factor

struct LaneRole {                      // fixed for the whole launch; recomputed by every phase from threadIdx (cheap)
  int w, l, lc, lq, blk, q, kind, sub, r9, k3, xe, xec, dup;
  bool h, g, e, inb, srow, xown;
};
__device__ __forceinline__ LaneRole lane_role(int N) {
  LaneRole r;
  const int tid = threadIdx.x;
  r.w = tid >> 6; r.l = tid & 63; r.lc = r.l & 15; r.lq = r.l >> 4; r.blk = tid >> 4; r.q = r.lc;
  r.h = r.lc & 8; r.g = r.lc & 4; r.e = r.lc & 2;
  r.inb = r.blk < N;
  r.kind = 0; r.sub = 0;               // 1 state row rr = sub, 2 command row c = sub, 3 rate row c = sub
  if (r.inb) {
    if (r.lc == 0 || r.lc == 2 || r.lc == 4) { r.kind = 1; r.sub = r.lc >> 1; }
    else if (r.lc == 8 || r.lc == 10 || r.lc == 12) { r.kind = 1; r.sub = 3 + ((r.lc - 8) >> 1); }
    else if (r.lc == 1 || r.lc == 3 || r.lc == 5) { r.kind = 2; r.sub = (r.lc - 1) >> 1; }
    else if (r.lc == 9 || r.lc == 11 || r.lc == 13) { r.kind = 3; r.sub = (r.lc - 9) >> 1; }
  }
  // x9 row of a state-row owner (kept rows: SROW[sub]; rows without bounds: lanes 6 -> phi, 14 -> theta, 7 -> lf1)
  r.r9 = r.kind == 1 ? (r.sub < 5 ? r.sub + 2 : 8) : (r.lc == 6 ? 0 : (r.lc == 14 ? 1 : (r.lc == 7 ? 7 : -1)));
  r.srow = r.inb && r.r9 >= 0;
  r.k3 = 3 * r.blk + r.sub;            // index of a command / rate row
  r.xown = r.inb && (r.lc == 0 || r.lc == 4 || r.lc == 8);
  r.xe = 3 * r.blk + (r.lc >> 2);
  r.xec = r.xe < FN ? r.xe : FN - 1;   // in-range index for lanes that own no variable
  r.dup = (r.kind == 1 && r.sub < 3) ? 6 : 0;      // state rows 0..2 are stored twice (stage1_load)
  return r;
}
__device__ __forceinline__ double *w_slot(const LaneRole &r, double *sP, double *sc, double *sr) {
  return r.kind == 1 ? sP + WROW * r.blk + r.sub : (r.kind == 2 ? sc + r.k3 : sr + r.k3);
}

// ---- Ruiz equilibration (OSQP scaling.c:scale_data).  Norms of the scaled matrices are formed from the ORIGINAL entries
// and the running D, E, c:
//   columns / rows of Pb = c D P D    from |P| held as matrix-core tiles (tile-row waves, DPP row maxima)
//   columns of Ab = E A D             the stage-1 pattern with (max, x) instead of (+, x) on |G_d| and E
//   rows of Ab                        the stage-3 pattern likewise on |G_d| and D
// All nine state rows take part (rows without bounds are rows of A as OSQP sees it).  On entry s_Dv / s_Es9 / s_Ecv / s_Erv
// hold D = E = 1 (zero padded); on return they hold the final D, E; out3 = this lane's D, E and the cost scaling c.
template <int NTT>
__device__ __noinline__ void ruiz_equilibrate(const double *Pg, int N, int passes, double qe, double *out3) {
  const int n = 3 * N;
  const LaneRole r = lane_role(N);
  const int w = r.w, lc = r.lc, lq = r.lq, blk = r.blk, q = r.q;
  double De = 1.0, Eo = 1.0, cs = 1.0;
  d4_t pt[NT];                                               // |P| tiles
  if (w < NTT) {
    load_packed_tiles<NTT>(pt, Pg, n, w, lc, lq);
#pragma unroll
    for (int J = 0; J < NTT; ++J)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int i = 16 * w + 4 * qq + lq, j = 16 * J + lc;
        pt[J][qq] = (i < n && j < n) ? fabs(pt[J][qq]) : 0.0;
      }
  }
  auto p_row_norms = [&]() {                                 // nPm[i] = D_i max_j |P_ij| D_j   (c applied by the reader)
    if (w < NTT) {
      double dj[NTT];
#pragma unroll
      for (int J = 0; J < NTT; ++J) dj[J] = s_Dv[XOFF + 16 * J + lc];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        double m = 0.0;
#pragma unroll
        for (int J = 0; J < NTT; ++J) m = fmax(m, pt[J][qq] * dj[J]);
        m = row_allmax(m);
        const int i = 16 * w + 4 * qq + lq;
        if (lc == qq && i < n) s_nPm[i] = s_Dv[XOFF + i] * m;
      }
    }
  };
  p_row_norms();
  __syncthreads();
  const int d0 = 2 * q, d1 = 2 * q + 1;
  // |G_d| of this lane's two Toeplitz blocks, all nine rows, zero beyond the horizon: read ONCE (54 LDS reads per lane that
  // every pass repeated -- the stores of a pass may alias them as far as the compiler can tell)
  double ag0[27], ag1[27];
  {
    const double *g0 = s_Gl + (d0 < N ? d0 : 0) * 27, *g1 = s_Gl + (d1 < N ? d1 : 0) * 27;
    const double m0 = d0 < N ? 1.0 : 0.0, m1 = d1 < N ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 27; ++k) { ag0[k] = fabs(g0[k]) * m0; ag1[k] = fabs(g1[k]) * m1; }
  }
  for (int pass = 0; pass < passes; ++pass) {
    // column norms of the state block of Ab (before the D of the column) and row norms (before the E of the row):
    // lane q covers the blocks d = 2q, 2q+1
    double colS[3] = {0.0, 0.0, 0.0}, rowS[9];
    const double *ep = s_Es9 + 9 * (blk + 2 * q), *dp = s_Dv + XOFF + 3 * (blk - 2 * q - 1);
    double dv[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) dv[k] = dp[k];               // D of step i-2q-1 (3), of step i-2q (3)
#pragma unroll
    for (int rr = 0; rr < 9; ++rr) {
      const double e0 = ep[rr], e1 = ep[9 + rr];
      double m = 0.0;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double a0 = ag0[rr * 3 + c], a1 = ag1[rr * 3 + c];
        colS[c] = fmax(colS[c], fmax(a0 * e0, a1 * e1));
        m = fmax(m, fmax(a0 * dv[3 + c], a1 * dv[c]));
      }
      rowS[rr] = row_allmax(m);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) colS[c] = row_allmax(colS[c]);
    double Dt = 1.0, Et = 1.0;
    if (r.xown) {
      const int c = lc >> 2;
      const double cS = c == 0 ? colS[0] : (c == 1 ? colS[1] : colS[2]);
      const double colA = De * fmax(fmax(cS, s_Ecv[r.xe]), fmax(s_Erv[r.xe], s_Erv[r.xe + 3]));
      Dt = 1.0 / sqrt(osqp_limit_scaling(fmax(cs * s_nPm[r.xe], colA)));
    }
    if (r.srow) {
      double rs = rowS[0];
#pragma unroll
      for (int rr = 1; rr < 9; ++rr) rs = r.r9 == rr ? rowS[rr] : rs;
      Et = 1.0 / sqrt(osqp_limit_scaling(Eo * rs));
    } else if (r.kind == 2) {
      Et = 1.0 / sqrt(osqp_limit_scaling(Eo * s_Dv[XOFF + r.k3]));
    } else if (r.kind == 3) {
      Et = 1.0 / sqrt(osqp_limit_scaling(Eo * fmax(s_Dv[XOFF + r.k3], s_Dv[XOFF + r.k3 - 3])));
    }
    __syncthreads();                                        // every lane has read the old D and E
    De *= Dt; Eo *= Et;
    if (r.xown) s_Dv[XOFF + r.xe] = De;
    if (r.srow) s_Es9[9 * blk + r.r9] = Eo;
    if (r.kind == 2) s_Ecv[r.k3] = Eo;
    if (r.kind == 3) s_Erv[r.k3] = Eo;
    __syncthreads();
    p_row_norms();                                           // with the new D: cost scaling now, column norms of the next pass
    __syncthreads();
    double v2[2] = {r.xown ? cs * s_nPm[r.xe] : 0.0, r.xown ? cs * De * fabs(qe) : 0.0};
    const bool s2[2] = {true, false};
    block_reduce<2>(v2, s2, s_red);
    cs *= 1.0 / fmax(osqp_limit_scaling(v2[0] / n), osqp_limit_scaling(v2[1]));
  }
  out3[0] = De; out3[1] = Eo; out3[2] = cs;
}

// ---- A'WA as matrix-core tiles.  A'WA = sum over the kept constraint rows of w a a': the state
// rows (6N x 3N block lower-triangular Toeplitz, never formed) as a Gram product on the matrix cores -- k-step kk covers rows
// 4kk..4kk+3, the operand of column tile T is G_(i-j)[S r][c] gathered from LDS, the SAME value serves as the A operand of
// tile row T and (times w) as the B operand of tile column T --, the command rows (identity) and rate rows (D = I - shift_3)
// as a diagonal / third-off-diagonal fix-up.  gram: tile row w of this lane (24 values).
// Causality: row (i, r) of the state block has entries only in columns of steps j <= i, so column tile T sees nothing of
// the k-steps before kk0(T) = 6 floor(16 T / 3) / 4 and tile (w, J) starts at kk0(max(w, J)): 620 instead of 1,620
// tile-k-steps at N = 30.  The k range is cut at these thresholds into SEGMENTS with a fixed set of live tiles (segment S:
// column tiles 0..S), so that a segment is straight-line code -- operands of the next k-step fetched from LDS under the
// matrix-core products of the current one.  (As ONE loop with a per-tile `if (kk >= kk0(J))` the compiler kept a second copy
// of every accumulator and waited out the full latency of each product and of each gather in turn: 57 k cycles for the
// 154 products of the busiest wave.)
__host__ __device__ constexpr int gram_kk0(int T) { return (6 * ((16 * T) / 3)) >> 2; }
struct GramOps { double a, b[NT]; };
struct GramAddr { int i, iw, ia, ib[NT]; };                  // row step, LDS indices of the weight and of the gathers
struct GramRaw { double wgt, ga, gb[NT]; };
// The operands of one k-step in three stages, so that the loop below can keep them in flight: indices (integer arithmetic),
// LDS reads, finish (weights and causality masks).
template <int NTT, int S>
__device__ __forceinline__ void gram_addr(GramAddr &A, int kk, int lq, int offW, const int (&offT)[NTT]) {
  const int rw = 4 * kk + lq, i = rw / 6, rr = rw - 6 * i;
  const int base = 27 * i + 3 * (rr < 5 ? rr + 2 : 8);
  A.i = i; A.iw = rw;
  // In segment S the column tiles 0..S-2 lie wholly below the diagonal (every row step i >= every column step j: index
  // >= 0, no mask); tiles S-1 and S straddle it.  Rows beyond 6N (last k-step) carry weight 0 and gather from the
  // zero-filled block N of s_Gl.
  const int ia = base + offW;
  A.ia = ia < 0 ? 0 : ia;
#pragma unroll
  for (int J = 0; J <= S; ++J) {
    const int ib = base + offT[J];
    A.ib[J] = (J >= S - 1 && ib < 0) ? 0 : ib;
  }
}
template <int S>
__device__ __forceinline__ void gram_read(GramRaw &R, const GramAddr &A) {
  R.wgt = s_Wg[A.iw];                                        // 0 beyond row 6N
  R.ga = s_Gl[A.ia];
#pragma unroll
  for (int J = 0; J <= S; ++J) R.gb[J] = s_Gl[A.ib[J]];
}
template <int NTT, int S>
__device__ __forceinline__ void gram_finish(GramOps &o, const GramAddr &A, const GramRaw &R, int jW, const int (&jT)[NTT]) {
  o.a = A.i >= jW ? R.ga : 0.0;
#pragma unroll
  for (int J = 0; J <= S; ++J) {
    const double v = R.gb[J] * R.wgt;
    o.b[J] = (J >= S - 1) ? (A.i >= jT[J] ? v : 0.0) : v;
  }
}
// One segment: the reads of k-step kk+1 and the indices of kk+2 are issued before the products of kk (left to itself the
// compiler waits for each k-step's gathers right after issuing them: ~850 clocks per k-step for ~200 clocks of products).
template <int NTT, int S>
__device__ __forceinline__ void gram_segment(d4_t (&acc)[NT], int lo, int hi, int lq, int N, int jW, int offW,
                                             const int (&jT)[NTT], const int (&offT)[NTT]) {
  (void)N;
  if (lo >= hi) return;                                      // (uniform)
  GramAddr A0, A1, A2;
  GramRaw R0, R1;
  GramOps o;
  const int last = hi - 1;
  gram_addr<NTT, S>(A0, lo, lq, offW, offT);
  gram_read<S>(R0, A0);
  gram_addr<NTT, S>(A1, lo + 1 < hi ? lo + 1 : last, lq, offW, offT);
  // two k-steps per trip with the raw operands in ping-pong registers (a rotating copy R0 = R1 makes the compiler wait for
  // the reads it has just issued)
  for (int kk = lo; kk < hi; kk += 2) {
    gram_read<S>(R1, A1);                                    // k-step kk + 1 (past the end: repeats hi - 1, harmless)
    F16_LDS_PHASE();
    gram_addr<NTT, S>(A2, kk + 2 < hi ? kk + 2 : last, lq, offW, offT);
    gram_finish<NTT, S>(o, A0, R0, jW, jT);
#pragma unroll
    for (int J = 0; J <= S; ++J) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a, o.b[J], acc[J], 0, 0, 0);
    if (kk + 1 < hi) {                                       // (uniform)
      gram_read<S>(R0, A2);                                  // k-step kk + 2
      F16_LDS_PHASE();
      gram_addr<NTT, S>(A0, kk + 3 < hi ? kk + 3 : last, lq, offW, offT);
      gram_finish<NTT, S>(o, A1, R1, jW, jT);
#pragma unroll
      for (int J = 0; J <= S; ++J) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a, o.b[J], acc[J], 0, 0, 0);
    }
    // next trip: step kk + 2 is (A2, R0), step kk + 3 is A0 (indices only so far)
    A1 = A0; A0 = A2;
  }
}
template <int NTT, int S>
struct GramSegments {
  static __device__ __forceinline__ void run(d4_t (&acc)[NT], int kw, int nk, int lq, int N, int jW, int offW,
                                             const int (&jT)[NTT], const int (&offT)[NTT]) {
    GramSegments<NTT, S - 1>::run(acc, kw, nk, lq, N, jW, offW, jT, offT);
    const int lo = kw > gram_kk0(S) ? kw : gram_kk0(S);
    const int top = S + 1 < NTT ? gram_kk0(S + 1) : (1 << 20);
    gram_segment<NTT, S>(acc, lo, top < nk ? top : nk, lq, N, jW, offW, jT, offT);
  }
};
template <int NTT>
struct GramSegments<NTT, -1> {
  static __device__ __forceinline__ void run(d4_t (&)[NT], int, int, int, int, int, int, const int (&)[NTT], const int (&)[NTT]) {}
};

// LOWER: only the tiles on and below the diagonal (J <= tile row) -- the matrix is symmetric, and tile row w then runs ONE
// segment (its tiles 0..w all start at kk0(w)): 45 x 1, 38 x 2, 30 x 3, 21 x 4, 14 x 5, 6 x 6 products on waves 0..5 at N = 30
// instead of 154 on wave 0 and 36 on wave 5.  The caller mirrors the result through the Gram workspace.
template <int NTT, bool LOWER = false>
__device__ __forceinline__ void gram_tiles(d4_t (&acc)[NT], int N) {
  const int n = 3 * N;
  const int tid = threadIdx.x, lc = tid & 15, lq = (tid & 63) >> 4;
  const TileOwn own = tile_own<NTT>();
  const int w = own.tr;
#pragma unroll
  for (int J = 0; J < NTT; ++J) acc[J] = d4_t{0.0, 0.0, 0.0, 0.0};
  if (own.j1 > 0) {
    int offT[NTT], jT[NTT];
#pragma unroll
    for (int T = 0; T < NTT; ++T) {
      const int col = 16 * T + lc;
      jT[T] = col < n ? col / 3 : 1 << 20;                   // padding columns: never reached (i >= jT fails)
      offT[T] = (col - 3 * (col / 3)) - 27 * (col / 3);
    }
    const int colw = 16 * w + lc;
    const int jW = colw < n ? colw / 3 : 1 << 20, offW = (colw - 3 * (colw / 3)) - 27 * (colw / 3);
    const int nk = (6 * N + 3) >> 2;
    if (LOWER) {
      switch (w) {
        case 0: gram_segment<NTT, 0>(acc, gram_kk0(0), nk, lq, N, jW, offW, jT, offT); break;
        case 1: if (NTT > 1) gram_segment<NTT, (NTT > 1 ? 1 : 0)>(acc, gram_kk0(1), nk, lq, N, jW, offW, jT, offT); break;
        case 2: if (NTT > 2) gram_segment<NTT, (NTT > 2 ? 2 : 0)>(acc, gram_kk0(2), nk, lq, N, jW, offW, jT, offT); break;
        case 3: if (NTT > 3) gram_segment<NTT, (NTT > 3 ? 3 : 0)>(acc, gram_kk0(3), nk, lq, N, jW, offW, jT, offT); break;
        case 4: if (NTT > 4) gram_segment<NTT, (NTT > 4 ? 4 : 0)>(acc, gram_kk0(4), nk, lq, N, jW, offW, jT, offT); break;
        default: if (NTT > 5) gram_segment<NTT, (NTT > 5 ? 5 : 0)>(acc, gram_kk0(5), nk, lq, N, jW, offW, jT, offT); break;
      }
    } else {
      GramSegments<NTT, NTT - 1>::run(acc, (6 * ((16 * w) / 3)) >> 2, nk, lq, N, jW, offW, jT, offT);
    }
#pragma unroll
    for (int J = 0; J < NTT; ++J)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int i = 16 * w + 4 * qq + lq, j = 16 * J + lc;
        if (F16_OWNS(J) && (!LOWER || J <= w) && i < n && j < n) {
          if (i == j) acc[J][qq] += s_Wcv[i] + s_Wrv[i] + s_Wrv[i + 3];
          else if (i == j + 3) acc[J][qq] -= s_Wrv[i];
          else if (j == i + 3) acc[J][qq] -= s_Wrv[j];
        }
      }
  }
}

// ---- One KKT factorisation: K = c P + sigma D^-2 + rho A'WA as matrix-core tiles (Gram product above; it does not depend
// on rho, so the first factorisation of a solve parks it in the workspace `gw` and the rho updates read it back), blocked
// sweep on the matrix cores (mfma_inverse), then the RE-LAYOUT for the iterations: lane q of DPP row blk receives
// K^-1[3 blk + c][6 q + cc] (c < 3 -- in the slot order of reduce3_slots: rows 2, 1, 0 in the upper half of a DPP row --,
// cc < 6; up to that order the layout of the cached P entries) through LDS, one tile row at a time, in
// the memory of the iteration vectors (dead during a factorisation).  rho <= 0 on entry: the builder's opt-in start value
// 2 sqrt(tr P / tr A'A) (no equilibration).  mrow: 18 values per lane.  Returns whether every pivot block was positive
// definite (uniform).
template <int NTT>
__device__ __noinline__ bool kkt_factorise(const double *Pg, double *gw, bool have_gram, int N, double cs, double sigma_unused,
                                           double *rho_io, double *mrow) {
  (void)sigma_unused;
  const int n = 3 * N;
  const LaneRole r = lane_role(N);
  const int lc = r.lc, lq = r.lq;
  const TileOwn own = tile_own<NTT>();
  const int w = own.tr;                                  // this wave's tile row; it holds tile columns [own.j0, own.j1)
  const bool has_tiles = own.j1 > 0;
  d4_t acc[NT], pp[NT];
#ifdef F16_EXP_STAMPM
  unsigned long long tF[5];
#define FSTAMP(i) { __builtin_amdgcn_s_waitcnt(0); tF[i] = __builtin_amdgcn_s_memtime(); }
#else
#define FSTAMP(i)
#endif
  FSTAMP(0)
#pragma unroll
  for (int J = 0; J < NTT; ++J) pp[J] = d4_t{0.0, 0.0, 0.0, 0.0};
  if (has_tiles) load_packed_tiles<NTT>(pp, Pg, n, w, lc, lq, own.j0, own.j1);   // P goes out first: its round trip hides under the Gram product
  double *const gwl = gw ? gw + (size_t)(w * NT * 4) * 64 + r.l : nullptr;      // tile (w, J), register qq at [(w NT + J) 4 + qq][lane]
  if (gwl && have_gram) {
#pragma unroll
    for (int J = 0; J < NTT; ++J)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) acc[J][qq] = F16_OWNS(J) ? gwl[(J * 4 + qq) * 64] : 0.0;
  } else {
#ifdef F16_EXP_GRAMSTAMP
    FSTAMP(0)
#endif
    if (gwl) {
      // With a Gram workspace (kept for the rho updates anyway): compute the tiles on and below the diagonal only -- balanced
      // over the waves --, park each tile AND its transpose, then every wave reads its whole tile row back (one L2 round trip).
      gram_tiles<NTT, true>(acc, N);
#pragma unroll
      for (int J = 0; J < NTT; ++J)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          if (F16_OWNS(J) && J <= w) {
            gwl[(J * 4 + qq) * 64] = acc[J][qq];
            // element (4 qq + lq, lc) of tile (w, J) = element (lc, 4 qq + lq) of tile (J, w): register lc / 4 of lane 16 (lc % 4) + 4 qq + lq
            if (J < w) gw[((size_t)(J * NT + w) * 4 + (lc >> 2)) * 64 + 16 * (lc & 3) + 4 * qq + lq] = acc[J][qq];
          }
      __syncthreads();                                      // (workgroup-scope visibility of the global stores)
#pragma unroll
      for (int J = 0; J < NTT; ++J)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) acc[J][qq] = F16_OWNS(J) ? gwl[(J * 4 + qq) * 64] : 0.0;
    } else {
      gram_tiles<NTT>(acc, N);
    }
#ifdef F16_EXP_GRAMSTAMP
    FSTAMP(1)
    if (blockIdx.x == 0 && threadIdx.x == 0) g_f_stamp[6] = (double)(tF[1] - tF[0]);
#endif
  }
  FSTAMP(1)
  double rho = *rho_io;
  if (!(rho > 0.0)) {   // the builder's opt-in start value (no equilibration): balance the two terms of P + rho A'A
    double tr[2] = {r.xown ? Pg[tri(r.xe, r.xe)] : 0.0, 0.0};
#pragma unroll
    for (int J = 0; J < NTT; ++J)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int i = 16 * w + 4 * qq + lq, j = 16 * J + lc;
        if (F16_OWNS(J) && i == j && i < n) tr[1] += acc[J][qq];
      }
    const bool sums[2] = {true, true};
    block_reduce<2>(tr, sums, s_red);
    rho = fmin(fmax(RHO_AUTO_SCALE * sqrt(tr[0] / tr[1]), OSQP_RHO_MIN), OSQP_RHO_MAX);
    *rho_io = rho;
  }
#pragma unroll
  for (int J = 0; J < NTT; ++J)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int i = 16 * w + 4 * qq + lq, j = 16 * J + lc;
      const double kij = (i < n && j < n) ? cs * pp[J][qq] + rho * acc[J][qq] + (i == j ? s_Dv[XOFF + (i < n ? i : 0)] : 0.0) : (i == j ? 1.0 : 0.0);
      acc[J][qq] = F16_OWNS(J) ? kij : 0.0;
    }
  FSTAMP(2)
  const bool ok = mfma_inverse<NTT>(acc, s_Cs, acc);
  FSTAMP(3)
  // re-layout (acc = MINUS the inverse, tile layout)
  double mr[3][6];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) mr[c][cc] = 0.0;
  for (int ww = 0; ww < NTT; ++ww) {
    __syncthreads();
    if (w == ww) {
#pragma unroll
      for (int J = 0; J < NTT; ++J)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          if (F16_OWNS(J)) Mst[(4 * qq + lq) * FN + 16 * J + lc] = -acc[J][qq];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int i = 3 * r.blk + (r.h ? 2 - c : c);          // slot order of reduce3_slots: the upper half of a row holds rows 2, 1, 0
      if ((i >> 4) == ww) {
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) mr[c][cc] = (6 * r.q + cc) < 16 * NTT ? Mst[(i & 15) * FN + 6 * r.q + cc] : 0.0;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) mrow[c * 6 + cc] = mr[c][cc];
  __syncthreads();                                          // Mst is read: its memory returns to the iteration vectors
  FSTAMP(4)
#ifdef F16_EXP_STAMPM
  if (blockIdx.x == 0 && threadIdx.x == 0) for (int i = 0; i < 4; ++i) g_f_stamp[i] = (double)(tF[i + 1] - tF[i]);
#endif
  return ok;
}

// ---- The iterations between two factorisations.
struct SolveState {
  double xs, z, y, dy;                 // this lane's variable (x owners) and constraint row: x, z, y = yb / E, last dy
  double rho, rp, rd;
  int it, to_check;
  int done, converged, infeasible;
};
struct IterSettings { double alpha, eps_abs, eps_rel, eps_prim_inf; int max_iter, check_every, rho_every, adaptive_rho; };

// One 512-lane workgroup per aircraft, three barriers per ADMM iteration:
//   A  stage 1 partials + row reduce -> rhs = sigma D^-2 x - c q + A' W (rho z - y)                | barrier
//   B  stage 2: x~ = K^-1 rhs, three rows x six columns per lane on all eight waves             | barrier
//   C  stage 3 partials + row reduce -> z~ = A x~ ; relaxation, projection, dual update, w = W (rho z - y) | barrier
// Lane roles inside DPP row blk (= horizon step): lanes 0,2,4,8,10,12 own the six kept state rows of step blk (that is
// where reduce6 leaves their z~), lanes 1,3,5 the command rows, 9,11,13 the rate rows, lanes 0,4,8 also own x[3 blk + c].
// Runs until the termination test is met, the problem is certified infeasible, max_iter is reached, or the rho estimate
// leaves the 5x band (then st->rho holds the new value and the caller re-factorises).  Returns 1 in the last case.
__device__ __noinline__ int admm_iterate(SolveState *st, const double *mrow_in, const double *Gg, int N, IterSettings o) {
  const LaneRole r = lane_role(N);
  const int blk = r.blk, q = r.q, kind = r.kind, k3 = r.k3, xe = r.xe, xec = r.xec, dup = r.dup;
  const bool h = r.h, g = r.g, e = r.e, xown = r.xown;
  const int lane = threadIdx.x;
  double mrow[3][6], Gd[2][6][3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) mrow[c][cc] = mrow_in[c * 6 + cc];
  {   // the lane's two Toeplitz blocks (utils.py:171-197: CC[i,j] = A^(i-j) B, rows S kept, slot order of reduce6_slots)
    constexpr int SR[6] = {2, 3, 4, 5, 6, 8};
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const int d = 2 * q + bb;
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) {
        const int srw = h ? SR[(rr + 3) % 6] : SR[rr];
#pragma unroll
        for (int c = 0; c < 3; ++c) Gd[bb][rr][c] = d < N ? Gg[d * 27 + srw * 3 + c] : 0.0;
      }
    }
  }
  double xs = st->xs, z = st->z, y = st->y, dy = st->dy, rho = st->rho, rp = st->rp, rd = st->rd;
  int it = st->it, to_check = st->to_check;
  bool done = false, converged = false, infeasible = false, refactor = false;
  // per-lane constants of the loop: bounds, row weight, this lane's entry of the rho vector
  const double lo = s_lc[lane], hi = s_lc[FT + lane], Wl = s_lc[2 * FT + lane], rho_o = rho * s_lc[3 * FT + lane], rinv = 1.0 / rho_o;
  const double sgl = s_sg2[xec], qcl = s_cq[xec];
  double *const wdst = w_slot(r, wsP, s_wc, s_wr);
  const double *const wsrc = wsP + WROW * (blk + 2 * q) + (h ? 3 : 0);
  // the iteration vectors (their memory carried the inverse during the re-layout): zero pads, then w of the current point
  for (int i = lane; i < 2 * WSP + 2 * XTP; i += FT) s_itv[i] = 0.0;
  __syncthreads();
  if (kind) { const double w0 = Wl * (rho_o * z - y); wdst[0] = w0; wdst[dup] = w0; }
  __syncthreads();
#ifdef F16_EXP_STAMPM
  unsigned long long tS[6] = {0, 0, 0, 0, 0, 0}, t0 = __builtin_amdgcn_s_memtime();
  const int it_in = it;
#define MSTAMP(i) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t1 = __builtin_amdgcn_s_memtime(); tS[i] += t1 - t0; t0 = t1; }
#else
#define MSTAMP(i)
#endif
  // The eight waves leave every barrier together and ask the one LDS pipe for their operands at once; it serves them in
  // wave order, so waves 4-7 start each phase ~300 clocks behind waves 0-3 and every barrier waits for them.  Raising their
  // issue priority takes ~1 % off a solve (measured A/B; alternating the priority per iteration was slower).
  if (r.w >= 4) __builtin_amdgcn_s_setprio(3);
  while (!done && !refactor) {
    ++it;
    // ---- A: rhs = sigma D^-2 x - c q + A' W (rho z - y)
    {
      double wv[12], o1[3];
      stage1_load(wsrc, wv);
      const double wce = s_wc[xec], wre = s_wr[xec], wrn = s_wr[xec + 3];
      MPC_PHASE();
      stage1_fma(Gd, wv, o1);
      const double t = reduce3(o1[0], o1[1], o1[2], h, g);
      if (xown) s_rhs[xe] = sgl * xs - qcl + (t + wce + (wre - wrn));
    }
    MSTAMP(0)
    __syncthreads();
    MSTAMP(1)
    // ---- B: x~ = K^-1 rhs, three rows x six columns per lane; reduce3 leaves x~[3 blk + c] in lane 4c of the row
    double xt_own;
    {
      double rj[6], p3[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) rj[cc] = s_rhs[6 * q + cc];
      MPC_PHASE();
#pragma unroll
      for (int cc = 0; cc < 6; ++cc)
#pragma unroll
        for (int c = 0; c < 3; ++c) p3[c] = fma(mrow[c][cc], rj[cc], p3[c]);
      xt_own = reduce3_slots(p3[0], p3[1], p3[2], g);
      if (xown) xtP[XOFF + xe] = xt_own;
    }
    MSTAMP(2)
    __syncthreads();
    MSTAMP(3)
    // ---- C: z~ = A x~, relaxation, projection, dual update (unscaled z, y = yb / E)
    {
      double xv[6], o3[6];
      stage3_load(xtP + XOFF + 3 * (blk - 2 * q - 1), xv);
      const double xk = xtP[XOFF + k3], xkm = xtP[XOFF + k3 - 3];
      MPC_PHASE();
      stage3_fma(Gd, xv, o3);
      const double zs = reduce6_slots(o3[0], o3[1], o3[2], o3[3], o3[4], o3[5], g, e);
      if (xown) xs = o.alpha * xt_own + (1 - o.alpha) * xs;
      if (kind) {
        const double zt = kind == 1 ? zs : (kind == 2 ? xk : xk - xkm);
        const double zr = o.alpha * zt + (1 - o.alpha) * z;
        const double zn = fmin(fmax(fma(y, rinv, zr), lo), hi);
        dy = rho_o * (zr - zn);
        y = y + dy;
        z = zn;
      }
    }
    const bool check = --to_check == 0 || it >= o.max_iter;      // it % check_every == 0, without the division
    if (to_check == 0) to_check = o.check_every;
    if (check) {
      // ---- residuals (OSQP termination test on the UNSCALED problem): A x, P x, A' W y / c.  x goes to its own zero-padded
      // buffer (x~ may still be read by slower waves), W y to the state-row layout; one barrier, then everything of the
      // test in one reduction (the two quantities of the primal-infeasibility certificate ride along)
      const double cinv = s_cinv;
      double *const ydst = w_slot(r, ysP, s_yc, s_yr);
      const double *const ysrc = ysP + WROW * (blk + 2 * q) + (h ? 3 : 0);
      if (kind) { const double ye = Wl * y; ydst[0] = ye; ydst[dup] = ye; }
      if (xown) xcP[XOFF + xe] = xs;
      __syncthreads();
      double o3[6], o1[3], px3[3] = {0.0, 0.0, 0.0};
      { double xv[6]; stage3_load(xcP + XOFF + 3 * (blk - 2 * q - 1), xv); stage3_fma(Gd, xv, o3); }
      const double axs = reduce6_slots(o3[0], o3[1], o3[2], o3[3], o3[4], o3[5], g, e);
      { double wv[12]; stage1_load(ysrc, wv); stage1_fma(Gd, wv, o1); }
      const double atys = reduce3(o1[0], o1[1], o1[2], h, g);
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) {                          // P x from the cached entries (zeros outside the matrix)
        const double xv = xcP[XOFF + 6 * q + cc];
#pragma unroll
        for (int c = 0; c < 3; ++c) px3[c] = fma(s_px[(cc * 3 + c) * FT + lane], xv, px3[c]);
      }
      const double px = reduce3(px3[0], px3[1], px3[2], h, g);
      const double qu = s_q[xec];
      double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};              // r1, |Ax|, |z|, r2, |Px|, |A'y|, |q|, |E dyb|, support(dyb)
      double ax = 0.0, aty = 0.0;
      if (kind) {
        ax = kind == 1 ? axs : (kind == 2 ? xcP[XOFF + k3] : xcP[XOFF + k3] - xcP[XOFF + k3 - 3]);
        v[0] = fabs(ax - z); v[1] = fabs(ax); v[2] = fabs(z);
        v[7] = Wl * fabs(dy); v[8] = Wl * (hi * fmax(dy, 0.0) + lo * fmin(dy, 0.0));
      }
      if (xown) {
        aty = cinv * (atys + s_yc[xe] + (s_yr[xe] - s_yr[xe + 3]));
        v[3] = fabs(px + qu + aty); v[4] = fabs(px); v[5] = fabs(aty); v[6] = fabs(qu);
      }
      const bool issum[9] = {false, false, false, false, false, false, false, false, true};
      block_reduce<9>(v, issum, s_red);
      rp = v[0]; rd = v[3];
      const double np_ = fmax(v[1], v[2]), nd_ = fmax(fmax(v[4], v[5]), v[6]);
      if (rp < o.eps_abs + o.eps_rel * np_ && rd < o.eps_abs + o.eps_rel * nd_) { done = true; converged = true; }
      else {
        // OSQP primal-infeasibility certificate on dy (auxil.c:is_primal_infeasible)
        const double ndy = v[7], supp = v[8];
        if (ndy > o.eps_prim_inf && supp < -o.eps_prim_inf * ndy) {
          if (kind) { const double de = Wl * dy; ydst[0] = de; ydst[dup] = de; }
          __syncthreads();
          { double wv[12]; stage1_load(ysrc, wv); stage1_fma(Gd, wv, o1); }
          const double t = reduce3(o1[0], o1[1], o1[2], h, g);
          double wv[1] = {xown ? fabs(t + s_yc[xe] + (s_yr[xe] - s_yr[xe + 3])) : 0.0};
          const bool km[1] = {false};
          block_reduce<1>(wv, km, s_red);
          if (wv[0] < o.eps_prim_inf * ndy) { done = true; infeasible = true; }
        }
        if (!done) {
          if (it >= o.max_iter) done = true;
          else if (o.adaptive_rho && it % o.rho_every == 0) {
            // auxil.c:compute_rho_estimate on the SCALED residuals: ||Ab xb - zb||, ||Pb xb + qb + Ab' yb|| and their norms
            const double cD = s_cD[xec], Er = sqrt(Wl);
            double sv[7] = {0, 0, 0, 0, 0, 0, 0};
            if (kind) { sv[0] = Er * fabs(ax - z); sv[1] = Er * fabs(ax); sv[2] = Er * fabs(z); }
            if (xown) { sv[3] = cD * fabs(px + qu + aty); sv[4] = cD * fabs(px); sv[5] = cD * fabs(aty); sv[6] = cD * fabs(qu); }
            const bool mx[7] = {false, false, false, false, false, false, false};
            block_reduce<7>(sv, mx, s_red);
            const double pr = sv[0] / (fmax(sv[2], sv[1]) + 1e-10), dr = sv[3] / (fmax(fmax(sv[6], sv[5]), sv[4]) + 1e-10);
            const double nw = fmin(fmax(rho * sqrt(pr / (dr + 1e-10)), OSQP_RHO_MIN), OSQP_RHO_MAX);
            if (nw > OSQP_ADAPTIVE_RHO_TOLERANCE * rho || nw < rho / OSQP_ADAPTIVE_RHO_TOLERANCE) { rho = nw; refactor = true; }
          }
        }
      }
    }
    // w = W (rho z - y) for the next iteration (after a rho update the next call rewrites it with the new rho)
    if (kind) { const double wn = Wl * (rho_o * z - y); wdst[0] = wn; wdst[dup] = wn; }
    MSTAMP(4)
    __syncthreads();
    MSTAMP(5)
  }
#ifdef F16_EXP_STAMPM
  if (blockIdx.x == 0 && (lane & 63) == 0) {
    for (int i = 0; i < 6; ++i) g_it_stamp[6 * (lane >> 6) + i] += (double)tS[i];
    if (lane == 0) g_f_stamp[5] += (double)(it - it_in);
  }
#endif
  __builtin_amdgcn_s_setprio(0);
  st->xs = xs; st->z = z; st->y = y; st->dy = dy; st->rho = rho; st->rp = rp; st->rd = rd;
  st->it = it; st->to_check = to_check;
  st->done = done; st->converged = converged; st->infeasible = infeasible;
  return refactor ? 1 : 0;
}

// The solve of one aircraft per workgroup (grid = B): prologue (loads, bounds, constants into LDS), equilibration, then
// factorise / iterate until done.  mode 0 one-shot; 1 prepare a plan without equilibration (factorise, keep the re-laid
// inverse and its rho, no iterations); 2 solve from such a plan.
template <int NTT>
__global__ __launch_bounds__(FT) void k_mpc_fast(MpcArgs a) {
  const int N = a.N, n = 3 * N;
  const LaneRole r = lane_role(N);
  const int tid = threadIdx.x, blk = r.blk, q = r.q, kind = r.kind, sub = r.sub, k3 = r.k3, xe = r.xe;
  const bool inb = r.inb, xown = r.xown, srow = r.srow;
  // zero what carries zero padding (the iteration vectors are zeroed by every admm_iterate call)
  for (int i = tid; i < XTP; i += FT) s_Dv[i] = 0.0;
  for (int i = tid; i < FN + 4; i += FT) { s_wr[i] = 0.0; s_yr[i] = 0.0; s_Erv[i] = 0.0; s_Wrv[i] = 0.0; }
  for (int i = tid; i < FN; i += FT) { s_rhs[i] = 0.0; s_wc[i] = 0.0; s_yc[i] = 0.0; s_Ecv[i] = 0.0; s_Wcv[i] = 0.0; s_nPm[i] = 0.0;
                                       s_sg2[i] = 0.0; s_cq[i] = 0.0; s_q[i] = 0.0; s_cD[i] = 0.0; }
  for (int i = tid; i < ES9; i += FT) s_Es9[i] = 0.0;
  for (int i = tid; i < WGN; i += FT) s_Wg[i] = 0.0;

  // one aircraft per workgroup.  readfirstlane: the index is uniform, so every per-aircraft pointer below lives in scalar
  // registers instead of a VGPR pair each
  const long b = a.order ? (long)__builtin_amdgcn_readfirstlane(a.order[blockIdx.x]) : (long)blockIdx.x;
  if (a.mode != 1 && a.mode != 3 && mpc_job_nonfinite(a.ext, a.N, b)) { mpc_write_nonfinite(a.ucmd, a.useq, a.info, a.iters_out, a.status, a.ld, a.N, a.s.rho, b, tid, FT); return; }      // (workgroup-uniform)
  double *const exw = a.ext + (size_t)b * mpc_ext_doubles(N);
  const double *Pg = a.Ppk + (size_t)b * (n * (n + 1) / 2);
  const double *Gg = exw + n, *pred = exw + n + 27 * N;
  double *const exm = exw + mpc_ext_model(N);                 // A | Q | Qbar | rho, ok of a prepared plan
  for (int i = tid; i < 27 * (N + 1); i += FT) s_Gl[i] = i < 27 * N ? Gg[i] : 0.0;      // (block N: what rows beyond 6N gather)
  const double qe = xown ? exw[xe] : 0.0;
  // the termination test needs P x: row blk's 16 lanes split the columns six apiece, so a lane touches the SAME 18
  // entries of P at every test -- fetched once here into per-lane LDS slots
#pragma unroll
  for (int cc = 0; cc < 6; ++cc) {
    const int col = 6 * q + cc;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int row = 3 * blk + c;
      s_px[(cc * 3 + c) * FT + tid] = (inb && col < n) ? Pg[row >= col ? tri(row, col) : tri(col, row)] : 0.0;
    }
  }
  double *const tl = a.tiles ? a.tiles + (size_t)b * MPC_TILE_DOUBLES + tid : nullptr;      // [18][FT]: the re-laid inverse of a plan
  double mrow[18];
  if (a.mode == 2) {
#pragma unroll
    for (int k = 0; k < 18; ++k) mrow[k] = tl[k * FT];
  }
  // ---- equilibration: De = D of this lane's variable (x owners), Eo = E of this lane's constraint row, cs = c
  // (a cached plan -- modes 1 and 2 -- exists only without equilibration: D = E = c = 1 there)
  double De = 1.0, Eo = 1.0, cs = 1.0;
  if (xown) s_Dv[XOFF + xe] = 1.0;
  if (srow) s_Es9[9 * blk + r.r9] = 1.0;
  if (kind == 2) s_Ecv[k3] = 1.0;
  if (kind == 3) s_Erv[k3] = 1.0;
  __syncthreads();                                            // zeros, G, D = E = 1 are in place
#ifdef F16_EXP_STAMPM
  if (blockIdx.x == 0) { if (tid < 48) g_it_stamp[tid] = 0.0; if (tid < 8) g_f_stamp[tid] = 0.0; }
  const unsigned long long tE0 = __builtin_amdgcn_s_memtime();
#endif
  if ((a.mode == 0 || a.mode == 3) && a.s.scaling > 0) {
    double o3[3];
    ruiz_equilibrate<NTT>(Pg, N, a.s.scaling, qe, o3);
    De = o3[0]; Eo = o3[1]; cs = o3[2];
  }
  if (a.mode == 3) {   // equilibration only: D | E of the kept state rows | E command | E rate | c for the wavefront solver
    double *const sc = a.gramws + (size_t)b * MPC_TILE_DOUBLES + WAVE_SCAL_OFF;
    if (xown) sc[xe] = De;
    if (kind == 1) sc[96 + 6 * blk + sub] = Eo;
    else if (kind == 2) sc[288 + k3] = Eo;
    else if (kind == 3) sc[384 + k3] = Eo;
    if (tid == 0) sc[480] = cs;
    return;
  }
#ifdef F16_EXP_STAMPM
  if (blockIdx.x == 0 && tid == 0) g_f_stamp[4] = (double)(__builtin_amdgcn_s_memtime() - tE0);
#endif
  if (kind == 0) Eo = 1.0;                                     // (lanes that stood in for an unbounded row)
  // ---- bounds of this lane's constraint row (utils.py:129-152; rows with two infinite bounds are not kept)
  double lo = 0.0, hi = 0.0;
  if (kind == 1) {
    const double pm = pred[blk * 9 + SROW[sub]];
    lo = a.pb.slb[sub] - pm; hi = a.pb.sub[sub] - pm;
  } else if (kind == 2) {
    lo = a.pb.ulb[sub]; hi = a.pb.uub[sub];
  } else if (kind == 3) {
    if (blk == 0) {
      const double act = a.x ? a.x[(13 + sub) * a.ld + b] : 0.0;      // (no state when a plan is prepared)
      lo = act + a.pb.rlb[sub] * a.dt; hi = act + a.pb.rub[sub] * a.dt;
    } else { lo = a.pb.rlb[sub]; hi = a.pb.rub[sub]; }               // reference quirk: not scaled by dt (utils.py:151-152)
  }
  // rho vector (osqp auxil.c:set_rho_vec): an equality row (E (u - l) < 1e-4, i.e. after scaling) carries 1e3 rho
  const double eqf = (kind && Eo * (hi - lo) < OSQP_RHO_TOL) ? OSQP_RHO_EQ_OVER_RHO_INEQ : 1.0;
  const double Wrow = kind ? Eo * Eo : 0.0;                    // the row weight E^2: all the iterations see of E
  if (kind == 1) s_Wg[6 * blk + sub] = Wrow * eqf;
  else if (kind == 2) s_Wcv[k3] = Wrow * eqf;
  else if (kind == 3) s_Wrv[k3] = Wrow * eqf;
  s_lc[tid] = lo; s_lc[FT + tid] = hi; s_lc[2 * FT + tid] = Wrow; s_lc[3 * FT + tid] = eqf;
  if (xown) {
    const double sg2 = a.s.sigma / (De * De);
    s_Dv[XOFF + xe] = sg2;                                     // s_Dv now holds sigma D^-2 for the KKT diagonal
    s_sg2[xe] = sg2; s_cq[xe] = cs * qe; s_q[xe] = qe; s_cD[xe] = cs * De;
  }
  if (tid == 0) s_cinv = 1.0 / cs;
  SolveState st;
  st.xs = 0.0; st.z = 0.0; st.y = 0.0; st.dy = 0.0; st.rp = INFINITY; st.rd = INFINITY;
  st.it = 0; st.to_check = a.s.check_every > 0 ? a.s.check_every : 1;
  st.done = 0; st.converged = 0; st.infeasible = 0;
  double *const wm = a.warm ? a.warm + (size_t)b * MPC_WARM_DOUBLES + tid : nullptr;
  if (wm && a.warm_load) {   // warm start: x, z, y of the previous solve of this plan, kept UNscaled (the equilibration of an
    const double x0 = wm[0], z0 = wm[FT], y0 = wm[2 * FT];      // OSQP-default plan is redone per solve: it depends on q)
    if (isfinite(x0) && isfinite(z0) && isfinite(y0)) { st.xs = x0; st.z = z0; st.y = kind ? cs * y0 / Wrow : 0.0; }
  }
  st.rho = a.mode == 2 ? exm[243] : a.s.rho;
  bool ok = true;
  if (a.mode == 2 && !(exm[244] > 0.5)) ok = false;
  __syncthreads();                                            // weights / sigma D^-2 / lane constants are in place
  double *const gw = a.gramws ? a.gramws + (size_t)b * MPC_TILE_DOUBLES : nullptr;
  IterSettings o;
  o.alpha = a.s.alpha; o.eps_abs = a.s.eps_abs; o.eps_rel = a.s.eps_rel; o.eps_prim_inf = a.s.eps_prim_inf;
  o.max_iter = a.s.max_iter; o.check_every = a.s.check_every; o.rho_every = a.s.rho_every; o.adaptive_rho = a.s.adaptive_rho;
  bool have_inverse = a.mode == 2, have_gram = false;
  bool done = a.s.max_iter < 0;                               // (max_iter == 0: factor only, used for timing)
  while (!done) {
    if (!have_inverse) {
      ok = kkt_factorise<NTT>(Pg, gw, have_gram, N, cs, 0.0, &st.rho, mrow) && ok;
      have_gram = true;
      if (a.mode == 1) {                                      // prepare: keep the inverse and its rho, no iterations
#pragma unroll
        for (int k = 0; k < 18; ++k) tl[k * FT] = mrow[k];
        if (tid == 0) { exm[243] = st.rho; exm[244] = ok ? 1.0 : 0.0; }
        return;
      }
    }
    have_inverse = false;
    if (!ok || a.s.max_iter <= 0) break;
    if (!admm_iterate(&st, mrow, Gg, N, o)) done = true;
  }
  const bool converged = st.converged != 0, infeasible = st.infeasible != 0;
  // res.x[0:3] (env.py:424); OSQP hands back NaN for a problem it certifies infeasible
  if (wm) {                                                  // keep the solution for the next warm start
    const bool good = converged && !infeasible;
    wm[0] = good ? st.xs : NAN; wm[FT] = good ? st.z : NAN; wm[2 * FT] = good ? Wrow * st.y / cs : NAN;
  }
  if (xown) {
    if (xe < 3) a.ucmd[xe * a.ld + b] = infeasible ? NAN : st.xs;
    if (a.useq) a.useq[xe * a.ld + b] = infeasible ? NAN : st.xs;
  }
  if (tid == 0) {
    if (a.iters_out) a.iters_out[b] = st.it;
    if (a.info) {
      a.info[0 * a.ld + b] = (double)st.it;
      a.info[1 * a.ld + b] = st.rp;
      a.info[2 * a.ld + b] = st.rd;
      a.info[3 * a.ld + b] = st.rho;
    }
    if (a.status && infeasible) a.status[b] |= F16_ST_QP_INFEASIBLE;
    else if (a.status && a.s.max_iter > 0 && (!converged || !ok)) a.status[b] |= F16_ST_QP_MAXITER;
  }
#ifdef F16_EXP_STAMPM
  __syncthreads();      // diagnostic build: the u_seq column of the aircraft solved by workgroup 0 is replaced by the stamps
  if (blockIdx.x == 0 && a.useq) {
    if (tid < 48) a.useq[tid * a.ld + b] = g_it_stamp[tid] / fmax(g_f_stamp[5], 1.0);
    if (tid < 8) a.useq[(66 + tid) * a.ld + b] = g_f_stamp[tid];
    if (tid < 16) a.useq[(50 + tid) * a.ld + b] = g_inv_stamp[tid];
  }
#endif
}
#undef wsP
#undef ysP
#undef xtP
#undef xcP
#undef Mst

// Dispatch order of a plan's next solve: aircraft sorted by the iteration count of the previous solve, longest first
// (counting sort over the termination-test buckets; one workgroup).  The hardware hands workgroups to CUs in index
// order as CUs free up, so mixed 25 / 50 / 75-iteration solves in arbitrary order leave a ragged tail: at B = 8192 a cold
// plan solve takes 2.56 ms in the caller's order, 2.27 ms longest-first (the balanced bound is ~2.1 ms).  The state of a
// closed loop moves little per step, so the previous counts predict the next ones.  Results do not depend on the order.
__global__ __launch_bounds__(1024) void k_plan_order(const int32_t *iters, int32_t *order, long B, int check_every) {
  constexpr int NB = 256;
  __shared__ int cnt[NB], base[NB];
  const int t = threadIdx.x;
  for (int i = t; i < NB; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  auto bucket = [&](int it) { const int k = (it + check_every - 1) / check_every; return NB - 1 - (k < NB - 1 ? (k < 0 ? 0 : k) : NB - 1); };
  for (long i = t; i < B; i += blockDim.x) atomicAdd(&cnt[bucket(iters[i])], 1);
  __syncthreads();
  if (t == 0) { int s = 0; for (int k = 0; k < NB; ++k) { base[k] = s; s += cnt[k]; } }     // bucket 0 = longest
  __syncthreads();
  for (long i = t; i < B; i += blockDim.x) order[atomicAdd(&base[bucket(iters[i])], 1)] = (int32_t)i;
}

// Dispatch order of a FIRST call (no iteration counts of a previous one): the aircraft by decreasing ||q||_inf of the QP the
// build kernel has just left in the workspace.  On the config-4 workload the correlation of log ||q||_inf with the iteration
// count OSQP's rules need is 0.66 -- enough for a longest-first order to recover four fifths of what the true counts would give
// (list-scheduling the measured counts on 1024 slots: 8.8 ms in a spread order, 7.3 ms by this score, 6.9 ms by the true counts).
// Scheduling only: results do not depend on the order.  score = 8 log2 ||q||_inf + 128, clamped to the 256 buckets of k_plan_order.
// (one wavefront per aircraft, four per workgroup: q is read with coalesced loads -- 45 us -> a few us per 4096)
__global__ __launch_bounds__(256) void k_first_order_score(const double *ext, size_t stride, int n, int32_t *score, long B) {
  const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int l = threadIdx.x & 63;
  if (b >= B) return;
  const double *q = ext + (size_t)b * stride;
  double m = 0.0;
  for (int i = l; i < n; i += 64) m = fmax(m, fabs(q[i]));      // (NaN entries are skipped by fmax: such a QP scores by its other entries)
  m = wave_reduce_dpp<false>(m);
  if (l == 0) {
    const int sc = m > 0.0 && isfinite(m) ? (int)floor(8.0 * log2(m)) + 128 : 0;
    score[b] = sc < 0 ? 0 : (sc > 255 ? 255 : sc);
  }
}
int mpc_first_order_launch(const MpcArgs &a, int32_t *scratch, int32_t *order, void *stream) {
  if (!a.ext || !scratch || !order) return F16_OK;
  hipLaunchKernelGGL(k_first_order_score, dim3((unsigned)((a.B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a.ext, mpc_ext_doubles(a.N),
                     3 * a.N, scratch, a.B);
  hipLaunchKernelGGL(k_plan_order, dim3(1), dim3(1024), 0, (hipStream_t)stream, scratch, order, a.B, 1);
  return hip_check(hipGetLastError(), "f16_mpc first-call order launch");
}

int mpc_plan_order_launch(const int32_t *iters, int32_t *order, long B, int check_every, void *stream) {
  hipLaunchKernelGGL(k_plan_order, dim3(1), dim3(1024), 0, (hipStream_t)stream, iters, order, B, check_every);
  return hip_check(hipGetLastError(), "f16_mpc_plan order launch");
}

int mpc_fast_solve_launch(f16_ctx *ctx, const MpcArgs &a, void *stream) {
  (void)ctx;
  if (a.N < 1 || a.N > FAST_MAXN) return set_error(F16_EINVAL, "fast MPC solver needs 1 <= N <= 32");
  // one aircraft per workgroup; the hardware queue balances the unequal iteration counts
  if (a.B > 0x7fffffffL) return set_error(F16_EINVAL, "batch too large for one launch");
  const unsigned grid = (unsigned)a.B;
  const int nt = (3 * a.N + 15) / 16;       // 16x16 tiles per side of the KKT matrix, instantiated for 2 / 4 / 6
  if (nt <= 2) hipLaunchKernelGGL(k_mpc_fast<2>, dim3(grid), dim3(FT), 0, (hipStream_t)stream, a);
  else if (nt <= 4) hipLaunchKernelGGL(k_mpc_fast<4>, dim3(grid), dim3(FT), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_mpc_fast<6>, dim3(grid), dim3(FT), 0, (hipStream_t)stream, a);
  return hip_check(hipGetLastError(), "f16_mpc_batch solve launch");
}

}  // namespace f16

extern "C" int f16_debug_spd_inverse(f16_ctx *ctx, const double *packed, double *out, int n, long B, void *stream) {
  using namespace f16;
  if (!ctx || !packed || !out || n < 1 || n > FN || B < 0) return set_error(F16_EINVAL, "bad argument");
  if (B == 0) return F16_OK;
  hipLaunchKernelGGL(k_dbg_inverse, dim3((unsigned)(B < 512 ? B : 512)), dim3(FT), 0, (hipStream_t)stream, packed, out, n, B);
  return hip_check(hipGetLastError(), "f16_debug_spd_inverse launch");
}
