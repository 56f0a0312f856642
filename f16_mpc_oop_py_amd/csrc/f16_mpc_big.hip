// f16_mpc_big.hip -- the OSQP solve of the condensed MPC QP (env.py:420-424) for LONG horizons, 33 <= N <= 150 (the reference
// sweeps N = 1..150 on the same states, env.py:426-436): one 512-lane WORKGROUP per aircraft.
//
// Same rules as the other solvers (f16_control.hip: k_mpc; f16_mpc_solve.hip; f16_mpc_wave.hip): Ruiz equilibration on the
// original entries and the running D, E, c; rho vector; x unscaled, zb / yb scaled; termination on the unscaled residuals; rho
// estimate on the scaled ones; primal-infeasibility certificate.  What changes is where things live: at N = 150 the KKT
// matrix is 450 x 450 (1.6 MB), far beyond registers and LDS, so
//   * the Gram matrix A'WA (it does not depend on rho) and the KKT matrix / its inverse are PACKED lower triangles in the
//     per-aircraft HBM workspace; the inverse is then re-laid as a padded HALF (rows padded to whole 64-column trips, the
//     diagonal halved) that serves x~ = K^-1 rhs with every entry read once: 0.93 MB per product at N = 150 -- a CU takes in
//     ~33 GB/s from the Infinity Cache (MI355X_MICROARCH.md, gather table), and that stream is the floor of an iteration;
//   * the inverse is the symmetric sweep, eight pivots per pass over the matrix (sweep_inverse_blocked), in its own function;
//   * the two block-Toeplitz products of an iteration run with lanes across the horizon steps and the blocks G_d as uniform
//     16-byte reads of an LDS copy (tp_partials);
//   * vectors (q, pred, x, rhs, E, D, the row vector w) live in LDS, the per-row values of a lane's (up to four) constraint
//     rows in registers.
// A sweep over horizons is ONE launch: resident workgroups take (horizon, aircraft) pairs from a work queue, longest horizons
// first (SweepArgs).  History: the one-wavefront kernel this replaced for N > 32 (k_mpc<false, true>) took 2.2 s for four
// aircraft at N = 150; the first workgroup version 2.15 s for 64 (80 us per iteration: full inverse streamed, both Toeplitz
// operands from LDS, one pass over the matrix per pivot) and 61 s for the reference's sweep at B = 64, a launch per horizon.
#include <hip/hip_runtime.h>
#include <math.h>

#include <mutex>

#include "f16_mpc.hpp"
#include "f16_smallmat.hpp"

namespace f16 {
namespace big {

#ifndef F16_BIG_BLK
#define F16_BIG_BLK 512
#endif
constexpr int BLK = F16_BIG_BLK, NW = BLK / 64;
constexpr int TM = (12 * BIG_MAXN + BLK - 1) / BLK;          // constraint rows per lane (4 at 512 lanes)

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

// LDS scratch shared by the three products of an iteration: [NW][n] column sums of the symmetric product, or the partial sums of
// a Toeplitz stage (at most NW + 2 wavefront segments x 64 steps x 6 rows)
constexpr int TP_SLOTS = NW + 2, TP_FWD = 6 * 64, TP_ADJ = 3 * 64;
constexpr int SWB = 8;                                   // pivots per block step of the KKT sweep
__host__ __device__ inline size_t part_doubles(int N) {
  const size_t a = (size_t)(NW > 2 * SWB ? NW : 2 * SWB) * 3 * N, b = (size_t)TP_SLOTS * TP_FWD;
  return a > b ? a : b;
}
// The prediction blocks G_d (9 state rows x 3 inputs) in LDS: GS doubles per lag, the six rows that carry bounds (SROW) first and
// in SROW's order -- 18 doubles, 16-byte aligned: what the Toeplitz stages read -- then rows 0, 1, 7 (equilibration only), one pad.
constexpr int GS = 28;
__host__ __device__ constexpr int gslot(int r) { return r == 0 ? 6 : (r == 1 ? 7 : (r == 7 ? 8 : (r == 8 ? 5 : r - 2))); }
struct Lds {
  double *G, *qv, *pred, *wbuf, *xs, *xt, *rhs, *tv, *Dg, *E9, *Ec, *Er, *cvec, *red, *part, *zpad;
};
__host__ __device__ inline size_t lds_doubles(int N) {
  const size_t n = 3 * (size_t)N, m = 12 * (size_t)N;
  auto ev = [](size_t v) { return (v + 1) & ~(size_t)1; };
  return ev(GS * N) + ev(n) + ev(9 * N) + ev(m) + 5 * ev(n) + ev(9 * N) + ev(n) + ev(n + 3) + ev(n) + ev(NW) + ev(part_doubles(N)) + 8;
}
__device__ __forceinline__ Lds carve(double *p, int N) {
  const int n = 3 * N, m = 12 * N;
  auto take = [&](int k) { double *r = p; p += (k + 1) & ~1; return r; };
  Lds L;
  L.G = take(GS * N); L.qv = take(n); L.pred = take(9 * N); L.wbuf = take(m);
  L.xs = take(n); L.xt = take(n); L.rhs = take(n); L.tv = take(n); L.Dg = take(n);
  L.E9 = take(9 * N); L.Ec = take(n); L.Er = take(n + 3); L.cvec = take(n); L.red = take(NW); L.part = take((int)part_doubles(N)); L.zpad = take(8);
  return L;
}

// ---- The two block-Toeplitz products of an iteration, z_i = sum_{j<=i} G_{i-j} u_j (rows SROW) and its adjoint, with LANES
// ACROSS THE STEPS: at lag d every lane of a wavefront needs the same block G_d -- nine 16-byte LDS reads of one address (the
// kept rows of the LDS copy; -DF16_TP_SMEM takes it through the scalar cache from the QP workspace instead: s_load into SGPRs,
// which v_fma_f64 reads directly -- as fast with 64 aircraft, four times slower with every CU streaming) -- and its own
// operand step, one LDS read of 3 (forward) or 6 (adjoint) doubles.  The (step block of 64, lag) pairs of the triangle are dealt
// to the NW wavefronts in equal contiguous shares (a share spans at most two step blocks: TP_SLOTS partial-sum slots); the
// partial sums go to LDS and are added in slot order by whoever consumes them.  (The first version read both operands of every
// product from LDS: 18 N^2 eight-byte reads per stage, 3.2 MB at N = 150 -- 27 k + 49 k cycles of a 185 k-cycle iteration.)
typedef const double __attribute__((address_space(4))) *cgptr_t;
#ifndef F16_TP_UNROLL
#define F16_TP_UNROLL 1
#endif
struct TpPlan { int lo[3], hi[3]; };                     // slots holding the partial sums of step block sb: lo[sb] .. hi[sb]
template <bool ADJ>
__device__ __forceinline__ int tp_count(int sb, int N) { return ADJ ? N - 64 * sb : (N < 64 * sb + 64 ? N : 64 * sb + 64); }
template <bool ADJ>
__device__ __forceinline__ TpPlan tp_plan(int N) {
  TpPlan P;
  const int nb = (N + 63) >> 6;
  int U = 0;
  for (int sb = 0; sb < nb; ++sb) U += tp_count<ADJ>(sb, N);
  int off = 0;
#pragma unroll
  for (int sb = 0; sb < 3; ++sb) {
    P.lo[sb] = 0; P.hi[sb] = -1;
    if (sb < nb) {
      const int c = tp_count<ADJ>(sb, N);
      int first = -1, last = -1;
      for (int w = 0; w < NW; ++w) {
        const int a0 = w * U / NW, a1 = (w + 1) * U / NW;
        if ((a0 > off ? a0 : off) < (a1 < off + c ? a1 : off + c)) { if (first < 0) first = w; last = w; }
      }
      P.lo[sb] = first + sb; P.hi[sb] = last + sb;
      off += c;
    }
  }
  return P;
}
#ifdef F16_TP_SMEM
#define F16_TP_G(d) cgptr_t g = Gc + 27 * (d); constexpr int GR[6] = {6, 9, 12, 15, 18, 24};      /* rows SROW of the workspace copy */
#else
// G_d from the LDS copy instead (18 doubles, nine 16-byte reads of one address): with every CU streaming its KKT inverse the
// scalar loads miss all the way to memory -- stages 24 k + 34 k cycles alone at N = 150, 50 k + 71 k with all 256 CUs busy
#define F16_TP_G(d) double g[18]; { const double2 *g2_ = reinterpret_cast<const double2 *>(Gl + GS * (d)); \
    _Pragma("unroll") for (int q_ = 0; q_ < 9; ++q_) { const double2 t_ = g2_[q_]; g[2 * q_] = t_.x; g[2 * q_ + 1] = t_.y; } } \
    constexpr int GR[6] = {0, 3, 6, 9, 12, 15};
#endif
template <bool ADJ>
__device__ __forceinline__ void tp_partials(double *part, cgptr_t Gc, const double *Gl, const double *vec, const double *zpad, int N) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
  const int nb = (N + 63) >> 6;
  int U = 0;
  for (int sb = 0; sb < nb; ++sb) U += tp_count<ADJ>(sb, N);
  const int u0 = w * U / NW, u1 = (w + 1) * U / NW;
  int off = 0;
  for (int sb = 0; sb < nb; ++sb) {
    const int c = tp_count<ADJ>(sb, N);
    const int d0 = (u0 > off ? u0 : off) - off, d1 = (u1 < off + c ? u1 : off + c) - off;
    off += c;
    if (d0 >= d1) continue;
    const int i = 64 * sb + l;
    double acc[ADJ ? 3 : 6];
#pragma unroll
    for (int k = 0; k < (ADJ ? 3 : 6); ++k) acc[k] = 0.0;
#pragma unroll F16_TP_UNROLL
    for (int d = d0; d < d1; ++d) {
      F16_TP_G(d)
      if (ADJ) {
        const int ii = i + d;
        const double2 *vp = reinterpret_cast<const double2 *>(ii < N ? vec + 6 * ii : zpad);
        const double2 a = vp[0], b = vp[1], e = vp[2];
        const double v[6] = {a.x, a.y, b.x, b.y, e.x, e.y};
#pragma unroll
        for (int rr = 0; rr < 6; ++rr)
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[k] = fma(g[GR[rr] + k], v[rr], acc[k]);
      } else {
        const int idx = i - d;
        const double *up = (idx >= 0 && i < N) ? vec + 3 * idx : zpad;
        const double x0 = up[0], x1 = up[1], x2 = up[2];
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) acc[rr] = fma(g[GR[rr] + 2], x2, fma(g[GR[rr] + 1], x1, fma(g[GR[rr]], x0, acc[rr])));
      }
    }
    double *out = part + (w + sb) * (ADJ ? TP_ADJ : TP_FWD) + l;
#pragma unroll
    for (int k = 0; k < (ADJ ? 3 : 6); ++k) out[64 * k] = acc[k];
  }
}
// element (step i, slot k) of a stage's result: the partial sums in slot order
template <bool ADJ>
__device__ __forceinline__ double tp_get(const double *part, const TpPlan &P, int i, int k) {
  const int sb = i >> 6;
  const int lo = sb == 0 ? P.lo[0] : (sb == 1 ? P.lo[1] : P.lo[2]), hi = sb == 0 ? P.hi[0] : (sb == 1 ? P.hi[1] : P.hi[2]);
  const double *p = part + k * 64 + (i & 63);
  double s = 0.0;
  for (int sl = lo; sl <= hi; ++sl) s += p[sl * (ADJ ? TP_ADJ : TP_FWD)];
  return s;
}

// ---- The symmetric HALF of a matrix as the iteration streams it (the inverse of the KKT matrix for x~ = K^-1 rhs, P for the
// termination test): row i of the lower triangle padded with zeros to whole 64-column trips (512-byte aligned rows), the
// diagonal entry HALVED, one 64-double block of zeros behind the last row.  A row then serves y_i += sum_j a_ij v_j and
// y_j += a_ij v_i with the same unmasked products (the diagonal contributes a_ii v_i / 2 to each), i.e. every entry is read
// ONCE per product: n^2 / 2 + 32 n doubles instead of n^2 (0.93 MB instead of 1.62 MB at N = 150).
__host__ __device__ inline size_t hoff(int i) {          // offset of row i
  const size_t q = (size_t)(i >> 6);
  return 2048 * q * (q + 1) + ((size_t)i - 64 * q) * (q + 1) * 64;
}
__host__ __device__ inline size_t half_doubles(int n) { return hoff(n) + 64; }
// H <- padded half of the packed symmetric matrix S (both in HBM); one wavefront per row
__device__ __forceinline__ void half_from_packed(double *H, const double *S, int n) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
  for (int i = w; i < n; i += NW) {
    const double *row = S + tri(i, 0);
    double *out = H + hoff(i);
    for (int j = l; j < 64 * ((i >> 6) + 1); j += 64) out[j] = j < i ? row[j] : (j == i ? 0.5 * row[j] : 0.0);
  }
  if (threadIdx.x < 64) H[hoff(n) + threadIdx.x] = 0.0;
}
// y = S v from the padded half H; v, y: LDS [n]; part: LDS [NW][n].  Wavefront w takes rows w, w + NW, ..., R at a time with
// every load of the R rows issued before the first product (R x KT independent 512-byte loads per wavefront in flight: the
// stream comes from the Infinity Cache or HBM at 1-3 us of latency); the column part accumulates in registers (lane l: columns
// l + 64 k) and is summed over the wavefronts in a fixed order at the end.  Ends with y complete and a barrier behind it.
// MAXABS: the same pass with (max, |a| x) in place of (+, x): y_j = max_i |S_ij| v_i, the diagonal at half weight (the column
// norms of the equilibration; the caller adds |S_jj| v_j).
// Loads are 16 bytes per lane: lanes 0..31 take one 64-column trip of a row (two columns each), lanes 32..63 the next one.  One CU
// streams 8-byte-per-lane loads at 25 / 50 / 76 GB/s (0.92 MB from the Infinity Cache / 0.43 MB and 0.92 MB from L2) and
// 16-byte ones at 45 / 127 / 150 GB/s (tools/micro/cu_stream.hip): the rate follows the load instructions, not the bytes in flight.
template <int KD, int R, bool MAXABS>                     // KD: pairs of trips covering n columns
__device__ __forceinline__ void half_symv_t(double *y, const double *H, const double *v, int n, double *part) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hh = l >> 5, m2 = 2 * (l & 31);
  const int zeros = (int)hoff(n) + m2;                    // (32-bit element offsets: one select per load, scalar base address)
  double vj[KD][2], acc[KD][2];
#pragma unroll
  for (int k = 0; k < KD; ++k)
#pragma unroll
    for (int c = 0; c < 2; ++c) { const int j = 128 * k + 64 * hh + m2 + c; vj[k][c] = j < n ? v[j] : 0.0; acc[k][c] = 0.0; }
  // two row groups in flight: the loads of the next group are issued before the products of this one
  auto load = [&](double2 (&a)[R][KD], int i0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ir = i0 + NW * r, irc = ir < n ? ir : i0, kr = irc >> 6;
      const int row = (int)hoff(irc) + 64 * hh + m2;
#pragma unroll
      for (int k = 0; k < KD; ++k) a[r][k] = *reinterpret_cast<const double2 *>(H + (2 * k + hh <= kr ? row + 128 * k : zeros));
    }
  };
  auto products = [&](const double2 (&a)[R][KD], int i0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ir = i0 + NW * r;
      const double vi = ir < n ? v[ir] : 0.0;
      double dot = 0.0;
#pragma unroll
      for (int k = 0; k < KD; ++k) {
        const double e[2] = {a[r][k].x, a[r][k].y};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (MAXABS) { const double aa = fabs(e[c]); dot = fmax(dot, aa * vj[k][c]); acc[k][c] = fmax(acc[k][c], aa * vi); }
          else { dot = fma(e[c], vj[k][c], dot); acc[k][c] = fma(e[c], vi, acc[k][c]); }
        }
      }
      dot = wave_reduce_dpp<!MAXABS>(dot);
      if (l == 0 && ir < n) y[ir] = dot;
    }
  };
  constexpr int STEP = NW * R;
  double2 bufA[R][KD], bufB[R][KD];
  if (w < n) load(bufA, w);
  for (int i0 = w; i0 < n; i0 += 2 * STEP) {
    const bool second = i0 + STEP < n;
    if (second) load(bufB, i0 + STEP);
    products(bufA, i0);
    if (second) {
      if (i0 + 2 * STEP < n) load(bufA, i0 + 2 * STEP);
      products(bufB, i0 + STEP);
    }
  }
#pragma unroll
  for (int k = 0; k < KD; ++k)
#pragma unroll
    for (int c = 0; c < 2; ++c) { const int j = 128 * k + 64 * hh + m2 + c; if (j < n) part[w * n + j] = acc[k][c]; }
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += BLK) {
    double s = y[j];
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) s = MAXABS ? fmax(s, part[ww * n + j]) : s + part[ww * n + j];
    y[j] = s;
  }
  __syncthreads();
}
template <bool MAXABS = false>
__device__ __forceinline__ void half_symv(double *y, const double *H, const double *v, int n, double *part) {
  switch ((n + 127) >> 7) {
    case 1: half_symv_t<1, 2, MAXABS>(y, H, v, n, part); break;
    case 2: half_symv_t<2, 2, MAXABS>(y, H, v, n, part); break;
    case 3: half_symv_t<3, 2, MAXABS>(y, H, v, n, part); break;
    default: half_symv_t<4, 2, MAXABS>(y, H, v, n, part); break;
  }
}
// In-place inverse of the packed SPD matrix S (HBM) by the symmetric sweep; c: LDS scratch [n].  Afterwards S = S^-1.
// One wavefront per row (lanes across the columns j <= i), the pivot column in LDS; returns false on a non-positive pivot.
__device__ __forceinline__ bool sweep_inverse_packed(double *S, int n, double *c) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
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

// In-place inverse of the packed SPD matrix S (HBM) by the symmetric sweep, SWB pivots per step.  Sweeping the pivot set K of a
// symmetric matrix [[A, B'], [B, C]] (A = S_KK) gives [[-A^-1, A^-1 B'], [B A^-1, C - B A^-1 B']] -- the composition of the
// scalar sweeps on its members -- so a step reads and writes the matrix ONCE for SWB pivots (the scalar version, one pass per
// pivot, spent 25 ms per factorisation at N = 150: 450 read-modify-write passes over 0.8 MB through one CU):
//   1. panel c_i = S_iK for every row i (LDS, p-major);  2. A^-1 by every wavefront for itself in registers (lane = entry of the
//   8 x 8 block, scalar sweeps over ds_bpermute / readlane);  3. T = c A^-1 (LDS, q-major);  4. S_ij -= T_i . c_j for i, j outside
//   K: one wavefront per row, lanes across the columns, the lane's c_j in registers for all rows of a column trip, four rows per
//   trip in flight;  5. S_iK = T_i, S_KK = -A^-1.
// cp, tp: LDS [SWB][n] each.  Afterwards S = S^-1; returns false on a non-positive pivot.
#ifdef F16_BIG_INLINE_SWEEP
#define F16_SWEEP_INLINE __forceinline__
#else
#define F16_SWEEP_INLINE __noinline__       // (its own register allocation: inlined, it pushed 180 spill reloads into every iteration)
#endif
__device__ F16_SWEEP_INLINE bool sweep_inverse_blocked(double *S_, int n, double *cp_, double *tp_) {
  // (a function of its own sees generic pointers: say where they point, or every LDS access becomes a flat one)
  typedef double __attribute__((address_space(3))) *lds_ptr_t;
  typedef double __attribute__((address_space(1))) *glb_ptr_t;
  const lds_ptr_t cp = (lds_ptr_t)cp_, tp = (lds_ptr_t)tp_;
  const glb_ptr_t S = (glb_ptr_t)S_;
  constexpr int RU = 8;                                   // rows of a wavefront in flight per column trip
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
  bool ok = true;
  for (int k0 = 0; k0 < n; k0 += SWB) {
    const int nb = n - k0 < SWB ? n - k0 : SWB;
    __syncthreads();                                      // the previous step's stores are done (block scope: also visible)
    // 1. the panel: rows above the block from the block's own rows (contiguous), rows from the block on from theirs (nb each)
    for (int e = threadIdx.x; e < nb * k0; e += BLK) {
      const int p = e / k0, i = e - p * k0;
      cp[p * n + i] = S[tri(k0 + p, i)];
    }
    for (int e = threadIdx.x; e < (n - k0) * nb; e += BLK) {
      const int i = k0 + e / nb, p = e - (i - k0) * nb, k = k0 + p;
      cp[p * n + i] = i >= k ? S[tri(i, k)] : S[tri(k, i)];
    }
    __syncthreads();
    // 2. A^-1 (every wavefront the same arithmetic): lane (r, c) of the 8 x 8 block, identity outside nb
    const int r = l >> 3, c = l & 7;
    double v = (r < nb && c < nb) ? cp[c * n + k0 + r] : (r == c ? 1.0 : 0.0);
    for (int p = 0; p < nb; ++p) {
      const double app = __shfl(v, 9 * p), arp = __shfl(v, 8 * r + p), apc = __shfl(v, 8 * p + c);
      if (!(app > 0.0)) ok = false;
      const double d = 1.0 / app;
      v = r == p ? (c == p ? -d : apc * d) : (c == p ? arp * d : v - arp * d * apc);
    }
    v = -v;                                               // A^-1 (rows / columns beyond nb: identity, never used)
    // 3. T = c A^-1
    for (int ib = 0; ib < n; ib += BLK) {                 // (whole wavefronts: the shuffles read lanes 0..63)
      const int ii = ib + threadIdx.x, i = ii < n ? ii : n - 1;
      double ci[SWB], ti[SWB];
#pragma unroll
      for (int p = 0; p < SWB; ++p) { ci[p] = p < nb ? cp[p * n + i] : 0.0; ti[p] = 0.0; }
#pragma unroll
      for (int p = 0; p < SWB; ++p)
#pragma unroll
        for (int q = 0; q < SWB; ++q) ti[q] = fma(ci[p], __shfl(v, 8 * p + q), ti[q]);
      if (ii < n) {
#pragma unroll
        for (int q = 0; q < SWB; ++q) tp[q * n + i] = ti[q];
      }
    }
    __syncthreads();
    // 4. the rows outside K
    for (int kt = 0; 64 * kt < n; ++kt) {
      const int j = l + 64 * kt;
      const bool jout = j < n && !(j >= k0 && j < k0 + nb);
      double cj[SWB];
#pragma unroll
      for (int p = 0; p < SWB; ++p) cj[p] = j < n ? cp[p * n + j] : 0.0;
      static_assert(64 % NW == 0, "rows w, w + NW, ... of a wavefront: the first one from 64 kt on is 64 kt + w");
      for (int i0 = 64 * kt + w; i0 < n; i0 += RU * NW) {
        double val[RU];
        bool on[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int i = i0 + u * NW;
          on[u] = i < n && !(i >= k0 && i < k0 + nb) && jout && j <= i;
          val[u] = S[on[u] ? tri(i, j) : 0];                 // (unconditional: a load under a lane condition becomes a branch)
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int i = i0 + u * NW < n ? i0 + u * NW : i0;
#pragma unroll
          for (int q = 0; q < SWB; ++q) val[u] = fma(-tp[q * n + i], cj[q], val[u]);
        }
#pragma unroll
        for (int u = 0; u < RU; ++u)
          if (on[u]) S[tri(i0 + u * NW, j)] = val[u];
      }
    }
    // 5. the block's rows and columns
    for (int e = threadIdx.x; e < nb * k0; e += BLK) {
      const int p = e / k0, i = e - p * k0;
      S[tri(k0 + p, i)] = tp[p * n + i];
    }
    for (int e = threadIdx.x; e < (n - k0 - nb) * nb; e += BLK) {
      const int i = k0 + nb + e / nb, q = e - (i - k0 - nb) * nb;
      S[tri(i, k0 + q)] = tp[q * n + i];
    }
    if (w == 0 && r < nb && c <= r) S[tri(k0 + r, k0 + c)] = -v;
    __threadfence_block();
  }
  __syncthreads();
  const int np = n * (n + 1) / 2;
  for (int e = threadIdx.x; e < np; e += BLK) S[e] = -S[e];
  __threadfence_block();
  __syncthreads();
  return __syncthreads_and(ok) != 0;
}

// The same blocked sweep on the PADDED layout (row i at hoff(i): 512-byte aligned, so the pass over the matrix moves 16 bytes
// per lane -- a CU streams those ~1.8 x faster than the 8-byte accesses the packed triangle forces, tools/micro/cu_stream.hip).
// Lanes 0..31 take one 64-column trip of a row (two columns each), lanes 32..63 the next; entries right of the diagonal inside a
// row's last trip are padding: computed along, never read as data.  S is left as MINUS the inverse (the copy into the streamed
// half negates: one pass over the matrix less).
__device__ __forceinline__ bool sweep_inverse_padded(double *S_, int n, double *cp_, double *tp_) {
  typedef double __attribute__((address_space(3))) *lds_ptr_t;
  typedef double __attribute__((address_space(1))) *glb_ptr_t;
  typedef double dbl2_t __attribute__((ext_vector_type(2)));
  typedef dbl2_t __attribute__((address_space(1))) *glb2_ptr_t;
  const lds_ptr_t cp = (lds_ptr_t)cp_, tp = (lds_ptr_t)tp_;
  const glb_ptr_t S = (glb_ptr_t)S_;
  constexpr int RU = 8;                                   // rows of a wavefront in flight per column trip
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hh = l >> 5, m2 = 2 * (l & 31);
  bool ok = true;
  for (int k0 = 0; k0 < n; k0 += SWB) {
    const int nb = n - k0 < SWB ? n - k0 : SWB;
    __syncthreads();                                      // the previous step's stores are done (block scope: also visible)
    // 1. the panel
    for (int e = threadIdx.x; e < nb * k0; e += BLK) {
      const int p = e / k0, i = e - p * k0;
      cp[p * n + i] = S[hoff(k0 + p) + i];
    }
    for (int e = threadIdx.x; e < (n - k0) * nb; e += BLK) {
      const int i = k0 + e / nb, p = e - (i - k0) * nb, k = k0 + p;
      cp[p * n + i] = i >= k ? S[hoff(i) + k] : S[hoff(k) + i];
    }
    __syncthreads();
    // 2. A^-1 (every wavefront the same arithmetic): lane (r, c) of the 8 x 8 block, identity outside nb
    const int r = l >> 3, c = l & 7;
    double v = (r < nb && c < nb) ? cp[c * n + k0 + r] : (r == c ? 1.0 : 0.0);
    for (int p = 0; p < nb; ++p) {
      const double app = __shfl(v, 9 * p), arp = __shfl(v, 8 * r + p), apc = __shfl(v, 8 * p + c);
      if (!(app > 0.0)) ok = false;
      const double d = 1.0 / app;
      v = r == p ? (c == p ? -d : apc * d) : (c == p ? arp * d : v - arp * d * apc);
    }
    v = -v;
    // 3. T = c A^-1
    for (int ib = 0; ib < n; ib += BLK) {                 // (whole wavefronts: the shuffles read lanes 0..63)
      const int ii = ib + threadIdx.x, i = ii < n ? ii : n - 1;
      double ci[SWB], ti[SWB];
#pragma unroll
      for (int p = 0; p < SWB; ++p) { ci[p] = p < nb ? cp[p * n + i] : 0.0; ti[p] = 0.0; }
#pragma unroll
      for (int p = 0; p < SWB; ++p)
#pragma unroll
        for (int q = 0; q < SWB; ++q) ti[q] = fma(ci[p], __shfl(v, 8 * p + q), ti[q]);
      if (ii < n) {
#pragma unroll
        for (int q = 0; q < SWB; ++q) tp[q * n + i] = ti[q];
      }
    }
    __syncthreads();
    // 4. the rows outside K (their entries in the block's columns are overwritten in 5)
    for (int kd = 0; 128 * kd < n; ++kd) {
      const int j = 128 * kd + 64 * hh + m2, trip = 2 * kd + hh;
      double cj[SWB][2];
#pragma unroll
      for (int p = 0; p < SWB; ++p) { cj[p][0] = j < n ? cp[p * n + j] : 0.0; cj[p][1] = j + 1 < n ? cp[p * n + j + 1] : 0.0; }
      for (int i0 = 128 * kd + w; i0 < n; i0 += RU * NW) {
        dbl2_t val[RU];
        bool on[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int i = i0 + u * NW;
          on[u] = i < n && !(i >= k0 && i < k0 + nb) && trip <= (i >> 6);
          val[u] = *(glb2_ptr_t)(S + (on[u] ? (int)hoff(i) + j : 0));          // (unconditional: no branch per load)
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int i = i0 + u * NW < n ? i0 + u * NW : i0;
#pragma unroll
          for (int q = 0; q < SWB; ++q) { const double t = -tp[q * n + i]; val[u].x = fma(t, cj[q][0], val[u].x); val[u].y = fma(t, cj[q][1], val[u].y); }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u)
          if (on[u]) *(glb2_ptr_t)(S + (int)hoff(i0 + u * NW) + j) = val[u];
      }
    }
    __syncthreads();                                      // (5 overwrites entries 4 has just updated: other wavefronts' rows)
    // 5. the block's rows and columns
    for (int e = threadIdx.x; e < nb * k0; e += BLK) {
      const int p = e / k0, i = e - p * k0;
      S[hoff(k0 + p) + i] = tp[p * n + i];
    }
    for (int e = threadIdx.x; e < (n - k0 - nb) * nb; e += BLK) {
      const int i = k0 + nb + e / nb, q = e - (i - k0 - nb) * nb;
      S[hoff(i) + k0 + q] = tp[q * n + i];
    }
    if (w == 0 && r < nb && c <= r) S[hoff(k0 + r) + k0 + c] = -v;
    __threadfence_block();
  }
  __syncthreads();
  return __syncthreads_and(ok) != 0;
}
// H <- the streamed half of MINUS the padded matrix S (what sweep_inverse_padded leaves): diagonal halved, padding zero
__device__ __forceinline__ void half_from_padded_neg(double *H, const double *S, int n) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
  for (int i = w; i < n; i += NW) {
    const double *row = S + hoff(i);
    double *out = H + hoff(i);
    for (int j = l; j < 64 * ((i >> 6) + 1); j += 64) out[j] = j < i ? -row[j] : (j == i ? -0.5 * row[j] : 0.0);
  }
  if (threadIdx.x < 64) H[hoff(n) + threadIdx.x] = 0.0;
}

// (c P + sigma D^-2 + r A'WA) -> its inverse as the streamed half Hinv; Minv: scratch for the padded matrix.  -DF16_BIG_PADDED_SWEEP
// only: measured, the factorisation itself is 28 % faster this way (6.9 M against 9.6 M cycles at N = 150), but with this code in
// the kernel the compiler keeps the per-row state of the lanes in scratch inside the iteration (w + projection 8 k -> 27 k cycles per
// iteration; the kernel spills ~300 scalar registers either way and small changes tip the vector allocation), a net loss:
// iteration 41 -> 54 us at N = 150.  The default stays the sweep on the packed triangle.
__device__ __noinline__ bool kkt_inverse(double *Minv, double *Hinv, const double *Pg_, const double *gram_, const double *Dg_, double cs,
                                         double r, int n, double *part) {
  typedef const double __attribute__((address_space(3))) *lds_cptr_t;
  typedef const double __attribute__((address_space(1))) *glb_cptr_t;
  typedef double __attribute__((address_space(1))) *glb_ptr_t;
  const lds_cptr_t Dg = (lds_cptr_t)Dg_;
  const glb_cptr_t Pg = (glb_cptr_t)Pg_, gram = (glb_cptr_t)gram_;
  const glb_ptr_t M = (glb_ptr_t)Minv;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), ll = threadIdx.x & 63;
  __syncthreads();
  for (int ia = w; ia < n; ia += NW)
    for (int ib = ll; ib <= ia; ib += 64) {
      const int e = tri(ia, ib);
      M[hoff(ia) + ib] = cs * Pg[e] + r * gram[e] + (ia == ib ? Dg[ia] : 0.0);
    }
  __threadfence_block();
  const bool good = sweep_inverse_padded(Minv, n, part, part + SWB * n);
  half_from_padded_neg(Hinv, Minv, n);
  __threadfence_block();
  __syncthreads();
  return good;
}

// per-aircraft HBM workspace of this solver (MpcArgs.bigws): Gram packed | K / minus its inverse, padded | padded half of the
// inverse | padded half of P
__host__ __device__ inline size_t ws_doubles(int N) {
  const size_t n = 3 * (size_t)N;
  return n * (n + 1) / 2 + 3 * half_doubles((int)n);
}

// A sweep over horizons lo..hi in ONE launch (env.py:426-436 solves the same states for every N): workgroup w solves aircraft
// w % B at horizon hi - w / B -- longest horizons first, so that the hardware's in-order dispatch packs the short solves
// behind the long ones.  Every horizon has its own workspace [B][np] P | [B][ext] | [B][ws] behind `base` (written by its build
// launch) and its own slice of the outputs ([hi - lo + 1][3][ld] commands, [..][4][ld] info, [..][ld] status words).
struct SweepArgs {
  int lo, hi;                 // hi = 0: not a sweep
  double *base;
  double *ucmd, *info;
  int32_t *status;
  unsigned int *next;         // work queue: the next (horizon, aircraft) pair not yet taken (zeroed before the launch)
  int32_t *iters;             // [pairs] iteration counts out (pair = (hi - N) * B + aircraft), or null
  const int32_t *order;       // [pairs] the pairs in queue order (a previous sweep's, costliest first), or null
};
__host__ __device__ inline size_t sweep_job_doubles(int N) {      // per aircraft
  const size_t n = 3 * (size_t)N;
  return n * (n + 1) / 2 + mpc_ext_doubles(N) + ws_doubles(N);
}

__global__ __launch_bounds__(BLK) void k_mpc_big(MpcArgs a, const SweepArgs sw) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int l = threadIdx.x;
  const long total = sw.hi ? (long)(sw.hi - sw.lo + 1) * a.B : a.B;
  __shared__ unsigned int s_next;
  for (long w = blockIdx.x;; w += gridDim.x) {
    if (sw.hi) {
      // The pairs of a sweep are TAKEN from a queue by as many workgroups as the chip holds, not dealt out by the launch: workgroup
      // ids go to the XCDs round-robin and are dispatched in order, so one XCD full of long solves stalls the dispatch for all
      // (measured: 200 of 256 CUs busy on average, the XCD of the hard aircraft 3 x longer than the rest before the rotation below)
      __syncthreads();
      if (l == 0) s_next = atomicAdd(sw.next, 1u);
      __syncthreads();
      w = s_next;
    }
    if (w >= total) break;
    long b = w;
    if (sw.hi) {
      const long pr = sw.order ? (long)__builtin_amdgcn_readfirstlane(sw.order[w]) : w;
      const int j = (int)(pr / a.B);
      b = sw.order ? pr - (long)j * a.B
                   : (pr - (long)j * a.B + j) % a.B;       // rotated per horizon (first version: workgroup ids dealt out by the
                                                           // launch went to the XCDs round-robin, and how hard an aircraft is
                                                           // repeats from horizon to horizon: three XCDs worked 3 x longer)
      a.N = sw.hi - j;
      size_t off = 0;
      for (int Nn = sw.hi; Nn > a.N; --Nn) off += sweep_job_doubles(Nn);
      const size_t npj = (size_t)(3 * a.N) * (3 * a.N + 1) / 2;
      a.Ppk = sw.base + off * (size_t)a.B;
      a.ext = a.Ppk + npj * (size_t)a.B;
      a.bigws = a.ext + mpc_ext_doubles(a.N) * (size_t)a.B;
      const size_t k = (size_t)(a.N - sw.lo);
      a.ucmd = sw.ucmd + k * 3 * (size_t)a.ld;
      a.info = sw.info ? sw.info + k * 4 * (size_t)a.ld : nullptr;
      a.status = sw.status ? sw.status + k * (size_t)a.ld : nullptr;
      a.useq = nullptr;
      a.iters_out = sw.iters ? sw.iters + (size_t)j * a.B : nullptr;
    }
#ifdef F16_EXP_STAMPG
    const unsigned long long wc0 = wall_clock64(), tjob0 = __builtin_amdgcn_s_memtime();
#endif
    if (mpc_job_nonfinite(a.ext, a.N, b)) { mpc_write_nonfinite(a.ucmd, a.useq, a.info, a.iters_out, a.status, a.ld, a.N, a.s.rho, b, l, BLK); __syncthreads(); continue; }      // (workgroup-uniform)
    const int N = a.N, n = 3 * N, np = n * (n + 1) / 2, ms = 6 * N, m = 12 * N;
    const Lds L = carve(smem, N);
    double *G = L.G, *qv = L.qv, *pred = L.pred, *wbuf = L.wbuf, *xs = L.xs, *xt = L.xt, *rhs = L.rhs, *tv = L.tv, *Dg = L.Dg,
           *E9 = L.E9, *Ec = L.Ec, *Er = L.Er;
    const double *exw = a.ext + (size_t)b * mpc_ext_doubles(N);
    const double *Pg = a.Ppk + (size_t)b * np;
    double *const gram = a.bigws + (size_t)b * ws_doubles(N), *const Minv = gram + np, *const Hinv = Minv + half_doubles(n),
                 *const HP = Hinv + half_doubles(n);
    __syncthreads();
    if (l < 8) L.zpad[l] = 0.0;
    const cgptr_t Gc = (cgptr_t)(exw + n);                  // G_k of the workspace (scalar-cache variant of the stages only)
    const TpPlan planF = tp_plan<false>(N), planA = tp_plan<true>(N);
    double *const part = L.part;
    for (int e = l; e < n; e += BLK) qv[e] = exw[e];
    for (int e = l; e < 27 * N; e += BLK) { const int d = e / 27, rc = e - 27 * d; G[d * GS + gslot(rc / 3) * 3 + rc % 3] = exw[n + e]; }
    for (int d = l; d < N; d += BLK) G[d * GS + 27] = 0.0;
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
        lo[t] = a.pb.slb[rr] - pm; hi[t] = a.pb.sub[rr] - pm;
      } else if (row < ms + n) {
        const int c = (row - ms) % 3;
        lo[t] = a.pb.ulb[c]; hi[t] = a.pb.uub[c];
      } else if (row < m) {
        const int k = row - ms - n, c = k % 3;
        if (k < 3) {
          const double act = a.x[(13 + c) * a.ld + b];
          lo[t] = act + a.pb.rlb[c] * a.dt; hi[t] = act + a.pb.rub[c] * a.dt;
        } else { lo[t] = a.pb.rlb[c]; hi[t] = a.pb.rub[c]; }          // reference quirk: not multiplied by dt (utils.py:151-152)
      }
    }
    // ---------------- equilibration (scaling.c:scale_data on the original entries and the running D, E, c)
    const double sigma = a.s.sigma, alpha = a.s.alpha;
    double cs = 1.0;
    for (int e = l; e < n; e += BLK) { Dg[e] = 1.0; Ec[e] = 1.0; Er[e] = 1.0; }
    for (int e = l; e < 3; e += BLK) Er[n + e] = 0.0;
    for (int e = l; e < 9 * N; e += BLK) E9[e] = 1.0;
    __syncthreads();
    // (the column norms of P D through the padded half of P: one coalesced pass per norm -- walking the packed triangle by
    // columns was 20 strided sweeps of 0.8 MB per solve, ~10 M cycles at N = 150; same maxima, bit for bit)
    half_from_packed(HP, Pg, n);
    __threadfence_block();
    __syncthreads();
    for (int pass = 0; pass < a.s.scaling; ++pass) {
      half_symv<true>(xs, HP, Dg, n, part);                // xs_e = max_i |P_ie| D_i (ends with a barrier)
      for (int e = l; e < n; e += BLK) {                   // column norms of [Pb; Ab]
        const int jb = e / 3, c = e - 3 * jb;
        double mp = fmax(xs[e], fabs(Pg[tri(e, e)]) * Dg[e]), ma = 0.0;
        for (int i = jb; i < N; ++i)
          for (int r = 0; r < 9; ++r) ma = fmax(ma, fabs(G[(i - jb) * GS + gslot(r) * 3 + c]) * E9[9 * i + r]);
        ma = fmax(fmax(ma, Ec[e]), fmax(Er[e], Er[e + 3]));
        tv[e] = 1.0 / sqrt(osqp_limit_scaling(Dg[e] * fmax(cs * mp, ma)));
      }
      for (int e = l; e < 9 * N; e += BLK) {               // row norms of the state block
        const int i = e / 9, r = e - 9 * i;
        double m_ = 0.0;
        for (int jb = 0; jb <= i; ++jb)
          for (int c = 0; c < 3; ++c) m_ = fmax(m_, fabs(G[(i - jb) * GS + gslot(r) * 3 + c]) * Dg[3 * jb + c]);
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
      half_symv<true>(xs, HP, Dg, n, part);
      for (int e = l; e < n; e += BLK) {
        const double mp = fmax(xs[e], fabs(Pg[tri(e, e)]) * Dg[e]);
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
    // ---------------- A'WA, packed, once (it does not depend on rho): one wavefront per BLOCK row (the three inputs of a step),
    // a lane per block column: nine entries share the operands of a step -- 42 LDS reads per step for 54 products, where an
    // entry of its own re-read 18 for 6 (the product was LDS-bound: 11 M of the 19 M set-up cycles at N = 150)
    {
      const int w = __builtin_amdgcn_readfirstlane(l >> 6), ll = l & 63;
      for (int ja = w; ja < N; ja += NW) {
        for (int jb = ll; jb <= ja; jb += 64) {
          double s[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
          for (int i = ja; i < N; ++i) {
            const double *ga = G + (i - ja) * GS, *gb = G + (i - jb) * GS, *wv = E9 + 9 * i;
#pragma unroll
            for (int rr = 0; rr < 6; ++rr) {                // (kept rows: slots 0..5; per entry the same order of sums as before)
              const double wr = wv[SROW[rr]];
#pragma unroll
              for (int ca = 0; ca < 3; ++ca) {
                const double wa = wr * ga[rr * 3 + ca];
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) s[ca][cb] += wa * gb[rr * 3 + cb];
              }
            }
          }
#pragma unroll
          for (int ca = 0; ca < 3; ++ca)
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
              const int ia = 3 * ja + ca, ib = 3 * jb + cb;
              if (ib <= ia) {
                double v = s[ca][cb];
                if (ia == ib) v += Ec[ia] + Er[ia] + Er[ia + 3];
                else if (ia == ib + 3) v -= Er[ia];
                gram[tri(ia, ib)] = v;
              }
            }
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
#ifdef F16_EXP_STAMPG
    unsigned long long tS[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    tS[10] = __builtin_amdgcn_s_memtime() - tjob0;        // set-up: equilibration, Gram product
#endif
    auto build_minv = [&](double r) {                     // Hinv <- padded half of (c P + sigma D^-2 + r A'WA)^-1
#ifdef F16_EXP_STAMPG
      const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
#endif
#ifndef F16_BIG_PADDED_SWEEP      // (default: the sweep on the packed triangle; see kkt_inverse)
      __syncthreads();
      {
        const int w = __builtin_amdgcn_readfirstlane(l >> 6), ll = l & 63;
        for (int ia = w; ia < n; ia += NW)
          for (int ib = ll; ib <= ia; ib += 64) {
            const int e = tri(ia, ib);
            Minv[e] = cs * Pg[e] + r * gram[e] + (ia == ib ? Dg[ia] : 0.0);
          }
      }
      __threadfence_block();
      const bool good = sweep_inverse_blocked(Minv, n, L.part, L.part + SWB * n);
      half_from_packed(Hinv, Minv, n);
      __threadfence_block();
      __syncthreads();
#else
      const bool good = kkt_inverse(Minv, Hinv, Pg, gram, Dg, cs, r, n, L.part);
#endif
#ifdef F16_EXP_STAMPG
      tS[8] += __builtin_amdgcn_s_memtime() - tb0; tS[9] += 1;
#endif
      return __syncthreads_and(good) != 0;
    };
    bool ok = build_minv(rho);
    for (int e = l; e < n; e += BLK) xs[e] = 0.0;
    __syncthreads();
    int it = 0;
    double rp = INFINITY, rd = INFINITY;
    bool converged = false, infeasible = false;
    bool done = !ok || a.s.max_iter <= 0;
    auto adjoint = [&](const double *wv_, int e) {        // (A' w)_e for w in the [6N | 3N | 3N] layout; CCs' w_s: partial sums
      return tp_get<true>(part, planA, e / 3, e % 3) + wv_[ms + e] + (wv_[ms + n + e] - (e + 3 < n ? wv_[ms + n + e + 3] : 0.0));
    };
#ifdef F16_EXP_STAMPG
    unsigned long long tq0 = __builtin_amdgcn_s_memtime();
#define GSTAMP(i) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long tq1 = __builtin_amdgcn_s_memtime(); tS[i] += tq1 - tq0; tq0 = tq1; }
#else
#define GSTAMP(i)
#endif
    while (!done) {
      ++it;
      GSTAMP(7)
      // w = E (rho zb - yb) -> t = A' w ; rhs = sigma D^-2 x - c q + t
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int row = l + BLK * t;
        if (row < m) wbuf[row] = Eo[t] * (rho * eqf[t] * z[t] - y[t]);
      }
      __syncthreads();
      GSTAMP(0)
      tp_partials<true>(part, Gc, G, wbuf, L.zpad, N);
      __syncthreads();
      GSTAMP(1)
      for (int e = l; e < n; e += BLK) rhs[e] = Dg[e] * xs[e] - cs * qv[e] + adjoint(wbuf, e);
      __syncthreads();
      GSTAMP(2)
      half_symv(xt, Hinv, rhs, n, L.part);                 // x~ (ends with a barrier)
      GSTAMP(3)
      tp_partials<false>(part, Gc, G, xt, L.zpad, N);
      __syncthreads();
      GSTAMP(5)
      // zb~ = E A x~ ; relaxation, projection, dual update
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int row = l + BLK * t;
        if (row < m) {
          double zt;
          if (row < ms) zt = tp_get<false>(part, planF, row / 6, row % 6);
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
      GSTAMP(4)
      if (it % a.s.check_every == 0 || it >= a.s.max_iter) {
        // residuals of the UNSCALED problem (OSQP termination test) + the scaled ones for the rho estimate
        double r1 = 0.0, nAx = 0.0, nz = 0.0, r1s = 0.0, nAxs = 0.0, nzs = 0.0;
        tp_partials<false>(part, Gc, G, xs, L.zpad, N);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const int row = l + BLK * t;
          if (row < m) {
            double ax;
            if (row < ms) ax = tp_get<false>(part, planF, row / 6, row % 6);
            else if (row < ms + n) ax = xs[row - ms];
            else { const int k = row - ms - n; ax = xs[k] - (k >= 3 ? xs[k - 3] : 0.0); }
            const double zu = z[t] / Eo[t];
            r1 = fmax(r1, fabs(ax - zu)); nAx = fmax(nAx, fabs(ax)); nz = fmax(nz, fabs(zu));
            r1s = fmax(r1s, fabs(Eo[t] * ax - z[t])); nAxs = fmax(nAxs, fabs(Eo[t] * ax)); nzs = fmax(nzs, fabs(z[t]));
            wbuf[row] = Eo[t] * y[t];
          }
        }
        __syncthreads();
        half_symv(xt, HP, xs, n, L.part);                  // P x
        tp_partials<true>(part, Gc, G, wbuf, L.zpad, N);
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
            tp_partials<true>(part, Gc, G, wbuf, L.zpad, N);
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
#ifdef F16_EXP_STAMPG
    __syncthreads();
    if (a.useq && l == 0) for (int e = 0; e < 12; ++e) a.useq[e * a.ld + b] = (double)tS[e];     // diagnostic build: cycles per phase (s_memtime)
#endif
    if (l == 0) {
      if (a.iters_out) a.iters_out[b] = it;
      if (a.info) {
        a.info[0 * a.ld + b] = (double)it;
        a.info[1 * a.ld + b] = rp;
        a.info[2 * a.ld + b] = rd;
        a.info[3 * a.ld + b] = rho;
#ifdef F16_EXP_STAMPG
        a.info[1 * a.ld + b] = (double)wc0;                // diagnostic build: start / end of this solve on the 100 MHz clock
        a.info[2 * a.ld + b] = (double)wall_clock64();
#endif
      }
      if (a.status && infeasible) a.status[b] |= F16_ST_QP_INFEASIBLE;
      else if (a.status && a.s.max_iter > 0 && (!converged || !ok)) a.status[b] |= F16_ST_QP_MAXITER;
    }
    __syncthreads();
  }
}

}  // namespace big

size_t mpc_big_ws_doubles(int N) { return big::ws_doubles(N); }

// k_mpc_big's dynamic LDS exceeds the default limit: opt in ONCE per device, to the BIG_MAXN size (never per launch; attribute
// calls are not legal under stream capture, so f16_mpc_plan_create calls this for wide plans before any solve can be captured).
int mpc_big_opt_in() {
  static std::mutex mu;
  static bool ready[64] = {};
  int dev = 0;
  if (int rc = hip_check(hipGetDevice(&dev), "hipGetDevice")) return rc;
  std::lock_guard<std::mutex> lk(mu);
  if (dev < 0 || dev >= 64 || ready[dev]) return F16_OK;
  if (int rc = hip_check(hipFuncSetAttribute((const void *)big::k_mpc_big, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(big::lds_doubles(BIG_MAXN) * sizeof(double))), "hipFuncSetAttribute(k_mpc_big)")) return rc;
  ready[dev] = true;
  return F16_OK;
}

static int big_launch(f16_ctx *ctx, const MpcArgs &a, void *stream, int sw_lo, int sw_hi, double *sw_base, double *sw_ucmd,
                      double *sw_info, int32_t *sw_status, unsigned int *sw_next, int32_t *sw_iters = nullptr,
                      const int32_t *sw_order = nullptr) {
  (void)ctx;
  {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = stream && hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    if (!capturing) { if (int rc = mpc_big_opt_in()) return rc; }      // (a captured plan solve: done by f16_mpc_plan_create)
  }
  const bool sweep = sw_hi > 0;
  if (sweep && (sw_lo < 1 || sw_hi < sw_lo || sw_hi > BIG_MAXN || !sw_base || !sw_ucmd || !sw_next)) return set_error(F16_EINVAL, "horizon sweep: bad arguments");
  big::SweepArgs sw{};
  sw.lo = sw_lo; sw.hi = sweep ? sw_hi : 0; sw.base = sw_base; sw.ucmd = sw_ucmd; sw.info = sw_info; sw.status = sw_status;
  sw.next = sw_next; sw.iters = sw_iters; sw.order = sw_order;
  const size_t lds = big::lds_doubles(sweep ? sw_hi : a.N) * sizeof(double);
  const long total = sweep ? (long)(sw_hi - sw_lo + 1) * a.B : a.B;
  long grid = total < 65536 ? total : 65536;
  if (sweep) {                                            // resident workgroups only (two per CU at most: LDS)
    int cus = 256, dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (grid > 2L * cus) grid = 2L * cus;
  }
  hipLaunchKernelGGL(big::k_mpc_big, dim3((unsigned)grid), dim3(big::BLK), lds, (hipStream_t)stream, a, sw);
  return hip_check(hipGetLastError(), "f16_mpc_batch long-horizon solve launch");
}

int mpc_big_solve_launch(f16_ctx *ctx, const MpcArgs &a, void *stream) {
  if (a.N < 1 || a.N > BIG_MAXN || !a.bigws || !a.ext || !a.Ppk) return set_error(F16_EINVAL, "long-horizon MPC solver: bad arguments");
  return big_launch(ctx, a, stream, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
}
size_t mpc_big_sweep_job_doubles(int N) { return big::sweep_job_doubles(N); }
int mpc_big_sweep_launch(f16_ctx *ctx, const MpcArgs &a, int lo, int hi, double *base, double *ucmd, double *info, int32_t *status,
                         unsigned int *next, int32_t *iters, const int32_t *order, void *stream) {
  return big_launch(ctx, a, stream, lo, hi, base, ucmd, info, status, next, iters, order);
}

// order <- the pairs of a sweep by decreasing cost estimate iterations x N^2 (256 buckets of the largest; one workgroup)
__global__ __launch_bounds__(1024) void k_sweep_order(const int32_t *iters, int32_t *order, long pairs, long B, int hi) {
  constexpr int NBK = 256;
  __shared__ int cnt[NBK], base[NBK];
  __shared__ float red[16];
  const int t = threadIdx.x;
  auto cost = [&](long p) { const float N = (float)(hi - (int)(p / B)); return (float)iters[p] * N * N; };
  float mx = 0.f;
  for (long p = t; p < pairs; p += 1024) mx = fmaxf(mx, cost(p));
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((t & 63) == 0) red[t >> 6] = mx;
  for (int i = t; i < NBK; i += 1024) cnt[i] = 0;
  __syncthreads();
  for (int i = 0; i < 16; ++i) mx = fmaxf(mx, red[i]);
  const float sc = mx > 0.f ? (NBK - 1) / mx : 0.f;
  auto bucket = [&](long p) { const int k = (int)(cost(p) * sc); return NBK - 1 - (k < 0 ? 0 : (k > NBK - 1 ? NBK - 1 : k)); };
  for (long p = t; p < pairs; p += 1024) atomicAdd(&cnt[bucket(p)], 1);
  __syncthreads();
  if (t == 0) { int s = 0; for (int k = 0; k < NBK; ++k) { base[k] = s; s += cnt[k]; } }     // bucket 0 = costliest
  __syncthreads();
  for (long p = t; p < pairs; p += 1024) order[atomicAdd(&base[bucket(p)], 1)] = (int32_t)p;
}
int mpc_big_sweep_order_launch(const int32_t *iters, int32_t *order, long pairs, long B, int hi, void *stream) {
  hipLaunchKernelGGL(k_sweep_order, dim3(1), dim3(1024), 0, (hipStream_t)stream, iters, order, pairs, B, hi);
  return hip_check(hipGetLastError(), "horizon sweep order launch");
}

}  // namespace f16
