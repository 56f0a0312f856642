// f16_control.hip -- batched control chain for gfx950: linearise -> ZOH -> DARE/LQR -> condensed QP -> ADMM.
//
//   k_linearise   env.py:294-342 (forward differences of _calc_xdot_na, eps 1e-5)          f16_linearise_batch
//   k_c2d         scipy.signal.cont2discrete zoh = expm([[A,B],[0,0]] dt) (env.py:50,351)  f16_c2d_batch
//   k_lqr         utils.py:219-245 dlqr (DARE + gain), Q = Cd'Cd, R = I (env.py:353-356)   f16_lqr_batch
//   k_mpc         utils.py:21-167 setup_OSQP + the OSQP solve of env.py:420-424            f16_mpc_batch
//
// Mapping.  k_linearise: one lane per (aircraft, perturbed column), tables in LDS as in the dynamics kernels.
// The other three: ONE WAVEFRONT PER AIRCRAFT (workgroup = 64 lanes); all per-aircraft matrices live in LDS:
// 9x9 blocks row-major, the 3N x 3N KKT inverse as a packed lower triangle (conflict-free row reads, see
// f16_smallmat.hpp).  The 9N x 3N prediction matrix CC of utils.py:171-197 is never formed: CC[i,j] = A^(i-j) B,
// so CC*U and CC'*v are causal (adjoint) convolutions with the N blocks G_k = A^k B, and CC'QQ CC is built by a
// diagonal recursion over the same blocks (DESIGN.md "QP build").
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <math.h>

#include <mutex>
#include <vector>

#include "../../include/f16_hip.h"
#include "f16_ctx.h"
#include "f16_plant.hpp"
#include "f16_smallmat.hpp"
#include "f16_mpc.hpp"
#include "f16_mpc_state.hpp"

namespace f16 {

// ------------------------------------------------------------------------------------ linearise
struct LinArgs {
  const double *tab, *lofi, *x, *u;
  double *Ac, *Bc, *Cc;
  int32_t *status;
  long B, ld;
  double eps, xcg;
  int fi;
  unsigned flags;
};

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_linearise(LinArgs a) {
  __shared__ __attribute__((aligned(16))) double tab[TABLE_IMAGE_DOUBLES];
  if (a.fi == 1) {
    const double2 *src = reinterpret_cast<const double2 *>(a.tab);
    double2 *dst = reinterpret_cast<double2 *>(tab);
    for (int i = threadIdx.x; i < TABLE_IMAGE_DOUBLES / 2; i += BLOCK) dst[i] = src[i];
    __syncthreads();
  }
  const long total = a.B * 12;
  for (long e = (long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long)gridDim.x * BLOCK) {
    const int c = (int)(e / a.B);          // perturbed column: 0..8 MPC states, 9..11 inputs
    const long b = e - (long)c * a.B;
    double sv[18], svp[18], f0[9], f1[9];
#pragma unroll
    for (int k = 0; k < 18; ++k) sv[k] = a.x[k * a.ld + b];
    // env.py:175-177: the three MPC inputs (u.values[1:4]) overwrite the actuator positions
    sv[13] = a.u[1 * a.ld + b]; sv[14] = a.u[2 * a.ld + b]; sv[15] = a.u[3 * a.ld + b];
    // column -> full-state index: mpc_x_idx = [3,4,7,8,9,10,11,17,16], inputs -> [13,14,15]
    const int idx = c == 0 ? 3 : c == 1 ? 4 : c == 2 ? 7 : c == 3 ? 8 : c == 4 ? 9 : c == 5 ? 10 : c == 6 ? 11
                  : c == 7 ? 17 : c == 8 ? 16 : 13 + (c - 9);
    double base = 0.0, pert = 0.0;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      svp[k] = sv[k] + (k == idx ? a.eps : 0.0);      // x + dx with dx = eps*e_c (env.py:327-330)
      if (k == idx) { base = sv[k]; pert = svp[k]; }
    }
    int st = 0;
    calc_xdot_na((const double *)tab, a.lofi, svp, f1, a.xcg, a.fi, a.flags, st);
    calc_xdot_na((const double *)tab, a.lofi, sv, f0, a.xcg, a.fi, a.flags, st);   // base point re-evaluated per column
    if (c < 9) {
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        a.Ac[(r * 9 + c) * a.ld + b] = (f1[r] - f0[r]) / a.eps;
        a.Cc[(r * 9 + c) * a.ld + b] = r == c ? (pert - base) / a.eps : 0.0;       // env.py:331 with _get_obs_na
      }
    } else {
#pragma unroll
      for (int r = 0; r < 9; ++r) a.Bc[(r * 3 + (c - 9)) * a.ld + b] = (f1[r] - f0[r]) / a.eps;
    }
    if (a.status && st) atomicOr(&a.status[b], st);
  }
}

// env.py:294-342 with the default _calc_xdot / get_obs (env.py:45): 18-state model, 22 perturbed columns.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_linearise_full(LinArgs a) {
  __shared__ __attribute__((aligned(16))) double tab[TABLE_IMAGE_DOUBLES];
  if (a.fi == 1) {
    const double2 *src = reinterpret_cast<const double2 *>(a.tab);
    double2 *dst = reinterpret_cast<double2 *>(tab);
    for (int i = threadIdx.x; i < TABLE_IMAGE_DOUBLES / 2; i += BLOCK) dst[i] = src[i];
    __syncthreads();
  }
  const long total = a.B * 22;
  for (long e = (long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long)gridDim.x * BLOCK) {
    const int c = (int)(e / a.B);          // 0..17 states, 18..21 inputs
    const long b = e - (long)c * a.B;
    double x[18], xp[18], u[4], up[4], f0[18], f1[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) { x[k] = a.x[k * a.ld + b]; xp[k] = x[k] + (k == c ? a.eps : 0.0); }
#pragma unroll
    for (int k = 0; k < 4; ++k) { u[k] = a.u[k * a.ld + b]; up[k] = u[k] + (k == c - 18 ? a.eps : 0.0); }
    int st = 0;
    calc_xdot((const double *)tab, a.lofi, xp, up, f1, a.xcg, a.fi, a.flags, st);
    calc_xdot((const double *)tab, a.lofi, x, u, f0, a.xcg, a.fi, a.flags, st);
    if (c < 18) {
#pragma unroll
      for (int r = 0; r < 18; ++r) a.Ac[(r * 18 + c) * a.ld + b] = (f1[r] - f0[r]) / a.eps;
      // C = d get_obs / dx, observed states [2,3,4,7,8,9,10,11,16,17] (parameters.py:134,160)
      const int OBS[10] = {2, 3, 4, 7, 8, 9, 10, 11, 16, 17};
#pragma unroll
      for (int r = 0; r < 10; ++r) {
        double d = 0.0;
#pragma unroll
        for (int k = 0; k < 18; ++k) if (k == OBS[r]) d = (xp[k] - x[k]) / a.eps;
        a.Cc[(r * 18 + c) * a.ld + b] = d;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 18; ++r) a.Bc[(r * 4 + (c - 18)) * a.ld + b] = (f1[r] - f0[r]) / a.eps;
    }
    if (a.status && st) atomicOr(&a.status[b], st);
  }
}

// ------------------------------------------------------------------------------------ small LDS allocator
struct Bump {
  double *p;
  __device__ double *take(int n) { double *r = p; p += (n + 1) & ~1; return r; }
};

__device__ __forceinline__ void load_soa(double *dst, const double *src, int n, long ld, long b) {
  for (int e = lane_id(); e < n; e += F16_WAVE) dst[e] = src[e * ld + b];
  __syncthreads();
}
__device__ __forceinline__ void store_soa(double *dst, const double *src, int n, long ld, long b) {
  for (int e = lane_id(); e < n; e += F16_WAVE) dst[e * ld + b] = src[e];
}

// ------------------------------------------------------------------------------------ c2d (ZOH)
// exp([[A,B],[0,0]] h) = [[E,F],[0,I]]: only the top 9x12 block [E F] is propagated.
// Scaling-and-squaring Taylor: X = A h/2^s, Y = B h/2^s, T_1 = [X Y], T_{k+1} = X T_k/(k+1), 13 terms
// (||X|| <= 0.5 => truncation < 1e-15 relative), then s squarings [E F] <- E [E F] + [0 F].
template <int NS, int NI>
__device__ void c2d_wave(const double *A, const double *Bm, double h, double *Ad, double *Bd, double *scr) {
  constexpr int NW = NS + NI, NE = NS * NW;
  Bump al{scr};
  double *X = al.take(NS * NS), *T = al.take(NE), *Tn = al.take(NE), *EF = al.take(NE);
  const int l = lane_id();
  double rs = 0.0;
  if (l < NS) {
    for (int j = 0; j < NS; ++j) rs += fabs(A[l * NS + j]);
    for (int j = 0; j < NI; ++j) rs += fabs(Bm[l * NI + j]);
    rs *= fabs(h);
  }
  const double nrm = wave_max(rs);
  int s = 0;
  if (nrm > 0.5) s = min(40, (int)ceil(log2(nrm / 0.5)));
  const double sc = ldexp(h, -s);
  for (int e = l; e < NS * NS; e += F16_WAVE) X[e] = A[e] * sc;
  for (int e = l; e < NE; e += F16_WAVE) {
    const int i = e / NW, j = e - i * NW;
    const double v = j < NS ? A[i * NS + j] * sc : Bm[i * NI + (j - NS)] * sc;
    T[e] = v;
    EF[e] = v + (j == i ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int k = 2; k <= 13; ++k) {
    mm<false, false>(Tn, X, T, NS, NS, NW, 1.0 / k);
    for (int e = l; e < NE; e += F16_WAVE) { T[e] = Tn[e]; EF[e] += Tn[e]; }
    __syncthreads();
  }
  for (int q = 0; q < s; ++q) {
    for (int e = l; e < NS * NS; e += F16_WAVE) X[e] = EF[(e / NS) * NW + (e % NS)];   // E
    __syncthreads();
    mm<false, false>(Tn, X, EF, NS, NS, NW);
    for (int e = l; e < NE; e += F16_WAVE) {
      const int j = e % NW;
      EF[e] = Tn[e] + (j >= NS ? EF[e] : 0.0);
    }
    __syncthreads();
  }
  for (int e = l; e < NS * NS; e += F16_WAVE) Ad[e] = EF[(e / NS) * NW + (e % NS)];
  for (int e = l; e < NS * NI; e += F16_WAVE) Bd[e] = EF[(e / NI) * NW + NS + (e % NI)];
  __syncthreads();
}

struct C2dArgs { const double *Ac, *Bc; double *Ad, *Bd; long B, ld; double dt; };

template <int NS, int NI>
__global__ __launch_bounds__(64) void k_c2d(C2dArgs a) {
  constexpr int NW = NS + NI;
  __shared__ double smem[2 * (NS * NS + 2) + 2 * (NS * NI + 2) + (NS * NS + 2) + 3 * (NS * NW + 2)];
  Bump al{smem};
  double *A = al.take(NS * NS), *Bm = al.take(NS * NI), *Ad = al.take(NS * NS), *Bd = al.take(NS * NI);
  double *scr = al.take(NS * NS + 2 + 3 * (NS * NW + 2) - 2);
  for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
    load_soa(A, a.Ac, NS * NS, a.ld, b);
    load_soa(Bm, a.Bc, NS * NI, a.ld, b);
    c2d_wave<NS, NI>(A, Bm, a.dt, Ad, Bd, scr);
    store_soa(a.Ad, Ad, NS * NS, a.ld, b);
    store_soa(a.Bd, Bd, NS * NI, a.ld, b);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ DARE / dlqr
// DARE  X = A'XA - A'XB (R+B'XB)^-1 B'XA + Q  by the structure-preserving doubling algorithm (SDA):
//   A0 = A, G0 = B R^-1 B', H0 = Q;  W = (I + G H)^-1;  A+ = A W A;  G+ = G + A W G A';  H+ = H + A' H W A.
// H_k -> X quadratically (21 doublings at the reference's trim point; scipy.linalg.solve_discrete_are's
// answer is reproduced to ~1e-11 relative).  R = I here (env.py:354, :405-407).
//
// Mapping: one wavefront per aircraft, every 9x9 operand lives in REGISTERS as a zero-padded 16x16 tile in the
// accumulator layout of v_mfma_f64_16x16x4_f64 (register q of lane l holds element (4q + l/16, l%16); rows 12..15 are
// padding and never stored), and every product is three chained MFMAs with no data movement at all, because
//   * the B operand of k-step s (lane l -> B[4s + l/16][l%16]) IS register s of the right factor's tile, and
//   * the A operand of k-step s (lane l -> A[l%16][4s + l/16]) IS register s of the tile of the left factor's TRANSPOSE,
// so the iteration carries A and A' (G, H are symmetric) and forms each intermediate in the orientation its consumer
// needs: 9 products = 27 MFMAs per doubling.  W = (I + G H)^-1 is a register-resident Gauss-Jordan (inverse9_tile).
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int DARE_SCRATCH = 9 * 18 + 16;

// D = C + L R, the left factor given as the tile of L'
__device__ __forceinline__ d4_t mm16(const d4_t &Lt, const d4_t &R, d4_t C) {
  C = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[0], R[0], C, 0, 0, 0);
  C = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[1], R[1], C, 0, 0, 0);
  C = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[2], R[2], C, 0, 0, 0);
  return C;
}
// tile of the row-major 9x9 LDS matrix M (or of its transpose)
__device__ __forceinline__ d4_t tile9(const double *M, bool transpose) {
  const int l = lane_id(), lc = l & 15, lq = l >> 4;
  d4_t t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int r = 4 * q + lq;
    if (r < 9 && lc < 9) t[q] = transpose ? M[lc * 9 + r] : M[r * 9 + lc];
  }
  return t;
}

// W <- M^-1 for the 9x9 tile M: Gauss-Jordan with partial pivoting on [M | I] held one COLUMN per lane (lanes 0..17,
// nine rows in registers).  A pivot step broadcasts the pivot column from its lane (v_readlane -> scalars), so the pivot
// search and the multipliers are wave-uniform and the step needs no LDS and no barrier; rows are not swapped, the row
// map is applied when the inverse is written back.  LDS (Wl, 81 doubles) only converts between the two layouts.
// PIVOT = false: the same elimination in natural order (no search, no row map) -- a third of the instructions.  The
// caller checks the result (residual of M W - I on the matrix cores) and repeats with PIVOT = true if it is not clean.
template <bool PIVOT>
__device__ __forceinline__ bool inverse9_tile(const d4_t &M, d4_t &W, double *Wl) {
  const int l = lane_id(), lc = l & 15, lq = l >> 4;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int r = 4 * q + lq;
    if (r < 9 && lc < 9) Wl[r * 9 + lc] = M[q];
  }
  __syncthreads();
  const int j = l < 18 ? l : 17;                       // lanes beyond 17 shadow column 17
  double r[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) r[i] = j < 9 ? Wl[i * 9 + j] : (i == j - 9 ? 1.0 : 0.0);
  unsigned done = 0;
  int pinv[9];                                         // pinv[i] = the pivot step that used row i
  bool ok = true;
#pragma unroll
  for (int p = 0; p < 9; ++p) {
    double c[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) c[i] = readlane_f64(r[i], p);
    if (PIVOT) {
      int piv = 0;
      double best = -1.0, cp = 1.0;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const double v = ((done >> i) & 1u) ? -1.0 : fabs(c[i]);
        const bool gt = v > best;
        best = gt ? v : best; piv = gt ? i : piv; cp = gt ? c[i] : cp;
      }
      ok = ok && best > 0.0;
      done |= 1u << piv;
      double rp = r[0];
#pragma unroll
      for (int i = 1; i < 9; ++i) rp = piv == i ? r[i] : rp;
      rp *= 1.0 / cp;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        r[i] = piv == i ? rp : fma(-c[i], rp, r[i]);
        pinv[i] = piv == i ? p : (p == 0 ? 0 : pinv[i]);
      }
    } else {
      ok = ok && c[p] != 0.0;
      const double rp = r[p] * (1.0 / c[p]);
#pragma unroll
      for (int i = 0; i < 9; ++i) r[i] = i == p ? rp : fma(-c[i], rp, r[i]);
    }
  }
  if (!PIVOT) {
#pragma unroll
    for (int i = 0; i < 9; ++i) pinv[i] = i;
  }
  __syncthreads();                                     // all reads of Wl are long done; reuse it for the inverse
  if (l >= 9 && l < 18) {
#pragma unroll
    for (int i = 0; i < 9; ++i) Wl[pinv[i] * 9 + (l - 9)] = r[i];
  }
  __syncthreads();
  W = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int rr = 4 * q + lq;
    if (rr < 9 && lc < 9) W[q] = Wl[rr * 9 + lc];
  }
  __syncthreads();
  return ok;
}

// Rinv: null = R is the identity (env.py:405-407), else the inverse of the input weight, 3 x 3 row-major (utils.py:219 `dlqr(A, B, Q, R)`)
__device__ __forceinline__ int dare_sda_wave(const double *A0, const double *Bm, const double *Q, double *X, double *scr, const double *Rinv = nullptr) {
  const int l = lane_id(), lc = l & 15, lq = l >> 4;
  const d4_t zero = {0.0, 0.0, 0.0, 0.0};
  d4_t A = tile9(A0, false), At = tile9(A0, true), H = tile9(Q, false);
  d4_t G;
  {
    d4_t Bt = zero;                                  // tile of B' (3x9): element (k, i) = B[i][k]
    if (lq < 3 && lc < 9) Bt[0] = Bm[lc * 3 + lq];
    double br = Bt[0];                               // element (lc, lq) of B R^-1
    if (Rinv && lq < 3 && lc < 9) br = Bm[lc * 3] * Rinv[lq] + Bm[lc * 3 + 1] * Rinv[3 + lq] + Bm[lc * 3 + 2] * Rinv[6 + lq];
    G = __builtin_amdgcn_mfma_f64_16x16x4f64(br, Bt[0], zero, 0, 0, 0);        // G0 = B R^-1 B'  (R = I: B B')
  }
  d4_t eye = zero;
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (4 * q + lq == lc && lc < 9) eye[q] = 1.0;
  int it = 0;
  for (; it < 60; ++it) {
    const d4_t M = mm16(G, H, eye);                  // I + G H          (G symmetric: its own transpose)
    const d4_t GAt = mm16(G, At, zero);              // G A'             (independent of W: overlaps the inverse)
    d4_t W;
    {
      // natural-order elimination first; accept it if || M W - I ||_max (M' = I + H G as the left factor) is at the level
      // cond(M) eps allows -- cond(M) reaches 3e7 late in the iteration, where partial pivoting leaves the same 1e-9 --
      // otherwise redo with partial pivoting.  Measured on the config-4 models (3,800 inversions): median residual
      // 1.4e-12, 99th percentile 6e-11, 0.05 % above the threshold; worst unpivoted pivot/column-max ratio 5e-5.
      bool good = inverse9_tile<false>(M, W, scr);
      const d4_t Mt = mm16(H, G, eye);
      d4_t neye = zero;
#pragma unroll
      for (int q = 0; q < 3; ++q) neye[q] = -eye[q];
      const d4_t E = mm16(Mt, W, neye);
      double emax = fmax(fmax(fabs(E[0]), fabs(E[1])), fabs(E[2]));
      emax = wave_max(emax);
      good = good && emax <= 2e-9;
      if (!good && !inverse9_tile<true>(M, W, scr)) { it = 60; break; }
    }
    const d4_t T1t = mm16(W, At, zero);              // (A W)' = W' A'   (left W' <- tile of W)
    const d4_t HWt = mm16(W, H, zero);               // (H W)' = W' H
    const d4_t An = mm16(T1t, A, zero);              // A W A            (left A W <- tile of (A W)')
    const d4_t Atn = mm16(A, T1t, zero);             // (A W A)' = A' (A W)'
    G = mm16(T1t, GAt, G);                           // G + A W G A'
    const d4_t HWA = mm16(HWt, A, zero);             // H W A            (left H W <- tile of (H W)')
    const d4_t dH = mm16(A, HWA, zero);              // A' H W A         (left A' <- tile of A)
    double dmax = 0.0, hmax = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      H[q] += dH[q];
      dmax = fmax(dmax, fabs(dH[q]));
      hmax = fmax(hmax, fabs(H[q]));
    }
    dmax = wave_max(dmax);
    hmax = wave_max(hmax);
    A = An; At = Atn;
    if (dmax <= 1e-16 * hmax) { ++it; break; }
  }
  // X = (H + H') / 2 through LDS (the caller wants it there)
  double *Xt = scr;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int r = 4 * q + lq;
    if (r < 9 && lc < 9) Xt[r * 9 + lc] = H[q];
  }
  __syncthreads();
  for (int e = l; e < 81; e += F16_WAVE) {
    const int i = e / 9, j = e - i * 9;
    X[e] = 0.5 * (Xt[e] + Xt[j * 9 + i]);
  }
  __syncthreads();
  return it;
}

// K = (B'XB + R)^-1 (B'XA)  (utils.py:244; Rw: null = R is the identity); scr >= 27+27+9+18+3 (+pad)
__device__ void lqr_gain_wave(const double *A, const double *Bm, const double *X, double *K, double *scr, const double *Rw = nullptr) {
  Bump al{scr};
  double *XB = al.take(27), *BXA = al.take(27), *S = al.take(9), *Wx = al.take(18), *f = al.take(3);
  mm<false, false>(XB, X, Bm, 9, 9, 3);        // X B   (9x3)
  mm<true, false>(S, Bm, XB, 3, 9, 3);         // B' X B
  if (Rw) { for (int e = lane_id(); e < 9; e += F16_WAVE) S[e] += Rw[e]; }
  else { for (int e = lane_id(); e < 3; e += F16_WAVE) S[e * 4] += 1.0; }
  __syncthreads();
  mm<true, false>(BXA, XB, A, 3, 9, 9);        // (X B)' A = B' X A   (X symmetric)
  inverse(S, 3, Wx, f);
  mm<false, false>(K, S, BXA, 3, 3, 9);
}

struct LqrArgs { const double *Ad, *Bd, *Cd; double *K, *Pare; int32_t *status; long B, ld; MpcProb pb; };

__global__ __launch_bounds__(64, 3) void k_lqr(LqrArgs a) {
  __shared__ double smem[82 * 5 + 28 * 2 + DARE_SCRATCH + 100];   // DARE scratch also serves lqr_gain_wave
  Bump al{smem};
  double *A = al.take(81), *Bm = al.take(27), *C = al.take(81), *Q = al.take(81), *X = al.take(81), *K = al.take(27);
  double *scr = al.take(DARE_SCRATCH + 90);
  for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
    load_soa(A, a.Ad, 81, a.ld, b);
    load_soa(Bm, a.Bd, 27, a.ld, b);
    load_soa(C, a.Cd, 81, a.ld, b);
    if (a.pb.custom_q) { for (int e = lane_id(); e < 81; e += F16_WAVE) Q[e] = a.pb.Q[e]; __syncthreads(); }      // utils.py:219 `Q`
    else mm<true, false>(Q, C, C, 9, 9, 9);            // Q = C'C (env.py:353)
    const int it = dare_sda_wave(A, Bm, Q, X, scr, a.pb.custom_r ? a.pb.Rinv : nullptr);
    lqr_gain_wave(A, Bm, X, K, scr, a.pb.custom_r ? a.pb.R : nullptr);
    for (int e = lane_id(); e < 27; e += F16_WAVE) a.K[e * a.ld + b] = -K[e];   // K = -dlqr(...) (env.py:356)
    if (a.Pare) store_soa(a.Pare, X, 81, a.ld, b);
    if (a.status && it >= 60 && lane_id() == 0) a.status[b] |= F16_ST_QP_MAXITER;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ MPC (QP build + ADMM)
// (CC U)[i][r] = sum_{j<=i} sum_c G_{i-j}[r][c] U[3j+c]
__device__ __forceinline__ double conv_forward_row(const double *G, const double *U, int i, int r) {
  double s = 0.0;
  for (int j = 0; j <= i; ++j) {
    const double *g = G + (i - j) * 27 + r * 3;
    s += g[0] * U[3 * j] + g[1] * U[3 * j + 1] + g[2] * U[3 * j + 2];
  }
  return s;
}

#ifdef F16_EXP_STAMPB   // diagnostic build: cycles per phase of the build kernel, aircraft b -> u_seq column b
#define BSTAMP(i) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t1_ = __builtin_amdgcn_s_memtime(); tB[i] += t1_ - tb0; tb0 = t1_; }
#else
#define BSTAMP(i)
#endif
// Per-lane values of the constraint rows a lane owns (rows l, l + 64, ...): registers up to MAXN; beyond (BIG: horizons up
// to BIG_MAXN, the reference's own sweep range env.py:426-436) they live in the per-aircraft global workspace, as does the
// packed KKT inverse -- a slow path that exists so that the horizon limit is not a hard wall.
template <bool BIG_> struct RowVec;
template <> struct RowVec<false> { double v[MAXT]; __device__ __forceinline__ double &operator[](int i) { return v[i]; } };
template <> struct RowVec<true> { double *p; __device__ __forceinline__ double &operator[](int i) { return p[i]; } };

template <bool SETUP_ONLY, bool BIG = false>
__global__ __launch_bounds__(64, SETUP_ONLY ? 2 : 1) void k_mpc(MpcArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int N = a.N, n = 3 * N, np = n * (n + 1) / 2, ms = 6 * N, m = 12 * N;
  const int l = lane_id();
  Bump al{smem};
  // R0 is time-shared: DARE scratch -> (build) pred | Q-weighted error | q -> (generic solver) packed KKT inverse.
  // Build-only launches keep the footprint at 17.8 KB for N = 30 (eight wavefronts per CU): the Q G_k / Qbar G_k blocks
  // of the P recursion are formed just in time (jit, 54 doubles) instead of being stored for all k.
  const int r0 = (SETUP_ONLY || BIG) ? 1100 : max(np, 1100);
  double *const R0 = al.take(r0);
  double *Minv = R0;                  // packed KKT inverse of the generic solver (BIG: in the global workspace, set per aircraft)
  const int TM = BIG ? (m + 63) / 64 : MAXT;     // constraint rows per lane
  double *G = al.take(N * 27);
  double *A = al.take(81), *Q = al.take(81), *Qb = al.take(81);
  double *jit = al.take(54);
  double *Bm = jit;                   // B is dead once G_0 is copied out, long before the P recursion uses jit
  double *x9 = al.take(9), *xref = al.take(9);
  double *qv, *wbuf, *pred;           // q | QQ (x_ref - MM x) | MM x: A^(i+1) x
  double *xs = nullptr, *xt = nullptr, *rhs = nullptr, *tv = nullptr;      // generic solver only
  double *Dg = nullptr, *E9 = nullptr, *Ec = nullptr, *Er = nullptr;       // generic solver: equilibration, then Gram weights
  if (SETUP_ONLY && !BIG) { pred = R0; wbuf = R0 + 9 * N; qv = R0 + 18 * N; }      // 21 N <= 840 < 1100
  else { qv = al.take(n); wbuf = al.take(m); pred = al.take(9 * N); }
  if (!SETUP_ONLY) { xs = al.take(n); xt = al.take(n); rhs = al.take(n); tv = al.take(n);
                     Dg = al.take(n); E9 = al.take(9 * N); Ec = al.take(n); Er = al.take(n + 3); }
  double *scr = R0, *X = R0 + 760;                  // DARE scratch (748 doubles), then X (82)

#ifdef F16_EXP_STAMPB
  unsigned long long tB[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tb0 = __builtin_amdgcn_s_memtime();
#endif
  for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
    // a prepared plan (mode 2) already holds everything that depends on the model only: A, Q, Qbar, G_k, P, A'A;
    // what is left per call is the state-dependent part, pred_i = A^(i+1) x and q
    const bool update_only = SETUP_ONLY && a.mode == 2;
    double *exm = a.ext ? a.ext + (size_t)b * mpc_ext_doubles(N) + mpc_ext_model(N) : nullptr;
    if (update_only) {
      for (int e = l; e < 81; e += F16_WAVE) { A[e] = exm[e]; Q[e] = exm[81 + e]; Qb[e] = exm[162 + e]; }
      const double *Gg = a.ext + (size_t)b * mpc_ext_doubles(N) + n;
      for (int e = l; e < N * 27; e += F16_WAVE) G[e] = Gg[e];
      __syncthreads();
    } else {
    // ---------------- model + weights (utils.py:82-105)
    {   // one global round trip for the three operands (189 strided loads in flight) instead of three
      const double a0 = a.Ad[(size_t)l * a.ld + b], a1 = l + 64 < 81 ? a.Ad[(size_t)(l + 64) * a.ld + b] : 0.0;
      const double c0 = a.Cd[(size_t)l * a.ld + b], c1 = l + 64 < 81 ? a.Cd[(size_t)(l + 64) * a.ld + b] : 0.0;
      const double b0 = l < 27 ? a.Bd[(size_t)l * a.ld + b] : 0.0;
      A[l] = a0; Qb[l] = c0;                           // Cd staged in Qb
      if (l + 64 < 81) { A[l + 64] = a1; Qb[l + 64] = c1; }
      if (l < 27) Bm[l] = b0;
      __syncthreads();
    }
    if (a.pb.custom_q) { for (int e = l; e < 81; e += F16_WAVE) Q[e] = a.pb.Q[e]; __syncthreads(); }       // utils.py:21 `Q`
    else mm<true, false>(Q, Qb, Qb, 9, 9, 9);         // Q = C'C (env.py:389)
    }
    if (l < 9) {
      const int MX[9] = {3, 4, 7, 8, 9, 10, 11, 17, 16};
      const double v = a.x ? a.x[MX[l] * a.ld + b] : 0.0;            // (no state when a plan is prepared)
      x9[l] = v;
      xref[l] = a.xref ? a.xref[l * a.ld + b]                                  // utils.py:21 `x_ref` as the caller gives it
                       : ((l >= 5 && l < 8 && a.dem) ? a.dem[(l - 5) * a.ld + b] : v);   // env.py:380-383 (x_ref[5:8] = demands)
    }
    __syncthreads();
    // a state that is not finite leaves no QP to solve (f16_mpc.hpp: mpc_job_nonfinite)
    bool fin_ = l < 9 ? (isfinite(x9[l]) && isfinite(xref[l])) : true;
    if (l < 3 && a.x) fin_ = fin_ && isfinite(a.x[(13 + l) * a.ld + b]);
    const bool nonfinite = __ballot(!fin_) != 0;
    if (a.ext && l == 0) a.ext[(size_t)b * mpc_ext_doubles(N) + mpc_ext_flag(N)] = nonfinite ? 1.0 : 0.0;
    BSTAMP(0)
    if (!update_only) {
    dare_sda_wave(A, Bm, Q, X, scr, a.pb.custom_r ? a.pb.Rinv : nullptr);
    BSTAMP(1)
    // (the gain K = -dlqr of utils.py:96 is not needed itself: it only enters through Q_bar)
    // Q_bar (utils.py:100) solves X = Phi' X Phi + Q + K'RK with Phi = A + B K: for the LQR gain K that equation IS the
    // DARE, so its solution is the DARE solution X itself.  (Measured on the reference's trim models: SDA's X agrees
    // with scipy.linalg.solve_discrete_lyapunov's Q_bar to 3e-13 relative -- closer than scipy's own DARE result.)
    copy(Qb, X, 81);
    if (exm) {
      for (int e = l; e < 81; e += F16_WAVE) { exm[e] = A[e]; exm[81 + e] = Q[e]; exm[162 + e] = Qb[e]; }
    }
    // ---------------- prediction blocks G_k = A^k B, pred_i = A^(i+1) x (utils.py:171-197 without forming CC/MM)
    {   // lane e = (r,c) < 27 carries G_k[r][c]; G_(k+1)[r][c] = sum_p A[r][p] G_k[p][c] takes the nine operands from the
        // lanes (p,c) by ds_bpermute: no LDS round trip + barrier per step of this N-long dependent chain
      const int e = l < 27 ? l : 0, r = e / 3, c = e - 3 * r;
      double ar[9], gv = Bm[e];
#pragma unroll
      for (int p = 0; p < 9; ++p) ar[p] = A[r * 9 + p];
      __syncthreads();                         // (Bm aliases pred: read before anything writes there)
      if (l < 27) G[l] = gv;
      for (int k = 1; k < N; ++k) {
        double sacc = 0.0;
#pragma unroll
        for (int p = 0; p < 9; ++p) sacc += ar[p] * __shfl(gv, p * 3 + c, 64);
        gv = sacc;
        if (l < 27) G[k * 27 + l] = gv;
      }
      __syncthreads();
    }
    BSTAMP(2)
    }
    // ---------------- pred_i = A^(i+1) x and q = -2 CC' QQ (x_ref - MM x)   (utils.py:112; f16_mpc_state.hpp: the closed-loop
    // rollout kernel of f16_mpc_wave.hip runs the very same code per step)
    mpc_state_vectors(A, Q, Qb, G, x9, xref, pred, wbuf, qv, N);
    BSTAMP(3)
    BSTAMP(4)
    // ---------------- P = 2 (CC' QQ CC + RR), packed lower, to the workspace.  (A'A is no longer formed here: the solvers
    // need the row-WEIGHTED Gram A'WA of the equilibrated problem, which has no Toeplitz recursion, and build it themselves.)
    // Block (j,l), j >= l, d = j-l:  T(j,l) = TQ(j,l) + G'_{N-1-j} Qbar G_{N-1-l},
    //   TQ(j,l) = TQ(j+1,l+1) + G'_{N-2-j} Q G_{N-2-l} (0 beyond N-2).
    // One chain per (diagonal d, element (ra,cb)), up to MAXCH per lane; all chains walk j together, so the two weighted
    // blocks a step needs (Q G_{N-2-j}, Qbar G_{N-1-j}) are the same for every lane and are formed once per step (jit).
    // This loop is LDS-bandwidth-bound (42 operand reads for 24 FMAs per chain element).  Tried and dropped: one lane per
    // block diagonal (216 FMAs for 27 + 72 reads per step, running sums in registers, no jit broadcast conflicts) --
    // fewer LDS bytes but only N active lanes and as many address computations for the packed stores: 20 % slower.
    double *Pg = a.Ppk + (size_t)b * np;
    if (!update_only) {
      constexpr int MAXCH = (9 * (BIG ? BIG_MAXN : MAXN) + F16_WAVE - 1) / F16_WAVE;
      double tq[MAXCH];
#pragma unroll
      for (int t = 0; t < MAXCH; ++t) tq[t] = 0.0;
      const int wh = l >= 27 ? 1 : 0, je = l - 27 * wh, jr = je / 3, jc = je - 3 * jr;      // jit roles of lanes 0..53
      const double *Qw = wh ? Qb : Q;
      __syncthreads();
      for (int j = N - 1; j >= 0; --j) {
        const int kq = wh ? N - 1 - j : N - 2 - j;
        if (l < 54 && kq >= 0) {
          double sj = 0.0;
#pragma unroll
          for (int p = 0; p < 9; ++p) sj += Qw[jr * 9 + p] * G[kq * 27 + p * 3 + jc];
          jit[l] = sj;
        }
        __syncthreads();
        const double *QGk = jit, *QbGk = jit + 27;
#pragma unroll
        for (int t = 0; t < MAXCH; ++t) {
          const int ch = l + F16_WAVE * t, d = ch / 9, ee = ch - 9 * d, ra = ee / 3, cb = ee - 3 * ra;
          __builtin_amdgcn_sched_barrier(0);      // one chain at a time: bounds the live operands (two waves per SIMD)
          if (ch < 9 * N && j >= d) {
            const int lcol = j - d;
            if (j <= N - 2) {
              double s = 0.0;
#pragma unroll
              for (int p = 0; p < 9; ++p) s += QGk[p * 3 + ra] * G[(N - 2 - lcol) * 27 + p * 3 + cb];
              tq[t] += s;
            }
            double gc[9];
#pragma unroll
            for (int p = 0; p < 9; ++p) gc[p] = G[(N - 1 - lcol) * 27 + p * 3 + cb];
            double sb = 0.0;
#pragma unroll
            for (int p = 0; p < 9; ++p) sb += QbGk[p * 3 + ra] * gc[p];
            const int gi = 3 * j + ra, gj = 3 * lcol + cb;
            // RR = blkdiag(R, ..., R) (utils.py:108-111); R = I (env.py:405-407) unless the caller gave one
            const double rr_ = a.pb.custom_r ? (d == 0 ? a.pb.R[ra * 3 + cb] : 0.0) : ((gi == gj) ? 1.0 : 0.0);
            if (gi >= gj) Pg[tri(gi, gj)] = 2.0 * (tq[t] + sb + rr_);
          }
        }
        __syncthreads();                                                    // jit is rewritten by the next step
      }
    }
    BSTAMP(5)
    // ---------------- bounds of the kept rows (utils.py:129-152): [6N state | 3N command | 3N rate]
    RowVec<BIG> lo, hi, z, y, dy, Eo, eqf;
    if constexpr (BIG) {
      double *rows = a.bigws + (size_t)b * mpc_big_doubles(N);
      Minv = rows + 7 * (size_t)(64 * TM);
      lo.p = rows; hi.p = rows + 64 * TM; z.p = rows + 2 * 64 * TM; y.p = rows + 3 * 64 * TM; dy.p = rows + 4 * 64 * TM;
      Eo.p = rows + 5 * 64 * TM; eqf.p = rows + 6 * 64 * TM;
    }
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int row = l + 64 * t, ri = BIG ? row : t;
      lo[ri] = 0.0; hi[ri] = 0.0; z[ri] = 0.0; y[ri] = 0.0; dy[ri] = 0.0;
      if (row < ms) {
        const int i = row / 6, rr = row - 6 * i;
        const double pm = pred[i * 9 + SROW[rr]];
        lo[ri] = a.pb.slb[rr] - pm;
        hi[ri] = a.pb.sub[rr] - pm;
      } else if (row < ms + n) {
        const int c = (row - ms) % 3;
        lo[ri] = a.pb.ulb[c]; hi[ri] = a.pb.uub[c];
      } else if (row < m) {
        const int k = row - ms - n, c = k % 3;
        if (k < 3) {
          const double act = a.x[(13 + c) * a.ld + b];
          lo[ri] = act + a.pb.rlb[c] * a.dt;
          hi[ri] = act + a.pb.rub[c] * a.dt;
        } else {
          lo[ri] = a.pb.rlb[c]; hi[ri] = a.pb.rub[c];              // reference quirk: not multiplied by dt (utils.py:151-152)
        }
      }
    }
    if (a.ext) {   // per-aircraft extras for the register-resident solver / the debug entry point
      double *ex = a.ext + (size_t)b * mpc_ext_doubles(N);
      for (int e = l; e < n; e += F16_WAVE) ex[e] = qv[e];
      if (!update_only) { for (int e = l; e < N * 27; e += F16_WAVE) ex[n + e] = G[e]; }
      for (int e = l; e < 9 * N; e += F16_WAVE) ex[n + N * 27 + e] = pred[e];
    }
    BSTAMP(6)
#ifdef F16_EXP_STAMPB
    if (SETUP_ONLY && a.useq && l == 0) for (int i = 0; i < 7; ++i) { a.useq[i * a.ld + b] = (double)tB[i]; tB[i] = 0; }
#endif
    if (SETUP_ONLY) { __syncthreads(); continue; }
    if (nonfinite) { mpc_write_nonfinite(a.ucmd, a.useq, a.info, a.iters_out, a.status, a.ld, a.N, a.s.rho, b, l, F16_WAVE); __syncthreads(); continue; }
    // ---------------- the solve (the published OSQP algorithm; same coordinates as f16_mpc_solve.hip:
    // x stays unscaled, the linear system is (c P + sigma D^-2 + rho A'WA) x~ = sigma D^-2 x - c q + A' E (rho zb - yb))
    const double sigma = a.s.sigma, alpha = a.s.alpha;
    double cs = 1.0;
    for (int e = l; e < n; e += F16_WAVE) { Dg[e] = 1.0; Ec[e] = 1.0; Er[e] = 1.0; }
    for (int e = l; e < 3; e += F16_WAVE) Er[n + e] = 0.0;
    for (int e = l; e < 9 * N; e += F16_WAVE) E9[e] = 1.0;
    __syncthreads();
    for (int pass = 0; pass < a.s.scaling; ++pass) {     // scaling.c:scale_data on the original entries and the running D, E, c
      for (int e = l; e < n; e += F16_WAVE) {            // column norms of [Pb; Ab]
        const int jb = e / 3, c = e - 3 * jb;
        double mp = 0.0, ma = 0.0;
        for (int i = 0; i < n; ++i) mp = fmax(mp, fabs(Pg[i >= e ? tri(i, e) : tri(e, i)]) * Dg[i]);
        for (int i = jb; i < N; ++i)
          for (int r = 0; r < 9; ++r) ma = fmax(ma, fabs(G[(i - jb) * 27 + r * 3 + c]) * E9[9 * i + r]);
        ma = fmax(fmax(ma, Ec[e]), fmax(Er[e], Er[e + 3]));
        tv[e] = 1.0 / sqrt(osqp_limit_scaling(Dg[e] * fmax(cs * mp, ma)));
      }
      for (int e = l; e < 9 * N; e += F16_WAVE) {        // row norms of the state block
        const int i = e / 9, r = e - 9 * i;
        double m_ = 0.0;
        for (int jb = 0; jb <= i; ++jb)
          for (int c = 0; c < 3; ++c) m_ = fmax(m_, fabs(G[(i - jb) * 27 + r * 3 + c]) * Dg[3 * jb + c]);
        wbuf[e] = 1.0 / sqrt(osqp_limit_scaling(E9[e] * m_));          // (m = 12N >= 9N)
      }
      for (int e = l; e < n; e += F16_WAVE) {
        rhs[e] = 1.0 / sqrt(osqp_limit_scaling(Ec[e] * Dg[e]));
        xt[e] = 1.0 / sqrt(osqp_limit_scaling(Er[e] * fmax(Dg[e], e >= 3 ? Dg[e - 3] : 0.0)));
      }
      __syncthreads();
      for (int e = l; e < n; e += F16_WAVE) { Dg[e] *= tv[e]; Ec[e] *= rhs[e]; Er[e] *= xt[e]; }
      for (int e = l; e < 9 * N; e += F16_WAVE) E9[e] *= wbuf[e];
      __syncthreads();
      double sm = 0.0, qn = 0.0;                          // cost scaling: mean column norm of Pb, ||qb||
      for (int e = l; e < n; e += F16_WAVE) {
        double mp = 0.0;
        for (int i = 0; i < n; ++i) mp = fmax(mp, fabs(Pg[i >= e ? tri(i, e) : tri(e, i)]) * Dg[i]);
        sm += cs * Dg[e] * mp;
        qn = fmax(qn, cs * Dg[e] * fabs(qv[e]));
      }
      sm = wave_sum(sm); qn = wave_max(qn);
      cs *= 1.0 / fmax(osqp_limit_scaling(sm / n), osqp_limit_scaling(qn));
      __syncthreads();
    }
    // per-row E, scaled bounds, rho-vector factor and Gram weight of the kept rows; sigma D^-2 per variable
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int row = l + 64 * t, ri = BIG ? row : t;
      Eo[ri] = 1.0; eqf[ri] = 1.0;
      if (row < m) {
        Eo[ri] = row < ms ? E9[9 * (row / 6) + SROW[row % 6]] : (row < ms + n ? Ec[row - ms] : Er[row - ms - n]);
        lo[ri] *= Eo[ri]; hi[ri] *= Eo[ri];
        eqf[ri] = (hi[ri] - lo[ri] < OSQP_RHO_TOL) ? OSQP_RHO_EQ_OVER_RHO_INEQ : 1.0;
      }
    }
    __syncthreads();                                      // E9 / Ec / Er are read; they now become the Gram weights W
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int row = l + 64 * t, ri = BIG ? row : t;
      if (row < ms) E9[9 * (row / 6) + SROW[row % 6]] = Eo[ri] * Eo[ri] * eqf[ri];
      else if (row < ms + n) Ec[row - ms] = Eo[ri] * Eo[ri] * eqf[ri];
      else if (row < m) Er[row - ms - n] = Eo[ri] * Eo[ri] * eqf[ri];
    }
    for (int e = l; e < n; e += F16_WAVE) Dg[e] = sigma / (Dg[e] * Dg[e]);     // Dg <- sigma D^-2  (c D of the rho estimate: sqrt back)
    __syncthreads();
    const double cinv = 1.0 / cs;
    auto build_minv = [&](double r) {                     // Minv <- (c P + sigma D^-2 + r A'WA)^-1, packed
      __syncthreads();
      for (int e = l; e < np; e += F16_WAVE) {
        int ia = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while (tri(ia + 1, 0) <= e) ++ia;
        while (tri(ia, 0) > e) --ia;
        const int ib = e - tri(ia, 0), ja = ia / 3, ca = ia - 3 * ja, jb = ib / 3, cb = ib - 3 * jb;
        double s = 0.0;
        for (int i = ja; i < N; ++i) {
          const double *ga = G + (i - ja) * 27 + ca, *gb = G + (i - jb) * 27 + cb, *wv = E9 + 9 * i;
#pragma unroll
          for (int rr = 0; rr < 6; ++rr) s += wv[SROW[rr]] * ga[SROW[rr] * 3] * gb[SROW[rr] * 3];
        }
        if (ia == ib) s += Ec[ia] + Er[ia] + Er[ia + 3];
        else if (ia == ib + 3) s -= Er[ia];
        Minv[e] = cs * Pg[e] + r * s + (ia == ib ? Dg[ia] : 0.0);
      }
      __syncthreads();
      return spd_inverse_packed(Minv, n, tv);
    };
    double rho = a.s.rho;
    if (!(rho > 0.0)) {   // the builder's opt-in start value (no equilibration): balance the two terms of P + rho A'A
      double tp = 0.0, ta = 0.0;
      for (int e = l; e < n; e += F16_WAVE) {
        const int jb = e / 3, c = e - 3 * jb;
        double s = Ec[e] + Er[e] + Er[e + 3];
        for (int i = jb; i < N; ++i)
          for (int rr = 0; rr < 6; ++rr) { const double gv = G[(i - jb) * 27 + SROW[rr] * 3 + c]; s += E9[9 * i + SROW[rr]] * gv * gv; }
        tp += Pg[tri(e, e)]; ta += s;
      }
      rho = fmin(fmax(RHO_AUTO_SCALE * sqrt(wave_sum(tp) / wave_sum(ta)), OSQP_RHO_MIN), OSQP_RHO_MAX);
    }
    bool ok = build_minv(rho);
    for (int e = l; e < n; e += F16_WAVE) xs[e] = 0.0;
    __syncthreads();
    int it = 0;
    double rp = INFINITY, rd = INFINITY;
    bool converged = false, infeasible = false;
    bool done = !ok || a.s.max_iter <= 0;
    auto adjoint = [&](const double *wv_, int e) {          // (A' w)_e for w in the [6N | 3N | 3N] layout, tv = CCs' w_s
      return tv[e] + wv_[ms + e] + (wv_[ms + n + e] - (e + 3 < n ? wv_[ms + n + e + 3] : 0.0));
    };
    while (!done) {
      ++it;
      // w = E (rho zb - yb) -> t = A' w ; rhs = sigma D^-2 x - c q + t
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int row = l + 64 * t, ri = BIG ? row : t;
        if (row < m) wbuf[row] = Eo[ri] * (rho * eqf[ri] * z[ri] - y[ri]);
      }
      __syncthreads();
      conv_adjoint<6>(tv, G, wbuf, N, SROW);
      for (int e = l; e < n; e += F16_WAVE) rhs[e] = Dg[e] * xs[e] - cs * qv[e] + adjoint(wbuf, e);
      __syncthreads();
      symv(xt, Minv, rhs, n);                               // x~
      // zb~ = E A x~ ; relaxation, projection, dual update
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int row = l + 64 * t, ri = BIG ? row : t;
        if (row < m) {
          double zt;
          if (row < ms) zt = conv_forward_row(G, xt, row / 6, SROW[row % 6]);
          else if (row < ms + n) zt = xt[row - ms];
          else { const int k = row - ms - n; zt = xt[k] - (k >= 3 ? xt[k - 3] : 0.0); }
          zt *= Eo[ri];
          const double ro = rho * eqf[ri];
          const double zr = alpha * zt + (1 - alpha) * z[ri];
          const double zn = fmin(fmax(zr + y[ri] / ro, lo[ri]), hi[ri]);
          dy[ri] = ro * (zr - zn);
          y[ri] = y[ri] + dy[ri];
          z[ri] = zn;
        }
      }
      for (int e = l; e < n; e += F16_WAVE) xs[e] = alpha * xt[e] + (1 - alpha) * xs[e];
      __syncthreads();
      if (it % a.s.check_every == 0 || it >= a.s.max_iter) {
        // residuals of the UNSCALED problem (OSQP termination test) + the scaled ones for the rho estimate
        double r1 = 0.0, nAx = 0.0, nz = 0.0, r1s = 0.0, nAxs = 0.0, nzs = 0.0;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const int row = l + 64 * t, ri = BIG ? row : t;
          if (row < m) {
            double ax;
            if (row < ms) ax = conv_forward_row(G, xs, row / 6, SROW[row % 6]);
            else if (row < ms + n) ax = xs[row - ms];
            else { const int k = row - ms - n; ax = xs[k] - (k >= 3 ? xs[k - 3] : 0.0); }
            const double zu = z[ri] / Eo[ri];
            r1 = fmax(r1, fabs(ax - zu)); nAx = fmax(nAx, fabs(ax)); nz = fmax(nz, fabs(zu));
            r1s = fmax(r1s, fabs(Eo[ri] * ax - z[ri])); nAxs = fmax(nAxs, fabs(Eo[ri] * ax)); nzs = fmax(nzs, fabs(z[ri]));
            wbuf[row] = Eo[ri] * y[ri];
          }
        }
        __syncthreads();
        symv(xt, Pg, xs, n);                                // P x (packed P from the workspace)
        conv_adjoint<6>(tv, G, wbuf, N, SROW);
        double r2 = 0.0, nPx = 0.0, nAty = 0.0, nq = 0.0, r2s = 0.0, nPxs = 0.0, nAtys = 0.0, nqs = 0.0;
        for (int e = l; e < n; e += F16_WAVE) {
          const double aty = cinv * adjoint(wbuf, e), rr_ = xt[e] + qv[e] + aty, cD = cs * sqrt(sigma / Dg[e]);
          r2 = fmax(r2, fabs(rr_)); nPx = fmax(nPx, fabs(xt[e])); nAty = fmax(nAty, fabs(aty)); nq = fmax(nq, fabs(qv[e]));
          r2s = fmax(r2s, cD * fabs(rr_)); nPxs = fmax(nPxs, cD * fabs(xt[e])); nAtys = fmax(nAtys, cD * fabs(aty)); nqs = fmax(nqs, cD * fabs(qv[e]));
        }
        rp = wave_max(r1);
        rd = wave_max(r2);
        const double np_ = fmax(wave_max(nAx), wave_max(nz));
        const double nd_ = fmax(fmax(wave_max(nPx), wave_max(nAty)), wave_max(nq));
        __syncthreads();
        if (rp < a.s.eps_abs + a.s.eps_rel * np_ && rd < a.s.eps_abs + a.s.eps_rel * nd_) { done = true; converged = true; }
        else {
          // OSQP primal-infeasibility certificate on dy (auxil.c:is_primal_infeasible)
          double ndy = 0.0, supp = 0.0;
#pragma unroll
          for (int t = 0; t < TM; ++t) {
            const int row = l + 64 * t, ri = BIG ? row : t;
            if (row < m) {
              ndy = fmax(ndy, fabs(Eo[ri] * dy[ri]));
              supp += hi[ri] * fmax(dy[ri], 0.0) + lo[ri] * fmin(dy[ri], 0.0);
              wbuf[row] = Eo[ri] * dy[ri];
            }
          }
          ndy = wave_max(ndy);
          supp = wave_sum(supp);
          __syncthreads();
          if (ndy > a.s.eps_prim_inf && supp < -a.s.eps_prim_inf * ndy) {
            conv_adjoint<6>(tv, G, wbuf, N, SROW);
            double nat = 0.0;
            for (int e = l; e < n; e += F16_WAVE) nat = fmax(nat, fabs(adjoint(wbuf, e)));
            nat = wave_max(nat);
            if (nat < a.s.eps_prim_inf * ndy) { done = true; infeasible = true; }
          }
          __syncthreads();
        }
        if (done) {}
        else if (it >= a.s.max_iter) done = true;
        else if (a.s.adaptive_rho && it % a.s.rho_every == 0) {     // auxil.c:compute_rho_estimate (scaled residuals)
          const double pr = wave_max(r1s) / (fmax(wave_max(nzs), wave_max(nAxs)) + 1e-10);
          const double dr = wave_max(r2s) / (fmax(fmax(wave_max(nqs), wave_max(nAtys)), wave_max(nPxs)) + 1e-10);
          const double nw = fmin(fmax(rho * sqrt(pr / (dr + 1e-10)), OSQP_RHO_MIN), OSQP_RHO_MAX);
          if (nw > OSQP_ADAPTIVE_RHO_TOLERANCE * rho || nw < rho / OSQP_ADAPTIVE_RHO_TOLERANCE) {
            rho = nw;
            if (!build_minv(rho)) done = true;
          }
        }
      }
    }
    // res.x[0:3] (env.py:424); OSQP hands back NaN for a problem it certifies infeasible
    for (int e = l; e < 3; e += F16_WAVE) a.ucmd[e * a.ld + b] = infeasible ? NAN : xs[e];
    if (a.useq) for (int e = l; e < n; e += F16_WAVE) a.useq[e * a.ld + b] = infeasible ? NAN : xs[e];
    if (l == 0) {
      if (a.info) {
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

static size_t mpc_lds_doubles(int N, bool setup_only, bool big = false) {      // mirrors the Bump allocations at the top of k_mpc
  const int n = 3 * N, np = n * (n + 1) / 2, m = 12 * N;
  auto ev = [](int v) { return (size_t)((v + 1) & ~1); };
  const int r0 = (!setup_only && !big && np > 1100) ? np : 1100;
  const size_t common = ev(r0) + ev(N * 27) + ev(81) * 3 + ev(54) + ev(9) * 2;
  const size_t vecs = ev(n) + ev(m) + ev(9 * N);             // q | weighted error | pred (build-only launches up to MAXN overlay them on r0)
  if (setup_only) return big ? common + vecs : common;
  return common + vecs + 4 * ev(n) + 2 * ev(n) + ev(9 * N) + ev(n + 3);
}

}  // namespace f16

using namespace f16;

extern "C" int f16_linearise_batch(f16_ctx *ctx, const double *x, const double *u, double *Ac, double *Bc, double *Cc,
                                   int32_t *status, long B, long ld, double eps, double xcg, int fi_flag, unsigned flags,
                                   void *stream) {
  if (!ctx || !x || !u || !Ac || !Bc || !Cc || B < 0 || ld < B) return set_error(F16_EINVAL, "bad argument to f16_linearise_batch");
  if (B == 0) return F16_OK;
  LinArgs a{ctx->d_tab, ctx->d_lofi, x, u, Ac, Bc, Cc, status, B, ld, eps, xcg, fi_flag, flags};
  const long lanes = B * 12;
  if (lanes <= 64L * 256) {
    hipLaunchKernelGGL(k_linearise<64>, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
  } else {
    long blocks = (lanes + 255) / 256;
    hipLaunchKernelGGL(k_linearise<256>, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, (hipStream_t)stream, a);
  }
  return hip_check(hipGetLastError(), "f16_linearise_batch launch");
}

static unsigned wave_grid(long B) { return (unsigned)(B < 256L * 32 ? B : 256L * 32); }

extern "C" int f16_c2d_batch(f16_ctx *ctx, const double *Ac, const double *Bc, double *Ad, double *Bd, long B, long ld,
                             double dt, void *stream) {
  if (!ctx || !Ac || !Bc || !Ad || !Bd || B < 0 || ld < B) return set_error(F16_EINVAL, "bad argument to f16_c2d_batch");
  if (B == 0) return F16_OK;
  C2dArgs a{Ac, Bc, Ad, Bd, B, ld, dt};
  hipLaunchKernelGGL((k_c2d<9, 3>), dim3(wave_grid(B)), dim3(64), 0, (hipStream_t)stream, a);
  return hip_check(hipGetLastError(), "f16_c2d_batch launch");
}

extern "C" int f16_linearise_full_batch(f16_ctx *ctx, const double *x, const double *u, double *Ac, double *Bc, double *Cc,
                                        int32_t *status, long B, long ld, double eps, double xcg, int fi_flag, unsigned flags,
                                        void *stream) {
  if (!ctx || !x || !u || !Ac || !Bc || !Cc || B < 0 || ld < B) return set_error(F16_EINVAL, "bad argument to f16_linearise_full_batch");
  if (B == 0) return F16_OK;
  LinArgs a{ctx->d_tab, ctx->d_lofi, x, u, Ac, Bc, Cc, status, B, ld, eps, xcg, fi_flag, flags};
  const long lanes = B * 22;
  if (lanes <= 64L * 256) {
    hipLaunchKernelGGL(k_linearise_full<64>, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
  } else {
    long blocks = (lanes + 255) / 256;
    hipLaunchKernelGGL(k_linearise_full<256>, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, (hipStream_t)stream, a);
  }
  return hip_check(hipGetLastError(), "f16_linearise_full_batch launch");
}

extern "C" int f16_c2d_full_batch(f16_ctx *ctx, const double *Ac, const double *Bc, double *Ad, double *Bd, long B, long ld,
                                  double dt, void *stream) {
  if (!ctx || !Ac || !Bc || !Ad || !Bd || B < 0 || ld < B) return set_error(F16_EINVAL, "bad argument to f16_c2d_full_batch");
  if (B == 0) return F16_OK;
  C2dArgs a{Ac, Bc, Ad, Bd, B, ld, dt};
  hipLaunchKernelGGL((k_c2d<18, 4>), dim3(wave_grid(B)), dim3(64), 0, (hipStream_t)stream, a);
  return hip_check(hipGetLastError(), "f16_c2d_full_batch launch");
}

// Weights and bounds as the caller gives them (utils.py:21, :219) -> the kernels' MpcProb; null = env.py's constants.
// +-INFINITY (or anything beyond OSQP's infinity 1e30) becomes +-1e30, as OSQP itself reads it.  The solvers keep the six state
// rows the reference bounds (alpha, beta, p, q, r, lf2: parameters.py:59-95) and leave phi, theta, lf1 out of the iteration, so
// THAT pattern is fixed: a finite bound on one of the three, or no bound at all on one of the six, is refused (the QP-build
// entry f16_mpc_qp_debug_w takes any pattern).
int f16::mpc_fill_prob(MpcProb *p, const f16_mpc_weights *w) {
  mpc_default_prob(p);
  if (!w) return F16_OK;
  auto inf = [](double v) { return !(fabs(v) < 1e20); };
  auto clip = [](double v) { return v > 1e30 ? 1e30 : (v < -1e30 ? -1e30 : v); };
  if (!w->q_from_cd) {
    for (int i = 0; i < 9; ++i)
      for (int j = 0; j < 9; ++j) {
        if (!isfinite(w->Q[i * 9 + j]) || fabs(w->Q[i * 9 + j] - w->Q[j * 9 + i]) > 1e-12 * (fabs(w->Q[i * 9 + j]) + fabs(w->Q[j * 9 + i])))
          return set_error(F16_EINVAL, "f16_mpc_weights: Q must be finite and symmetric");
        p->Q[i * 9 + j] = 0.5 * (w->Q[i * 9 + j] + w->Q[j * 9 + i]);
      }
    p->custom_q = 1;
  }
  {
    const double *R = w->R;
    bool ident = true;
    for (int i = 0; i < 9; ++i) {
      if (!isfinite(R[i])) return set_error(F16_EINVAL, "f16_mpc_weights: R must be finite");
      ident = ident && R[i] == ((i % 4 == 0) ? 1.0 : 0.0);
    }
    if (!ident) {
      if (fabs(R[1] - R[3]) + fabs(R[2] - R[6]) + fabs(R[5] - R[7]) > 1e-12 * (fabs(R[0]) + fabs(R[4]) + fabs(R[8])))
        return set_error(F16_EINVAL, "f16_mpc_weights: R must be symmetric");
      const double c00 = R[4] * R[8] - R[5] * R[7], c01 = R[5] * R[6] - R[3] * R[8], c02 = R[3] * R[7] - R[4] * R[6];
      const double det = R[0] * c00 + R[1] * c01 + R[2] * c02;
      if (!(R[0] > 0) || !(R[0] * R[4] - R[1] * R[3] > 0) || !(det > 0)) return set_error(F16_EINVAL, "f16_mpc_weights: R must be positive definite");
      const double inv[9] = {c00, R[2] * R[7] - R[1] * R[8], R[1] * R[5] - R[2] * R[4],
                             c01, R[0] * R[8] - R[2] * R[6], R[2] * R[3] - R[0] * R[5],
                             c02, R[1] * R[6] - R[0] * R[7], R[0] * R[4] - R[1] * R[3]};
      for (int i = 0; i < 9; ++i) { p->R[i] = R[i]; p->Rinv[i] = inv[i] / det; }
      p->custom_r = 1;
    }
  }
  static const int srow[6] = {2, 3, 4, 5, 6, 8}, free_rows[3] = {0, 1, 7};
  for (int k = 0; k < 3; ++k)
    if (!inf(w->x_lb[free_rows[k]]) || !inf(w->x_ub[free_rows[k]]))
      return set_error(F16_EINVAL, "f16_mpc_weights: phi, theta and lf1 (MPC states 0, 1, 7) carry no bounds in the solvers (use f16_mpc_qp_debug_w for the QP alone)");
  for (int k = 0; k < 6; ++k) {
    const double lo = w->x_lb[srow[k]], hi = w->x_ub[srow[k]];
    if (lo != lo || hi != hi || (inf(lo) && inf(hi)) || lo > hi) return set_error(F16_EINVAL, "f16_mpc_weights: a bounded state row needs lb <= ub and at least one finite bound");
    p->slb[k] = clip(lo); p->sub[k] = clip(hi);
  }
  for (int c = 0; c < 3; ++c) {
    const double v[4] = {w->u_lb[c], w->u_ub[c], w->udot_lb[c], w->udot_ub[c]};
    if (v[0] != v[0] || v[1] != v[1] || v[2] != v[2] || v[3] != v[3] || v[0] > v[1] || v[2] > v[3] || (inf(v[0]) && inf(v[1])) || (inf(v[2]) && inf(v[3])))
      return set_error(F16_EINVAL, "f16_mpc_weights: command and rate rows need lb <= ub and at least one finite bound");
    p->ulb[c] = clip(v[0]); p->uub[c] = clip(v[1]); p->rlb[c] = clip(v[2]); p->rub[c] = clip(v[3]);
  }
  return F16_OK;
}

extern "C" void f16_mpc_default_weights(f16_mpc_weights *w) {
  if (!w) return;
  *w = f16_mpc_weights{};
  w->q_from_cd = 1;
  static const double xlb[9] = {-INFINITY, -INFINITY, -20., -30., -300., -100., -50., -INFINITY, 0.};      // parameters.py:59-95 in MPC-state order
  static const double xub[9] = {INFINITY, INFINITY, 90., 30., 300., 100., 50., INFINITY, 25.};
  static const double ulb[3] = {-25., -21.5, -30.}, uub[3] = {25., 21.5, 30.}, rlb[3] = {-60., -80., -120.}, rub[3] = {60., 80., 120.};
  for (int i = 0; i < 9; ++i) { w->Q[i * 10] = 1.0; w->x_lb[i] = xlb[i]; w->x_ub[i] = xub[i]; }
  for (int i = 0; i < 3; ++i) { w->R[i * 4] = 1.0; w->u_lb[i] = ulb[i]; w->u_ub[i] = uub[i]; w->udot_lb[i] = rlb[i]; w->udot_ub[i] = rub[i]; }
}

extern "C" int f16_lqr_batch_w(f16_ctx *ctx, const double *Ad, const double *Bd, const double *Cd, const f16_mpc_weights *h_w,
                               double *K, double *Pare, int32_t *status, long B, long ld, void *stream) {
  if (!ctx || !Ad || !Bd || !Cd || !K || B < 0 || ld < B) return set_error(F16_EINVAL, "bad argument to f16_lqr_batch");
  if (B == 0) return F16_OK;
  LqrArgs a{Ad, Bd, Cd, K, Pare, status, B, ld, MpcProb{}};
  if (int rc = mpc_fill_prob(&a.pb, h_w)) return rc;
  hipLaunchKernelGGL(k_lqr, dim3(wave_grid(B)), dim3(64), 0, (hipStream_t)stream, a);
  return hip_check(hipGetLastError(), "f16_lqr_batch launch");
}
extern "C" int f16_lqr_batch(f16_ctx *ctx, const double *Ad, const double *Bd, const double *Cd, double *K, double *Pare,
                             int32_t *status, long B, long ld, void *stream) {
  return f16_lqr_batch_w(ctx, Ad, Bd, Cd, nullptr, K, Pare, status, B, ld, stream);
}

extern "C" void f16_qp_default_settings(f16_qp_settings *s) {
  // what env.py:420-422 invokes: osqp.OSQP().setup(..., max_iter=40000, polish=False), every other setting at OSQP's
  // default; the adaptive-rho interval is OSQP's no-timer constant (its default is wall-clock based)
  s->rho = 0.1; s->scaling = 10;
  s->sigma = 1e-6; s->alpha = 1.6; s->eps_abs = 1e-3; s->eps_rel = 1e-3; s->eps_prim_inf = 1e-4;
  s->max_iter = 40000;            // env.py:421
  s->check_every = 25; s->rho_every = 100; s->adaptive_rho = 1;
}

// k_mpc's dynamic LDS exceeds 64 KB for the generic solver at large N: opt in ONCE per device, to the MAXN size (never
// per launch: a concurrent call with a smaller horizon must not shrink the limit under a larger launch, and attribute
// calls are not legal under stream capture).
static int mpc_lds_opt_in() {
  static std::mutex mu;
  static bool ready[64] = {};
  int dev = 0;
  if (int rc = hip_check(hipGetDevice(&dev), "hipGetDevice")) return rc;
  std::lock_guard<std::mutex> lk(mu);
  if (dev < 0 || dev >= 64 || ready[dev]) return F16_OK;
  hipError_t e = hipFuncSetAttribute((const void *)k_mpc<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(mpc_lds_doubles(MAXN, false) * sizeof(double)));
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void *)k_mpc<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(mpc_lds_doubles(MAXN, true) * sizeof(double)));
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void *)k_mpc<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(mpc_lds_doubles(BIG_MAXN, false, true) * sizeof(double)));
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void *)k_mpc<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(mpc_lds_doubles(BIG_MAXN, true, true) * sizeof(double)));
  if (int rc = hip_check(e, "hipFuncSetAttribute(k_mpc)")) return rc;
  ready[dev] = true;
  return F16_OK;
}

// Per-call QP workspace, stream-ordered (see f16_ctx.h): [B][np] P | [B][ext] extras | [B][tiles] A'WA of the fast solver.
static int mpc_work_alloc(f16_ctx *ctx, MpcArgs &a, bool with_ext, void *stream, void **block, bool with_gram = false,
                          bool with_pblk = false) {
  const size_t np = (size_t)(3 * a.N) * (3 * a.N + 1) / 2;
  const bool big = a.N > FAST_MAXN;                    // long horizons: operands in HBM (the workgroup solver; beyond MAXN also the one-wave slow path)
  const size_t bigd = big ? (mpc_big_ws_doubles(a.N) > mpc_big_doubles(a.N) ? mpc_big_ws_doubles(a.N) : mpc_big_doubles(a.N)) : 0;
  const size_t need = (np + (with_ext ? mpc_ext_doubles(a.N) : 0) + (with_gram ? MPC_TILE_DOUBLES : 0) + bigd +
                       (with_pblk ? WAVE_PBLK_DOUBLES : 0)) * (size_t)a.B * sizeof(double);
  *block = nullptr;
  // Not under stream capture.  What round 3 established (profiles/r03_capture_pool.log, r03_capture_probe.log): on ROCm 7.2 a
  // plain-HIP graph with mem-alloc / mem-free nodes (tools/micro/capture_pool.hip, none of this library's kernels) replays
  // WRONGLY beside eager allocations from the same pool (302,520 wrong elements at replay 11 of 12); the mechanism is not
  // established -- the logged eager blocks never overlap the graph's block.  tools/gpu_capture_probe.py replays the one-shot call
  // with no host state in the capture (F16_MPC_DISPATCH_ORDER=0): every replay is bit-identical to the eager call EXCEPT the
  // ones with an eager call of the same context enqueued behind them (NaN in 20-200 of 256 aircraft), with either solver.
  // The fault is the runtime's, not in what the capture bakes in; the refusal stays.  A prepared plan owns its workspace for
  // its lifetime and IS capturable (f16_mpc_plan_solve; tests).
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  static const bool allow_capture = [] { const char *e = getenv("F16_MPC_ALLOW_CAPTURE"); return e && e[0] == '1'; }();   // (diagnosis only: tools/gpu_capture_probe.py)
  if (!allow_capture && stream && hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
    return set_error(F16_EINVAL, "one-shot MPC calls cannot be captured into a HIP graph (per-call workspace): use f16_mpc_plan_solve");
  if (int rc = hip_check(hipMallocFromPoolAsync(block, need, ctx->pool, (hipStream_t)stream), "hipMallocFromPoolAsync QP workspace")) return rc;
  a.Ppk = (double *)*block;
  a.ext = with_ext ? a.Ppk + np * (size_t)a.B : nullptr;
  a.gramws = with_gram ? a.Ppk + (np + mpc_ext_doubles(a.N)) * (size_t)a.B : nullptr;
  a.bigws = big ? a.Ppk + (np + (with_ext ? mpc_ext_doubles(a.N) : 0) + (with_gram ? MPC_TILE_DOUBLES : 0)) * (size_t)a.B : nullptr;
  a.pblk = with_pblk ? a.Ppk + (np + (with_ext ? mpc_ext_doubles(a.N) : 0) + (with_gram ? MPC_TILE_DOUBLES : 0) + bigd) * (size_t)a.B : nullptr;
  return F16_OK;
}
static int mpc_work_free(void *block, void *stream) {
  return block ? hip_check(hipFreeAsync(block, (hipStream_t)stream), "hipFreeAsync QP workspace") : F16_OK;
}

// Dispatch-order history of one-shot calls, per (stream, batch size); nullptr = none available (run in caller's order).
static f16_ctx::sched_entry *mpc_sched_entry(f16_ctx *ctx, void *stream, long B, int tag = 0) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  for (int i = 0; i < ctx->n_sched; ++i)
    if (ctx->sched[i].stream == stream && ctx->sched[i].B == B && ctx->sched[i].tag == tag) return &ctx->sched[i];
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (stream && hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return nullptr;
  int32_t *buf = nullptr;
  if (hipMalloc(&buf, 2 * (size_t)B * sizeof(int32_t)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  int slot = ctx->n_sched;
  if (slot >= F16_MAX_SCHED) {
    // all history slots taken (more than F16_MAX_SCHED (stream, batch size) pairs on one context): the oldest one is recycled, round
    // robin -- hipFree waits for whatever still reads its buffer; a caller that cycles through more pairs than slots loses the
    // ordering gain on the evicted ones, never a result
    static int victim = 0;
    slot = victim++ % F16_MAX_SCHED;
    (void)hipFree(ctx->sched[slot].buf);
  } else {
    ctx->n_sched++;
  }
  f16_ctx::sched_entry &e = ctx->sched[slot];
  e.stream = stream; e.B = B; e.buf = buf; e.valid = 0; e.tag = tag;
  return &e;
}

// The solve behind a build: one wavefront per aircraft where that solver applies (k_mpc_fast then only equilibrates, mode 3),
// else the 512-lane workgroup per aircraft.  F16_MPC_WAVE=0 keeps the latter everywhere (A/B runs, cross-checks).
static int mpc_solve_dispatch(f16_ctx *ctx, const MpcArgs &a, void *stream) {
  if (a.mode == 0 && mpc_wave_enabled(a)) {
    static const bool ruiz_outside = [] { const char *e = getenv("F16_WAVE_RUIZ"); return e && e[0] == '0'; }();
    MpcArgs w = a;
    w.wave_ruiz = ruiz_outside ? 0 : 1;
    if (ruiz_outside) {
      MpcArgs e = a;
      e.mode = 3; e.order = nullptr; e.iters_out = nullptr; e.warm = nullptr;
      if (int rc = mpc_fast_solve_launch(ctx, e, stream)) return rc;
    }
    return mpc_wave_solve_launch(ctx, w, stream);
  }
  return mpc_fast_solve_launch(ctx, a, stream);
}

// mode 0: generic one-wave kernel (build + ADMM); 1: build only (workspace P, A'A, q|G|pred; *keep receives the block,
// the caller frees it); 2: build, then the register-resident 512-thread solver (N <= 32); 3: build, then the 512-lane workgroup
// solver for long horizons (N > 32, f16_mpc_big.hip).
static int mpc_launch(f16_ctx *ctx, MpcArgs &a, void *stream, int mode, void **keep = nullptr) {
  const int N = a.N;
  const bool big = N > MAXN;
  if (N < 1 || N > BIG_MAXN || (N > FAST_MAXN && mode == 2)) return set_error(F16_EINVAL, "horizon must be 1..150 (plans: 1..40)");
  const size_t lds = mpc_lds_doubles(N, mode != 0, big) * sizeof(double);
  if (lds > 160 * 1024) return set_error(F16_EINVAL, "horizon too large for LDS");
  if (int rc = mpc_lds_opt_in()) return rc;
  void *block = nullptr;
  // the wavefront solver (f16_mpc_wave.hip: N <= 30, equilibrated solves) keeps its per-aircraft workspace in the Gram block
  const bool wave_ok = mode == 2 && N <= WAVE_MAXN && a.s.scaling > 0 && a.s.max_iter > 0;
  if (int rc = mpc_work_alloc(ctx, a, mode != 0, stream, &block, mode == 2 && (a.s.adaptive_rho || wave_ok), wave_ok)) return rc;
  int rc = F16_OK;
  if (mode == 0) {
    if (big) hipLaunchKernelGGL((k_mpc<false, true>), dim3(wave_grid(a.B)), dim3(64), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_mpc<false>, dim3(wave_grid(a.B)), dim3(64), lds, (hipStream_t)stream, a);
    rc = hip_check(hipGetLastError(), "f16_mpc_batch launch");
  } else {
    if (big) hipLaunchKernelGGL((k_mpc<true, true>), dim3(wave_grid(a.B)), dim3(64), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_mpc<true>, dim3(wave_grid(a.B)), dim3(64), lds, (hipStream_t)stream, a);
    rc = hip_check(hipGetLastError(), "f16_mpc_batch setup launch");
#ifdef F16_EXP_STAMPB
    mode = 1;
#endif
    if (!rc && mode == 3) {
      MpcArgs w = a;
      w.mode = 0;
      rc = mpc_big_solve_launch(ctx, w, stream);
    }
    if (!rc && mode == 2) {
      // Dispatch order (see k_plan_order): the reference's closed loops call calc_MPC_action once per step on states that
      // move little, so the iteration counts of the previous call of the same batch size ON THE SAME STREAM predict this
      // one's; any order is valid, a stale one only loses the gain.  F16_MPC_DISPATCH_ORDER=0 keeps the caller's order.
      // A first call has no such counts: its order comes from the QPs themselves (mpc_first_order_launch: by ||q||_inf).
      // F16_MPC_DISPATCH_ORDER=first treats every call as a first one (benchmarks: what BASELINE config 4, one call on an unseen
      // batch, costs); =0 switches every ordering off.
      const char *ev = getenv("F16_MPC_DISPATCH_ORDER");
      const bool always_first = ev && ev[0] == 'f';
      f16_ctx::sched_entry *se = (ev && ev[0] == '0') ? nullptr : mpc_sched_entry(ctx, stream, a.B);
      if (se) { a.iters_out = se->buf; a.order = (se->valid && !always_first) ? se->buf + a.B : nullptr; }
      if (se && !a.order && a.B > 1024) {
        rc = mpc_first_order_launch(a, se->buf, se->buf + a.B, stream);
        if (!rc) a.order = se->buf + a.B;
      }
      if (!rc) rc = mpc_solve_dispatch(ctx, a, stream);
      if (!rc && se) {
        rc = mpc_plan_order_launch(se->buf, se->buf + a.B, a.B, a.s.check_every, stream);
        if (!rc) se->valid = 1;
      }
    }
  }
  if (keep && !rc) { *keep = block; return F16_OK; }
  const int rf = mpc_work_free(block, stream);
  return rc ? rc : rf;
}

extern "C" int f16_mpc_batch(f16_ctx *ctx, const double *Ad, const double *Bd, const double *Cd, const double *x,
                             const double *dem, double *u_cmd, double *u_seq, double *info, int32_t *status, long B,
                             long ld, int hzn, double dt, const f16_qp_settings *s, void *stream) {
  return f16_mpc_batch_w(ctx, Ad, Bd, Cd, x, dem, nullptr, nullptr, u_cmd, u_seq, info, status, B, ld, hzn, dt, s, stream);
}

extern "C" int f16_mpc_batch_w(f16_ctx *ctx, const double *Ad, const double *Bd, const double *Cd, const double *x,
                               const double *dem, const double *x_ref, const f16_mpc_weights *h_w, double *u_cmd, double *u_seq,
                               double *info, int32_t *status, long B, long ld, int hzn, double dt, const f16_qp_settings *s,
                               void *stream) {
  if (!ctx || !Ad || !Bd || !Cd || !x || (!dem && !x_ref) || !u_cmd || B < 0 || ld < B) return set_error(F16_EINVAL, "bad argument to f16_mpc_batch");
  if (B == 0) return F16_OK;
  MpcArgs a{};
  a.Ad = Ad; a.Bd = Bd; a.Cd = Cd; a.x = x; a.dem = dem; a.xref = x_ref; a.ucmd = u_cmd; a.useq = u_seq; a.info = info; a.status = status;
  a.B = B; a.ld = ld; a.N = hzn; a.dt = dt;
  if (int rc = mpc_fill_prob(&a.pb, h_w)) return rc;
  if (s) a.s = *s; else f16_qp_default_settings(&a.s);
  if (a.s.check_every < 1 || a.s.rho_every < 1 || !(a.s.rho >= 0) || !(a.s.sigma > 0) || a.s.scaling < 0 || a.s.scaling > 100 ||
      (a.s.scaling > 0 && !(a.s.rho > 0)))
    return set_error(F16_EINVAL, "bad QP settings (the automatic start value of rho, rho = 0, needs scaling = 0)");
  // the rho estimate is formed from the residuals of a termination test, so it can only run on an iteration that has one
  // (osqp.c adapts independently of check_termination; here an interval that is not a multiple would silently become the
  // least common multiple)
  if (a.s.adaptive_rho && a.s.rho_every % a.s.check_every != 0)
    return set_error(F16_EINVAL, "bad QP settings: rho_every must be a multiple of check_every");
  // solver selection: N <= 32 the register-resident / one-wavefront solvers, longer horizons the workgroup solver with its
  // operands in HBM; a negative max_iter forces the generic one-wave kernel (tests: the cross-check of every other solver)
  const bool generic = a.s.max_iter < 0;
  if (a.s.max_iter < 0) a.s.max_iter = -a.s.max_iter;
  return mpc_launch(ctx, a, stream, generic ? 0 : (hzn > FAST_MAXN ? 3 : 2));
}

// The reference's horizon sweep (env.py:426-436: calc_MPC_action(0, 0, 0, N) of the SAME states for N = 1..150) as a
// throughput call.  Horizons up to FAST_MAXN go through the one-shot path one after the other (milliseconds each); the long
// ones are built per horizon and then solved by ONE launch of the workgroup solver over every (horizon, aircraft) pair: the
// iteration counts of this family grow with N and spread widely (N = 150: mean 2,100, max 26,850), so a launch per horizon
// lasts as long as its slowest aircraft on B of the 256 CUs, while the pairs of all horizons together keep every CU busy.
// Outputs: u_cmd [hi - lo + 1][3][ld], info [..][4][ld] (may be null), status [..][ld] (may be null; OR-ed into).
extern "C" int f16_mpc_hzn_sweep(f16_ctx *ctx, const double *Ad, const double *Bd, const double *Cd, const double *x,
                                 const double *dem, double *u_cmd, double *info, int32_t *status, long B, long ld, int hzn_lo,
                                 int hzn_hi, double dt, const f16_qp_settings *s, void *stream) {
  if (!ctx || !Ad || !Bd || !Cd || !x || !dem || !u_cmd || B < 0 || ld < B || hzn_lo < 1 || hzn_hi < hzn_lo || hzn_hi > BIG_MAXN)
    return set_error(F16_EINVAL, "bad argument to f16_mpc_hzn_sweep (horizons 1..150)");
  if (B == 0) return F16_OK;
  f16_qp_settings st;
  if (s) st = *s; else f16_qp_default_settings(&st);
  int N = hzn_lo;
  for (; N <= hzn_hi && (N <= FAST_MAXN || st.max_iter < 0); ++N) {
    const size_t k = (size_t)(N - hzn_lo);
    if (int rc = f16_mpc_batch(ctx, Ad, Bd, Cd, x, dem, u_cmd + k * 3 * ld, nullptr, info ? info + k * 4 * ld : nullptr,
                               status ? status + k * ld : nullptr, B, ld, N, dt, &st, stream)) return rc;
  }
  if (N > hzn_hi) return F16_OK;
  MpcArgs a{};
  mpc_default_prob(&a.pb);
  a.Ad = Ad; a.Bd = Bd; a.Cd = Cd; a.x = x; a.dem = dem; a.B = B; a.ld = ld; a.dt = dt; a.s = st;
  if (a.s.check_every < 1 || a.s.rho_every < 1 || !(a.s.rho >= 0) || !(a.s.sigma > 0) || a.s.scaling < 0 || a.s.scaling > 100 ||
      (a.s.scaling > 0 && !(a.s.rho > 0)) || (a.s.adaptive_rho && a.s.rho_every % a.s.check_every != 0))
    return set_error(F16_EINVAL, "bad QP settings");
  if (int rc = mpc_lds_opt_in()) return rc;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (stream && hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
    return set_error(F16_EINVAL, "f16_mpc_hzn_sweep cannot be captured into a HIP graph (per-call workspace)");
  // groups of horizons, longest first, each within the workspace budget (N = 150: 4.45 MB per aircraft = mpc_big_sweep_job_doubles(150) x 8)
  static const size_t budget = [] { const char *e = getenv("F16_SWEEP_WS_GB"); const double g = e ? atof(e) : 0; return (size_t)((g > 0 ? g : 32.0) * (1ull << 30)); }();
  int hi = hzn_hi;
  while (hi >= N) {
    int lo = hi;
    size_t doubles = mpc_big_sweep_job_doubles(hi);
    while (lo - 1 >= N && (doubles + mpc_big_sweep_job_doubles(lo - 1)) * (size_t)B * sizeof(double) <= budget) doubles += mpc_big_sweep_job_doubles(--lo);
    void *block = nullptr;
    const size_t wbytes = doubles * (size_t)B * sizeof(double);
    if (int rc = hip_check(hipMallocFromPoolAsync(&block, wbytes + 256, ctx->pool, (hipStream_t)stream),
                           "hipMallocFromPoolAsync sweep workspace")) return rc;
    unsigned int *next = (unsigned int *)((char *)block + wbytes);      // work-queue counter of the solve launch
    int rc = hip_check(hipMemsetAsync(next, 0, 256, (hipStream_t)stream), "hipMemsetAsync sweep queue");
    size_t off = 0;
    for (int Nn = hi; Nn >= lo && !rc; --Nn) {           // the builds (one wavefront per aircraft each)
      MpcArgs b = a;
      const size_t np = (size_t)(3 * Nn) * (3 * Nn + 1) / 2, k = (size_t)(Nn - hzn_lo);
      b.N = Nn;
      b.Ppk = (double *)block + off * (size_t)B;
      b.ext = b.Ppk + np * (size_t)B;
      b.bigws = b.ext + mpc_ext_doubles(Nn) * (size_t)B;
      b.ucmd = u_cmd + k * 3 * ld;
      b.status = status ? status + k * ld : nullptr;
      const bool big = Nn > MAXN;
      const size_t lds = mpc_lds_doubles(Nn, true, big) * sizeof(double);
      if (big) hipLaunchKernelGGL((k_mpc<true, true>), dim3(wave_grid(B)), dim3(64), lds, (hipStream_t)stream, b);
      else hipLaunchKernelGGL(k_mpc<true>, dim3(wave_grid(B)), dim3(64), lds, (hipStream_t)stream, b);
      rc = hip_check(hipGetLastError(), "f16_mpc_hzn_sweep build launch");
      off += mpc_big_sweep_job_doubles(Nn);
    }
    // Queue order: the pairs of a previous sweep of the same horizons on this stream, costliest first (iterations x N^2) -- a
    // sweep on states that moved little then ends with its packed phase instead of waiting for a straggler taken late (the
    // first call: longest horizons first, 2.25 s at B = 64; repeated: 1.6 s; bench.py hzn_sweep).  Scheduling only.
    const long npairs = (long)(hi - lo + 1) * B;
    const char *ev = getenv("F16_MPC_DISPATCH_ORDER");
    const bool one_group = lo == N && hi == hzn_hi;        // (a sweep cut into many groups must not use up the context's few history slots)
    f16_ctx::sched_entry *se = (ev && ev[0] == '0') || npairs > 0x3fffffffL || !one_group ? nullptr
                                                                                          : mpc_sched_entry(ctx, stream, npairs, (lo << 16) | hi);
    if (!rc) rc = mpc_big_sweep_launch(ctx, a, lo, hi, (double *)block, u_cmd + (size_t)(lo - hzn_lo) * 3 * ld,
                                       info ? info + (size_t)(lo - hzn_lo) * 4 * ld : nullptr,
                                       status ? status + (size_t)(lo - hzn_lo) * ld : nullptr, next, se ? se->buf : nullptr,
                                       se && se->valid ? se->buf + npairs : nullptr, stream);
    if (!rc && se) {
      rc = mpc_big_sweep_order_launch(se->buf, se->buf + npairs, npairs, B, hi, stream);
      if (!rc) se->valid = 1;
    }
    const int rf = mpc_work_free(block, stream);
    if (rc || rf) return rc ? rc : rf;
    hi = lo - 1;
  }
  return F16_OK;
}

// ---- prepared plans: everything of calc_MPC_action that depends on the model only (the reference freezes the model at
// construction, env.py:49-60, but rebuilds the whole QP on every call) is computed once; a solve then costs the
// state-dependent vectors + the ADMM iterations.
struct f16_mpc_plan {
  f16_ctx *ctx;
  long B, ld;
  int N;
  double dt;
  f16_qp_settings s;
  double *buf;
  double *warm;        // x, z, y of the previous solve (allocated when warm start is switched on)
  bool warm_on, have_prev;
  int32_t *sched;      // [2][B]: iteration counts of the last solve | dispatch order of the next (longest first)
  bool have_order;
  void *last_stream;   // the stream of the creation / the last solve: what f16_mpc_plan_destroy waits for
  void *roll_sync;     // f16_rollout_mpc: ticket counter + per-aircraft progress counters (allocated by its first call)
  MpcArgs a;
};

static int plan_launch_build(f16_mpc_plan *p, MpcArgs &a, void *stream) {
  const size_t lds = mpc_lds_doubles(p->N, true) * sizeof(double);
  if (a.mode == 1) {      // once per device (the solve path may run under stream capture)
    if (int rc = mpc_lds_opt_in()) return rc;
  }
  hipLaunchKernelGGL(k_mpc<true>, dim3(wave_grid(a.B)), dim3(64), lds, (hipStream_t)stream, a);
  return hip_check(hipGetLastError(), "f16_mpc_plan build launch");
}

extern "C" int f16_mpc_plan_create(f16_ctx *ctx, f16_mpc_plan **plan, const double *Ad, const double *Bd, const double *Cd,
                                   long B, long ld, int hzn, double dt, const f16_qp_settings *s, void *stream) {
  return f16_mpc_plan_create_w(ctx, plan, Ad, Bd, Cd, nullptr, B, ld, hzn, dt, s, stream);
}

extern "C" int f16_mpc_plan_create_w(f16_ctx *ctx, f16_mpc_plan **plan, const double *Ad, const double *Bd, const double *Cd,
                                     const f16_mpc_weights *h_w, long B, long ld, int hzn, double dt, const f16_qp_settings *s,
                                     void *stream) {
  if (!ctx || !plan || !Ad || !Bd || !Cd || B < 1 || ld < B) return set_error(F16_EINVAL, "bad argument to f16_mpc_plan_create");
  MpcProb pb;
  if (int rc = mpc_fill_prob(&pb, h_w)) return rc;
  if (hzn < 1 || hzn > MAXN) return set_error(F16_EINVAL, "prepared plans need 1 <= hzn <= 40");
  f16_mpc_plan *p = new f16_mpc_plan();
  p->ctx = ctx; p->B = B; p->ld = ld; p->N = hzn; p->dt = dt;
  p->warm = nullptr; p->warm_on = false; p->have_prev = false; p->sched = nullptr; p->have_order = false; p->last_stream = stream;
  p->roll_sync = nullptr;
  if (s) p->s = *s; else f16_qp_default_settings(&p->s);
  if (p->s.check_every < 1 || p->s.rho_every < 1 || !(p->s.rho >= 0) || !(p->s.sigma > 0) || p->s.max_iter < 1 || p->s.scaling < 0 ||
      p->s.scaling > 100 || (p->s.scaling > 0 && !(p->s.rho > 0)) || (p->s.adaptive_rho && p->s.rho_every % p->s.check_every != 0)) {
    delete p;
    return set_error(F16_EINVAL, "bad QP settings (rho_every must be a multiple of check_every)");
  }
  const size_t np = (size_t)(3 * hzn) * (3 * hzn + 1) / 2;
  const bool wide = hzn > FAST_MAXN;                   // horizons 33..40: the model part is kept, every solve runs the workgroup solver
  if (wide) {                                          // its LDS opt-in now: the first solve of the plan may already be under capture
    if (int rc = mpc_big_opt_in()) { delete p; return rc; }
  }
  const size_t per = np + mpc_ext_doubles(hzn) + 2 * MPC_TILE_DOUBLES + (wide ? mpc_big_ws_doubles(hzn) : 0);      // P | extras | inverse (scaling = 0) | A'WA | long-horizon operands
  if (int rc = hip_check(hipMalloc(&p->buf, per * (size_t)B * sizeof(double)), "hipMalloc MPC plan")) { delete p; return rc; }
  if (int rc = hip_check(hipMalloc(&p->sched, 2 * (size_t)B * sizeof(int32_t)), "hipMalloc MPC plan")) { (void)hipFree(p->buf); delete p; return rc; }
  MpcArgs &a = p->a;
  a = MpcArgs{};
  a.pb = pb;
  a.Ad = Ad; a.Bd = Bd; a.Cd = Cd; a.B = B; a.ld = ld; a.N = hzn; a.dt = dt; a.s = p->s;
  a.Ppk = p->buf; a.ext = a.Ppk + np * (size_t)B;
  a.tiles = a.ext + mpc_ext_doubles(hzn) * (size_t)B;
  a.gramws = a.tiles + MPC_TILE_DOUBLES * (size_t)B;
  a.pblk = a.tiles;                                    // (equilibrated plans keep no inverse: the block is the wavefront solver's)
  a.bigws = wide ? a.gramws + MPC_TILE_DOUBLES * (size_t)B : nullptr;
  a.mode = 1;
  int rc = plan_launch_build(p, a, stream);
  // Without equilibration the start value of rho and the KKT factorisation depend on the model only and are cached too.
  // OSQP's equilibration depends on q (the cost scaling c looks at ||q||, and D, E at c), i.e. on the state of the call:
  // a plan with scaling > 0 keeps the model part (DARE, G_k, P) and redoes equilibration + factorisation per solve.
  if (!rc && p->s.scaling == 0 && !wide) rc = mpc_fast_solve_launch(ctx, a, stream);
  a.Ad = a.Bd = a.Cd = nullptr;                        // not retained
  if (rc) { (void)hipFree(p->buf); (void)hipFree(p->sched); delete p; return rc; }
  *plan = p;
  return F16_OK;
}

extern "C" int f16_mpc_plan_solve(f16_mpc_plan *p, const double *x, const double *dem, double *u_cmd, double *u_seq,
                                  double *info, int32_t *status, void *stream) {
  return f16_mpc_plan_solve_w(p, x, dem, nullptr, u_cmd, u_seq, info, status, stream);
}

extern "C" int f16_mpc_plan_solve_w(f16_mpc_plan *p, const double *x, const double *dem, const double *x_ref, double *u_cmd,
                                    double *u_seq, double *info, int32_t *status, void *stream) {
  if (!p || !x || (!dem && !x_ref) || !u_cmd) return set_error(F16_EINVAL, "bad argument to f16_mpc_plan_solve");
  MpcArgs a = p->a;
  a.x = x; a.dem = dem; a.xref = x_ref; a.ucmd = u_cmd; a.useq = u_seq; a.info = info; a.status = status;
  p->last_stream = stream;
  a.mode = 2;
  a.warm = p->warm_on ? p->warm : nullptr;
  a.warm_load = p->warm_on && p->have_prev;
  a.iters_out = p->sched;
  a.order = p->have_order ? p->sched + p->B : nullptr;
  if (int rc = plan_launch_build(p, a, stream)) return rc;
  if (!a.order && p->B > 1024 && p->N <= FAST_MAXN) {      // first solve of the plan: longest-first by ||q||_inf of the QPs just built
    if (int rc = mpc_first_order_launch(a, p->sched, p->sched + p->B, stream)) return rc;
    a.order = p->sched + p->B;
  }
  if (p->s.scaling > 0 || p->N > FAST_MAXN) a.mode = 0;   // nothing cached beyond the model part: full solver prologue
  if (p->N > FAST_MAXN) {
    a.warm = nullptr;
    if (int rc = mpc_big_solve_launch(p->ctx, a, stream)) return rc;
  } else if (int rc = mpc_solve_dispatch(p->ctx, a, stream)) return rc;
  if (int rc = mpc_plan_order_launch(p->sched, p->sched + p->B, p->B, p->s.check_every, stream)) return rc;
  p->have_order = true;
  p->have_prev = p->warm_on;
  return F16_OK;
}

extern "C" int f16_rollout_mpc(f16_mpc_plan *p, double *x, double *u, const double *dem, double *traj, double *cmd_traj,
                               int32_t *iters_traj, int32_t *status, int nsteps, int traj_every, double xcg, int fi_flag,
                               unsigned flags, void *stream) {
  if (!p || !x || !u || !dem) return set_error(F16_EINVAL, "bad argument to f16_rollout_mpc");
  if (p->N > WAVE_MAXN || p->s.scaling <= 0)
    return set_error(F16_EINVAL, "f16_rollout_mpc needs a plan with hzn <= 30 and equilibrated solves (scaling > 0); "
                                 "other plans run the host loop (f16_mpc_plan_solve + f16_rollout per step)");
  if (nsteps < 0 || (traj && (traj_every < 1 || nsteps % traj_every != 0)))
    return set_error(F16_EINVAL, "nsteps must be >= 0 and a multiple of traj_every >= 1 when traj is given");
  if (nsteps == 0) return F16_OK;
  if (!p->roll_sync) {
    if (int rc = hip_check(hipMalloc(&p->roll_sync, 8 + (size_t)p->B * sizeof(int32_t)), "hipMalloc f16_rollout_mpc counters")) return rc;
  }
  p->last_stream = stream;
  RolloutMpcCall c{};
  c.x = x; c.u = u; c.dem = dem; c.traj = traj; c.cmd_traj = cmd_traj; c.iters_traj = iters_traj; c.status = status;
  c.sync = p->roll_sync; c.T = nsteps; c.every = traj ? traj_every : nsteps + 1; c.xcg = xcg; c.fi = fi_flag; c.flags = flags;
  c.warm = p->warm_on ? p->warm : nullptr;
  c.warm_load = p->warm_on && p->have_prev;
  const int rc = mpc_wave_rollout_launch(p->ctx, p->a, c, stream);
  if (!rc) p->have_prev = p->warm_on;
  return rc;
}

extern "C" int f16_mpc_plan_warm_start(f16_mpc_plan *p, int on) {
  if (!p) return set_error(F16_EINVAL, "bad argument to f16_mpc_plan_warm_start");
  if (on && p->N > FAST_MAXN) return set_error(F16_EINVAL, "warm start needs hzn <= 32 (the long-horizon solver starts cold)");
  if (on && !p->warm) {
    if (int rc = hip_check(hipMalloc(&p->warm, (size_t)p->B * MPC_WARM_DOUBLES * sizeof(double)), "hipMalloc warm start")) return rc;
  }
  p->warm_on = on != 0;
  p->have_prev = false;
  return F16_OK;
}

extern "C" void f16_mpc_plan_destroy(f16_mpc_plan *p) {
  if (!p) return;
  // wait for the plan's own work only (the stream of its creation / last solve), not for every stream of the device
  (void)hipStreamSynchronize((hipStream_t)p->last_stream);
  (void)hipFree(p->buf);
  if (p->sched) (void)hipFree(p->sched);
  if (p->warm) (void)hipFree(p->warm);
  if (p->roll_sync) (void)hipFree(p->roll_sync);
  delete p;
}

extern "C" int f16_mpc_qp_debug(f16_ctx *ctx, const double *Ad, const double *Bd, const double *Cd, const double *x,
                                const double *dem, long b, long ld, int hzn, double dt, double *h_P, double *h_q, double *h_A,
                                double *h_l, double *h_u) {
  return f16_mpc_qp_debug_w(ctx, Ad, Bd, Cd, x, dem, nullptr, nullptr, b, ld, hzn, dt, h_P, h_q, h_A, h_l, h_u);
}

extern "C" int f16_mpc_qp_debug_w(f16_ctx *ctx, const double *Ad, const double *Bd, const double *Cd, const double *x,
                                  const double *dem, const double *x_ref, const f16_mpc_weights *h_w, long b, long ld, int hzn,
                                  double dt, double *h_P, double *h_q, double *h_A, double *h_l, double *h_u) {
  if (!ctx || !Ad || !Bd || !Cd || !x || (!dem && !x_ref) || b < 0 || b >= ld || !h_P || !h_q || !h_A || !h_l || !h_u)
    return set_error(F16_EINVAL, "bad argument to f16_mpc_qp_debug");
  f16_mpc_weights wd;
  f16_mpc_default_weights(&wd);
  const f16_mpc_weights *wq = h_w ? h_w : &wd;          // (the bounds of the dense QP below: any pattern)
  f16_mpc_weights wk = *wq;                              // (for the kernel: cost only -- its bounds are not used by the build)
  for (int i = 0; i < 9; ++i) { wk.x_lb[i] = wd.x_lb[i]; wk.x_ub[i] = wd.x_ub[i]; }
  for (int i = 0; i < 3; ++i) { wk.u_lb[i] = wd.u_lb[i]; wk.u_ub[i] = wd.u_ub[i]; wk.udot_lb[i] = wd.udot_lb[i]; wk.udot_ub[i] = wd.udot_ub[i]; }
  const int N = hzn, n = 3 * N, rows = 15 * N;
  if (N < 1 || N > BIG_MAXN) return set_error(F16_EINVAL, "horizon must be 1..150");
  const size_t np = (size_t)n * (n + 1) / 2;
  const size_t ndbg = mpc_ext_doubles(N);
  double *d_u = nullptr;
  int rc;
  if ((rc = hip_check(hipMalloc(&d_u, 3 * ld * sizeof(double)), "hipMalloc dbg u"))) return rc;
  MpcArgs a{};
  a.Ad = Ad; a.Bd = Bd; a.Cd = Cd; a.x = x; a.dem = dem; a.xref = x_ref; a.ucmd = d_u; a.B = b + 1; a.ld = ld; a.N = N; a.dt = dt;
  if ((rc = mpc_fill_prob(&a.pb, &wk))) { (void)hipFree(d_u); return rc; }
  f16_qp_default_settings(&a.s);
  void *block = nullptr;
  rc = mpc_launch(ctx, a, nullptr, 1, &block);      // build only; the workspace stays ours until read back
  std::vector<double> dbg(ndbg), Ppk(np), xcol(18);
  if (!rc) rc = hip_check(hipDeviceSynchronize(), "sync");
  if (!rc) rc = hip_check(hipMemcpy(dbg.data(), a.ext + ndbg * (size_t)b, ndbg * sizeof(double), hipMemcpyDeviceToHost), "copy ext");
  if (!rc) rc = hip_check(hipMemcpy(Ppk.data(), a.Ppk + np * (size_t)b, np * sizeof(double), hipMemcpyDeviceToHost), "copy P");
  for (int k = 0; k < 18 && !rc; ++k)
    rc = hip_check(hipMemcpy(&xcol[k], x + k * ld + b, sizeof(double), hipMemcpyDeviceToHost), "copy x");
  (void)hipFree(d_u);
  (void)mpc_work_free(block, nullptr);
  if (rc) return rc;
  // reference-format QP (utils.py:111-165): P dense, A = [CC; I; D] (15N x 3N), l/u with +-inf rows kept
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) h_P[i * n + j] = i >= j ? Ppk[(size_t)i * (i + 1) / 2 + j] : Ppk[(size_t)j * (j + 1) / 2 + i];
  for (int i = 0; i < n; ++i) h_q[i] = dbg[i];
  const double *G = dbg.data() + n, *pred = dbg.data() + n + N * 27;
  for (size_t i = 0; i < (size_t)rows * n; ++i) h_A[i] = 0.0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j <= i; ++j)
      for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 3; ++c) h_A[(size_t)(9 * i + r) * n + 3 * j + c] = G[(i - j) * 27 + r * 3 + c];
  for (int k = 0; k < n; ++k) {
    h_A[(size_t)(9 * N + k) * n + k] = 1.0;
    h_A[(size_t)(12 * N + k) * n + k] = 1.0;
    if (k >= 3) h_A[(size_t)(12 * N + k) * n + k - 3] = -1.0;
  }
  const double *xlb = wq->x_lb, *xub = wq->x_ub, *ulb = wq->u_lb, *uub = wq->u_ub, *rlb = wq->udot_lb, *rub = wq->udot_ub;
  for (int i = 0; i < N; ++i)
    for (int r = 0; r < 9; ++r) {
      h_l[9 * i + r] = xlb[r] - pred[9 * i + r];
      h_u[9 * i + r] = xub[r] - pred[9 * i + r];
    }
  for (int k = 0; k < n; ++k) {
    h_l[9 * N + k] = ulb[k % 3];
    h_u[9 * N + k] = uub[k % 3];
    if (k < 3) {
      h_l[12 * N + k] = xcol[13 + k] + rlb[k] * dt;
      h_u[12 * N + k] = xcol[13 + k] + rub[k] * dt;
    } else {
      h_l[12 * N + k] = rlb[k % 3];
      h_u[12 * N + k] = rub[k % 3];
    }
  }
  return F16_OK;
}
