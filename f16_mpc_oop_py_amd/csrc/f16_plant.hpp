// f16_plant.hpp -- device-side F-16 plant for gfx950: one wavefront lane = one aircraft.
//
// What it computes (same arithmetic, expression by expression, as the reference CPU path):
//   plant<>()      C/nlplant.c:23-457 Nlplant (hifi: C/hifi_F16_AeroData.c:1871-1934 group functions over
//                  C/mexndinterp.c:97-265 interpolation; lofi: C/lofi_F16_AeroData.c:12-368)
//   atmos_dev()    C/nlplant.c:467-490
//   actuators      utils.py:289-330 (upd_thrust/dstab/ail/rud/lef)
//   calc_xdot()    env.py:65-103          calc_xdot_na(): env.py:152-193
// How it differs structurally from the reference (results are unchanged):
//   * the 5 axis brackets (alpha on ALPHA1/ALPHA2, beta, el on DH1/DH2) are found once and shared by
//     all lookups; the reference searches them again inside each of its 58 interpn() calls;
//   * 58 interpn() calls collapse to 48 distinct evaluations on node-major table groups staged in LDS
//     (f16_tables.h); `_Cy` x4, `_Cn/_Cl(..,0)` x3 are evaluated once;
//   * an exact grid-node hit needs no special case: lambda is exactly 0 (or 1), and
//     lambda*f2 + (1-lambda)*f1 then returns f1 (or f2) exactly -- the value the reference's
//     degenerate-axis rule (mexndinterp.c:126-133,195-200) produces;
//   * no malloc, no file I/O, sin/cos evaluated once per angle (accels() re-evaluates the same
//     arguments, C/nlplant.c:525-545).
// Off-grid coordinates (UB in the reference) are clamped to the grid edge and flagged.
#pragma once
#include <hip/hip_runtime.h>

#include "f16_tables.h"

namespace f16 {

constexpr int ST_ALPHA1 = 1, ST_ALPHA2 = 2, ST_BETA = 4, ST_EL = 8, ST_ENVELOPE = 16, ST_NONFINITE = 32;
constexpr unsigned FLAG_FIX_CLR = 1u, FLAG_NO_ENVELOPE = 2u;

#define F16_DEV __device__ __forceinline__

// Division by a compile-time constant.  Strict build: IEEE division as the reference does.  Default build
// (F16_FAST_DIV): multiplication by the correctly rounded reciprocal -- <= 1 ulp from the quotient, ~10 VALU
// instructions cheaper each (an fp64 division is a 10-deep dependent chain on gfx950).
#ifdef F16_FAST_DIV
#define F16_DIVC(x, c) ((x) * (1.0 / (c)))
#else
#define F16_DIVC(x, c) ((x) / (c))
#endif

// 1/x for a normal, finite x.  Strict build: IEEE division.  Default build: v_rcp_f64 + two Newton steps (<= 1 ulp),
// 5 instructions instead of the ~15 of the division sequence.
__device__ __forceinline__ double f16_rcp(double x) {
#ifdef F16_FAST_DIV
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
#else
  return 1.0 / x;
#endif
}

struct Axis {
  int j;       // lower node of the bracketing cell, 0 .. n-2
  double l;    // lambda = (v - X[j]) / (X[j+1] - X[j])     (mexndinterp.c:196)
  double m;    // 1 - lambda
};

F16_DEV double lerp(double f1, double f2, const Axis &a) { return a.l * f2 + a.m * f1; }  // mexndinterp.c:197

// C/nlplant.c:467-490
F16_DEV void atmos_dev(double alt, double vt, double &mach, double &qbar, double &ps) {
  const double rho0 = 2.377e-3;
  const double tfac = 1 - .703e-5 * alt;
  double temp = 519.0 * tfac;
  if (alt >= 35000.0) temp = 390;
#ifdef F16_FAST_POW
  // tfac^4.14 = (tfac^2)^2 * exp(0.14 log tfac): |0.14 log tfac| < 0.2 on the flight envelope keeps the
  // exp-of-log error below 1 ulp; two exact-to-0.5ulp squarings on top => <= 2 ulp vs libm pow.
  const double t2 = tfac * tfac;
  const double rho = rho0 * ((t2 * t2) * exp(0.14 * log(tfac)));
#else
  const double rho = rho0 * pow(tfac, 4.14);
#endif
  mach = vt / sqrt(1.4 * 1716.3 * temp);
  qbar = .5 * rho * (vt * vt);
  ps = 1715.0 * rho * temp;
  if (ps == 0) ps = 1715;
}


#ifdef F16_FAST_TRIG
// Branch-free sin/cos pair: 3-part Cody-Waite reduction by pi/2 (exact for |x| < ~1e6 rad) + Taylor polynomials on
// |r| <= pi/4 (coefficients are the exact 1/k! values; truncation < 5e-17).  Measured <= 2 ulp from libm on
// [-1e5, 1e5].  Unlike the libm call it contains no large-argument branch, so the five independent evaluations of a
// plant step sit in ONE basic block and the scheduler interleaves their dependency chains (the kernel is bound by
// fp64 dependent-issue latency at one wave per SIMD).
F16_DEV void sincos_bf(double x, double *sn, double *cs) {
  const double n = rint(x * 6.36619772367581382433e-01);
  double r = fma(-n, 1.57079632673412561417e+00, x);
  r = fma(-n, 6.07710050630396597660e-11, r);
  r = fma(-n, 2.02226624879595063154e-21, r);
  const double z = r * r;
  const double S[8] = {-1.66666666666666657e-01, 8.33333333333333322e-03, -1.98412698412698413e-04, 2.75573192239858925e-06, -2.50521083854417202e-08, 1.60590438368216133e-10, -7.64716373181981641e-13, 2.81145725434552060e-15};
  const double C[9] = {-5.00000000000000000e-01, 4.16666666666666644e-02, -1.38888888888888894e-03, 2.48015873015873016e-05, -2.75573192239858883e-07, 2.08767569878681002e-09, -1.14707455977297245e-11, 4.77947733238738525e-14, -1.56192069685862253e-16};
  double p = S[7], q = C[8];
#pragma unroll
  for (int k = 6; k >= 0; --k) p = fma(p, z, S[k]);
#pragma unroll
  for (int k = 7; k >= 0; --k) q = fma(q, z, C[k]);
  const double s0 = fma(r * z, p, r), c0 = fma(z, q, 1.0);
  const int k = (int)n & 3;
  const double a = (k & 1) ? c0 : s0, b = (k & 1) ? s0 : c0;
  *sn = (k & 2) ? -a : a;
  *cs = ((k + 1) & 2) ? -b : b;
}
#define F16_SINCOS(x, s, c) sincos_bf(x, s, c)
#else
#define F16_SINCOS(x, s, c) sincos(x, s, c)
#endif

struct Aero {  // everything C/nlplant.c:185-240 (or :245-323) hands to the coefficient build-up
  double Cx, Cz, Cm, Cy, Cn, Cl;
  double Cxq, Cyr, Cyp, Czq, Clr, Clp, Cmq, Cnr, Cnp;
  double dCx_lef, dCz_lef, dCm_lef, dCy_lef, dCn_lef, dCl_lef;
  double dCxq_lef, dCyr_lef, dCyp_lef, dCzq_lef, dClr_lef, dClp_lef, dCmq_lef, dCnr_lef, dCnp_lef;
  double dCy_r30, dCn_r30, dCl_r30;
  double dCy_a20, dCy_a20_lef, dCn_a20, dCn_a20_lef, dCl_a20, dCl_a20_lef;
  double dCnbeta, dClbeta, dCm, eta_el;
};

// ---- hifi lookups: hifi_C, hifi_damping, hifi_C_lef, hifi_damping_lef, hifi_rudder, hifi_ailerons, hifi_other_coeffs
// (C/hifi_F16_AeroData.c:1871-1934) fused, in scheduling phases.  With one wavefront per SIMD nothing hides an LDS
// round trip, and left to itself the compiler emits lookups in source order (read four corners, wait, interpolate, next
// table: ~14 dependent round trips per role).  Here: (1) ALL breakpoint reads, (2) cell indices -> ALL table-corner
// reads, with the lambda divisions issued behind them, (3) the interpolation arithmetic.
//
// Bracketing v on breakpoints X[0..n-1] (monotone): `guess` must be within one cell of the true cell (the arithmetic
// guesses are exact up to a rounding at a node); the ends and the four breakpoints around the guess are read at once
// (one LDS round trip), the cell is then fixed up with compares and selects.
struct BrRaw { double lo, hi, xm, xg, xg1, xp; int g; };
template <typename TP>
F16_DEV BrRaw br_load(TP X, int n, int guess) {                 // the six breakpoint reads
  BrRaw r;
  r.g = min(max(guess, 0), n - 2);
  r.lo = X[0]; r.hi = X[n - 1];
  r.xm = X[max(r.g - 1, 0)]; r.xg = X[r.g]; r.xg1 = X[r.g + 1]; r.xp = X[min(r.g + 2, n - 1)];
  return r;
}
struct BrCell { int j; double v, x0, x1; };
F16_DEV BrCell br_cell(const BrRaw &r, int n, double v, bool &off) {   // clamp to the grid, cell fix-up
  off = !(v >= r.lo && v <= r.hi);
  v = fmin(fmax(v, r.lo), r.hi);
  const bool down = r.g > 0 && v < r.xg, up = r.g < n - 2 && v >= r.xg1;
  BrCell c;
  c.j = r.g - (down ? 1 : 0) + (up ? 1 : 0);
  c.v = v;
  c.x0 = down ? r.xm : (up ? r.xg1 : r.xg);
  c.x1 = down ? r.xg : (up ? r.xp : r.xg1);
  return c;
}
F16_DEV Axis br_axis(const BrCell &c) {                          // lambda = (v - X[j]) / (X[j+1] - X[j]), mexndinterp.c:196
  Axis a;
  a.j = c.j;
#ifdef F16_FAST_DIV
  // (the Newton-refined reciprocal is <= 1 ulp off 1/d, so d * rcp(d) need not be exactly 1: a hit on the LAST node of an
  //  axis -- the only node reached as the upper end of a cell -- is pinned, every other node gives lambda = 0 by itself)
  a.l = c.v == c.x1 ? 1.0 : (c.v - c.x0) * f16_rcp(c.x1 - c.x0);
#else
  a.l = (c.v - c.x0) / (c.x1 - c.x0);
#endif
  a.m = 1 - a.l;
  return a;
}
struct Q4 { double f00, f10, f01, f11; };                        // corners (a,b), (a+1,b), (a,b+1), (a+1,b+1)
template <typename TP>
F16_DEV Q4 ld4(TP p, int sa, int sb) { Q4 c; c.f00 = p[0]; c.f10 = p[sa]; c.f01 = p[sb]; c.f11 = p[sb + sa]; return c; }
F16_DEV double bil4(const Q4 &c, const Axis &a, const Axis &b) {   // alpha collapsed first, then beta (mexndinterp.c:178-209)
  return lerp(lerp(c.f00, c.f10, a), lerp(c.f01, c.f11, a), b);
}
// Default build: a bilinear interpolation as a weighted sum of the four corners.  The weights depend on the (alpha, beta)
// cell only, so the 31 bilinear evaluations of a plant call share two weight sets (ALPHA1 x BETA1, ALPHA2 x BETA1): four
// operations each instead of the six of the nested form, <= 2 ulp away from it, and still EXACT on a grid node (the
// weights are then 0 / 1).  F16_STRICT keeps the reference's nested form (mexndinterp.c:178-209).
struct W4 { double w00, w10, w01, w11; };
F16_DEV W4 bil_weights(const Axis &a, const Axis &b) { W4 w; w.w00 = a.m * b.m; w.w10 = a.l * b.m; w.w01 = a.m * b.l; w.w11 = a.l * b.l; return w; }
#ifdef F16_FAST_DIV
F16_DEV double bil4w(const Q4 &c, const Axis &, const Axis &, const W4 &w) { return w.w00 * c.f00 + w.w10 * c.f10 + w.w01 * c.f01 + w.w11 * c.f11; }
#else
F16_DEV double bil4w(const Q4 &c, const Axis &a, const Axis &b, const W4 &) { return bil4(c, a, b); }
#endif
#ifndef F16_PHASE_MASK
#define F16_PHASE_MASK 0x7        // LDS / memory instructions stay in their phase, ALU instructions may float (2 % over 0)
#endif
#define F16_PHASE() __builtin_amdgcn_sched_barrier(F16_PHASE_MASK)

F16_DEV int alpha_guess(double alpha) { return (int)((fmin(fmax(alpha, -20.0), 90.0) + 20.0) * 0.2); }
F16_DEV int beta_guess(double beta) {
  const double bc = fmin(fmax(beta, -30.0), 30.0);
  return bc < -10.0 ? (int)((bc + 30.0) * 0.2) : (bc < 10.0 ? 4 + (int)((bc + 10.0) * 0.5) : 14 + (int)((bc - 10.0) * 0.2));
}


// The six totals of C/nlplant.c:333-377 for ONE lane = one aircraft, in four load / compute phases whose live ranges
// stay inside the register file (written as one expression tree, all 168 vertex reads are hoisted to the top of
// the step and the allocator parks ~200 values in AGPRs: ~360 accvgpr moves per step, and a 512-lane workgroup spills
// to scratch).  Same terms, same differences-first order.  Measured at B = 262,144: 14.2 -> 15.8 G steps/s with 256-lane
// workgroups, 5.7 -> 17.1 with 512-lane ones (two waves per SIMD).
struct TotalsOut { double Cx, Cz, Cm, Cy, Cn, Cl; };

// Two table images serve the one-lane plant (f16_tables.h): fp64 values (TP = const double *) and scaled integers
// (TP = TabI32: half the LDS bytes per vertex, one exact int -> double conversion per vertex).  Lay<TP> gives the group
// offsets / node strides of the image and a view whose operator[] yields a double; with the integer image the lookups
// run on k = 1e5 x value and the six totals -- linear in the table values once eta_el is scaled -- are scaled at the end.
struct TabI32 { const int *base; };
struct I32View {
  const int *p;
  F16_DEV I32View operator+(int o) const { return I32View{p + o}; }
  F16_DEV double operator[](int i) const { return (double)p[i]; }
};
template <typename TP> struct Lay {
  static constexpr int G3A = OFF_G3A, SG3A = S_G3A, G3B = OFF_G3B, SG3B = S_G3B, G2A = OFF_G2A, SG2A = S_G2A, G2B = OFF_G2B,
                       SG2B = S_G2B, G1A = OFF_G1A, SG1A = S_G1A, G1B = OFF_G1B, SG1B = S_G1B, ETA = OFF_ETA;
  static constexpr bool SCALED = false;
  static F16_DEV TP tab(TP T) { return T; }
  static F16_DEV TP bp(TP T) { return T; }
};
template <> struct Lay<TabI32> {
  static constexpr int G3A = i32::OFF_G3A, SG3A = i32::S_G3A, G3B = i32::OFF_G3B, SG3B = i32::S_G3B, G2A = i32::OFF_G2A,
                       SG2A = i32::S_G2A, G2B = i32::OFF_G2B, SG2B = i32::S_G2B, G1A = i32::OFF_G1A, SG1A = i32::S_G1A,
                       G1B = i32::OFF_G1B, SG1B = i32::S_G1B, ETA = i32::OFF_ETA;
  static constexpr bool SCALED = true;
  static F16_DEV I32View tab(TabI32 T) { return I32View{T.base}; }
  static F16_DEV const double *bp(TabI32 T) { return reinterpret_cast<const double *>(T.base); }
};

template <typename TP>
F16_DEV void aero_totals_phased(TP T0, const double *xu, double xcg, unsigned flags, TotalsOut &t, int &status) {
  using L = Lay<TP>;
  const auto T = L::tab(T0);
  const auto BP = L::bp(T0);
  constexpr int S_G3A = L::SG3A, S_G3B = L::SG3B, S_G2A = L::SG2A, S_G2B = L::SG2B, S_G1A = L::SG1A, S_G1B = L::SG1B;
  constexpr int OFF_G3A = L::G3A, OFF_G3B = L::G3B, OFF_G2A = L::G2A, OFF_G2B = L::G2B, OFF_G1A = L::G1A, OFF_G1B = L::G1B,
                OFF_ETA = L::ETA;
  const double B = 30.0, cbar = 11.32, xcgr = 0.35, r2d = 180.0 / 3.141592653589793;
  double vt = xu[6];
  if (vt <= 0.01) vt = 0.01;
  const double alpha = xu[7] * r2d, beta = xu[8] * r2d, P = xu[9], Q = xu[10], R = xu[11], el = xu[13];
  const double dail = F16_DIVC(xu[14], 21.5), drud = F16_DIVC(xu[15], 30.0), dlef = 1 - F16_DIVC(xu[16], 25.0);
  const double r2vt = f16_rcp(2 * vt), kq = cbar * r2vt, kb = B * r2vt;
  // (1) breakpoints
  const BrRaw ra = br_load(BP + OFF_BP_A1, N_A1, alpha_guess(alpha));
  const BrRaw rb = br_load(BP + OFF_BP_B1, N_B1, beta_guess(beta));
  const BrRaw r1 = br_load(BP + OFF_BP_D1, N_D1, (el >= -10.0) + (el >= 0.0) + (el >= 10.0));
  const BrRaw r2 = br_load(BP + OFF_BP_D2, N_D2, (int)(el >= 0.0));
  const double a45 = BP[OFF_BP_A1 + N_A2 - 1];
  F16_PHASE();
  bool offa, offb, off1, off2;
  const BrCell ca = br_cell(ra, N_A1, alpha, offa), cb = br_cell(rb, N_B1, beta, offb);
  const BrCell c1 = br_cell(r1, N_D1, el, off1), c2 = br_cell(r2, N_D2, el, off2);
  if (offa) status |= ST_ALPHA1 | ST_ALPHA2;
  if (offb) status |= ST_BETA;
  if (off1) status |= ST_EL;
  const bool hi_a = ca.j > N_A2 - 2;
  if (hi_a && alpha > a45) status |= ST_ALPHA2;
  const int j2 = hi_a ? N_A2 - 2 : ca.j;
  const int n1 = cb.j * N_A1 + ca.j, n2 = cb.j * N_A2 + j2;
  const Axis a1 = br_axis(ca), b = br_axis(cb), d1 = br_axis(c1), d2 = br_axis(c2);
  Axis a2 = a1;
  if (hi_a) { a2.j = N_A2 - 2; a2.l = 1.0; a2.m = 0.0; }
  // (each weight set is formed inside the phase that uses it -- W1 twice, W2 in phase 4 only -- to keep live ranges short:
  //  the 512-lane workgroup has 256 registers per lane)
  const auto g = T + OFF_G1A + ca.j * S_G1A, h = T + OFF_G1B + j2 * S_G1B;
  // (2) longitudinal: G3A (3-D + el = 0 plane), G2B lef, pitch damping, eta_el
  {
    constexpr int SA = S_G3A, SB = S_G3A * N_A1, SD = S_G3A * N_A1 * N_B1;
    const auto p = T + OFF_G3A + n1 * SA;
    Q4 qlo[3], qhi[3], q0[3], ql[3];
    double g0[3], g1[3], h0[3], h1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      qlo[k] = ld4(p + k + c1.j * SD, SA, SB); qhi[k] = ld4(p + k + (c1.j + 1) * SD, SA, SB);
      q0[k] = ld4(p + k + D1_ZERO_NODE * SD, SA, SB);
      ql[k] = ld4(T + OFF_G2B + n2 * S_G2B + k, S_G2B, S_G2B * N_A2);
      g0[k] = g[3 * k]; g1[k] = g[S_G1A + 3 * k]; h0[k] = h[3 * k]; h1[k] = h[S_G1B + 3 * k];
    }
    const double m0 = g[11], m1 = g[S_G1A + 11], e0 = T[OFF_ETA + c1.j], e1 = T[OFF_ETA + c1.j + 1];
    F16_PHASE();
    const W4 W1 = bil_weights(a1, b);
    double Cf[3], dC[3], dQ[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      Cf[k] = lerp(bil4w(qlo[k], a1, b, W1), bil4w(qhi[k], a1, b, W1), d1);
      dC[k] = bil4(ql[k], a2, b) - bil4w(q0[k], a1, b, W1);                            // hifi_C_lef
      const double Cq = lerp(g0[k], g1[k], a1), dq = lerp(h0[k], h1[k], a2);
      dQ[k] = kq * (Cq + (k == 1 ? dC[k] : dq) * dlef);                        // dXdQ, dZdQ (reference quirk :339), dMdQ
    }
    // same order of additions as C/nlplant.c:333-347
    t.Cx = Cf[0] + dC[0] * dlef + dQ[0] * Q;
    t.Cz = Cf[1] + dC[1] * dlef + dQ[1] * Q;
    t.Cm = Cf[2] * (L::SCALED ? i32::SCALE * lerp(e0, e1, d1) : lerp(e0, e1, d1)) + t.Cz * (xcgr - xcg) + dC[2] * dlef + dQ[2] * Q + lerp(m0, m1, a1);
  }
  F16_PHASE();
  // (3) lateral, first half: G3B (3-D + plane), G2A
  double base[3], base0[3], dr30[3], da20[3];
  {
    constexpr int SA3 = S_G3B, SB3 = S_G3B * N_A1, SD3 = S_G3B * N_A1 * N_B1;
    const auto p3 = T + OFF_G3B + n1 * SA3;
    Q4 qlo[2], qhi[2], q0[2], qa[7];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      qlo[k] = ld4(p3 + k + c2.j * SD3, SA3, SB3); qhi[k] = ld4(p3 + k + (c2.j + 1) * SD3, SA3, SB3);
      q0[k] = ld4(p3 + k + D2_ZERO_NODE * SD3, SA3, SB3);
    }
    const auto pa = T + OFF_G2A + n1 * S_G2A;
#pragma unroll
    for (int k = 0; k < 7; ++k) qa[k] = ld4(pa + k, S_G2A, S_G2A * N_A1);
    F16_PHASE();
    const W4 W1 = bil_weights(a1, b);
    const double Cy = bil4w(qa[0], a1, b, W1);
    base[0] = Cy; base0[0] = Cy;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      base[k + 1] = lerp(bil4w(qlo[k], a1, b, W1), bil4w(qhi[k], a1, b, W1), d2);
      base0[k + 1] = bil4w(q0[k], a1, b, W1);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      dr30[k] = bil4w(qa[1 + k], a1, b, W1) - base0[k];                              // hifi_rudder
      da20[k] = bil4w(qa[4 + k], a1, b, W1) - base0[k];                              // hifi_ailerons
    }
  }
  F16_PHASE();
  // (4) lateral, second half: G2B lef tables, yaw / roll damping, sideslip corrections
  {
    const auto pb = T + OFF_G2B + n2 * S_G2B;
    Q4 ql[3], qal[3];
    double r0[3], r1v[3], p0[3], p1[3], hr0[3], hr1[3], hp0[3], hp1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      ql[k] = ld4(pb + 3 + k, S_G2B, S_G2B * N_A2); qal[k] = ld4(pb + 6 + k, S_G2B, S_G2B * N_A2);
      const int ir = k == 0 ? 1 : (k == 1 ? 7 : 4);
      r0[k] = g[ir]; r1v[k] = g[S_G1A + ir]; p0[k] = g[ir + 1]; p1[k] = g[S_G1A + ir + 1];
      hr0[k] = h[ir]; hr1[k] = h[S_G1B + ir]; hp0[k] = h[ir + 1]; hp1[k] = h[S_G1B + ir + 1];
    }
    const double nb0 = g[9], nb1 = g[S_G1A + 9], lb0 = g[10], lb1 = g[S_G1A + 10];
    F16_PHASE();
    const W4 W2 = bil_weights(a2, b);
    double dl[3], dA[3], dR[3], dP[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double Clef = bil4w(ql[k], a2, b, W2), Ca20lef = bil4w(qal[k], a2, b, W2);
      double Cr = lerp(r0[k], r1v[k], a1);
      if (k == 2 && !(flags & FLAG_FIX_CLR)) Cr = 0.0;                          // reference defect: _CLr never loaded
      const double Cp = lerp(p0[k], p1[k], a1), dCr = lerp(hr0[k], hr1[k], a2), dCp = lerp(hp0[k], hp1[k], a2);
      dl[k] = Clef - base0[k];                                                  // hifi_C_lef
      dA[k] = da20[k] + (Ca20lef - Clef - da20[k]) * dlef;                      // dYdail, dNdail, dLdail
      dR[k] = kb * (Cr + dCr * dlef);
      dP[k] = kb * (Cp + dCp * dlef);
    }
    // same order of additions as C/nlplant.c:353-377
    t.Cy = base[0] + dl[0] * dlef + dA[0] * dail + dr30[0] * drud + dR[0] * R + dP[0] * P;
    t.Cn = base[1] + dl[1] * dlef - t.Cy * (xcgr - xcg) * (cbar / B) + dA[1] * dail + dr30[1] * drud + dR[1] * R + dP[1] * P +
           lerp(nb0, nb1, a1) * beta;
    t.Cl = base[2] + dl[2] * dlef + dA[2] * dail + dr30[2] * drud + dR[2] * R + dP[2] * P + lerp(lb0, lb1, a1) * beta;
  }
  if (L::SCALED) { t.Cx *= i32::SCALE; t.Cz *= i32::SCALE; t.Cm *= i32::SCALE; t.Cy *= i32::SCALE; t.Cn *= i32::SCALE; t.Cl *= i32::SCALE; }
}

// The same totals as sums of four partial triples, so that four wavefronts can each look up one table family
// (k_rollout_4w), every part in load / compute phases:
//   PART 1  longitudinal, 3-D / 2-D tables  {Cx, Cz, Cm}_s        PART 3  longitudinal damping (1-D)  {Cx, Cz, Cm}_d
//   PART 2  lateral, 3-D / 2-D tables       {Cy, Cn, Cl}_s        PART 4  lateral damping (1-D)       {Cy, Cn, Cl}_d
// compose_totals() adds them and applies the cg-offset couplings of C/nlplant.c:347,367.  The terms are exactly those
// of C/nlplant.c:333-377; only the order of the additions differs from the single-expression form (ulp level).
template <int PART, typename TP>
F16_DEV void aero_part(TP T, const double *xu, unsigned flags, double *out, int &status) {
  const double B = 30.0, cbar = 11.32, r2d = 180.0 / 3.141592653589793;
  double vt = xu[6];
  if (vt <= 0.01) vt = 0.01;
  const double alpha = xu[7] * r2d, beta = xu[8] * r2d, P = xu[9], Q = xu[10], R = xu[11], el = xu[13];
  const double dail = F16_DIVC(xu[14], 21.5), drud = F16_DIVC(xu[15], 30.0), dlef = 1 - F16_DIVC(xu[16], 25.0);
  const double r2vt = f16_rcp(2 * vt), kq = cbar * r2vt, kb = B * r2vt;
  constexpr bool TWO_D = PART == 1 || PART == 2;                 // needs beta (and an elevator axis)
  // (1) breakpoints
  const BrRaw ra = br_load(T + OFF_BP_A1, N_A1, alpha_guess(alpha));
  const double a45 = T[OFF_BP_A1 + N_A2 - 1];
  BrRaw rb = ra, rd = ra;
  if (TWO_D) {
    rb = br_load(T + OFF_BP_B1, N_B1, beta_guess(beta));
    rd = PART == 1 ? br_load(T + OFF_BP_D1, N_D1, (el >= -10.0) + (el >= 0.0) + (el >= 10.0))
                   : br_load(T + OFF_BP_D2, N_D2, (int)(el >= 0.0));
  }
  F16_PHASE();
  bool offa, offb = false, offd = false;
  const BrCell ca = br_cell(ra, N_A1, alpha, offa);
  BrCell cb = ca, cd = ca;
  if (TWO_D) { cb = br_cell(rb, N_B1, beta, offb); cd = br_cell(rd, PART == 1 ? N_D1 : N_D2, el, offd); }
  if (offa) status |= ST_ALPHA1 | ST_ALPHA2;
  if (offb) status |= ST_BETA;
  if (offd) status |= ST_EL;
  const bool hi_a = ca.j > N_A2 - 2;
  if (hi_a && alpha > a45) status |= ST_ALPHA2;
  const int j2 = hi_a ? N_A2 - 2 : ca.j;
  const int n1 = cb.j * N_A1 + ca.j, n2 = cb.j * N_A2 + j2;
  TP g = T + OFF_G1A + ca.j * S_G1A, h = T + OFF_G1B + j2 * S_G1B;
  if (PART == 1) {
    constexpr int SA = S_G3A, SB = S_G3A * N_A1, SD = S_G3A * N_A1 * N_B1;
    TP p = T + OFF_G3A + n1 * SA;
    Q4 qlo[3], qhi[3], q0[3], ql[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      qlo[k] = ld4(p + k + cd.j * SD, SA, SB); qhi[k] = ld4(p + k + (cd.j + 1) * SD, SA, SB);
      q0[k] = ld4(p + k + D1_ZERO_NODE * SD, SA, SB);
      ql[k] = ld4(T + OFF_G2B + n2 * S_G2B + k, S_G2B, S_G2B * N_A2);
    }
    const double e0 = T[OFF_ETA + cd.j], e1 = T[OFF_ETA + cd.j + 1];
    F16_PHASE();
    const Axis a1 = br_axis(ca), b = br_axis(cb), d1 = br_axis(cd);
    Axis a2 = a1;
    if (hi_a) { a2.j = N_A2 - 2; a2.l = 1.0; a2.m = 0.0; }
    const W4 W1 = bil_weights(a1, b), W2 = bil_weights(a2, b);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double Cf = lerp(bil4w(qlo[k], a1, b, W1), bil4w(qhi[k], a1, b, W1), d1);
      const double dC = bil4w(ql[k], a2, b, W2) - bil4w(q0[k], a1, b, W1);             // hifi_C_lef :1892-1899
      out[k] = (k == 2 ? Cf * lerp(e0, e1, d1) : Cf) + dC * dlef + (k == 1 ? kq * (dC * dlef) * Q : 0.0);   // dZdQ quirk
    }
  } else if (PART == 3) {
    double g0[3], g1[3], h0[3], h1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { g0[k] = g[3 * k]; g1[k] = g[S_G1A + 3 * k]; h0[k] = h[3 * k]; h1[k] = h[S_G1B + 3 * k]; }
    const double m0 = g[11], m1 = g[S_G1A + 11];
    F16_PHASE();
    const Axis a1 = br_axis(ca);
    Axis a2 = a1;
    if (hi_a) { a2.j = N_A2 - 2; a2.l = 1.0; a2.m = 0.0; }
    out[0] = kq * (lerp(g0[0], g1[0], a1) + lerp(h0[0], h1[0], a2) * dlef) * Q;
    out[1] = kq * lerp(g0[1], g1[1], a1) * Q;
    out[2] = kq * (lerp(g0[2], g1[2], a1) + lerp(h0[2], h1[2], a2) * dlef) * Q + lerp(m0, m1, a1);
  } else if (PART == 2) {
    constexpr int SA3 = S_G3B, SB3 = S_G3B * N_A1, SD3 = S_G3B * N_A1 * N_B1;
    TP p3 = T + OFF_G3B + n1 * SA3;
    Q4 qlo[2], qhi[2], q0[2], qa[7];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      qlo[k] = ld4(p3 + k + cd.j * SD3, SA3, SB3); qhi[k] = ld4(p3 + k + (cd.j + 1) * SD3, SA3, SB3);
      q0[k] = ld4(p3 + k + D2_ZERO_NODE * SD3, SA3, SB3);
    }
    TP pa = T + OFF_G2A + n1 * S_G2A;
#pragma unroll
    for (int k = 0; k < 7; ++k) qa[k] = ld4(pa + k, S_G2A, S_G2A * N_A1);
    F16_PHASE();
    const Axis a1 = br_axis(ca), b = br_axis(cb), d2 = br_axis(cd);
    Axis a2 = a1;
    if (hi_a) { a2.j = N_A2 - 2; a2.l = 1.0; a2.m = 0.0; }
    const W4 W1 = bil_weights(a1, b), W2 = bil_weights(a2, b);
    double base[3], base0[3], dr30[3], da20[3];
    base[0] = base0[0] = bil4w(qa[0], a1, b, W1);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      base[k + 1] = lerp(bil4w(qlo[k], a1, b, W1), bil4w(qhi[k], a1, b, W1), d2);
      base0[k + 1] = bil4w(q0[k], a1, b, W1);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { dr30[k] = bil4w(qa[1 + k], a1, b, W1) - base0[k]; da20[k] = bil4w(qa[4 + k], a1, b, W1) - base0[k]; }
    F16_PHASE();
    TP pb = T + OFF_G2B + n2 * S_G2B;
    Q4 ql[3], qal[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { ql[k] = ld4(pb + 3 + k, S_G2B, S_G2B * N_A2); qal[k] = ld4(pb + 6 + k, S_G2B, S_G2B * N_A2); }
    F16_PHASE();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double Clef = bil4w(ql[k], a2, b, W2), Ca20lef = bil4w(qal[k], a2, b, W2);
      out[k] = base[k] + (Clef - base0[k]) * dlef + (da20[k] + (Ca20lef - Clef - da20[k]) * dlef) * dail + dr30[k] * drud;
    }
  } else {
    double r0[3], r1v[3], p0[3], p1[3], hr0[3], hr1[3], hp0[3], hp1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ir = k == 0 ? 1 : (k == 1 ? 7 : 4);
      r0[k] = g[ir]; r1v[k] = g[S_G1A + ir]; p0[k] = g[ir + 1]; p1[k] = g[S_G1A + ir + 1];
      hr0[k] = h[ir]; hr1[k] = h[S_G1B + ir]; hp0[k] = h[ir + 1]; hp1[k] = h[S_G1B + ir + 1];
    }
    const double nb0 = g[9], nb1 = g[S_G1A + 9], lb0 = g[10], lb1 = g[S_G1A + 10];
    F16_PHASE();
    const Axis a1 = br_axis(ca);
    Axis a2 = a1;
    if (hi_a) { a2.j = N_A2 - 2; a2.l = 1.0; a2.m = 0.0; }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double Cr = lerp(r0[k], r1v[k], a1);
      if (k == 2 && !(flags & FLAG_FIX_CLR)) Cr = 0.0;                          // reference defect: _CLr never loaded
      out[k] = kb * (Cr + lerp(hr0[k], hr1[k], a2) * dlef) * R + kb * (lerp(p0[k], p1[k], a1) + lerp(hp0[k], hp1[k], a2) * dlef) * P;
    }
    out[1] += lerp(nb0, nb1, a1) * beta;
    out[2] += lerp(lb0, lb1, a1) * beta;
  }
}

// ---- lofi (C/lofi_F16_AeroData.c), tables in global/constant memory ---------------------------
F16_DEV int sgn_i(double v) { return (v > 0) - (v < 0); }
F16_DEV int fix_i(double v) { return (int)trunc(v); }

struct LofiAlpha { int k, L; double da; };
F16_DEV LofiAlpha lofi_alpha(double alpha) {   // :31-45
  const double s = .2 * alpha;
  int k = fix_i(s);
  if (k <= -2) k = -1; else if (k >= 9) k = 8;
  LofiAlpha r;
  r.da = s - k;
  r.L = k + fix_i(1.1 * sgn_i(r.da)) + 3;
  r.k = k + 3;
  return r;
}
F16_DEV double lofi_bilin(const double *T, int m, int n, const LofiAlpha &a, double db) {
  const double t = T[(m - 1) * 12 + a.k - 1], u = T[(n - 1) * 12 + a.k - 1];
  const double v = t + fabs(a.da) * (T[(m - 1) * 12 + a.L - 1] - t);
  const double w = u + fabs(a.da) * (T[(n - 1) * 12 + a.L - 1] - u);
  return v + (w - v) * db;
}

F16_DEV void aero_lofi(const double *__restrict__ LT, double alpha, double beta, double el, double dail, double drud,
                       Aero &c, int &status) {
  // off-table guard: the reference indexes outside its arrays for alpha < -10, alpha > 45, |beta| >= 30
  if (!(alpha >= -10.0 && alpha <= 45.0)) status |= ST_ALPHA1;
  if (!(fabs(beta) < 30.0)) status |= ST_BETA;
  if (!(fabs(el) <= 25.0)) status |= ST_EL;
  alpha = fmin(fmax(alpha, -10.0), 45.0);
  const LofiAlpha a = lofi_alpha(alpha);
  {  // damping :12-56
    const double *A = LT + LOFI_DAMP;
    double d[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = A[i * 12 + a.k - 1] + fabs(a.da) * (A[i * 12 + a.L - 1] - A[i * 12 + a.k - 1]);
    c.Cxq = d[0]; c.Cyr = d[1]; c.Cyp = d[2]; c.Czq = d[3]; c.Clr = d[4]; c.Clp = d[5]; c.Cmq = d[6]; c.Cnr = d[7]; c.Cnp = d[8];
  }
  {  // dmomdcon :59-183
    const double s = 0.2 * fabs(beta);
    int m = fix_i(s);
    if (m >= 7) m = 6;
    const double db = s - m;
    int n = m + 1;
    m = m + 1; n = n + 1;
    if (n > 7) n = 7;
    c.dCl_a20 = lofi_bilin(LT + LOFI_DLDA, m, n, a, db);
    c.dCl_r30 = lofi_bilin(LT + LOFI_DLDR, m, n, a, db);
    c.dCn_a20 = lofi_bilin(LT + LOFI_DNDA, m, n, a, db);
    c.dCn_r30 = lofi_bilin(LT + LOFI_DNDR, m, n, a, db);
  }
  {  // clcn :185-262
    const double s = .2 * fabs(beta);
    int m = fix_i(s);
    if (m == 0) m = 1; else if (m >= 6) m = 5;
    const double db = s - m;
    int n = m + fix_i(1.1 * sgn_i(db));
    m = m + 1; n = n + 1;
    c.Cl = lofi_bilin(LT + LOFI_CL, m, n, a, fabs(db)) * sgn_i(beta);
    c.Cn = lofi_bilin(LT + LOFI_CN, m, n, a, fabs(db)) * sgn_i(beta);
  }
  {  // cxcm :265-336
    const double s = el / 12.0;
    int m = fix_i(s);
    if (m <= -2) m = -1; else if (m >= 2) m = 1;
    const double de = s - m;
    int n = m + fix_i(1.1 * sgn_i(de));
    m = m + 3; n = n + 3;
    n = min(max(n, 1), 5);
    c.Cx = lofi_bilin(LT + LOFI_CX, m, n, a, fabs(de));
    c.Cm = lofi_bilin(LT + LOFI_CM, m, n, a, fabs(de));
  }
  c.Cy = -.02 * beta + .021 * dail + .086 * drud;   // nlplant.c:283
  {  // cz :339-368
    const double *A = LT + LOFI_CZ;
    const double s = A[a.k - 1] + fabs(a.da) * (A[a.L - 1] - A[a.k - 1]);
    const double b573 = beta / 57.3;
    c.Cz = s * (1 - b573 * b573) - .19 * (el) / 25;
  }
  // nlplant.c:295-319
  c.dCx_lef = c.dCz_lef = c.dCm_lef = c.dCy_lef = c.dCn_lef = c.dCl_lef = 0.0;
  c.dCxq_lef = c.dCyr_lef = c.dCyp_lef = c.dCzq_lef = c.dClr_lef = c.dClp_lef = c.dCmq_lef = c.dCnr_lef = c.dCnp_lef = 0.0;
  c.dCy_r30 = c.dCy_a20 = c.dCy_a20_lef = c.dCn_a20_lef = c.dCl_a20_lef = 0.0;
  c.dCnbeta = c.dClbeta = c.dCm = 0.0;
  c.eta_el = 1.0;
}

// ---- C/nlplant.c:23-457 in three pieces ------------------------------------------------------------------------
// (the single-wave kernels run them back to back; k_rollout_4w spreads them over four wavefronts, see f16_dynamics.hip)
struct Totals { double Cx, Cz, Cm, Cy, Cn, Cl; };   // C*_tot of C/nlplant.c:333-377
struct Pre {                                        // trigonometry, atmosphere, body velocities
  double sa, ca, sb, cb, st, ct, sphi, cphi;
  double U, V, W, vt, mach, qbar, ps;
};

// Table lookups + coefficient build-up (C/nlplant.c:183-377).  Needs xu[6..11], xu[13..16] only.
// FI: 1 / 0 = fidelity fixed at compile time, -1 = decided by fi_flag at run time.
// hifi: the phased lookups above; lofi: Stevens & Lewis tables + the same totals with every lef term zero (:295-319).
template <int FI = -1, typename TP>
F16_DEV void aero_totals(TP T, const double *__restrict__ LT, const double *xu, double xcg, int fi_flag, unsigned flags,
                         Totals &t, int &status) {
  const bool hifi = FI < 0 ? fi_flag == 1 : FI == 1;
  if (hifi) {
    TotalsOut o;
    aero_totals_phased(T, xu, xcg, flags, o, status);
    t.Cx = o.Cx; t.Cz = o.Cz; t.Cm = o.Cm; t.Cy = o.Cy; t.Cn = o.Cn; t.Cl = o.Cl;
    return;
  }
  const double B = 30.0, cbar = 11.32, xcgr = 0.35;
  const double r2d = 180.0 / 3.141592653589793;   // 180.0/acos(-1)
  double vt = xu[6];
  const double alpha = xu[7] * r2d, beta = xu[8] * r2d;
  const double P = xu[9], Q = xu[10], R = xu[11];
  if (vt <= 0.01) vt = 0.01;
  const double el = xu[13], ail = xu[14], rud = xu[15];
  const double dail = F16_DIVC(ail, 21.5);
  const double drud = F16_DIVC(rud, 30.0);
  const double dlef = 0.0;
  Aero c;
  aero_lofi(LT, alpha, beta, el, dail, drud, c, status);
  // totals, C/nlplant.c:333-377 (dZdQ uses delta_Cz_lef, as the reference does)
#ifdef F16_FAST_DIV
  const double r2vt = 1.0 / (2 * vt);
  const double kq = cbar * r2vt, kb = B * r2vt;       // cbar/(2 vt), B/(2 vt) off one reciprocal
#else
  const double kq = cbar / (2 * vt), kb = B / (2 * vt);
#endif
  const double dXdQ = kq * (c.Cxq + c.dCxq_lef * dlef);
  t.Cx = c.Cx + c.dCx_lef * dlef + dXdQ * Q;
  const double dZdQ = kq * (c.Czq + c.dCz_lef * dlef);
  t.Cz = c.Cz + c.dCz_lef * dlef + dZdQ * Q;
  const double dMdQ = kq * (c.Cmq + c.dCmq_lef * dlef);
  t.Cm = c.Cm * c.eta_el + t.Cz * (xcgr - xcg) + c.dCm_lef * dlef + dMdQ * Q + c.dCm;
  const double dYdail = c.dCy_a20 + c.dCy_a20_lef * dlef;
  const double dYdR = kb * (c.Cyr + c.dCyr_lef * dlef);
  const double dYdP = kb * (c.Cyp + c.dCyp_lef * dlef);
  t.Cy = c.Cy + c.dCy_lef * dlef + dYdail * dail + c.dCy_r30 * drud + dYdR * R + dYdP * P;
  const double dNdail = c.dCn_a20 + c.dCn_a20_lef * dlef;
  const double dNdR = kb * (c.Cnr + c.dCnr_lef * dlef);
  const double dNdP = kb * (c.Cnp + c.dCnp_lef * dlef);
  t.Cn = c.Cn + c.dCn_lef * dlef - t.Cy * (xcgr - xcg) * (cbar / B) + dNdail * dail + c.dCn_r30 * drud + dNdR * R + dNdP * P +
         c.dCnbeta * beta;
  const double dLdail = c.dCl_a20 + c.dCl_a20_lef * dlef;
  const double dLdR = kb * (c.Clr + c.dClr_lef * dlef);
  const double dLdP = kb * (c.Clp + c.dClp_lef * dlef);
  t.Cl = c.Cl + c.dCl_lef * dlef + dLdail * dail + c.dCl_r30 * drud + dLdR * R + dLdP * P + c.dClbeta * beta;
}


F16_DEV void compose_totals(const double *ls, const double *ld, const double *ts, const double *td, double xcg, Totals &t) {
  const double B = 30.0, cbar = 11.32, xcgr = 0.35;
  t.Cx = ls[0] + ld[0];
  t.Cz = ls[1] + ld[1];
  t.Cm = ls[2] + ld[2] + t.Cz * (xcgr - xcg);
  t.Cy = ts[0] + td[0];
  t.Cn = ts[1] + td[1] - t.Cy * (xcgr - xcg) * (cbar / B);
  t.Cl = ts[2] + td[2];
}

// Trigonometry, atmosphere, navigation + kinematic equations (C/nlplant.c:90-176): xdot[0..5].
// sin / cos of alpha, beta, theta, phi, psi handed in by a caller that keeps them up to date itself (rollout kernels: trig_advance)
struct Trig5 { double sa, ca, sb, cb, st, ct, sphi, cphi, spsi, cpsi; };
F16_DEV void trig_exact(const double *x, Trig5 &g) {
  F16_SINCOS(x[7], &g.sa, &g.ca);
  F16_SINCOS(x[8], &g.sb, &g.cb);
  F16_SINCOS(x[4], &g.st, &g.ct);
  F16_SINCOS(x[3], &g.sphi, &g.cphi);
  F16_SINCOS(x[5], &g.spsi, &g.cpsi);
}
// (sin, cos)(a + d) from (sin, cos)(a) by the angle-sum formulas with sin d, cos d from their series: for |d| <= 4e-3 rad (an Euler
// step of 1 ms moves an angle by |rate| x 1e-3) the terms dropped are d^7 / 5040 < 4e-21 and d^6 / 720 < 6e-18.  11 operations per
// angle against ~42 for the full evaluation (range reduction, two polynomials, quadrant selects).
F16_DEV void trig_rotate(double &s, double &c, double d) {
  const double d2 = d * d;
  const double sd = d * fma(d2, fma(d2, 1.0 / 120.0, -1.0 / 6.0), 1.0);
  const double cd = fma(d2, fma(d2, 1.0 / 24.0, -0.5), 1.0);
  const double s1 = fma(c, sd, s * cd), c1 = fma(-s, sd, c * cd);
  s = s1; c = c1;
}
// The rollout kernels carry the five pairs from step to step in lane-indexed LDS slots (TrigSlots: slot k of this lane at
// base[k * stride]).  A stale set (every 32nd step, or after an increment beyond the series' range) is re-evaluated exactly INTO the
// slots at the top of the step -- a block of its own with nothing live behind it -- and plant_pre always loads the ten doubles where
// it needs them (behind the lookups): a merge of "evaluated" and "loaded" values in front of the lookups was spilled across them.
struct TrigSlots { double *base; int stride; };
F16_DEV void trig_store(const TrigSlots &ts, const Trig5 &g) {
  double *b = ts.base; const int n = ts.stride;
  b[0] = g.sa; b[n] = g.ca; b[2 * n] = g.sb; b[3 * n] = g.cb; b[4 * n] = g.st; b[5 * n] = g.ct; b[6 * n] = g.sphi; b[7 * n] = g.cphi;
  b[8 * n] = g.spsi; b[9 * n] = g.cpsi;
}
// after an Euler step: the pairs follow their angles by the EXACT increment of the stored state (xn - xo: a difference of
// neighbouring doubles), so nothing but the rounding of the rotations (~1e-16 a step) separates them from the state.  Returns
// whether an increment left the series' range (the caller then marks the slots stale: the next step evaluates exactly).
F16_DEV bool trig_advance(const double *xo5, const double *xn, const TrigSlots &ts) {
  const double da = xn[7] - xo5[0], db = xn[8] - xo5[1], dt = xn[4] - xo5[2], dp = xn[3] - xo5[3], ds = xn[5] - xo5[4];
  const double big = fmax(fmax(fabs(da), fabs(db)), fmax(fmax(fabs(dt), fabs(dp)), fabs(ds)));
  double *b = ts.base; const int n = ts.stride;
  { double s = b[0], c = b[n]; trig_rotate(s, c, da); b[0] = s; b[n] = c; }
  { double s = b[2 * n], c = b[3 * n]; trig_rotate(s, c, db); b[2 * n] = s; b[3 * n] = c; }
  { double s = b[4 * n], c = b[5 * n]; trig_rotate(s, c, dt); b[4 * n] = s; b[5 * n] = c; }
  { double s = b[6 * n], c = b[7 * n]; trig_rotate(s, c, dp); b[6 * n] = s; b[7 * n] = c; }
  { double s = b[8 * n], c = b[9 * n]; trig_rotate(s, c, ds); b[8 * n] = s; b[9 * n] = c; }
  return !(big <= 4e-3);
}

template <bool ATMOS = true, bool GIVEN = false>
F16_DEV void plant_pre(const double *xu, Pre &p, double *xdot, const TrigSlots *tg = nullptr) {
  const double alt = xu[2], phi = xu[3], theta = xu[4], psi = xu[5];
  const double P = xu[9], Q = xu[10], R = xu[11];
  double vt = xu[6];
  if (vt <= 0.01) vt = 0.01;
  p.vt = vt;
  double spsi, cpsi;
  if (GIVEN) {        // (the slots are current: the caller has refreshed them where they were stale)
    const double *b = tg->base; const int n = tg->stride;
    p.sa = b[0]; p.ca = b[n]; p.sb = b[2 * n]; p.cb = b[3 * n]; p.st = b[4 * n]; p.ct = b[5 * n]; p.sphi = b[6 * n]; p.cphi = b[7 * n];
    spsi = b[8 * n]; cpsi = b[9 * n];
    (void)phi; (void)psi;
  } else {
    F16_SINCOS(xu[7], &p.sa, &p.ca);
    F16_SINCOS(xu[8], &p.sb, &p.cb);
    F16_SINCOS(theta, &p.st, &p.ct);
    F16_SINCOS(phi, &p.sphi, &p.cphi);
    F16_SINCOS(psi, &spsi, &cpsi);
  }
#ifdef F16_FAST_DIV
  const double rct = f16_rcp(p.ct);
  const double tt = p.st * rct;
#elif defined(F16_FAST_TAN)
  const double tt = p.st / p.ct;
#else
  const double tt = tan(theta);
#endif
  if (ATMOS) atmos_dev(alt, vt, p.mach, p.qbar, p.ps);
  else { p.mach = p.qbar = p.ps = 0.0; (void)alt; }
  p.U = vt * p.ca * p.cb; p.V = vt * p.sb; p.W = vt * p.sa * p.cb;
  const double U = p.U, V = p.V, W = p.W, st = p.st, ct = p.ct, sphi = p.sphi, cphi = p.cphi;
  xdot[0] = U * (ct * cpsi) + V * (sphi * cpsi * st - cphi * spsi) + W * (cphi * st * cpsi + sphi * spsi);
  xdot[1] = U * (ct * spsi) + V * (sphi * spsi * st + cphi * cpsi) + W * (cphi * st * spsi - sphi * cpsi);
  xdot[2] = U * st - V * (sphi * ct) - W * (cphi * ct);
  xdot[3] = P + tt * (Q * sphi + R * cphi);
  xdot[4] = Q * cphi - R * sphi;
#ifdef F16_FAST_DIV
  xdot[5] = (Q * sphi + R * cphi) * rct;
#else
  xdot[5] = (Q * sphi + R * cphi) / ct;
#endif
}

// Force equations (C/nlplant.c:383-405): xdot[6..8].
F16_DEV void plant_forces(const double *xu, const Pre &p, double Cx_tot, double Cy_tot, double Cz_tot, double *xdot) {
  const double g = 32.17, m = 636.94, S = 300.0;
  const double P = xu[9], Q = xu[10], R = xu[11], Thr = xu[12];
  const double U = p.U, V = p.V, W = p.W, vt = p.vt, qbar = p.qbar;
  const double st = p.st, ct = p.ct, sphi = p.sphi, cphi = p.cphi, cb = p.cb;
  const double Udot = R * V - Q * W - g * st + F16_DIVC(qbar * S * Cx_tot, m) + F16_DIVC(Thr, m);
  const double Vdot = P * W - R * U + g * ct * sphi + F16_DIVC(qbar * S * Cy_tot, m);
  const double Wdot = Q * U - P * V + g * ct * cphi + F16_DIVC(qbar * S * Cz_tot, m);
#ifdef F16_FAST_DIV      // reciprocals by v_rcp_f64 + two Newton steps (<= 1 ulp) instead of three IEEE division sequences
  xdot[6] = (U * Udot + V * Vdot + W * Wdot) * f16_rcp(vt);
  xdot[7] = (U * Wdot - W * Udot) * f16_rcp(U * U + W * W);
  xdot[8] = (Vdot * vt - V * xdot[6]) * f16_rcp(vt * vt * cb);
#else
  xdot[6] = (U * Udot + V * Vdot + W * Wdot) / vt;
  xdot[7] = (U * Wdot - W * Udot) / (U * U + W * W);
  xdot[8] = (Vdot * vt - V * xdot[6]) / (vt * vt * cb);
#endif
}

// Moment equations (C/nlplant.c:413-436): xdot[9..11].  Needs the body rates, qbar and Cl, Cm, Cn only.
F16_DEV void plant_moments(double P, double Q, double R, double qbar, double Cl_tot, double Cm_tot, double Cn_tot, double *xdot) {
  const double B = 30.0, S = 300.0, cbar = 11.32;
  const double Heng = 0.0;
  const double Jy = 55814.0, Jxz = 982.0, Jz = 63100.0, Jx = 9496.0;
  const double L_tot = Cl_tot * qbar * S * B;
  const double M_tot = Cm_tot * qbar * S * cbar;
  const double N_tot = Cn_tot * qbar * S * B;
#ifdef F16_FAST_DIV
  const double rdenom = 1.0 / (9496.0 * 63100.0 - 982.0 * 982.0);
#define F16_DIV_DENOM *rdenom
#else
  const double denom = Jx * Jz - Jxz * Jxz;
#define F16_DIV_DENOM / denom
#endif
  xdot[9] = (Jz * L_tot + Jxz * N_tot - (Jz * (Jz - Jy) + Jxz * Jxz) * Q * R + Jxz * (Jx - Jy + Jz) * P * Q + Jxz * Q * Heng) F16_DIV_DENOM;
  xdot[10] = F16_DIVC(M_tot + (Jz - Jx) * P * R - Jxz * (P * P - R * R) - R * Heng, Jy);
  xdot[11] = (Jx * N_tot + Jxz * L_tot + (Jx * (Jx - Jy) + Jxz * Jxz) * P * Q - Jxz * (Jx - Jy + Jz) * Q * R + Jx * Q * Heng) F16_DIV_DENOM;
}

// Force and moment equations (C/nlplant.c:383-436): xdot[6..11] (+ accels outputs 12..17 when OUTPUTS).
template <bool OUTPUTS>
F16_DEV void plant_post(const double *xu, const Pre &p, const Totals &t, double *xdot) {
  const double P = xu[9], Q = xu[10], R = xu[11];
  const double qbar = p.qbar, st = p.st, ct = p.ct, sphi = p.sphi, cphi = p.cphi, cb = p.cb;
  plant_forces(xu, p, t.Cx, t.Cy, t.Cz, xdot);
  plant_moments(P, Q, R, qbar, t.Cl, t.Cm, t.Cn, xdot);

  if (OUTPUTS) {  // accels(), C/nlplant.c:512-552 (uses the UNclamped state[6])
    const double grav = 32.174;
    const double v6 = xu[6];
    const double sa = p.sa, ca = p.ca, sb = p.sb;
    const double vel_u = v6 * cb * ca, vel_v = v6 * sb, vel_w = v6 * cb * sa;
    const double u_dot = cb * ca * xdot[6] - v6 * sb * ca * xdot[8] - v6 * cb * sa * xdot[7];
    const double v_dot = sb * xdot[6] + v6 * cb * xdot[8];
    const double w_dot = cb * sa * xdot[6] - v6 * sb * sa * xdot[8] + v6 * cb * ca * xdot[7];
    xdot[12] = 1.0 / grav * (u_dot + Q * vel_w - R * vel_v) + st;
    xdot[13] = 1.0 / grav * (v_dot + R * vel_u - P * vel_w) - ct * sphi;
    xdot[14] = -1.0 / grav * (w_dot + P * vel_v - Q * vel_u) + ct * cphi;
    xdot[15] = p.mach;
    xdot[16] = qbar;
    xdot[17] = p.ps;
  }
}

// C/nlplant.c:23-457.  xu[0..16] in, xdot[0..11] out (+ xdot[12..17] = nx,ny,nz,mach,qbar,ps when OUTPUTS).
// Returns qbar/ps of the clamped-vt atmosphere call for reuse by the lef model.  The lookups and the six coefficient
// totals come FIRST: the 45 interpolated values collapse to 6 doubles before the register-hungry sincos/pow code runs.
template <bool OUTPUTS, int FI = -1, typename TP, bool GIVEN = false>
F16_DEV void plant(TP T, const double *__restrict__ LT, const double *xu, double *xdot, double xcg, int fi_flag,
                   unsigned flags, int &status, double &qbar_out, double &ps_out, const TrigSlots *tg = nullptr) {
  Totals t;
  aero_totals<FI>(T, LT, xu, xcg, fi_flag, flags, t, status);
  Pre p;
  plant_pre<true, GIVEN>(xu, p, xdot, tg);
  qbar_out = p.qbar; ps_out = p.ps;
  plant_post<OUTPUTS>(xu, p, t, xdot);
}

// IEEE minNum / maxNum clamp: DROPS a NaN (returns the bound).  For callers whose operand is finite by construction (trim).
F16_DEV double clipd(double a, double lo, double hi) { return fmin(fmax(a, lo), hi); }

// The reference's actuators saturate with np.clip (utils.py:303-330), and np.clip PROPAGATES NaN: a NaN command (what OSQP hands
// back for a QP it certifies infeasible, env.py:420-424) or a NaN state gives a NaN derivative -- fmin / fmax alone would turn it
// into a hard-over at full rate.  One command-saturated, rate-saturated first-order lag
//     clip(k * (clip(cmd, lo, hi) - state), -rate, rate)
// with that rule: the fast clamps, then ONE unordered compare of the command with the unclamped rate (NaN if either is) + a select.
F16_DEV double actuator_rate(double cmd, double lo, double hi, double k, double state, double rate) {
  const double t = k * (clipd(cmd, lo, hi) - state);
  const double r = clipd(t, -rate, rate);
  return __builtin_isunordered(cmd, t) ? __builtin_nan("") : r;
}

// utils.py:289-306 -> lf1_dot (7.25*LF_err), lf2_dot (lef_err).  qbar/ps are those of atmos(h, V) with the
// RAW V (utils.py:291); plant() evaluates atmos with vt clamped to >= 0.01, identical whenever V > 0.01.
F16_DEV void upd_lef_dev(double h, double V, double alpha, double lf1, double lf2, double qbar_p, double ps_p,
                         double &lf1_dot, double &lf2_dot) {
  double qbar = qbar_p, ps = ps_p;
  if (V <= 0.01) {
    double mach;
    atmos_dev(h, V, mach, qbar, ps);
  }
  const double atmos_out = qbar / ps * 9.05;
  const double alpha_deg = F16_DIVC(alpha * 180, 3.141592653589793);
  const double LF_err = alpha_deg - (lf1 + (2 * alpha_deg));
  const double LF_out = (lf1 + (2 * alpha_deg)) * 1.38;
  const double lef_cmd = LF_out + 1.45 - atmos_out;
  lf2_dot = actuator_rate(lef_cmd, 0., 25, 1 / 0.136, lf2, 25);     // utils.py:303-305 (np.clip: NaN in -> NaN out)
  lf1_dot = LF_err * 7.25;
}

// utils.py:308-330 + env.py:90-102: actuator and leading-edge-flap state derivatives xdot[12..17].
F16_DEV void actuators_dev(const double *x, const double *u, double qbar, double ps, double *xdot) {
  xdot[12] = actuator_rate(u[0], 1000, 19000, 1.0, x[12], 10000);      // utils.py:308-312
  xdot[13] = actuator_rate(u[1], -25, 25, 20.2, x[13], 60);            // :314-318
  xdot[14] = actuator_rate(u[2], -21.5, 21.5, 20.2, x[14], 80);        // :320-324
  xdot[15] = actuator_rate(u[3], -30., 30, 20.2, x[15], 120);          // :326-330
  double lf1_dot, lf2_dot;
  upd_lef_dev(x[2], x[6], x[7], x[17], x[16], qbar, ps, lf1_dot, lf2_dot);
  xdot[16] = lf2_dot;   // env.py:98,102: temp[4] -> xdot[16]
  xdot[17] = lf1_dot;
}

// env.py:65-103: xdot[18] of the full actuated model.
template <int FI = -1, typename TP, bool GIVEN = false>
F16_DEV void calc_xdot(TP T, const double *__restrict__ LT, const double *x, const double *u, double *xdot, double xcg,
                       int fi_flag, unsigned flags, int &status, const TrigSlots *tg = nullptr) {
  double qbar, ps;
  plant<false, FI, TP, GIVEN>(T, LT, x, xdot, xcg, fi_flag, flags, status, qbar, ps, tg);
  actuators_dev(x, u, qbar, ps, xdot);
}

// env.py:152-193: sv = full state with the 9 MPC states / 3 actuator positions already scattered in.
// Returns the 9 derivatives in MPC order {phi,theta,alpha,beta,p,q,r,lf1,lf2}; the two lef derivatives
// land swapped exactly as in the reference (env.py:184,189 vs :98,102).
template <typename TP>
F16_DEV void calc_xdot_na(TP T, const double *__restrict__ LT, const double *sv, double *xdot9, double xcg, int fi_flag,
                          unsigned flags, int &status) {
  double xd[12], qbar, ps;
  plant<false>(T, LT, sv, xd, xcg, fi_flag, flags, status, qbar, ps);
  double lf1_dot, lf2_dot;
  upd_lef_dev(sv[2], sv[6], sv[7], sv[17], sv[16], qbar, ps, lf1_dot, lf2_dot);
  xdot9[0] = xd[3]; xdot9[1] = xd[4]; xdot9[2] = xd[7]; xdot9[3] = xd[8];
  xdot9[4] = xd[9]; xdot9[5] = xd[10]; xdot9[6] = xd[11];
  xdot9[7] = lf2_dot;   // state_vector_dot[17] = lf_state2_dot
  xdot9[8] = lf1_dot;   // state_vector_dot[16] = lf_state1_dot
}

// env.py:117-124 box check (parameters.py:122-123, mixed units as in the reference)
// Which states are outside their box (env.py:117-124 tests all eighteen; six of them carry no finite limit): bit 8 + k for state
// k -- evaluated only once an aircraft has been found outside (outside_envelope), not in the per-step path
F16_DEV int envelope_state_bits(const double *x) {
#ifdef F16_NO_ENV_BITS      // (A/B builds: what the which-state bits cost the rollout kernels)
  return 0;
#endif
  int m = 0;
  m |= (x[2] < 0 || x[2] > 100000) ? 1 << (8 + 2) : 0;
  m |= (x[6] < 0 || x[6] > 900) ? 1 << (8 + 6) : 0;
  m |= (x[7] < -20. || x[7] > 90) ? 1 << (8 + 7) : 0;
  m |= (x[8] < -30. || x[8] > 30) ? 1 << (8 + 8) : 0;
  m |= (x[9] < -300 || x[9] > 300) ? 1 << (8 + 9) : 0;
  m |= (x[10] < -100 || x[10] > 100) ? 1 << (8 + 10) : 0;
  m |= (x[11] < -50 || x[11] > 50) ? 1 << (8 + 11) : 0;
  m |= (x[12] < 1000 || x[12] > 19000) ? 1 << (8 + 12) : 0;
  m |= (x[13] < -25 || x[13] > 25) ? 1 << (8 + 13) : 0;
  m |= (x[14] < -21.5 || x[14] > 21.5) ? 1 << (8 + 14) : 0;
  m |= (x[15] < -30. || x[15] > 30) ? 1 << (8 + 15) : 0;
  m |= (x[16] < 0. || x[16] > 25) ? 1 << (8 + 16) : 0;
  return m;
}
F16_DEV bool outside_envelope(const double *x) {
  bool bad = x[2] < 0 || x[2] > 100000 || x[6] < 0 || x[6] > 900 || x[7] < -20. || x[7] > 90 || x[8] < -30. || x[8] > 30 ||
             x[9] < -300 || x[9] > 300 || x[10] < -100 || x[10] > 100 || x[11] < -50 || x[11] > 50 || x[12] < 1000 ||
             x[12] > 19000 || x[13] < -25 || x[13] > 25 || x[14] < -21.5 || x[14] > 21.5 || x[15] < -30. || x[15] > 30 ||
             x[16] < 0. || x[16] > 25;
  return bad;
}

// ONE explicit-Euler step of ONE aircraft (env.py:105-130 without the envelope test; exact trigonometry), compiled OUT OF LINE: a
// function of its own has one instruction sequence whoever calls it, so the kernels that step through it -- the F16_FLAG_ONE_LANE
// rollout (f16_dynamics.hip: k_rollout_exact) and the closed MPC loop (f16_mpc_wave.hip: k_rollout_mpc) -- agree bit for bit.  (The
// same source inlined into two kernels does not: FMA contraction follows the surrounding code, and the navigation states came out an
// ulp apart.)  tab: the fp64 table image through a generic pointer (LDS or global); xio[18] in / out, uin[4], *stio |= the grid bits.
static __device__ __noinline__ void euler_step_exact(const double *tab, const double *lofi, double *xio, const double *uin, double dt,
                                                     double xcg, int fi, unsigned flags, int *stio) {
  double x[18], u[4], xd[18];
#pragma unroll
  for (int k = 0; k < 18; ++k) x[k] = xio[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) u[k] = uin[k];
  int st = 0;
  calc_xdot<-1>(tab, lofi, x, u, xd, xcg, fi, flags, st);
#pragma unroll
  for (int k = 0; k < 18; ++k) xio[k] = x[k] + xd[k] * dt;   // env.py:126
  *stio |= st;
}

}  // namespace f16
