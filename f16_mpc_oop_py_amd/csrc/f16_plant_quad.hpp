// f16_plant_quad.hpp -- the hifi aerodynamic build-up with FOUR lanes per aircraft (used by k_rollout_q).
//
// At the reference's batch of 4096 there are fewer aircraft than SIMD lanes on the chip, so a rollout step is bound by
// how many wave-instructions ONE aircraft group needs, not by throughput.  Here a quad of lanes (l = 4 a + s) shares an
// aircraft and sub-lane s evaluates ONE member of each coefficient family -- the node-major table image keeps the
// members of a family next to each other, so the sub-lanes run the same instructions on payload offset s:
//     longitudinal wave   s = 0,1,2  ->  Cx, Cz, Cm      (C/nlplant.c:333-347)
//     lateral wave        s = 0,1,2  ->  Cy, Cn, Cl      (C/nlplant.c:353-377)
// (sub-lane 3 shadows sub-lane 2).  Same terms as aero_totals_phased(), hifi_F16_AeroData.c:1871-1934; the
// cg-offset couplings (Cm needs Cz_tot, Cn needs Cy_tot) cross the quad with one DPP broadcast.
#pragma once
#include "f16_plant.hpp"

namespace f16 {

// value of sub-lane J of this lane's quad
template <int J>
F16_DEV double quad_bcast(double v) {
  constexpr int ctrl = J | (J << 2) | (J << 4) | (J << 6);      // quad_perm:[J,J,J,J]
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

template <int J>
F16_DEV int quad_bcast_i(int v) {
  constexpr int ctrl = J | (J << 2) | (J << 4) | (J << 6);
  return __builtin_amdgcn_mov_dpp(v, ctrl, 0xF, 0xF, true);
}

// The three axis brackets of a role, ONE PER SUB-LANE (0: alpha on ALPHA1, 1: beta, 2/3: elevator on DH1 or DH2), shared
// across the quad by DPP broadcasts: first the cell indices (the table addresses need nothing else), later lambda.
struct QuadBr { BrCell c; int ja, jb, jd; bool offa, offb, offd; };
template <bool USE_D2, typename TP>
F16_DEV BrRaw quad_br_load(TP T, double alpha, double beta, double el, int s, int &nX, double &vX) {
  const int ax = s < 2 ? s : 2;
  const int gel = USE_D2 ? (int)(el >= 0.0) : (el >= -10.0) + (el >= 0.0) + (el >= 10.0);
  const int offX = ax == 0 ? OFF_BP_A1 : (ax == 1 ? OFF_BP_B1 : (USE_D2 ? OFF_BP_D2 : OFF_BP_D1));
  nX = ax == 0 ? N_A1 : (ax == 1 ? N_B1 : (USE_D2 ? N_D2 : N_D1));
  vX = ax == 0 ? alpha : (ax == 1 ? beta : el);
  const int gX = ax == 0 ? alpha_guess(alpha) : (ax == 1 ? beta_guess(beta) : gel);
  return br_load(T + offX, nX, gX);
}
F16_DEV QuadBr quad_br_cells(const BrRaw &r, int nX, double vX) {
  QuadBr q;
  bool off;
  q.c = br_cell(r, nX, vX, off);
  q.ja = quad_bcast_i<0>(q.c.j); q.jb = quad_bcast_i<1>(q.c.j); q.jd = quad_bcast_i<2>(q.c.j);
  const int o = off ? 1 : 0;
  q.offa = quad_bcast_i<0>(o) != 0; q.offb = quad_bcast_i<1>(o) != 0; q.offd = quad_bcast_i<2>(o) != 0;
  return q;
}
F16_DEV void quad_br_axes(const QuadBr &q, Axis &a, Axis &b, Axis &d) {
  const Axis own = br_axis(q.c);                                  // one division per lane
  a.j = q.ja; a.l = quad_bcast<0>(own.l); a.m = quad_bcast<0>(own.m);
  b.j = q.jb; b.l = quad_bcast<1>(own.l); b.m = quad_bcast<1>(own.m);
  d.j = q.jd; d.l = quad_bcast<2>(own.l); d.m = quad_bcast<2>(own.m);
}

struct QuadIn {            // what both aerodynamic waves derive from the published state (C/nlplant.c:84-125)
  double alpha, beta, el, dail, drud, dlef, P, Q, R, kq, kb;
};
F16_DEV QuadIn quad_inputs(const double *xu) {
  const double B = 30.0, cbar = 11.32, r2d = 180.0 / 3.141592653589793;
  QuadIn in;
  double vt = xu[6];
  if (vt <= 0.01) vt = 0.01;
  in.alpha = xu[7] * r2d; in.beta = xu[8] * r2d;
  in.P = xu[9]; in.Q = xu[10]; in.R = xu[11];
  in.el = xu[13];
  in.dail = F16_DIVC(xu[14], 21.5);
  in.drud = F16_DIVC(xu[15], 30.0);
  in.dlef = 1 - F16_DIVC(xu[16], 25.0);
#ifdef F16_FAST_DIV
  const double r2vt = f16_rcp(2 * vt);
  in.kq = cbar * r2vt; in.kb = B * r2vt;
#else
  in.kq = cbar / (2 * vt); in.kb = B / (2 * vt);
#endif
  return in;
}

// Cx_tot / Cz_tot / Cm_tot on sub-lanes 0 / 1 / 2 (C/nlplant.c:333-347, hifi_C, hifi_damping, hifi_C_lef,
// hifi_damping_lef, hifi_other_coeffs).  dZdQ uses delta_Cz_lef exactly as the reference does (:339).
// latd: the rate-damping part of the LATERAL member k (Cy, Cn, Cl) -- 1-D tables on the same alpha cell, evaluated here
// to balance the two aerodynamic waves: kb (Cr + dCr_lef dlef) R + kb (Cp + dCp_lef dlef) P (+ dC_beta beta), :353-377.
#ifdef F16_EXP_STAMPQ2      // diagnostic build: cycles per phase of the longitudinal role (workgroup 0), tools/gpu_dyn_stamps2.py
__device__ unsigned long long g_qstamp[8];
#define Q2STAMP(i) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_qstamp[i] += t1_ - tq_; tq_ = t1_; }
#else
#define Q2STAMP(i)
#endif
template <typename TP>
F16_DEV double quad_long(TP T, const double *xu, int s, double xcg, unsigned flags, double &latd, int &status) {
#ifdef F16_EXP_STAMPQ2
  unsigned long long tq_ = __builtin_amdgcn_s_memtime();
#endif
  const QuadIn in = quad_inputs(xu);
  Q2STAMP(0)
  const int k = s < 2 ? s : 2;
  // (1) breakpoints: one axis per sub-lane
  int nX; double vX;
  const BrRaw rx = quad_br_load<false>(T, in.alpha, in.beta, in.el, s, nX, vX);
  const double a45 = T[OFF_BP_A1 + N_A2 - 1];
  F16_PHASE();
  Q2STAMP(1)
  const QuadBr qb = quad_br_cells(rx, nX, vX);
  if (qb.offa) status |= ST_ALPHA1 | ST_ALPHA2;
  if (qb.offb) status |= ST_BETA;
  if (qb.offd) status |= ST_EL;
  const bool hi_a = qb.ja > N_A2 - 2;                            // ALPHA2 ends at 45 deg: last cell, lambda = 1
  if (hi_a && in.alpha > a45) status |= ST_ALPHA2;
  const int j2 = hi_a ? N_A2 - 2 : qb.ja;
  const int n1 = qb.jb * N_A1 + qb.ja, n2 = qb.jb * N_A2 + j2;
  struct { int j; } ca = {qb.ja}, cd = {qb.jd};
  // (2) every table corner this role needs
  constexpr int SA = S_G3A, SB = S_G3A * N_A1, SD = S_G3A * N_A1 * N_B1;
  TP p = T + OFF_G3A + n1 * SA + k;
  const Q4 qlo = ld4(p + cd.j * SD, SA, SB), qhi = ld4(p + (cd.j + 1) * SD, SA, SB), q0 = ld4(p + D1_ZERO_NODE * SD, SA, SB);
  const Q4 qlef = ld4(T + OFF_G2B + n2 * S_G2B + k, S_G2B, S_G2B * N_A2);
  TP g = T + OFF_G1A + ca.j * S_G1A, h = T + OFF_G1B + j2 * S_G1B;
  const double g0 = g[3 * k], g1 = g[S_G1A + 3 * k], m0 = g[11], m1 = g[S_G1A + 11];
  const double h0 = h[3 * k], h1 = h[S_G1B + 3 * k];
  const double e0 = T[OFF_ETA + cd.j], e1 = T[OFF_ETA + cd.j + 1];
  const int ir = k == 0 ? 1 : (k == 1 ? 7 : 4), ib = k == 1 ? 9 : 10;     // CYr CYp | CNr CNp | CLr CLp; dCNbeta, dCLbeta
  const double r0 = g[ir], r1 = g[S_G1A + ir], p0 = g[ir + 1], p1 = g[S_G1A + ir + 1], b0 = g[ib], b1 = g[S_G1A + ib];
  const double hr0 = h[ir], hr1 = h[S_G1B + ir], hp0 = h[ir + 1], hp1 = h[S_G1B + ir + 1];
  Q2STAMP(2)
  F16_PHASE();
  // (3) arithmetic
  Axis a1, b, d1;
  quad_br_axes(qb, a1, b, d1);
  Axis a2 = a1;
  if (hi_a) { a2.j = N_A2 - 2; a2.l = 1.0; a2.m = 0.0; }
  const W4 W1 = bil_weights(a1, b), W2 = bil_weights(a2, b);
  const double Cf = lerp(bil4w(qlo, a1, b, W1), bil4w(qhi, a1, b, W1), d1);
  const double C0 = bil4w(q0, a1, b, W1);
  const double dC = bil4w(qlef, a2, b, W2) - C0;                       // hifi_C_lef :1892-1899
  const double Cq = lerp(g0, g1, a1), dCm = lerp(m0, m1, a1), dq = lerp(h0, h1, a2), eta = lerp(e0, e1, d1);
  const double dql = k == 1 ? dC : dq;                            // reference quirk: dZdQ uses delta_Cz_lef
  double tot = Cf * (k == 2 ? eta : 1.0) + dC * in.dlef + in.kq * (Cq + dql * in.dlef) * in.Q + (k == 2 ? dCm : 0.0);
  const double Cz_tot = quad_bcast<1>(tot);
  if (k == 2) tot += Cz_tot * (0.35 - xcg);                       // :347
  double Cr = lerp(r0, r1, a1);
  if (k == 2 && !(flags & FLAG_FIX_CLR)) Cr = 0.0;                // reference defect: _CLr is never loaded
  const double Cp = lerp(p0, p1, a1), Cb = lerp(b0, b1, a1);
  const double dCr = lerp(hr0, hr1, a2), dCp = lerp(hp0, hp1, a2);
  latd = in.kb * (Cr + dCr * in.dlef) * in.R + in.kb * (Cp + dCp * in.dlef) * in.P + (k == 0 ? 0.0 : Cb * in.beta);
  Q2STAMP(3)
  return tot;
}

// Static part (3-D / 2-D tables) of Cy_tot / Cn_tot / Cl_tot on sub-lanes 0 / 1 / 2 (C/nlplant.c:353-377, hifi_C,
// hifi_C_lef, hifi_rudder, hifi_ailerons); the damping part comes from quad_long, the cg coupling of Cn (:367) is
// applied by the consumer once both parts of Cy_tot are known.
template <typename TP>
F16_DEV double quad_lat(TP T, const double *xu, int s, int &status) {
  const QuadIn in = quad_inputs(xu);
  const int k = s < 2 ? s : 2;
  // (1) breakpoints: one axis per sub-lane
  int nX; double vX;
  const BrRaw rx = quad_br_load<true>(T, in.alpha, in.beta, in.el, s, nX, vX);
  const double a45 = T[OFF_BP_A1 + N_A2 - 1];
  F16_PHASE();
  const QuadBr qb = quad_br_cells(rx, nX, vX);
  if (qb.offa) status |= ST_ALPHA1 | ST_ALPHA2;
  if (qb.offb) status |= ST_BETA;
  if (qb.offd) status |= ST_EL;
  const bool hi_a = qb.ja > N_A2 - 2;
  if (hi_a && in.alpha > a45) status |= ST_ALPHA2;
  const int j2 = hi_a ? N_A2 - 2 : qb.ja;
  const int n1 = qb.jb * N_A1 + qb.ja, n2 = qb.jb * N_A2 + j2;
  struct { int j; } cd = {qb.jd};
  // (2) every table corner this role needs
  constexpr int SA3 = S_G3B, SB3 = S_G3B * N_A1, SD3 = S_G3B * N_A1 * N_B1;
  TP p3 = T + OFF_G3B + n1 * SA3 + (k > 0 ? k - 1 : 0);          // sub-lane 0 shadows Cn; its base is Cy
  const Q4 qlo = ld4(p3 + cd.j * SD3, SA3, SB3), qhi = ld4(p3 + (cd.j + 1) * SD3, SA3, SB3), q0 = ld4(p3 + D2_ZERO_NODE * SD3, SA3, SB3);
  constexpr int SA2 = S_G2A, SB2 = S_G2A * N_A1;
  TP pa = T + OFF_G2A + n1 * SA2;
  const Q4 qy = ld4(pa, SA2, SB2), qr = ld4(pa + 1 + k, SA2, SB2), qa = ld4(pa + 4 + k, SA2, SB2);
  constexpr int SAB = S_G2B, SBB = S_G2B * N_A2;
  TP pb = T + OFF_G2B + n2 * SAB;
  const Q4 ql = ld4(pb + 3 + k, SAB, SBB), qal = ld4(pb + 6 + k, SAB, SBB);
  F16_PHASE();
  // (3) arithmetic
  Axis a1, b, d2;
  quad_br_axes(qb, a1, b, d2);
  Axis a2 = a1;
  if (hi_a) { a2.j = N_A2 - 2; a2.l = 1.0; a2.m = 0.0; }
  const W4 W1 = bil_weights(a1, b), W2 = bil_weights(a2, b);
  const double C3 = lerp(bil4w(qlo, a1, b, W1), bil4w(qhi, a1, b, W1), d2), C30 = bil4w(q0, a1, b, W1);
  const double Cy = bil4w(qy, a1, b, W1), Cr30 = bil4w(qr, a1, b, W1), Ca20 = bil4w(qa, a1, b, W1);
  const double Clef = bil4w(ql, a2, b, W2), Ca20lef = bil4w(qal, a2, b, W2);
  const double base = k == 0 ? Cy : C3, base0 = k == 0 ? Cy : C30;
  const double dlefC = Clef - base0;                             // hifi_C_lef
  const double dr30 = Cr30 - base0;                              // hifi_rudder
  const double da20 = Ca20 - base0, da20lef = Ca20lef - Clef - da20;   // hifi_ailerons
  return base + dlefC * in.dlef + (da20 + da20lef * in.dlef) * in.dail + dr30 * in.drud;
}

}  // namespace f16
