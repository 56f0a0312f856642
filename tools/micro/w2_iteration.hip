// Microbenchmark (diagnostic): what an ADMM iteration of the condensed MPC QP (N = 30) would cost with TWO
// wavefronts per aircraft (eight waves per CU, two per SIMD, six two-wave barriers per iteration) against the shipped mapping (one
// wavefront per aircraft, four per CU, no barrier).  Both variants are SYNTHETIC: the instruction mix of one iteration as counted in
// the shipped kernel's ISA / as laid out in DESIGN.md for the two-wave mapping -- fp64 FMAs with three register operands, fp64 adds,
// DPP moves, accumulation-register reads, LDS reads / writes of the real widths with real write -> read hand-overs through LDS --
// on dummy data.  The one-wave variant calibrates the method against the stamped kernel (6.9 k cycles per iteration).
//   hipcc --offload-arch=gfx950 -O3 -o w2_iteration w2_iteration.hip && ./w2_iteration
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef double d2_t __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm(double v, int src) {
  const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
#define SYNC1() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// ---------------------------------------------------------------------------------------------------------------------------
// one wavefront per aircraft (the shipped mapping): per iteration 150 + 144 + 150 FMAs, ~150 other fp64, 18 + 9 + 36 + ... LDS
// operations, 24 bpermutes, five hand-overs through LDS
// ABL: ablations of the one-wave model (what each ingredient of an iteration costs under real contention): 1 no ds_bpermute in the
// mat-vec tail, 2 stage-1 totals from the lane's own sums (no records round trip), 3 the same for stage 3, 4 no projection arithmetic,
// 5 both halves of the inverse in registers (no 16-byte LDS reads of the B blocks), 6 the transposed mat-vec parts exchanged through LDS
// records (6 + 6 accesses of 16 bytes) instead of 24 ds_bpermute
template <int ABL>
__global__ __launch_bounds__(64, 1) void k_one(double *out, int iters) {
  __shared__ __attribute__((aligned(16))) double lds[5120];
  const int l = threadIdx.x;
  for (int i = l; i < 5120; i += 64) lds[i] = 1e-3 * (i % 97);
  __syncthreads();
  double G[30], KA[36], KB2[ABL == 5 ? 36 : 1], x[3] = {0.1, 0.2, 0.3}, z[6] = {0, 0, 0, 0, 0, 0}, y[6] = {0, 0, 0, 0, 0, 0}, lc[24];
#pragma unroll
  for (int i = 0; i < 30; ++i) G[i] = 1e-3 * (i + l);
#pragma unroll
  for (int i = 0; i < 36; ++i) KA[i] = 1e-3 * (i + 2 * l);
#pragma unroll
  for (int i = 0; i < 24; ++i) lc[i] = 1.0 + 1e-3 * i;
  if (ABL == 5) {
#pragma unroll
    for (int i = 0; i < 36; ++i) KB2[i] = 2e-3 * (i + l);
  }
  const int job = l % 21, sp = l / 21;
  double *const rec1 = lds + 660 + l * 22, *const rec3 = lds + (l < 63 ? job : 21) * 30 + 2 * (sp % 3);
  const double *const w1 = lds + 4320 + 30 * (job % 6) + 2 * (sp % 3), *const w3 = lds + 4952 + 4 * (5 * (job % 6) + 3);
  const int c1 = 660 + ((l * 7) % 63) * 22, c3 = ((l * 5) % 21) * 30 + 3 * (l & 1);
  for (int it = 0; it < iters; ++it) {
    {   // stage 1: 9 window reads of 16 B, 150 FMAs, 5 x (16 B + 8 B) writes
      double acc[15];
#pragma unroll
      for (int i = 0; i < 15; ++i) acc[i] = 0.0;
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        const d2_t a = *reinterpret_cast<const d2_t *>(w1 + 6 * m);
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int e = m - u;
          if (e >= 0 && e < 5) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { acc[3 * e + c] = fma(G[6 * u + c], a.x, acc[3 * e + c]); acc[3 * e + c] = fma(G[6 * u + 3 + c], a.y, acc[3 * e + c]); }
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 5; ++e) { *reinterpret_cast<d2_t *>(rec1 + 4 * e) = d2_t{acc[3 * e], acc[3 * e + 1]}; rec1[4 * e + 2] = acc[3 * e + 2]; }
    }
    SYNC1();
    {   // totals: 9 x (16 B + 8 B) reads, 27 adds, one DPP exchange; rhs
      double t[3] = {0, 0, 0};
      if (ABL == 2) { t[0] = rec1[0]; t[1] = rec1[1]; t[2] = rec1[2]; }
      else {
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const d2_t a = *reinterpret_cast<const d2_t *>(lds + c1 + 22 * (k % 3) + 4 * (k / 3));
        t[0] += a.x; t[1] += a.y; t[2] += lds[c1 + 22 * (k % 3) + 4 * (k / 3) + 2];
      }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) { if (ABL != 2) t[c] += dpp<0xB1>(t[c]); lds[4764 + 3 * (l >> 1) + c] = lc[c] * x[c] - lc[3 + c] + t[c]; }
    }
    SYNC1();
    double xk[3];
    {   // x~ = K^-1 rhs: 18 B-block reads of 16 B, 9 vector reads, 144 FMAs (A blocks from accumulation registers: modelled by
        // the allocator itself -- 36 live doubles too many), 24 bpermutes, quad reduction, write x~
      double Bk[36], xr[6], xa[6], xb[6];
      const d2_t *ki = reinterpret_cast<const d2_t *>(lds + 2160) + (l < 60 ? l : 59);
      if (ABL == 5) {
#pragma unroll
        for (int m = 0; m < 36; ++m) Bk[m] = KB2[m];
      } else {
#pragma unroll
      for (int m = 0; m < 18; ++m) { const d2_t b = ki[m * 60]; Bk[2 * m] = b.x; Bk[2 * m + 1] = b.y; }
      }
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const d2_t a = *reinterpret_cast<const d2_t *>(lds + 4764 + 6 * (l >> 2) % 84 + 2 * m), b = *reinterpret_cast<const d2_t *>(lds + 4764 + 6 * ((l >> 2) + 3) % 84 + 2 * m),
                   c = *reinterpret_cast<const d2_t *>(lds + 4764 + 6 * ((l >> 2) + 5) % 84 + 2 * m);
        xr[2 * m] = a.x; xr[2 * m + 1] = a.y; xa[2 * m] = b.x; xa[2 * m + 1] = b.y; xb[2 * m] = c.x; xb[2 * m + 1] = c.y;
      }
      double yd[6], ta[6], tb[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) { ta[j] = 0; tb[j] = 0; }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          s0 = fma(KA[6 * i + j], xa[j], s0); s1 = fma(Bk[6 * i + j], xb[j], s1);
          ta[j] = fma(KA[6 * i + j], xr[i], ta[j]); tb[j] = fma(Bk[6 * i + j], xr[i], tb[j]);
        }
        yd[i] = s0 + s1;
      }
      double y6[6];
      if (ABL == 6) {   // the transposed parts through LDS records (14 doubles per lane: 28 banks) instead of 24 ds_bpermute
        double *o = lds + l * 14;
#pragma unroll
        for (int m = 0; m < 3; ++m) { *reinterpret_cast<d2_t *>(o + 2 * m) = d2_t{ta[2 * m], ta[2 * m + 1]}; *reinterpret_cast<d2_t *>(o + 6 + 2 * m) = d2_t{tb[2 * m], tb[2 * m + 1]}; }
        SYNC1();
        const double *pa = lds + ((l + 4) & 63) * 14, *pb = lds + ((l + 8) & 63) * 14 + 6;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const d2_t a = *reinterpret_cast<const d2_t *>(pa + 2 * m), b = *reinterpret_cast<const d2_t *>(pb + 2 * m);
          ta[2 * m] = a.x; ta[2 * m + 1] = a.y; tb[2 * m] = b.x; tb[2 * m + 1] = b.y;
        }
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double sm = (ABL == 1 || ABL == 6) ? yd[j] + ta[j] + tb[j] : yd[j] + bperm(ta[j], (l + 4) & 63) + bperm(tb[j], (l + 8) & 63);
        sm += dpp<0xB1>(sm); sm += dpp<0x4E>(sm);
        y6[j] = sm;
      }
      if ((l & 3) == 0 && l < 60) {
        double *o = lds + 4952 + 4 * (2 * (l >> 2) + 7);
        *reinterpret_cast<d2_t *>(o) = d2_t{y6[0], y6[1]}; o[2] = y6[2];
        *reinterpret_cast<d2_t *>(o + 4) = d2_t{y6[3], y6[4]}; o[6] = y6[5];
      }
      xk[0] = y6[0]; xk[1] = y6[1]; xk[2] = y6[2];
    }
    SYNC1();
    {   // stage 3: 9 x (16 B + 8 B) window reads, 150 FMAs, 5 writes of 16 B
      double acc[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) acc[i] = 0.0;
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        const d2_t a = *reinterpret_cast<const d2_t *>(w3 + 4 * m);
        const double b = w3[4 * m + 2];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int e = m + u - 4;
          if (e >= 0 && e < 5) {
#pragma unroll
            for (int r = 0; r < 2; ++r) { acc[2 * e + r] = fma(G[6 * u + 3 * r], a.x, acc[2 * e + r]); acc[2 * e + r] = fma(G[6 * u + 3 * r + 1], a.y, acc[2 * e + r]); acc[2 * e + r] = fma(G[6 * u + 3 * r + 2], b, acc[2 * e + r]); }
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 5; ++e) *reinterpret_cast<d2_t *>(rec3 + 6 * e) = d2_t{acc[2 * e], acc[2 * e + 1]};
    }
    SYNC1();
    {   // totals (18 reads of 8 B, 15 adds), projection of six rows, w writes
      double z3[3] = {0, 0, 0};
      if (ABL == 3) { z3[0] = rec3[0]; z3[1] = rec3[1]; z3[2] = rec3[6]; }
      else {
#pragma unroll
      for (int T = 0; T < 6; ++T)
#pragma unroll
        for (int c = 0; c < 3; ++c) z3[c] += lds[c3 + 30 * (T % 3) + c + 6 * (T / 3)];
      }
      if (ABL == 4) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { x[c] = xk[c]; z[c] = z3[c]; lds[4320 + 6 * (l >> 1) % 200 + c] = z3[c]; lds[4764 + 96 + 3 * (l >> 1) % 90 + c] = xk[c]; }
      } else
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        x[c] = 1.6 * xk[c] - 0.6 * x[c];
        const double zr = 1.6 * z3[c] - 0.6 * z[c];
        const double zn = fmin(fmax(fma(y[c], lc[6], zr), lc[7 + c]), lc[10 + c]);
        const double d = lc[13] * (zr - zn);
        y[c] += d; z[c] = zn;
        lds[4320 + 6 * (l >> 1) % 200 + c] = lc[14 + c] * (lc[13] * zn - y[c]);
        const double zr2 = 1.6 * xk[c] - 0.6 * z[3 + c];
        const double zn2 = fmin(fmax(fma(y[3 + c], lc[6], zr2), lc[17 + c]), lc[20 + c]);
        const double d2 = lc[13] * (zr2 - zn2);
        y[3 + c] += d2; z[3 + c] = zn2;
        lds[4764 + 96 + 3 * (l >> 1) % 90 + c] = lc[23] * (lc[13] * zn2 - y[3 + c]);
      }
    }
    SYNC1();
  }
  out[blockIdx.x * 64 + l] = x[0] + x[1] + x[2] + z[0] + z[5] + y[1] + y[4];
}

// ---------------------------------------------------------------------------------------------------------------------------
// two wavefronts per aircraft: per wave and iteration 75 + 72 + 75 FMAs (one kept row of a Toeplitz block, one 6 x 6
// block of the symmetric inverse per lane, everything in architectural registers), partial sums and the transposed mat-vec parts
// through LDS records, six workgroup barriers
template <bool BARRIER>
__global__ __launch_bounds__(128, 4) void k_two(double *out, int iters) {
  __shared__ __attribute__((aligned(16))) double lds[5120];
  const int l = threadIdx.x;
  for (int i = l; i < 5120; i += 128) lds[i] = 1e-3 * (i % 97);
  __syncthreads();
  double G[15], KB[36], x[3] = {0.1, 0.2, 0.3}, z[3] = {0, 0, 0}, y[3] = {0, 0, 0}, lc[12];
#pragma unroll
  for (int i = 0; i < 15; ++i) G[i] = 1e-3 * (i + l);
#pragma unroll
  for (int i = 0; i < 36; ++i) KB[i] = 1e-3 * (i + 2 * l);
#pragma unroll
  for (int i = 0; i < 12; ++i) lc[i] = 1.0 + 1e-3 * i;
  const int job = l % 21;
  double *const rec1 = lds + 700 + (l % 126) * 22, *const rec3 = lds + job * 30 + (l % 6);
  const double *const w1 = lds + 4500 + 30 * (job % 6) + (l % 6), *const w3 = lds + 4900 + 4 * (5 * (job % 6) + 3);
  const int c1 = 700 + ((l * 7) % 120) * 22, c3 = ((l * 5) % 21) * 30 + 3 * (l & 1), cm = 2400 + ((l * 3) % 100) * 18;
  for (int it = 0; it < iters; ++it) {
    {   // stage 1: one kept row per lane: 9 window reads of 8 B, 75 FMAs, 5 x (16 B + 8 B) writes
      double acc[15];
#pragma unroll
      for (int i = 0; i < 15; ++i) acc[i] = 0.0;
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        const double a = w1[6 * m];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int e = m - u;
          if (e >= 0 && e < 5) {
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[3 * e + c] = fma(G[3 * u + c], a, acc[3 * e + c]);
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 5; ++e) { *reinterpret_cast<d2_t *>(rec1 + 4 * e) = d2_t{acc[3 * e], acc[3 * e + 1]}; rec1[4 * e + 2] = acc[3 * e + 2]; }
    }
    if (BARRIER) __syncthreads(); else SYNC1();
    {   // totals: four lanes per step, nine records each, two DPP levels; rhs
      double t[3] = {0, 0, 0};
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const d2_t a = *reinterpret_cast<const d2_t *>(lds + c1 + 22 * (k % 3) + 4 * (k / 3));
        t[0] += a.x; t[1] += a.y; t[2] += lds[c1 + 22 * (k % 3) + 4 * (k / 3) + 2];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) { t[c] += dpp<0xB1>(t[c]); t[c] += dpp<0x4E>(t[c]); }
      if ((l & 3) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) lds[4700 + 3 * (l >> 2) + c] = lc[c] * x[c] - lc[3 + c] + t[c];
      }
    }
    if (BARRIER) __syncthreads(); else SYNC1();
    {   // x~ = K^-1 rhs: one block per lane used both ways (72 FMAs), 6 vector reads of 16 B, 6 record writes of 16 B
      double xr[6], xc[6];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const d2_t a = *reinterpret_cast<const d2_t *>(lds + 4700 + 6 * (l >> 3) % 84 + 2 * m), b = *reinterpret_cast<const d2_t *>(lds + 4700 + 6 * ((l >> 3) + (l & 7) + 1) % 84 + 2 * m);
        xr[2 * m] = a.x; xr[2 * m + 1] = a.y; xc[2 * m] = b.x; xc[2 * m + 1] = b.y;
      }
      double yd[6], yt[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) yt[j] = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double s0 = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) { s0 = fma(KB[6 * i + j], xc[j], s0); yt[j] = fma(KB[6 * i + j], xr[i], yt[j]); }
        yd[i] = s0;
      }
      double *o = lds + 2400 + (l % 120) * 18;
#pragma unroll
      for (int m = 0; m < 3; ++m) { *reinterpret_cast<d2_t *>(o + 2 * m) = d2_t{yd[2 * m], yd[2 * m + 1]}; *reinterpret_cast<d2_t *>(o + 6 + 2 * m) = d2_t{yt[2 * m], yt[2 * m + 1]}; }
    }
    if (BARRIER) __syncthreads(); else SYNC1();
    double xk[3];
    {   // x~ totals: four lanes per step, four records of three doubles each, two DPP levels; write x~
      double t[3] = {0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const d2_t a = *reinterpret_cast<const d2_t *>(lds + cm + 18 * k);
        t[0] += a.x; t[1] += a.y; t[2] += lds[cm + 18 * k + 2];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) { t[c] += dpp<0xB1>(t[c]); t[c] += dpp<0x4E>(t[c]); xk[c] = t[c]; }
      if ((l & 3) == 0) { double *o = lds + 4900 + 4 * ((l >> 2) + 7); *reinterpret_cast<d2_t *>(o) = d2_t{t[0], t[1]}; o[2] = t[2]; }
    }
    if (BARRIER) __syncthreads(); else SYNC1();
    {   // stage 3: 9 x (16 B + 8 B) window reads, 75 FMAs, 5 writes of 8 B
      double acc[5] = {0, 0, 0, 0, 0};
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        const d2_t a = *reinterpret_cast<const d2_t *>(w3 + 4 * m);
        const double b = w3[4 * m + 2];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int e = m + u - 4;
          if (e >= 0 && e < 5) { acc[e] = fma(G[3 * u], a.x, acc[e]); acc[e] = fma(G[3 * u + 1], a.y, acc[e]); acc[e] = fma(G[3 * u + 2], b, acc[e]); }
        }
      }
#pragma unroll
      for (int e = 0; e < 5; ++e) rec3[6 * e] = acc[e];
    }
    if (BARRIER) __syncthreads(); else SYNC1();
    {   // totals of three rows (18 reads of 8 B, 15 adds) -- the command / rate lanes read x~ instead --, projection of three rows, w
      double z3[3] = {0, 0, 0};
      if ((l & 2) == 0) {
#pragma unroll
        for (int T = 0; T < 6; ++T)
#pragma unroll
          for (int c = 0; c < 3; ++c) z3[c] += lds[c3 + 30 * (T % 3) + c + 6 * (T / 3)];
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) z3[c] = lds[4900 + 4 * ((l >> 2) + 7) + c] - lds[4900 + 4 * ((l >> 2) + 6) + c];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        x[c] = 1.6 * xk[c] - 0.6 * x[c];
        const double zr = 1.6 * z3[c] - 0.6 * z[c];
        const double zn = fmin(fmax(fma(y[c], lc[6], zr), lc[7]), lc[8 + c % 2]);
        const double d = lc[10] * (zr - zn);
        y[c] += d; z[c] = zn;
        lds[4500 + 6 * (l >> 2) % 200 + 3 * (l & 1) + c] = lc[11] * (lc[10] * zn - y[c]);
      }
    }
    if (BARRIER) __syncthreads(); else SYNC1();
  }
  out[blockIdx.x * 128 + l] = x[0] + x[1] + x[2] + z[0] + z[2] + y[1];
}

template <typename F>
static void run(const char *name, F kern, int block, int grid, int iters) {
  double *d;
  (void)hipMalloc(&d, sizeof(double) * block * grid);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, d, 10);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double aircraft = grid, rounds = aircraft / 1024.0;          // 1024 aircraft resident on the chip in either mapping
  const double us_per_iter = ms * 1e3 / iters / rounds;
  printf("%-34s %d aircraft x %d iterations: %.3f ms -> %.2f us per iteration with 1024 aircraft in flight (%.0f cycles at 2.3 GHz); "
         "4096 solves of 447 iterations + 22 %% other: %.2f ms\n", name, grid, iters, ms, us_per_iter, us_per_iter * 2300, us_per_iter * 447 * 4 / 0.78 / 1e3);
  (void)hipFree(d);
}

int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run("one wavefront per aircraft", k_one<0>, 64, 4096, 400);
    if (rep == 1) {
      run("  - without the 24 ds_bpermute", k_one<1>, 64, 4096, 400);
      run("  - stage-1 totals without records", k_one<2>, 64, 4096, 400);
      run("  - stage-3 totals without records", k_one<3>, 64, 4096, 400);
      run("  - without the projection", k_one<4>, 64, 4096, 400);
      run("  - B blocks in registers too", k_one<5>, 64, 4096, 400);
      run("  - transposed parts through LDS", k_one<6>, 64, 4096, 400);
    }
    run("two wavefronts per aircraft", k_two<true>, 128, 4096, 400);
    run("  ... without its six barriers (bound)", k_two<false>, 128, 4096, 400);
  }
  return 0;
}
