// f16_smallmat.hpp -- dense small-matrix building blocks executed by ONE wavefront on LDS-resident,
// row-major fp64 matrices (9x9 / 9x3 / 3x3 / 12x12 of the control chain: scipy/numpy calls in
// env.py:46-50,351 and utils.py:96-112,242-244).  A workgroup is exactly one 64-lane wave, so
// __syncthreads() is a (cheap) wave-level LDS fence between producer and consumer lanes.
#pragma once
#include <hip/hip_runtime.h>

namespace f16 {

#define F16_WAVE 64

__device__ __forceinline__ int lane_id() { return threadIdx.x; }

// C[m x n] = alpha * op(A) * op(B) + beta * C0   (C must not alias A or B; C0 may be C)
// op(A) is m x k: A stored [m][k] (TA=false) or [k][m] (TA=true); op(B) is k x n likewise.
template <bool TA, bool TB>
__device__ __forceinline__ void mm(double *C, const double *A, const double *B, int m, int k, int n, double alpha = 1.0,
                                   double beta = 0.0) {
  for (int e = lane_id(); e < m * n; e += F16_WAVE) {
    const int i = e / n, j = e - i * n;
    double s = 0.0;
    for (int p = 0; p < k; ++p) s += (TA ? A[p * m + i] : A[i * k + p]) * (TB ? B[j * k + p] : B[p * n + j]);
    C[e] = alpha * s + (beta != 0.0 ? beta * C[e] : 0.0);
  }
  __syncthreads();
}

__device__ __forceinline__ void copy(double *dst, const double *src, int n) {
  for (int e = lane_id(); e < n; e += F16_WAVE) dst[e] = src[e];
  __syncthreads();
}

// wave-wide sum / max (of NON-NEGATIVE values) on the DPP network: four butterfly steps inside each 16-lane row, two row
// broadcasts, the total read from lane 63 -- no LDS traffic (__shfl_xor is ds_bpermute: ~100 cycles per step and a
// share of the one LDS pipe; the Riccati doubling reduces three values per step, the ADMM termination test nine)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_masked_f64(double v) {      // lanes outside ROWMASK receive 0
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_all_f64(double v) {         // every row takes part: no `old` operand to set up
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <bool SUM>
__device__ __forceinline__ double wave_reduce_dpp(double v) {
  auto op = [](double a, double b) { return SUM ? a + b : fmax(a, b); };
  v = op(v, dpp_all_f64<0xB1>(v));               // quad_perm [1,0,3,2]
  v = op(v, dpp_all_f64<0x4E>(v));               // quad_perm [2,3,0,1]
  v = op(v, dpp_all_f64<0x141>(v));              // row_half_mirror
  v = op(v, dpp_all_f64<0x140>(v));              // row_mirror: every lane of a row holds the row total
  v = op(v, dpp_masked_f64<0x142, 0xA>(v));      // row_bcast:15 -> rows 1, 3
  v = op(v, dpp_masked_f64<0x143, 0xC>(v));      // row_bcast:31 -> rows 2, 3
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v) { return wave_reduce_dpp<false>(v); }      // v >= 0
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce_dpp<true>(v); }

// In-place inverse of the n x n matrix M (n <= 12) by Gauss-Jordan with partial pivoting.
// W: scratch [n][2n], f: scratch [n].  Result overwrites M.  Returns false on a zero pivot.
__device__ __forceinline__ bool inverse(double *M, int n, double *W, double *f) {
  const int w = 2 * n;
  for (int e = lane_id(); e < n * w; e += F16_WAVE) {
    const int i = e / w, j = e - i * w;
    W[e] = j < n ? M[i * n + j] : (j - n == i ? 1.0 : 0.0);
  }
  __syncthreads();
  bool ok = true;
  for (int p = 0; p < n; ++p) {
    int piv = p;
    double best = fabs(W[p * w + p]);
    for (int r = p + 1; r < n; ++r) {
      const double v = fabs(W[r * w + p]);
      if (v > best) { best = v; piv = r; }
    }
    if (!(best > 0.0)) ok = false;
    __syncthreads();
    if (piv != p) {
      for (int e = lane_id(); e < w; e += F16_WAVE) {
        const double t = W[p * w + e];
        W[p * w + e] = W[piv * w + e];
        W[piv * w + e] = t;
      }
      __syncthreads();
    }
    const double d = W[p * w + p];
    for (int e = lane_id(); e < n; e += F16_WAVE) f[e] = W[e * w + p];
    __syncthreads();
    for (int e = lane_id(); e < w; e += F16_WAVE) W[p * w + e] = W[p * w + e] / d;
    __syncthreads();
    for (int e = lane_id(); e < n * w; e += F16_WAVE) {
      const int i = e / w, j = e - i * w;
      if (i != p) W[e] -= f[i] * W[p * w + j];
    }
    __syncthreads();
  }
  for (int e = lane_id(); e < n * n; e += F16_WAVE) {
    const int i = e / n, j = e - i * n;
    M[e] = W[i * w + n + j];
  }
  __syncthreads();
  return ok;
}


// ---- packed symmetric storage (lower triangle, row-major): element (i,j), i>=j at i(i+1)/2 + j.
// With one row per lane the row reads hit bank slots (T_i + k) mod 32, T_i triangular numbers: a
// permutation over 32 consecutive i, so ds_read_b64 is conflict-free; column reads are contiguous.
__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// y = S * v for packed symmetric S (n x n); v, y in LDS (y must not alias v).  S may live in LDS or global.
__device__ __forceinline__ void symv(double *y, const double *S, const double *v, int n) {
  for (int i = lane_id(); i < n; i += F16_WAVE) {
    double s = 0.0;
    const double *row = S + tri(i, 0);
    for (int j = 0; j <= i; ++j) s += row[j] * v[j];
    for (int j = i + 1; j < n; ++j) s += S[tri(j, i)] * v[j];
    y[i] = s;
  }
  __syncthreads();
}

// In-place inverse of a packed SPD matrix by the symmetric sweep operator (Gauss-Jordan without pivoting,
// stable for SPD): after sweeping every pivot the array holds -S^{-1}; the sign is flipped at the end.
// c: scratch [n].  Returns false if a pivot is not positive.
__device__ __forceinline__ bool spd_inverse_packed(double *S, int n, double *c) {
  bool ok = true;
  for (int k = 0; k < n; ++k) {
    const double piv = S[tri(k, k)];
    if (!(piv > 0.0)) ok = false;
    const double d = 1.0 / piv;
    __syncthreads();
    for (int i = lane_id(); i < n; i += F16_WAVE) c[i] = i >= k ? S[tri(i, k)] : S[tri(k, i)];
    __syncthreads();
    for (int i = lane_id(); i < n; i += F16_WAVE) {
      double *row = S + tri(i, 0);
      const double ci = c[i];
      if (i == k) {
        for (int j = 0; j < k; ++j) row[j] = c[j] * d;
        row[k] = -d;
      } else {
        const double cid = ci * d;
        for (int j = 0; j <= i; ++j) {
          if (j == k) row[j] = cid;
          else row[j] -= cid * c[j];
        }
      }
    }
    __syncthreads();
  }
  for (int e = lane_id(); e < n * (n + 1) / 2; e += F16_WAVE) S[e] = -S[e];
  __syncthreads();
  return ok;
}

}  // namespace f16
