// f16_tables.cpp -- host-side construction of the node-major table image (see f16_tables.h).
#include "f16_tables.h"

#include <string.h>

#include "f16_tables_data.inc"

namespace f16 {

static const int g3a[3] = {F16_T_Cx, F16_T_Cz, F16_T_Cm};
static const int g3b[2] = {F16_T_Cn, F16_T_Cl};
static const int g2a[7] = {F16_T_Cy, F16_T_Cy_r30, F16_T_Cn_r30, F16_T_Cl_r30, F16_T_Cy_a20, F16_T_Cn_a20, F16_T_Cl_a20};
static const int g2b[9] = {F16_T_Cx_lef, F16_T_Cz_lef, F16_T_Cm_lef, F16_T_Cy_lef, F16_T_Cn_lef, F16_T_Cl_lef,
                           F16_T_Cy_a20_lef, F16_T_Cn_a20_lef, F16_T_Cl_a20_lef};
static const int g1a[12] = {F16_T_CXq, F16_T_CYr, F16_T_CYp, F16_T_CZq, F16_T_CLr, F16_T_CLp,
                            F16_T_CMq, F16_T_CNr, F16_T_CNp, F16_T_dCNbeta, F16_T_dCLbeta, F16_T_dCm};
static const int g1b[9] = {F16_T_dCXq_lef, F16_T_dCYr_lef, F16_T_dCYp_lef, F16_T_dCZq_lef, F16_T_dCLr_lef,
                           F16_T_dCLp_lef, F16_T_dCMq_lef, F16_T_dCNr_lef, F16_T_dCNp_lef};

static inline double hv(const int32_t *t, int i) { return (double)t[i] / F16_HIFI_SCALE; }

int build_table_images(double *img, double *lofi) {
  memset(img, 0, sizeof(double) * TABLE_IMAGE_DOUBLES);
  // structural assumptions the kernels rely on
  for (int i = 0; i < N_A2; ++i)
    if (f16_bp_alpha2[i] != f16_bp_alpha1[i]) return -1;          // ALPHA2 is a prefix of ALPHA1
  if (f16_bp_dh1[D1_ZERO_NODE] != 0 || f16_bp_dh2[D2_ZERO_NODE] != 0) return -1;
  if (f16_bp_dh2[0] != f16_bp_dh1[0] || f16_bp_dh2[2] != f16_bp_dh1[4]) return -1;

  for (int i = 0; i < N_A1; ++i) img[OFF_BP_A1 + i] = hv(f16_bp_alpha1, i);
  for (int i = 0; i < N_B1; ++i) img[OFF_BP_B1 + i] = hv(f16_bp_beta1, i);
  for (int i = 0; i < N_D1; ++i) img[OFF_BP_D1 + i] = hv(f16_bp_dh1, i);
  for (int i = 0; i < N_D2; ++i) img[OFF_BP_D2 + i] = hv(f16_bp_dh2, i);

  auto interleave = [&](int off, int stride, const int *ids, int nid, int nodes) {
    for (int c = 0; c < nid; ++c) {
      if (f16_hifi_sizes[ids[c]] != nodes) return -1;
      for (int n = 0; n < nodes; ++n) img[off + n * stride + c] = hv(f16_hifi_tables[ids[c]], n);
    }
    return 0;
  };
  int rc = 0;
  rc |= interleave(OFF_G3A, S_G3A, g3a, 3, N_A1 * N_B1 * N_D1);
  rc |= interleave(OFF_G3B, S_G3B, g3b, 2, N_A1 * N_B1 * N_D2);
  rc |= interleave(OFF_G2A, S_G2A, g2a, 7, N_A1 * N_B1);
  rc |= interleave(OFF_G2B, S_G2B, g2b, 9, N_A2 * N_B1);
  rc |= interleave(OFF_G1A, S_G1A, g1a, 12, N_A1);
  rc |= interleave(OFF_G1B, S_G1B, g1b, 9, N_A2);
  if (f16_hifi_sizes[F16_T_eta_el] != N_D1) return -1;
  for (int n = 0; n < N_D1; ++n) img[OFF_ETA + n] = hv(f16_tab_eta_el, n);
  if (rc) return -1;

  struct { const int32_t *src; int off, n; } L[] = {
      {f16_lofi_damp, LOFI_DAMP, 108}, {f16_lofi_dlda, LOFI_DLDA, 84}, {f16_lofi_dldr, LOFI_DLDR, 84},
      {f16_lofi_dnda, LOFI_DNDA, 84},  {f16_lofi_dndr, LOFI_DNDR, 84}, {f16_lofi_cl, LOFI_CL, 84},
      {f16_lofi_cn, LOFI_CN, 84},      {f16_lofi_cx, LOFI_CX, 60},     {f16_lofi_cm, LOFI_CM, 60},
      {f16_lofi_cz, LOFI_CZ, 12}};
  for (auto &e : L)
    for (int i = 0; i < e.n; ++i) lofi[e.off + i] = (double)e.src[i] / F16_LOFI_SCALE;
  return 0;
}

void build_table_image_i32(int32_t *img) {
  memset(img, 0, sizeof(int32_t) * i32::IMAGE_INTS);
  double *bp = reinterpret_cast<double *>(img);
  for (int i = 0; i < N_A1; ++i) bp[OFF_BP_A1 + i] = hv(f16_bp_alpha1, i);
  for (int i = 0; i < N_B1; ++i) bp[OFF_BP_B1 + i] = hv(f16_bp_beta1, i);
  for (int i = 0; i < N_D1; ++i) bp[OFF_BP_D1 + i] = hv(f16_bp_dh1, i);
  for (int i = 0; i < N_D2; ++i) bp[OFF_BP_D2 + i] = hv(f16_bp_dh2, i);
  auto interleave = [&](int off, int stride, const int *ids, int nid, int nodes) {
    for (int c = 0; c < nid; ++c)
      for (int n = 0; n < nodes; ++n) img[off + n * stride + c] = f16_hifi_tables[ids[c]][n];
  };
  interleave(i32::OFF_G3A, i32::S_G3A, g3a, 3, N_A1 * N_B1 * N_D1);
  interleave(i32::OFF_G3B, i32::S_G3B, g3b, 2, N_A1 * N_B1 * N_D2);
  interleave(i32::OFF_G2A, i32::S_G2A, g2a, 7, N_A1 * N_B1);
  interleave(i32::OFF_G2B, i32::S_G2B, g2b, 9, N_A2 * N_B1);
  interleave(i32::OFF_G1A, i32::S_G1A, g1a, 12, N_A1);
  interleave(i32::OFF_G1B, i32::S_G1B, g1b, 9, N_A2);
  for (int n = 0; n < N_D1; ++n) img[i32::OFF_ETA + n] = f16_tab_eta_el[n];
}

}  // namespace f16
