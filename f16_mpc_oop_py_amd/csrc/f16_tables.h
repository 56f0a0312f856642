// f16_tables.h -- layout of the fp64 aerodynamic table image that every dynamics workgroup
// stages into LDS (112,928 B of the 160 KiB/CU), and the host routine that builds it.
//
// The reference keeps 43 separate tables (C/hifi_F16_AeroData.c:109-1861), each with its own
// interpn() call (58 per Nlplant evaluation).  Here tables that share a grid are interleaved
// NODE-MAJOR: all coefficients of one grid node sit next to each other, so one aircraft fetches a
// cell corner of a whole table group with consecutive wide LDS reads and one address computation,
// and the five axis brackets are computed once per evaluation instead of once per table.
//
//   group  grid                     node payload (doubles)                                  nodes
//   G3A    ALPHA1 x BETA1 x DH1     Cx Cz Cm                                          3     1900
//   G3B    ALPHA1 x BETA1 x DH2     Cn Cl                                             2     1140
//   G2A    ALPHA1 x BETA1           Cy Cy_r30 Cn_r30 Cl_r30 Cy_a20 Cn_a20 Cl_a20 pad  8      380
//   G2B    ALPHA2 x BETA1           Cx_lef Cz_lef Cm_lef Cy_lef Cn_lef Cl_lef
//                                   Cy_a20_lef Cn_a20_lef Cl_a20_lef pad             10      266
//   G1A    ALPHA1                   CXq CYr CYp CZq CLr CLp CMq CNr CNp dCNbeta dCLbeta dCm 12  20
//   G1B    ALPHA2                   9 x delta_C*_lef damping, pad                    10       14
//   ETA    DH1                      eta_el                                            1        5
// Node order inside a group is the reference's: alpha fastest, then beta, then el (getLinIndex,
// C/mexndinterp.c:149-159).
#pragma once
#include <stdint.h>

namespace f16 {

constexpr int N_A1 = 20, N_A2 = 14, N_B1 = 19, N_D1 = 5, N_D2 = 3;

constexpr int OFF_BP_A1 = 0;                         // ALPHA1 breakpoints (ALPHA2 = its first 14)
constexpr int OFF_BP_B1 = OFF_BP_A1 + N_A1;          // 20
constexpr int OFF_BP_D1 = OFF_BP_B1 + N_B1;          // 39
constexpr int OFF_BP_D2 = OFF_BP_D1 + N_D1;          // 44
constexpr int OFF_G3A = 48;
constexpr int S_G3A = 3;
constexpr int OFF_G3B = OFF_G3A + S_G3A * N_A1 * N_B1 * N_D1;   // 5748
constexpr int S_G3B = 2;
constexpr int OFF_G2A = OFF_G3B + S_G3B * N_A1 * N_B1 * N_D2;   // 8028
constexpr int S_G2A = 8;
constexpr int OFF_G2B = OFF_G2A + S_G2A * N_A1 * N_B1;          // 11068
constexpr int S_G2B = 10;
constexpr int OFF_G1A = OFF_G2B + S_G2B * N_A2 * N_B1;          // 13728
constexpr int S_G1A = 12;
constexpr int OFF_G1B = OFF_G1A + S_G1A * N_A1;                 // 13968
constexpr int S_G1B = 10;
constexpr int OFF_ETA = OFF_G1B + S_G1B * N_A2;                 // 14108
constexpr int TABLE_IMAGE_DOUBLES = OFF_ETA + 8;                // 14116 (16-byte multiple)
static_assert(TABLE_IMAGE_DOUBLES % 2 == 0, "image is copied with 16-byte loads");
static_assert(TABLE_IMAGE_DOUBLES * 8 <= 160 * 1024, "must fit one CU's LDS");

// The same groups as SCALED INTEGERS (int32, value = k / 1e5 exactly -- the reference's data files carry five decimals),
// for the large-batch rollout kernels: half the LDS bytes per vertex gather (the LDS pipe is one of their two ceilings),
// node payloads padded to whole 16-byte reads.  The breakpoints stay doubles at the front of the image (same OFF_BP_*
// offsets, in doubles); table offsets / strides below are in ints.  Interpolating k instead of k / 1e5 and scaling the
// six totals once is linear algebra on the same numbers, <= a few ulp from the fp64-image path (default build only).
namespace i32 {
constexpr int BP_DOUBLES = 48;
constexpr int OFF_G3A = 2 * BP_DOUBLES, S_G3A = 4;
constexpr int OFF_G3B = OFF_G3A + S_G3A * N_A1 * N_B1 * N_D1, S_G3B = 2;
constexpr int OFF_G2A = OFF_G3B + S_G3B * N_A1 * N_B1 * N_D2, S_G2A = 8;
constexpr int OFF_G2B = OFF_G2A + S_G2A * N_A1 * N_B1, S_G2B = 12;
constexpr int OFF_G1A = OFF_G2B + S_G2B * N_A2 * N_B1, S_G1A = 12;
constexpr int OFF_G1B = OFF_G1A + S_G1A * N_A1, S_G1B = 12;
constexpr int OFF_ETA = OFF_G1B + S_G1B * N_A2;
constexpr int IMAGE_INTS = OFF_ETA + 8;                          // 16,624 ints = 66,496 B
constexpr double SCALE = 1e-5;
static_assert(OFF_G3B % 4 == 0 && OFF_G2A % 4 == 0 && OFF_G2B % 4 == 0 && OFF_G1A % 4 == 0 && OFF_G1B % 4 == 0 &&
              OFF_ETA % 4 == 0 && IMAGE_INTS % 4 == 0, "node payloads are read with 16-byte loads");
}  // namespace i32

// el = 0 is node 2 of DH1 and node 1 of DH2 (checked at image build time): the reference's
// `_Cx(alpha,beta,0)`-style calls (hifi_F16_AeroData.c:1892-1925) are plain 2-D lookups on that plane.
constexpr int D1_ZERO_NODE = 2, D2_ZERO_NODE = 1;

// lofi (Stevens & Lewis) image, kept in global/constant memory (2.9 KB as doubles x 744)
constexpr int LOFI_DAMP = 0, LOFI_DLDA = 108, LOFI_DLDR = 192, LOFI_DNDA = 276, LOFI_DNDR = 360;
constexpr int LOFI_CL = 444, LOFI_CN = 528, LOFI_CX = 612, LOFI_CM = 672, LOFI_CZ = 732;
constexpr int LOFI_IMAGE_DOUBLES = 744;

// Builds both images on the host (IEEE division int/scale == the reference's strtod, see
// tools/pack_tables.py).  Returns 0, or -1 when a structural assumption on the data fails.
int build_table_images(double *hifi /*[TABLE_IMAGE_DOUBLES]*/, double *lofi /*[LOFI_IMAGE_DOUBLES]*/);
// The scaled-integer image (call after build_table_images succeeded: same structural assumptions).
void build_table_image_i32(int32_t *img /*[i32::IMAGE_INTS]*/);

}  // namespace f16
