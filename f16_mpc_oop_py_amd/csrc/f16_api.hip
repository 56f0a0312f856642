// f16_api.hip -- context lifetime, error reporting and the two drop-in symbols of the reference .so.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/f16_hip.h"
#include "f16_ctx.h"
#include "f16_tables.h"

namespace f16 {

static thread_local char g_err[512] = "";

int set_error(int code, const char *msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}

int hip_check(hipError_t e, const char *what) {
  if (e == hipSuccess) return F16_OK;
  snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
  return F16_EHIP;
}

}  // namespace f16

using namespace f16;

extern "C" const char *f16_last_error(void) { return g_err; }
extern "C" size_t f16_table_image_doubles(void) { return TABLE_IMAGE_DOUBLES; }

extern "C" int f16_create(f16_ctx **out, int device) {
  if (!out) return set_error(F16_EINVAL, "out is NULL");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return set_error(F16_ENOGPU, "no HIP device visible");
  if (device < 0 || device >= n) return set_error(F16_EINVAL, "device index out of range");
  int rc;
  if ((rc = hip_check(hipSetDevice(device), "hipSetDevice"))) return rc;
  std::vector<double> img(TABLE_IMAGE_DOUBLES), lofi(LOFI_IMAGE_DOUBLES);
  if (build_table_images(img.data(), lofi.data())) return set_error(F16_EINVAL, "table data violates a layout assumption");
  std::vector<int32_t> img32(i32::IMAGE_INTS);
  build_table_image_i32(img32.data());
  f16_ctx *c = new f16_ctx();
  c->device = device;
  if ((rc = hip_check(hipMalloc(&c->d_tab, sizeof(double) * TABLE_IMAGE_DOUBLES), "hipMalloc tables")) ||
      (rc = hip_check(hipMalloc(&c->d_lofi, sizeof(double) * LOFI_IMAGE_DOUBLES), "hipMalloc lofi")) ||
      (rc = hip_check(hipMalloc(&c->d_tab32, sizeof(int32_t) * i32::IMAGE_INTS), "hipMalloc tables (int)")) ||
      (rc = hip_check(hipMemcpy(c->d_tab32, img32.data(), sizeof(int32_t) * i32::IMAGE_INTS, hipMemcpyHostToDevice), "upload tables (int)")) ||
      (rc = hip_check(hipMalloc(&c->d_one, sizeof(double) * 36), "hipMalloc scratch")) ||
      (rc = hip_check(hipHostMalloc(&c->h_one, sizeof(double) * 36), "hipHostMalloc scratch")) ||
      (rc = hip_check(hipMemcpy(c->d_tab, img.data(), sizeof(double) * TABLE_IMAGE_DOUBLES, hipMemcpyHostToDevice), "upload tables")) ||
      (rc = hip_check(hipMemcpy(c->d_lofi, lofi.data(), sizeof(double) * LOFI_IMAGE_DOUBLES, hipMemcpyHostToDevice), "upload lofi"))) {
    f16_destroy(c);
    return rc;
  }
  {   // stream-ordered pool for per-call workspaces: keep freed blocks cached instead of returning them to the driver
    hipMemPoolProps props = {};
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = device;
    if ((rc = hip_check(hipMemPoolCreate(&c->pool, &props), "hipMemPoolCreate"))) { c->pool = nullptr; f16_destroy(c); return rc; }
    // Finite release threshold: the fast-path workspace of the documented batch sizes (1.9 GB at B = 8192, N = 30, wavefront
    // solver) stays cached between calls; what a long-horizon sweep leaves behind (mpc_big_sweep_job_doubles(150) = 4.45 MB per
    // aircraft at N = 150 -- 3.59 MB of it the solver's own operands, mpc_big_ws_doubles(150) --, 10.4 GB for N = 33..150 at B = 64) goes back to the driver at the next synchronisation instead of
    // staying invisible to torch's allocator until f16_destroy.  F16_POOL_KEEP_GB overrides the 4 GiB (e.g. 16 for repeated
    // horizon sweeps or one-shot calls above ~17,000 aircraft at N = 30, whose workspace would otherwise be re-mapped per call).
    uint64_t keep = 4ull << 30;
    if (const char *e = getenv("F16_POOL_KEEP_GB")) { const double g = atof(e); if (g > 0) keep = (uint64_t)(g * 1073741824.0); }
    (void)hipMemPoolSetAttribute(c->pool, hipMemPoolAttrReleaseThreshold, &keep);
  }
  *out = c;
  return F16_OK;
}

extern "C" void f16_destroy(f16_ctx *c) {
  if (!c) return;
  if (c->d_tab) (void)hipFree(c->d_tab);
  if (c->d_lofi) (void)hipFree(c->d_lofi);
  if (c->d_tab32) (void)hipFree(c->d_tab32);
  if (c->d_one) (void)hipFree(c->d_one);
  (void)hipDeviceSynchronize();       // in-flight calls may still own pool blocks / schedule buffers
  for (int i = 0; i < c->n_sched; ++i) if (c->sched[i].buf) (void)hipFree(c->sched[i].buf);
  if (c->pool) (void)hipMemPoolDestroy(c->pool);
  if (c->h_one) (void)hipHostFree(c->h_one);
  delete c;
}

extern "C" int f16_debug_read_tables(f16_ctx *ctx, double *h_out) {
  if (!ctx || !h_out) return set_error(F16_EINVAL, "NULL argument");
  return hip_check(hipMemcpy(h_out, ctx->d_tab, sizeof(double) * TABLE_IMAGE_DOUBLES, hipMemcpyDeviceToHost), "read tables");
}

extern "C" size_t f16_table_image_i32_ints(void) { return (size_t)i32::IMAGE_INTS; }
extern "C" int f16_debug_read_tables_i32(f16_ctx *ctx, int32_t *h_out) {
  if (!ctx || !h_out) return set_error(F16_EINVAL, "bad argument to f16_debug_read_tables_i32");
  return hip_check(hipMemcpy(h_out, ctx->d_tab32, sizeof(int32_t) * i32::IMAGE_INTS, hipMemcpyDeviceToHost), "read tables (int)");
}

// ------------------------------------------------------------------ drop-in symbols
// The reference loads one of two binaries (parameters.py:108-111); here xcg is configuration.
static std::mutex g_dropin_mu;
static f16_ctx *g_dropin_ctx = nullptr;
static double g_dropin_xcg = 0.25;
static unsigned g_dropin_flags = 0;

extern "C" void f16_dropin_config(double xcg, unsigned flags) {
  std::lock_guard<std::mutex> lk(g_dropin_mu);
  g_dropin_xcg = xcg;
  g_dropin_flags = flags;
}

// C/nlplant.c:23.  One aircraft through the same device code as the batched path: H2D 17 doubles,
// one 64-lane launch (tables read from L2), D2H 18 doubles.  Fails loudly (NaN outputs + stderr)
// when no GPU is present -- there is no CPU fallback in this library.
extern "C" void Nlplant(double *xu, double *xdot, int fidelity) {
  std::lock_guard<std::mutex> lk(g_dropin_mu);
  if (!g_dropin_ctx) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (f16_create(&g_dropin_ctx, dev) != F16_OK) {
      fprintf(stderr, "libf16hip Nlplant: %s\n", f16_last_error());
      for (int i = 0; i < 18; ++i) xdot[i] = NAN;
      return;
    }
  }
  f16_ctx *c = g_dropin_ctx;
  memcpy(c->h_one, xu, sizeof(double) * 17);
  hipError_t e = hipMemcpyAsync(c->d_one, c->h_one, sizeof(double) * 17, hipMemcpyHostToDevice, 0);
  int rc = F16_OK;
  if (e == hipSuccess) rc = f16_nlplant_batch(c, c->d_one, c->d_one + 18, nullptr, 1, 1, g_dropin_xcg, fidelity, g_dropin_flags, nullptr);
  if (e == hipSuccess && rc == F16_OK) e = hipMemcpyAsync(c->h_one + 18, c->d_one + 18, sizeof(double) * 18, hipMemcpyDeviceToHost, 0);
  if (e == hipSuccess && rc == F16_OK) e = hipStreamSynchronize(0);
  if (e != hipSuccess || rc != F16_OK) {
    fprintf(stderr, "libf16hip Nlplant: %s\n", e != hipSuccess ? hipGetErrorString(e) : f16_last_error());
    for (int i = 0; i < 18; ++i) xdot[i] = NAN;
    return;
  }
  memcpy(xdot, c->h_one + 18, sizeof(double) * 18);
}

// C/nlplant.c:467-490.  Three scalars of closed-form arithmetic: evaluated on the host, in the same
// expression order; the device twin is f16::atmos_dev (f16_plant.hpp).
extern "C" void atmos(double alt, double vt, double *coeff) {
  const double rho0 = 2.377e-3;
  const double tfac = 1 - .703e-5 * alt;
  double temp = 519.0 * tfac;
  if (alt >= 35000.0) temp = 390;
  const double rho = rho0 * pow(tfac, 4.14);
  const double mach = vt / sqrt(1.4 * 1716.3 * temp);
  const double qbar = .5 * rho * pow(vt, 2);
  double ps = 1715.0 * rho * temp;
  if (ps == 0) ps = 1715;
  coeff[0] = mach;
  coeff[1] = qbar;
  coeff[2] = ps;
}
