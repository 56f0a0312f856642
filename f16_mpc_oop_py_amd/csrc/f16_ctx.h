// f16_ctx.h -- library context shared by the translation units of libf16hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct f16_ctx {
  int device;
  double *d_tab;    // [TABLE_IMAGE_DOUBLES] hifi node-major image
  double *d_lofi;   // [LOFI_IMAGE_DOUBLES]
  // single-aircraft scratch for the drop-in Nlplant symbol
  double *d_one;    // [18 + 18]
  double *h_one;    // pinned mirror
  void *d_work;     // QP workspace (packed P and A'A per aircraft), grown on demand
  size_t work_bytes;
  // one-shot MPC calls: iteration counts of the last call | dispatch order derived from them ([2][sched_B] int32)
  int32_t *d_sched;
  long sched_B;
  int sched_valid;
};

namespace f16 {
int set_error(int code, const char *msg);
int hip_check(hipError_t e, const char *what);
}  // namespace f16
