// f16_ctx.h -- library context shared by the translation units of libf16hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct f16_ctx {
  int device;
  double *d_tab;    // [TABLE_IMAGE_DOUBLES] hifi node-major image
  double *d_lofi;   // [LOFI_IMAGE_DOUBLES]
  int *d_tab32;     // [i32::IMAGE_INTS] hifi image as scaled integers (large-batch rollout)
  // single-aircraft scratch for the drop-in Nlplant symbol
  double *d_one;    // [18 + 18]
  double *h_one;    // pinned mirror
  // One-shot MPC calls keep NO mutable state that two calls could share:
  //  * the QP workspace (packed P and A'A + extras per aircraft) is allocated and freed PER CALL, stream-ordered
  //    (hipMallocFromPoolAsync / hipFreeAsync on the caller's stream; refused under stream capture), from this pool, whose
  //    release threshold keeps the memory cached between calls;
  //  * the dispatch-order history (iteration counts of the previous call | order derived from them, [2][B] int32) is
  //    kept per (stream, batch size): calls on one stream are ordered, calls on different streams never touch the same
  //    buffer.  Beyond F16_MAX_SCHED entries the oldest entry is recycled (round robin).
  hipMemPool_t pool;
  struct sched_entry { void *stream; long B; int32_t *buf; int valid; int tag; } sched[16];   // tag 0: a batch of B aircraft;
                                                        // (lo << 16) | hi: the pairs of a horizon sweep (B = their number)
  int n_sched;
};
#define F16_MAX_SCHED 16

namespace f16 {
int set_error(int code, const char *msg);
int hip_check(hipError_t e, const char *what);
}  // namespace f16
