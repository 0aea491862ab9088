// Internal helpers shared by the .hip translation units of liborigin_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/origin_hip.h"

struct origin_ctx {
  int device;
  hipStream_t stream;
  hipEvent_t ev_start[64];
  hipEvent_t ev_stop[64];
  bool ev_made[64];
  // grow-only scratch for partial reductions
  void *scratch;
  size_t scratch_bytes;
  int num_cu;
  // cached DCT cosine table (dct.hip)
  double *ctab;
  int ctab_nz, ctab_order;
};

void origin_set_error(const char *fmt, ...);
int origin_scratch(origin_ctx *ctx, size_t bytes, void **out);

#define ORIGIN_CHECK_ARG(cond, ...)       \
  do {                                    \
    if (!(cond)) {                        \
      origin_set_error(__VA_ARGS__);      \
      return ORIGIN_E_ARG;                \
    }                                     \
  } while (0)

#define ORIGIN_HIP(call)                                                              \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) {                                                           \
      origin_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                       __LINE__);                                                     \
      return e_ == hipErrorOutOfMemory ? ORIGIN_E_NOMEM : ORIGIN_E_HIP;               \
    }                                                                                 \
  } while (0)

#define ORIGIN_LAUNCH_CHECK() ORIGIN_HIP(hipGetLastError())

static inline int origin_use(origin_ctx *ctx) {
  if (!ctx) {
    origin_set_error("null context");
    return ORIGIN_E_ARG;
  }
  ORIGIN_HIP(hipSetDevice(ctx->device));
  return ORIGIN_OK;
}

#define ORIGIN_USE(ctx)            \
  do {                             \
    int r_ = origin_use(ctx);      \
    if (r_ != ORIGIN_OK) return r_; \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
