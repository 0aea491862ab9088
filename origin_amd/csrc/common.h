// Internal helpers shared by the .hip translation units of liborigin_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/origin_hip.h"

// kernel classes timed by the built-in HIP-event profiler (origin_prof_*)
enum OriginKernelId {
  K_DCT_FIT = 0,
  K_DCT_SUMS,
  K_DCT_STANDARDIZE,
  K_DCT_CONTINUUM,
  K_O2,
  K_PCA_SELECT,
  K_PCA_BMEAN,
  K_PCA_GATHER,
  K_PCA_PROJECT,
  K_PCA_GRAM,
  K_PCA_EIG,
  K_PCA_UVEC,
  K_PCA_DEFLATE_DOT,
  K_PCA_DEFLATE_UPDATE,
  K_PCA_FLUSH,
  K_GLR_SPATIAL,
  K_GLR_SPECTRAL,
  K_GLR_BORDER,
  K_GLR_TABLES,
  K_LOCAL_MAX,
  K_SMALL,
  K_PCA_TOTAL,  // the whole greedy PCA as one scope (level 1; its kernels are level-2 scopes)
  K_COUNT
};

struct OriginProfEvent {
  hipEvent_t a, b;
  int id;
};

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
  // per-kernel-class timing with HIP events on the stream the kernels run on
  int prof_level;  // 0 off, 1 coarse (one scope per large kernel / per PCA run), 2 + every PCA kernel
  std::vector<OriginProfEvent> prof_pending;
  std::vector<hipEvent_t> prof_free;
  double prof_ms[K_COUNT];
  long prof_n[K_COUNT];
  // persistent workspace of origin_pca_run (owned by pca.hip)
  void *pca_ws;
  void (*pca_ws_free)(void *);
  // Auxiliary low-priority stream: HBM-bound passes nothing downstream waits for (the cont_dct
  // cube; the final F = X - U C of areas that have finished) run here in the shadow of the greedy
  // PCA's latency-bound kernels.  aux_join is recorded after the last piece of aux work;
  // origin_aux_join() makes the main stream wait for it, origin_sync() waits for both streams.
  hipStream_t aux_stream;
  hipEvent_t aux_fork, aux_join;
  bool aux_pending;
  void *aux_scratch;
  size_t aux_scratch_bytes;
  // pinned staging of origin_d2h_f32_as_f64 / origin_h2d_f64_as_f32: two 64 MiB buffers and two
  // events, made by the first conversion call on this context (all or nothing) and freed with it
  float *cvt_stage[2];
  hipEvent_t cvt_ev[2];
  bool cvt_ready;
  // Side stream of the row-band GLR (origin_glr_run_rows with ORIGIN_GLR_SIDE): restricted to the
  // first num_cu - reserve compute units, so that a band started while the greedy PCA still
  // iterates over its last areas leaves CUs to the PCA's small kernels.  side_join is recorded
  // behind the band; origin_glr_run_finish / origin_sync wait for it.
  hipStream_t side_stream;
  hipEvent_t side_fork, side_join;
  bool side_pending;
  // greedy PCA: called once, when at most pca_tail_max areas still iterate, after the areas that
  // have finished were written to the output (origin_pca_set_tail_hook, pca.hip)
  void (*pca_tail_hook)(void *user, int n_active, const int *areas);
  void *pca_tail_user;
  int pca_tail_max;
  // device blocks released by origin_free and kept for the next origin_malloc of their size
  // (ctx.hip: a hipMalloc of a 5 GB cube takes ~40 ms, a hipFree synchronises the device)
  void *alloc_cache;
};

// side stream plumbing (ctx.hip)
int origin_side_begin(origin_ctx *ctx);  // side waits for the main stream's work so far
int origin_side_end(origin_ctx *ctx);    // marks the end of the side work enqueued
int origin_side_join(origin_ctx *ctx);   // main waits for the side work

// aux stream plumbing (ctx.hip)
int origin_aux_begin(origin_ctx *ctx);                        // aux waits for the main stream's work so far
int origin_aux_end(origin_ctx *ctx);                          // marks the end of the aux work enqueued
int origin_aux_scratch(origin_ctx *ctx, size_t bytes, void **out);

// An event pair costs ~10 us of stream time on this hardware (barrier packets): 13 scopes in each
// of the 56 PCA iterations add 5 ms to a 95 ms step.  Scopes therefore carry a level; bench.py
// times its steps at level 1 (a handful of pairs per step) and takes the per-kernel PCA detail
// from an extra, untimed step at level 2.
int origin_prof_begin(origin_ctx *ctx, int id);  // returns the entry index
void origin_prof_end(origin_ctx *ctx, int entry);

// RAII: times everything enqueued on ctx->stream during its lifetime as kernel class `id`
struct ProfScope {
  origin_ctx *ctx;
  int entry;
  ProfScope(origin_ctx *c, int id, int level = 1) : ctx(c), entry(-1) {
    if (ctx->prof_level >= level) entry = origin_prof_begin(ctx, id);
  }
  // close the scope and open a new one of class `id` (same level)
  void next(int id) {
    if (entry >= 0) {
      origin_prof_end(ctx, entry);
      entry = origin_prof_begin(ctx, id);
    }
  }
  ~ProfScope() {
    if (entry >= 0) origin_prof_end(ctx, entry);
  }
};

void origin_set_error(const char *fmt, ...);
int origin_scratch(origin_ctx *ctx, size_t bytes, void **out);

// glr_spatial_mfma.hip: matrix-core spatial GLR stage (one field, or one weighted field of a
// mosaic: W its weight map, accf = add to what the fields before left in out)
int origin_spatial_mfma_ok(int Ny, int Nx, int P);
// (ry0, nry / rx0, nrx: rows / columns of 64 x 64 regions to run, <= 0 = all of them)
int origin_spatial_mfma_launch(origin_ctx *ctx, int terms, const float *A, const float *W,
                               const float *taps, int Nz, int Ny, int Nx, int P, int accf,
                               float *out, int ry0 = 0, int nry = 0, int rx0 = 0, int nrx = 0);
long origin_spatial_mfma_count(int terms, int Nz, int Ny, int Nx, int P);

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

// hipFuncSetAttribute (the dynamic-LDS ceiling of a kernel) holds per DEVICE, and a process may
// drive several devices from several threads (origin_amd/session.py: one context and one thread
// per GPU): the statements run once per device, under a lock -- a second thread on the same device
// must not launch before the attribute is there.  They may leave through ORIGIN_HIP.
struct OriginPerDeviceOnce {
  std::mutex mu;
  unsigned long long done = 0;
};
#define ORIGIN_ONCE_PER_DEVICE(ctx, state, ...)                          \
  do {                                                                   \
    std::lock_guard<std::mutex> lk_((state).mu);                         \
    const unsigned long long bit_ = 1ull << ((ctx)->device & 63);        \
    if (!((state).done & bit_)) {                                        \
      __VA_ARGS__;                                                       \
      (state).done |= bit_;                                              \
    }                                                                    \
  } while (0)
