// Context, device memory, copies and HIP-event timers of liborigin_hip.so.
#include <sys/mman.h>

#include <algorithm>
#include <cstdlib>
#include <map>
#include <unordered_map>

#include <functional>

#include <cstring>
#include "common.h"

void origin_host_pool_run(int n, const std::function<void(int)> &task);  // thresh.hip (C++ linkage)

static thread_local char g_err[1024] = "";

static void alloc_cache_release(origin_ctx *ctx, bool destroy);  // (allocation cache, below)
static void alloc_cache_make(origin_ctx *ctx);
static size_t alloc_cache_spare(origin_ctx *ctx);

// strided (nz, ny, rowbytes) box copy, device to device; 16 bytes per thread when aligned
__global__ __launch_bounds__(256) void copy_box_kernel(char *__restrict__ dst, long dpy, long dpz,
                                                       const char *__restrict__ src, long spy,
                                                       long spz, int ny, long rowbytes, int vec) {
  const int z = blockIdx.z;
  const int y = blockIdx.y;
  const long x = ((long)blockIdx.x * 256 + threadIdx.x) * vec;
  if (y >= ny || x >= rowbytes) return;
  const char *s = src + (long)z * spz + (long)y * spy + x;
  char *d = dst + (long)z * dpz + (long)y * dpy + x;
  if (vec == 16) {
    *reinterpret_cast<uint4 *>(d) = *reinterpret_cast<const uint4 *>(s);
  } else if (vec == 4) {
    *reinterpret_cast<unsigned *>(d) = *reinterpret_cast<const unsigned *>(s);
  } else {
    *d = *s;
  }
}

void origin_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int origin_scratch(origin_ctx *ctx, size_t bytes, void **out) {
  if (bytes > ctx->scratch_bytes) {
    if (ctx->scratch) {
      ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
      ORIGIN_HIP(hipFree(ctx->scratch));
      ctx->scratch = nullptr;
      ctx->scratch_bytes = 0;
    }
    size_t want = bytes + bytes / 4 + (1 << 20);
    ORIGIN_HIP(hipMalloc(&ctx->scratch, want));
    ctx->scratch_bytes = want;
  }
  *out = ctx->scratch;
  return ORIGIN_OK;
}

// The side stream, made on first use: every CU but the last `reserve` (ORIGIN_GLR_SIDE_RESERVE,
// default an eighth of the chip: 32 of 256).  Measured at 3681 x 600 x 600 with the GLR's early
// bands beside the greedy PCA's tail (tools/tail_overlap_tune.sh, profiles/r03_tail_overlap_tune.txt):
// step 55.8 ms with no reserve (the PCA's small kernels wait for a GLR workgroup to leave a CU: no
// better than running the two in sequence, 55.9), 53.8 / 53.3 / 53.1 with 8 / 16 / 24 CUs,
// 51.9 with 32, 52.4 / 52.7 / 53.0 with 40 / 64 / 96.  Work enqueued on the stream afterwards starts
// behind everything the main stream has been given so far.
int origin_side_begin(origin_ctx *ctx) {
  if (!ctx->side_stream) {
    const char *e = getenv("ORIGIN_GLR_SIDE_RESERVE");
    const int ncu = ctx->num_cu > 0 ? std::min(ctx->num_cu, 256) : 256;
    int reserve = e ? atoi(e) : ncu / 8;
    reserve = std::max(0, std::min(reserve, ncu - 8));
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ncu - reserve; ++i) mask[i >> 5] |= 1u << (i & 31);
    if (hipExtStreamCreateWithCUMask(&ctx->side_stream, 8, mask) != hipSuccess) {
      // (no CU masks on this runtime: a stream of the lowest priority still gives correct results,
      // only less of an overlap -- the main stream's small kernels wait for CUs)
      (void)hipGetLastError();
      int lo = 0, hi = 0;
      ORIGIN_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      ORIGIN_HIP(hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, lo));
    }
    ORIGIN_HIP(hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
    ORIGIN_HIP(hipEventCreateWithFlags(&ctx->side_join, hipEventDisableTiming));
  }
  ORIGIN_HIP(hipEventRecord(ctx->side_fork, ctx->stream));
  ORIGIN_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0));
  return ORIGIN_OK;
}

int origin_side_end(origin_ctx *ctx) {
  ORIGIN_HIP(hipEventRecord(ctx->side_join, ctx->side_stream));
  ctx->side_pending = true;
  return ORIGIN_OK;
}

// the main stream waits for what the side stream was given
int origin_side_join(origin_ctx *ctx) {
  if (ctx->side_stream && ctx->side_pending) {
    ORIGIN_HIP(hipStreamWaitEvent(ctx->stream, ctx->side_join, 0));
    ctx->side_pending = false;
  }
  return ORIGIN_OK;
}

// The aux stream is created on first use, with the lowest priority the device offers: its
// thousands of HBM-bound workgroups must not sit in front of the one-block kernels of the PCA.
int origin_aux_begin(origin_ctx *ctx) {
  if (!ctx->aux_stream) {
    // ORIGIN_AUX_CUS=n: restrict the stream to n CUs (hipExtStreamCreateWithCUMask), leaving the
    // rest of the chip to the main stream's small kernels; default: lowest priority, all CUs
    const char *cus = getenv("ORIGIN_AUX_CUS");
    if (cus && atoi(cus) > 0) {
      const int n = std::min(atoi(cus), ctx->num_cu > 0 ? ctx->num_cu : 256);
      uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < n; ++i) mask[i >> 5] |= 1u << (i & 31);
      ORIGIN_HIP(hipExtStreamCreateWithCUMask(&ctx->aux_stream, 8, mask));
    } else {
      int lo = 0, hi = 0;
      ORIGIN_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo = least urgent
      ORIGIN_HIP(hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, lo));
    }
    ORIGIN_HIP(hipEventCreateWithFlags(&ctx->aux_fork, hipEventDisableTiming));
    ORIGIN_HIP(hipEventCreateWithFlags(&ctx->aux_join, hipEventDisableTiming));
  }
  ORIGIN_HIP(hipEventRecord(ctx->aux_fork, ctx->stream));
  ORIGIN_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->aux_fork, 0));
  return ORIGIN_OK;
}

int origin_aux_end(origin_ctx *ctx) {
  ORIGIN_HIP(hipEventRecord(ctx->aux_join, ctx->aux_stream));
  ctx->aux_pending = true;
  return ORIGIN_OK;
}

int origin_aux_scratch(origin_ctx *ctx, size_t bytes, void **out) {
  if (bytes > ctx->aux_scratch_bytes) {
    if (ctx->aux_scratch) {
      if (ctx->aux_stream) ORIGIN_HIP(hipStreamSynchronize(ctx->aux_stream));
      ORIGIN_HIP(hipFree(ctx->aux_scratch));
      ctx->aux_scratch = nullptr;
      ctx->aux_scratch_bytes = 0;
    }
    ORIGIN_HIP(hipMalloc(&ctx->aux_scratch, bytes + bytes / 4 + (1 << 20)));
    ctx->aux_scratch_bytes = bytes + bytes / 4 + (1 << 20);
  }
  *out = ctx->aux_scratch;
  return ORIGIN_OK;
}

static hipEvent_t prof_event(origin_ctx *ctx) {
  if (!ctx->prof_free.empty()) {
    hipEvent_t e = ctx->prof_free.back();
    ctx->prof_free.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

static void prof_drain(origin_ctx *ctx) {
  for (auto &p : ctx->prof_pending) {
    float ms = 0.f;
    if (p.a && p.b && hipEventSynchronize(p.b) == hipSuccess &&
        hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      ctx->prof_ms[p.id] += ms;
      ctx->prof_n[p.id] += 1;
    }
    if (p.a) ctx->prof_free.push_back(p.a);
    if (p.b) ctx->prof_free.push_back(p.b);
  }
  ctx->prof_pending.clear();
}

int origin_prof_begin(origin_ctx *ctx, int id) {
  OriginProfEvent p;
  p.a = prof_event(ctx);
  p.b = nullptr;
  p.id = id;
  if (p.a) (void)hipEventRecord(p.a, ctx->stream);
  ctx->prof_pending.push_back(p);
  return (int)ctx->prof_pending.size() - 1;
}

void origin_prof_end(origin_ctx *ctx, int entry) {
  if (entry < 0 || entry >= (int)ctx->prof_pending.size()) return;
  OriginProfEvent &p = ctx->prof_pending[entry];
  p.b = prof_event(ctx);
  if (p.b) (void)hipEventRecord(p.b, ctx->stream);
}

static const char *kKernelNames[K_COUNT] = {
    "dct_fit",         "dct_plane_sums",     "dct_standardize", "dct_continuum", "o2",
    "pca_select",      "pca_bmean",          "pca_gather",      "pca_project",   "pca_gram",
    "pca_eig",         "pca_uvec",
    "pca_deflate_dot", "pca_deflate_finish", "pca_flush", "glr_spatial",     "glr_spectral",  "glr_border", "glr_tables",
    "local_max",       "small",              "pca_total"};

extern "C" {

int origin_prof_enable(origin_ctx *ctx, int on) {
  ORIGIN_USE(ctx);
  prof_drain(ctx);
  ctx->prof_level = on < 0 ? 0 : (on > 2 ? 2 : on);
  return ORIGIN_OK;
}

int origin_prof_reset(origin_ctx *ctx) {
  ORIGIN_USE(ctx);
  prof_drain(ctx);
  memset(ctx->prof_ms, 0, sizeof(ctx->prof_ms));
  memset(ctx->prof_n, 0, sizeof(ctx->prof_n));
  return ORIGIN_OK;
}

int origin_prof_count(void) { return K_COUNT; }

int origin_prof_get(origin_ctx *ctx, int id, const char **name, double *total_ms, long *launches) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(id >= 0 && id < K_COUNT, "kernel id out of range");
  prof_drain(ctx);
  if (name) *name = kKernelNames[id];
  if (total_ms) *total_ms = ctx->prof_ms[id];
  if (launches) *launches = ctx->prof_n[id];
  return ORIGIN_OK;
}

const char *origin_last_error(void) { return g_err; }

int origin_abi_version(void) { return 1; }

int origin_device_count(int *count) {
  ORIGIN_CHECK_ARG(count, "count is null");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    origin_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return ORIGIN_E_NODEVICE;
  }
  *count = n;
  return ORIGIN_OK;
}

int origin_ctx_create(int device, origin_ctx **out) {
  ORIGIN_CHECK_ARG(out, "out is null");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    origin_set_error("no HIP device visible");
    return ORIGIN_E_NODEVICE;
  }
  ORIGIN_CHECK_ARG(device >= 0 && device < n, "device %d out of range [0,%d)", device, n);
  ORIGIN_HIP(hipSetDevice(device));
  origin_ctx *ctx = new origin_ctx();
  ctx->device = device;
  ctx->stream = nullptr;
  memset(ctx->ev_made, 0, sizeof(ctx->ev_made));
  ctx->scratch = nullptr;
  ctx->scratch_bytes = 0;
  ctx->num_cu = 0;
  ctx->ctab = nullptr;
  ctx->ctab_nz = ctx->ctab_order = 0;
  ctx->prof_level = false;
  ctx->pca_ws = nullptr;
  ctx->pca_ws_free = nullptr;
  ctx->aux_stream = nullptr;
  ctx->aux_fork = ctx->aux_join = nullptr;
  ctx->aux_pending = false;
  ctx->aux_scratch = nullptr;
  ctx->aux_scratch_bytes = 0;
  ctx->cvt_stage[0] = ctx->cvt_stage[1] = nullptr;
  ctx->cvt_ready = false;
  memset(ctx->prof_ms, 0, sizeof(ctx->prof_ms));
  memset(ctx->prof_n, 0, sizeof(ctx->prof_n));
  // ORIGIN_CTX_PRIORITY=high|low: queue priority of this context's main stream (measurements of
  // concurrent contexts: a latency-bound chain beside a chip-filling pass, tools/pca_glr_overlap.py)
  hipError_t e;
  const char *prio = getenv("ORIGIN_CTX_PRIORITY");
  // ORIGIN_CTX_CUS=n: this context's main stream may use the first n compute units only
  // (hipExtStreamCreateWithCUMask) -- leaves the others to a concurrent context's small kernels
  const char *cus = getenv("ORIGIN_CTX_CUS");
  int masked_cus = 0;
  if (cus && atoi(cus) > 0 && atoi(cus) < 256) {
    masked_cus = atoi(cus);
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < masked_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    e = hipExtStreamCreateWithCUMask(&ctx->stream, 8, mask);
  } else if (prio && (!strcmp(prio, "high") || !strcmp(prio, "low"))) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = least urgent, hi = most urgent
    e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio[0] == 'h' ? hi : lo);
  } else {
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  }
  if (e != hipSuccess) {
    delete ctx;
    origin_set_error("hipStreamCreate: %s", hipGetErrorString(e));
    return ORIGIN_E_HIP;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cu = prop.multiProcessorCount;
  if (ctx->num_cu <= 0) ctx->num_cu = 256;
  if (masked_cus > 0 && masked_cus < ctx->num_cu) ctx->num_cu = masked_cus;  // (grids are sized by it)
  alloc_cache_make(ctx);
  *out = ctx;
  return ORIGIN_OK;
}

int origin_ctx_destroy(origin_ctx *ctx) {
  if (!ctx) return ORIGIN_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (int i = 0; i < 64; ++i)
    if (ctx->ev_made[i]) {
      hipEventDestroy(ctx->ev_start[i]);
      hipEventDestroy(ctx->ev_stop[i]);
    }
  if (ctx->aux_stream) {
    hipStreamSynchronize(ctx->aux_stream);
    hipStreamDestroy(ctx->aux_stream);
    hipEventDestroy(ctx->aux_fork);
    hipEventDestroy(ctx->aux_join);
  }
  if (ctx->side_stream) {
    hipStreamSynchronize(ctx->side_stream);
    hipStreamDestroy(ctx->side_stream);
    hipEventDestroy(ctx->side_fork);
    hipEventDestroy(ctx->side_join);
  }
  if (ctx->cvt_ready) {
    for (int b = 0; b < 2; ++b) {
      hipHostFree(ctx->cvt_stage[b]);
      hipEventDestroy(ctx->cvt_ev[b]);
    }
  }
  alloc_cache_release(ctx, true);
  if (ctx->aux_scratch) hipFree(ctx->aux_scratch);
  if (ctx->scratch) hipFree(ctx->scratch);
  if (ctx->ctab) hipFree(ctx->ctab);
  if (ctx->pca_ws && ctx->pca_ws_free) ctx->pca_ws_free(ctx->pca_ws);
  prof_drain(ctx);
  for (hipEvent_t e : ctx->prof_free) hipEventDestroy(e);
  hipStreamDestroy(ctx->stream);
  delete ctx;
  return ORIGIN_OK;
}

int origin_sync(origin_ctx *ctx) {
  ORIGIN_USE(ctx);
  if (ctx->side_stream && ctx->side_pending) {
    ORIGIN_HIP(hipStreamSynchronize(ctx->side_stream));
    ctx->side_pending = false;
  }
  ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->aux_stream && ctx->aux_pending) {
    ORIGIN_HIP(hipStreamSynchronize(ctx->aux_stream));
    ctx->aux_pending = false;
  }
  return ORIGIN_OK;
}

int origin_aux_join(origin_ctx *ctx) {
  ORIGIN_USE(ctx);
  if (ctx->aux_stream && ctx->aux_pending) {
    ORIGIN_HIP(hipStreamWaitEvent(ctx->stream, ctx->aux_join, 0));
    ctx->aux_pending = false;
  }
  return ORIGIN_OK;
}

int origin_device_name(origin_ctx *ctx, char *buf, int buflen) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(buf && buflen > 0, "bad buffer");
  hipDeviceProp_t prop;
  ORIGIN_HIP(hipGetDeviceProperties(&prop, ctx->device));
  snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return ORIGIN_OK;
}

int origin_mem_info(origin_ctx *ctx, size_t *free_bytes, size_t *total_bytes) {
  ORIGIN_USE(ctx);
  size_t f = 0, t = 0;
  ORIGIN_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f + alloc_cache_spare(ctx);  // (spare blocks go back when memory runs out)
  if (total_bytes) *total_bytes = t;
  return ORIGIN_OK;
}

int origin_stream(origin_ctx *ctx, void **stream) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(stream, "stream is null");
  *stream = (void *)ctx->stream;
  return ORIGIN_OK;
}

// Blocks of at least 1 MiB that origin_free releases are kept (up to ORIGIN_ALLOC_CACHE_GB, default
// 96, 0 = off) and handed to the next origin_malloc that asks for their size or up to an eighth
// less.  The Step seam allocates every output of every step afresh (the reference's steps return
// new arrays): at 3681 x 600 x 600 that was 27 allocations of up to 5.3 GB at ~40 ms each -- 1.2 s
// of a 3.2 s pass (tools/e2e_profile.py) -- and a device-wide synchronisation per release.  Reuse
// is ordered by the context's stream: origin_free makes it wait for the auxiliary and side streams,
// and every kernel of the library runs on one of the three.
struct AllocCache {
  std::mutex mu;
  std::unordered_map<void *, size_t> live;     // blocks handed out (>= ALLOC_MIN): their sizes
  std::multimap<size_t, void *> spare;         // released blocks by size
  size_t spare_bytes = 0, cap_bytes = 0;
};
constexpr size_t ALLOC_MIN = (size_t)1 << 20;

static void alloc_cache_make(origin_ctx *ctx) {  // (origin_ctx_create)
  auto *c = new AllocCache();
  const char *e = getenv("ORIGIN_ALLOC_CACHE_GB");
  c->cap_bytes = (size_t)((e ? atof(e) : 96.0) * 1e9);
  ctx->alloc_cache = c;
}
static AllocCache *alloc_cache(origin_ctx *ctx) { return (AllocCache *)ctx->alloc_cache; }
static size_t alloc_cache_spare(origin_ctx *ctx) {
  auto *c = alloc_cache(ctx);
  if (!c) return 0;
  std::lock_guard<std::mutex> lk(c->mu);
  return c->spare_bytes;
}

static void alloc_cache_release(origin_ctx *ctx, bool destroy) {
  auto *c = (AllocCache *)ctx->alloc_cache;
  if (!c) return;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->spare.empty()) (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : c->spare) (void)hipFree(kv.second);
    c->spare.clear();
    c->spare_bytes = 0;
  }
  if (destroy) {
    delete c;
    ctx->alloc_cache = nullptr;
  }
}

int origin_malloc(origin_ctx *ctx, size_t bytes, void **d_ptr) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_ptr, "d_ptr is null");
  *d_ptr = nullptr;
  if (bytes == 0) bytes = 16;
  AllocCache *c = alloc_cache(ctx);
  if (bytes >= ALLOC_MIN && c->cap_bytes > 0) {
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->spare.lower_bound(bytes);
    if (it != c->spare.end() && it->first - bytes <= bytes / 8) {
      *d_ptr = it->second;
      c->live[it->second] = it->first;
      c->spare_bytes -= it->first;
      c->spare.erase(it);
      return ORIGIN_OK;
    }
  }
  hipError_t e = hipMalloc(d_ptr, bytes);
  if (e == hipErrorOutOfMemory && c->spare_bytes > 0) {  // give the spare blocks back and try again
    (void)hipGetLastError();
    alloc_cache_release(ctx, false);
    e = hipMalloc(d_ptr, bytes);
  }
  if (e != hipSuccess) {
    origin_set_error("hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? ORIGIN_E_NOMEM : ORIGIN_E_HIP;
  }
  if (bytes >= ALLOC_MIN && c->cap_bytes > 0) {
    std::lock_guard<std::mutex> lk(c->mu);
    c->live[*d_ptr] = bytes;
  }
  return ORIGIN_OK;
}

int origin_free(origin_ctx *ctx, void *d_ptr) {
  ORIGIN_USE(ctx);
  if (!d_ptr) return ORIGIN_OK;
  AllocCache *c = alloc_cache(ctx);
  {
    std::unique_lock<std::mutex> lk(c->mu);
    auto it = c->live.find(d_ptr);
    if (it != c->live.end()) {
      const size_t sz = it->second;
      c->live.erase(it);
      if (c->spare_bytes + sz <= c->cap_bytes) {
        // whoever gets the block next uses it behind everything enqueued so far, on any stream
        lk.unlock();
        if (ctx->aux_stream && ctx->aux_pending) {
          ORIGIN_HIP(hipStreamWaitEvent(ctx->stream, ctx->aux_join, 0));
          ctx->aux_pending = false;
        }
        if (ctx->side_stream && ctx->side_pending) {
          int rj = origin_side_join(ctx);
          if (rj) return rj;
        }
        lk.lock();
        c->spare.emplace(sz, d_ptr);
        c->spare_bytes += sz;
        return ORIGIN_OK;
      }
    }
  }
  ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
  ORIGIN_HIP(hipFree(d_ptr));
  return ORIGIN_OK;
}

int origin_memset(origin_ctx *ctx, void *d_ptr, int byte, size_t bytes) {
  ORIGIN_USE(ctx);
  ORIGIN_HIP(hipMemsetAsync(d_ptr, byte, bytes, ctx->stream));
  return ORIGIN_OK;
}

static int cvt_staging(origin_ctx *ctx);
constexpr size_t STAGED_MIN = (size_t)32 << 20;   // copies of at least this go through pinned staging
constexpr size_t STAGED_CH = (size_t)64 << 20;    // bytes per staging buffer (= CVT_CH floats)

// Large pageable copies, both ways, in 64 MiB chunks through the context's two pinned staging
// buffers, with the host side of every chunk -- a memcpy between the caller's pages and the
// staging buffer -- spread over the host worker pool while the other buffer is on the bus.  A
// pageable hipMemcpy does that memcpy on one runtime thread: 21-30 GB/s, and 22 GB/s into a fresh
// destination whose pages fault one by one (tools/pagefault_probe.py); the float64 hand-over, which
// works this way, reaches 45 GB/s into fresh pages.
static int staged_h2d(origin_ctx *ctx, char *d_dst, const char *h_src, size_t bytes) {
  int rcs = cvt_staging(ctx);
  if (rcs) return rcs;
  char *const stage[2] = {(char *)ctx->cvt_stage[0], (char *)ctx->cvt_stage[1]};
  hipEvent_t *ev = ctx->cvt_ev;
  const size_t nch = (bytes + STAGED_CH - 1) / STAGED_CH;
  for (size_t c = 0; c < nch; ++c) {
    const size_t o = c * STAGED_CH, m = std::min(STAGED_CH, bytes - o);
    char *dst = stage[c & 1];
    ORIGIN_HIP(hipEventSynchronize(ev[c & 1]));  // the copy that last read this buffer is done
    constexpr size_t PIECE = (size_t)1 << 20;
    const int np = (int)((m + PIECE - 1) / PIECE);
    origin_host_pool_run(np, [&](int p) {
      const size_t a = (size_t)p * PIECE, b = std::min(m, a + PIECE);
      memcpy(dst + a, h_src + o + a, b - a);
    });
    ORIGIN_HIP(hipMemcpyAsync(d_dst + o, dst, m, hipMemcpyHostToDevice, ctx->stream));
    ORIGIN_HIP(hipEventRecord(ev[c & 1], ctx->stream));
  }
  ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
  return ORIGIN_OK;
}

int origin_h2d(origin_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
  ORIGIN_USE(ctx);
  if (bytes == 0) return ORIGIN_OK;
  if (bytes >= STAGED_MIN) return staged_h2d(ctx, (char *)d_dst, (const char *)h_src, bytes);
  ORIGIN_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
  return ORIGIN_OK;
}

// Reads of device arrays by the host wait for pending work of the auxiliary stream too (cont_dct /
// ima_dct of origin_dct_cont_std_async): a caller that forgot origin_aux_join would otherwise get
// a half-written array without any error.
static int aux_before_read(origin_ctx *ctx) {
  if (ctx->aux_stream && ctx->aux_pending) {
    ORIGIN_HIP(hipStreamWaitEvent(ctx->stream, ctx->aux_join, 0));
    ctx->aux_pending = false;
  }
  return ORIGIN_OK;
}

// a large destination is usually a fresh np.empty: every 4 KiB page of it faults on its first
// write (23 against 56 GB/s for a pageable copy of 1.3 GB, tools/pagefault_probe.py).  Ask for
// transparent huge pages on the part that covers whole 2 MiB pages -- a hint, ignored where the
// system does not offer them
static void hint_huge_pages(void *dst, size_t bytes) {
  if (bytes < ((size_t)8 << 20)) return;
  const uintptr_t m = ((uintptr_t)2 << 20) - 1;
  const uintptr_t a = ((uintptr_t)dst + m) & ~m, b = ((uintptr_t)dst + bytes) & ~m;
  if (b > a) (void)madvise((void *)a, (size_t)(b - a), MADV_HUGEPAGE);
}

int origin_d2h(origin_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
  ORIGIN_USE(ctx);
  if (bytes == 0) return ORIGIN_OK;
  { int rca = aux_before_read(ctx); if (rca) return rca; }
  hint_huge_pages(h_dst, bytes);
  if (bytes >= STAGED_MIN) {
    int rcs = cvt_staging(ctx);
    if (rcs) return rcs;
    char *const stage[2] = {(char *)ctx->cvt_stage[0], (char *)ctx->cvt_stage[1]};
    hipEvent_t *ev = ctx->cvt_ev;
    ORIGIN_HIP(hipEventSynchronize(ev[0]));  // (an upload may still be reading the buffers)
    ORIGIN_HIP(hipEventSynchronize(ev[1]));
    const size_t nch = (bytes + STAGED_CH - 1) / STAGED_CH;
    auto issue = [&](size_t c) -> int {
      const size_t o = c * STAGED_CH, m = std::min(STAGED_CH, bytes - o);
      ORIGIN_HIP(hipMemcpyAsync(stage[c & 1], (const char *)d_src + o, m, hipMemcpyDeviceToHost,
                                ctx->stream));
      ORIGIN_HIP(hipEventRecord(ev[c & 1], ctx->stream));
      return ORIGIN_OK;
    };
    int rc = issue(0);
    if (rc) return rc;
    for (size_t c = 0; c < nch; ++c) {
      ORIGIN_HIP(hipEventSynchronize(ev[c & 1]));
      if (c + 1 < nch && (rc = issue(c + 1))) return rc;
      const size_t o = c * STAGED_CH, m = std::min(STAGED_CH, bytes - o);
      const char *src = stage[c & 1];
      char *dst = (char *)h_dst + o;
      constexpr size_t PIECE = (size_t)1 << 20;
      const int np = (int)((m + PIECE - 1) / PIECE);
      origin_host_pool_run(np, [&](int p) {
        const size_t a = (size_t)p * PIECE, b = std::min(m, a + PIECE);
        memcpy(dst + a, src + a, b - a);
      });
    }
    return ORIGIN_OK;
  }
  ORIGIN_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
  return ORIGIN_OK;
}

// Staging of the two conversions below: owned by the context (its device, its stream), marked
// ready only when both buffers and both events exist; a partial failure is rolled back.
constexpr size_t CVT_CH = (size_t)16 << 20;  // elements per chunk: 64 MiB of float32
static int cvt_staging(origin_ctx *ctx) {
  if (ctx->cvt_ready) return ORIGIN_OK;
  float *st[2] = {nullptr, nullptr};
  hipEvent_t ev[2];
  int nev = 0;
  hipError_t e = hipSuccess;
  for (int b = 0; b < 2 && e == hipSuccess; ++b)
    e = hipHostMalloc((void **)&st[b], CVT_CH * sizeof(float), hipHostMallocDefault);
  for (int b = 0; b < 2 && e == hipSuccess; ++b) {
    e = hipEventCreateWithFlags(&ev[b], hipEventDisableTiming);
    if (e == hipSuccess) ++nev;
  }
  for (int b = 0; b < 2 && e == hipSuccess; ++b) e = hipEventRecord(ev[b], ctx->stream);
  if (e != hipSuccess) {
    for (int b = 0; b < nev; ++b) (void)hipEventDestroy(ev[b]);
    for (int b = 0; b < 2; ++b)
      if (st[b]) (void)hipHostFree(st[b]);
    origin_set_error("conversion staging: %s", hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? ORIGIN_E_NOMEM : ORIGIN_E_HIP;
  }
  for (int b = 0; b < 2; ++b) ctx->cvt_stage[b] = st[b], ctx->cvt_ev[b] = ev[b];
  ctx->cvt_ready = true;
  return ORIGIN_OK;
}

// Device float32 -> host float64 (the reference's arrays are float64: every cube that leaves
// through the function seam or a LazyCube is widened).  Chunks of the device array land in two
// pinned staging buffers by turns; while chunk i + 1 is in flight the host worker pool widens chunk
// i into the destination -- instead of a pageable copy followed by a single-threaded astype.
int origin_d2h_f32_as_f64(origin_ctx *ctx, double *h_dst, const float *d_src, size_t n) {
  ORIGIN_USE(ctx);
  if (n == 0) return ORIGIN_OK;
  ORIGIN_CHECK_ARG(h_dst && d_src, "null pointer");
  constexpr size_t CH = CVT_CH;
  int rcs = cvt_staging(ctx);
  if (rcs) return rcs;
  if ((rcs = aux_before_read(ctx))) return rcs;
  float *const *stage = ctx->cvt_stage;
  hipEvent_t *ev = ctx->cvt_ev;
  // (a narrowing upload may still be reading the buffers)
  ORIGIN_HIP(hipEventSynchronize(ev[0]));
  ORIGIN_HIP(hipEventSynchronize(ev[1]));
  const size_t nch = (n + CH - 1) / CH;
  hint_huge_pages(h_dst, n * sizeof(double));
  auto issue = [&](size_t c) -> int {
    const size_t o = c * CH, m = std::min(CH, n - o);
    ORIGIN_HIP(hipMemcpyAsync(stage[c & 1], d_src + o, m * sizeof(float), hipMemcpyDeviceToHost,
                              ctx->stream));
    ORIGIN_HIP(hipEventRecord(ev[c & 1], ctx->stream));
    return ORIGIN_OK;
  };
  int rc = issue(0);
  if (rc) return rc;
  for (size_t c = 0; c < nch; ++c) {
    ORIGIN_HIP(hipEventSynchronize(ev[c & 1]));
    if (c + 1 < nch && (rc = issue(c + 1))) return rc;
    const size_t o = c * CH, m = std::min(CH, n - o);
    const float *src = stage[c & 1];
    double *dst = h_dst + o;
    constexpr size_t PIECE = (size_t)1 << 18;  // 1 MiB of float32 per task
    const int np = (int)((m + PIECE - 1) / PIECE);
    origin_host_pool_run(np, [&](int p) {
      const size_t a = (size_t)p * PIECE, b = std::min(m, a + PIECE);
      for (size_t i = a; i < b; ++i) dst[i] = (double)src[i];
    });
  }
  return ORIGIN_OK;
}

// Host float64 -> device float32 (the reference hands float64 cubes to the function seam): the
// host pool narrows chunk i + 1 into a pinned staging buffer while chunk i is on its way.
int origin_h2d_f64_as_f32(origin_ctx *ctx, float *d_dst, const double *h_src, size_t n) {
  ORIGIN_USE(ctx);
  if (n == 0) return ORIGIN_OK;
  ORIGIN_CHECK_ARG(d_dst && h_src, "null pointer");
  constexpr size_t CH = CVT_CH;
  int rcs = cvt_staging(ctx);
  if (rcs) return rcs;
  float *const *stage = ctx->cvt_stage;
  hipEvent_t *ev = ctx->cvt_ev;
  const size_t nch = (n + CH - 1) / CH;
  for (size_t c = 0; c < nch; ++c) {
    const size_t o = c * CH, m = std::min(CH, n - o);
    float *dst = stage[c & 1];
    const double *src = h_src + o;
    ORIGIN_HIP(hipEventSynchronize(ev[c & 1]));  // the copy that last read this buffer is done
    constexpr size_t PIECE = (size_t)1 << 18;
    const int np = (int)((m + PIECE - 1) / PIECE);
    origin_host_pool_run(np, [&](int p) {
      const size_t a = (size_t)p * PIECE, b = std::min(m, a + PIECE);
      for (size_t i = a; i < b; ++i) dst[i] = (float)src[i];
    });
    ORIGIN_HIP(hipMemcpyAsync(d_dst + o, dst, m * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    ORIGIN_HIP(hipEventRecord(ev[c & 1], ctx->stream));
  }
  ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
  return ORIGIN_OK;
}

int origin_d2d(origin_ctx *ctx, void *d_dst, const void *d_src, size_t bytes) {
  ORIGIN_USE(ctx);
  if (bytes == 0) return ORIGIN_OK;
  ORIGIN_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return ORIGIN_OK;
}

int origin_copy_box(origin_ctx *ctx, int kind, void *dst, long dst_pitch_y, long dst_pitch_z,
                    const void *src, long src_pitch_y, long src_pitch_z, int nz, int ny,
                    int nx, int elem) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(kind >= 0 && kind <= 2, "kind must be 0 (h2d), 1 (d2h) or 2 (d2d)");
  ORIGIN_CHECK_ARG(elem == 1 || elem == 2 || elem == 4 || elem == 8, "elem must be 1,2,4,8");
  ORIGIN_CHECK_ARG(nz >= 0 && ny >= 0 && nx >= 0, "negative extent");
  ORIGIN_CHECK_ARG(dst_pitch_y >= nx && src_pitch_y >= nx, "row pitch smaller than nx");
  ORIGIN_CHECK_ARG(dst_pitch_z >= dst_pitch_y * (long)(ny > 0 ? 1 : 0) &&
                       src_pitch_z >= src_pitch_y * (long)(ny > 0 ? 1 : 0),
                   "plane pitch smaller than a row");
  if (nz == 0 || ny == 0 || nx == 0) return ORIGIN_OK;
  if (kind == 1) {
    int rca = aux_before_read(ctx);
    if (rca) return rca;
  }
  hipMemcpyKind k = kind == 0   ? hipMemcpyHostToDevice
                    : kind == 1 ? hipMemcpyDeviceToHost
                                : hipMemcpyDeviceToDevice;
  if (kind == 2 && nz <= 65535 && ny <= 65535) {
    const long rowbytes = (long)nx * elem;
    const long dpy = dst_pitch_y * elem, dpz = dst_pitch_z * elem;
    const long spy = src_pitch_y * elem, spz = src_pitch_z * elem;
    auto aligned = [&](long a) {
      return ((uintptr_t)dst % a) == 0 && ((uintptr_t)src % a) == 0 && rowbytes % a == 0 &&
             dpy % a == 0 && dpz % a == 0 && spy % a == 0 && spz % a == 0;
    };
    const int vec = aligned(16) ? 16 : aligned(4) ? 4 : 1;
    dim3 grid((unsigned)((rowbytes / vec + 255) / 256), ny, nz);
    hipLaunchKernelGGL(copy_box_kernel, grid, dim3(256), 0, ctx->stream, (char *)dst, dpy, dpz,
                       (const char *)src, spy, spz, ny, rowbytes, vec);
    ORIGIN_LAUNCH_CHECK();
    return ORIGIN_OK;
  }
  // one 2-D copy per plane: rows of nx*elem bytes at the given pitches
  for (int z = 0; z < nz; ++z) {
    const char *s = (const char *)src + (size_t)z * src_pitch_z * elem;
    char *d = (char *)dst + (size_t)z * dst_pitch_z * elem;
    ORIGIN_HIP(hipMemcpy2DAsync(d, (size_t)dst_pitch_y * elem, s, (size_t)src_pitch_y * elem,
                                (size_t)nx * elem, ny, k, ctx->stream));
  }
  if (kind != 2) ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
  return ORIGIN_OK;
}

int origin_timer_start(origin_ctx *ctx, int slot) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(slot >= 0 && slot < 64, "timer slot out of range");
  if (!ctx->ev_made[slot]) {
    ORIGIN_HIP(hipEventCreate(&ctx->ev_start[slot]));
    ORIGIN_HIP(hipEventCreate(&ctx->ev_stop[slot]));
    ctx->ev_made[slot] = true;
  }
  ORIGIN_HIP(hipEventRecord(ctx->ev_start[slot], ctx->stream));
  return ORIGIN_OK;
}

int origin_timer_stop(origin_ctx *ctx, int slot) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(slot >= 0 && slot < 64 && ctx->ev_made[slot], "timer slot not started");
  ORIGIN_HIP(hipEventRecord(ctx->ev_stop[slot], ctx->stream));
  return ORIGIN_OK;
}

int origin_timer_ms(origin_ctx *ctx, int slot, float *ms) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(slot >= 0 && slot < 64 && ctx->ev_made[slot] && ms, "timer slot not started");
  ORIGIN_HIP(hipEventSynchronize(ctx->ev_stop[slot]));
  ORIGIN_HIP(hipEventElapsedTime(ms, ctx->ev_start[slot], ctx->ev_stop[slot]));
  return ORIGIN_OK;
}

}  // extern "C"
