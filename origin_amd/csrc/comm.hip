// Inter-GPU exchange of the tiled hot path: RCCL called directly on the
// context's stream.
//
// The reference has no multi-GPU (or multi-node) path at all -- its only parallelism is a
// joblib process pool over profiles (lib_origin.py:1150-1160) -- so there is no call pattern
// to follow.  A tiled field needs exactly two exchanges (origin_amd/multigpu.py): the sum of
// 2*Nz float64 over all tiles for the per-channel mean (steps.py:442) and the halo strips of
// cube_faint before the GLR.  Strips go GPU to GPU over xGMI with grouped ncclSend/ncclRecv.
//
// librccl.so (573 MB) is opened lazily with dlopen on the first origin_comm_* call, so the
// one-GPU path neither loads nor needs it.  The unique id travels between processes by
// whatever the host has (origin_amd/multigpu.py broadcasts it over a gloo group).
#include <dlfcn.h>

#include <mutex>
#include <string>

#include "common.h"

namespace {

// the subset of rccl.h this file uses (opaque handle, id blob, enums by value)
typedef struct ncclComm *ncclComm_t;
typedef struct {
  char internal[ORIGIN_COMM_ID_BYTES];
} ncclUniqueId;
typedef int ncclResult_t;
enum { kNcclSuccess = 0 };
enum { kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclUint8 = 1 };  // ncclDataType_t
enum { kNcclSum = 0 };                                        // ncclRedOp_t

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t,
                            hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  const char *(*GetLastError)(ncclComm_t) = nullptr;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
char g_rccl_err[256] = "";

template <typename F>
bool sym(void *h, const char *name, F &out) {
  out = (F)dlsym(h, name);
  if (!out) snprintf(g_rccl_err, sizeof g_rccl_err, "librccl.so has no symbol %s", name);
  return out != nullptr;
}

// RCCL must drive the SAME HIP runtime as this library.  A process that has PyTorch loaded
// holds the wheel's bundled libamdhip64/librccl next to /opt/rocm's: whichever libamdhip64 the
// dynamic loader bound this library to (the first one loaded), its sibling librccl is the one
// built against it -- so look next to the runtime our own HIP calls resolve to first.
void load_rccl() {
  std::string sibling1, sibling2;
  Dl_info di;
  if (dladdr((void *)&hipStreamSynchronize, &di) && di.dli_fname) {
    std::string dir(di.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos) {
      dir.resize(slash + 1);
      sibling1 = dir + "librccl.so.1";
      sibling2 = dir + "librccl.so";
    }
  }
  const char *names[] = {getenv("ORIGIN_RCCL_LIB"), sibling1.c_str(), sibling2.c_str(),
                         "librccl.so.1", "librccl.so"};
  void *h = nullptr;
  for (const char *n : names) {
    if (!n || !*n) continue;
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) {
    snprintf(g_rccl_err, sizeof g_rccl_err, "cannot open librccl.so: %s", dlerror());
    return;
  }
  Rccl r;
  r.handle = h;
  bool ok = sym(h, "ncclGetUniqueId", r.GetUniqueId) && sym(h, "ncclCommInitRank", r.CommInitRank) &&
            sym(h, "ncclCommDestroy", r.CommDestroy) && sym(h, "ncclCommAbort", r.CommAbort) &&
            sym(h, "ncclGroupStart", r.GroupStart) && sym(h, "ncclGroupEnd", r.GroupEnd) &&
            sym(h, "ncclSend", r.Send) && sym(h, "ncclRecv", r.Recv) &&
            sym(h, "ncclAllReduce", r.AllReduce) && sym(h, "ncclGetErrorString", r.GetErrorString);
  if (!ok) return;
  r.GetLastError = (const char *(*)(ncclComm_t))dlsym(h, "ncclGetLastError");  // optional
  g_rccl = r;
}

int need_rccl() {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.handle) {
    origin_set_error("RCCL unavailable: %s", g_rccl_err);
    return ORIGIN_E_STATE;
  }
  return ORIGIN_OK;
}

}  // namespace

struct origin_comm {
  origin_ctx *ctx;
  ncclComm_t comm;
  int rank, world;
};

#define ORIGIN_RCCL(comm_, call)                                                             \
  do {                                                                                       \
    ncclResult_t r_ = (call);                                                                \
    if (r_ != kNcclSuccess) {                                                                \
      const char *last_ = g_rccl.GetLastError ? g_rccl.GetLastError(comm_) : "";             \
      origin_set_error("%s failed: %s %s (%s:%d)", #call, g_rccl.GetErrorString(r_),          \
                       last_ ? last_ : "", __FILE__, __LINE__);                              \
      return ORIGIN_E_HIP;                                                                   \
    }                                                                                        \
  } while (0)

extern "C" {

int origin_comm_unique_id(char *id) {
  ORIGIN_CHECK_ARG(id, "null id buffer");
  int rc = need_rccl();
  if (rc) return rc;
  ncclUniqueId u;
  ORIGIN_RCCL(nullptr, g_rccl.GetUniqueId(&u));
  memcpy(id, u.internal, ORIGIN_COMM_ID_BYTES);
  return ORIGIN_OK;
}

int origin_comm_create(origin_ctx *ctx, const char *id, int rank, int world, origin_comm **out) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(id && out && world >= 1 && rank >= 0 && rank < world, "bad arguments");
  int rc = need_rccl();
  if (rc) return rc;
  ncclUniqueId u;
  memcpy(u.internal, id, ORIGIN_COMM_ID_BYTES);
  ncclComm_t c = nullptr;
  ORIGIN_RCCL(nullptr, g_rccl.CommInitRank(&c, world, u, rank));
  *out = new origin_comm{ctx, c, rank, world};
  return ORIGIN_OK;
}

int origin_comm_destroy(origin_comm *comm) {
  if (!comm) return ORIGIN_OK;
  if (comm->comm && g_rccl.handle) {
    (void)hipStreamSynchronize(comm->ctx->stream);
    (void)g_rccl.CommDestroy(comm->comm);
  }
  delete comm;
  return ORIGIN_OK;
}

int origin_comm_allreduce_f64(origin_comm *comm, double *d_buf, long n) {
  ORIGIN_CHECK_ARG(comm && d_buf && n > 0, "bad arguments");
  ORIGIN_USE(comm->ctx);
  ORIGIN_RCCL(comm->comm, g_rccl.AllReduce(d_buf, d_buf, (size_t)n, kNcclFloat64, kNcclSum,
                                           comm->comm, comm->ctx->stream));
  return ORIGIN_OK;
}

int origin_comm_exchange(origin_comm *comm, int nsend, const int *send_peer,
                         const void *const *d_send, const long *send_bytes, int nrecv,
                         const int *recv_peer, void *const *d_recv, const long *recv_bytes) {
  ORIGIN_CHECK_ARG(comm && nsend >= 0 && nrecv >= 0, "bad arguments");
  ORIGIN_CHECK_ARG(nsend == 0 || (send_peer && d_send && send_bytes), "null send lists");
  ORIGIN_CHECK_ARG(nrecv == 0 || (recv_peer && d_recv && recv_bytes), "null receive lists");
  ORIGIN_USE(comm->ctx);
  for (int i = 0; i < nsend; ++i)
    ORIGIN_CHECK_ARG(send_peer[i] >= 0 && send_peer[i] < comm->world && d_send[i] &&
                         send_bytes[i] > 0,
                     "bad send entry %d", i);
  for (int i = 0; i < nrecv; ++i)
    ORIGIN_CHECK_ARG(recv_peer[i] >= 0 && recv_peer[i] < comm->world && d_recv[i] &&
                         recv_bytes[i] > 0,
                     "bad receive entry %d", i);
  if (nsend + nrecv == 0) return ORIGIN_OK;
  hipStream_t st = comm->ctx->stream;
  // one group: every rank posts all its receives and sends together, so no ordering between
  // neighbours can deadlock
  ORIGIN_RCCL(comm->comm, g_rccl.GroupStart());
  ncclResult_t bad = kNcclSuccess;
  for (int i = 0; i < nrecv && bad == kNcclSuccess; ++i)
    bad = g_rccl.Recv(d_recv[i], (size_t)recv_bytes[i], kNcclUint8, recv_peer[i], comm->comm, st);
  for (int i = 0; i < nsend && bad == kNcclSuccess; ++i)
    bad = g_rccl.Send(d_send[i], (size_t)send_bytes[i], kNcclUint8, send_peer[i], comm->comm, st);
  ncclResult_t end = g_rccl.GroupEnd();
  ORIGIN_RCCL(comm->comm, bad);
  ORIGIN_RCCL(comm->comm, end);
  return ORIGIN_OK;
}

}  // extern "C"
