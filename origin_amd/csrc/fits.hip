// FITS data-unit codec on the device.
//
// Step.dump / Step.load of the reference (muse_origin/steps.py:301-352) write every cube and
// image of a step through mpdaf's Cube.write(convert_float32=False) -- float64 FITS image
// extensions -- and reload them lazily (DataObj.__get__, steps.py:131-160).  A FITS data unit
// is the array in big-endian byte order (IEEE-754 for BITPIX < 0); astropy does the widening
// and the byte swap on the host, one more pass over every cube.  Here the step outputs already
// sit in HBM as float32 / uint8 / int32, so the conversion (widen to the file type, swap) is a
// streaming kernel and the host only moves file bytes.
//
// Pure byte work, HBM bound: 4 + 8 bytes per voxel for float32 -> BITPIX -64.
#include <errno.h>
#include <unistd.h>

#include "common.h"

namespace {

enum { T_F32 = 0, T_U8 = 1, T_I32 = 2, T_F64 = 3 };

__device__ __forceinline__ unsigned bswap32(unsigned v) { return __builtin_bswap32(v); }
__device__ __forceinline__ unsigned long long bswap64(unsigned long long v) {
  return __builtin_bswap64(v);
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static __device__ double as_f64(float v) { return (double)v; }
  static __device__ float as_f32(float v) { return v; }
  static __device__ long long as_i64(float v) { return (long long)v; }
};
template <>
struct Elem<double> {
  static __device__ double as_f64(double v) { return v; }
  static __device__ float as_f32(double v) { return (float)v; }
  static __device__ long long as_i64(double v) { return (long long)v; }
};
template <>
struct Elem<int> {
  static __device__ double as_f64(int v) { return (double)v; }
  static __device__ float as_f32(int v) { return (float)v; }
  static __device__ long long as_i64(int v) { return v; }
};
template <>
struct Elem<uint8_t> {
  static __device__ double as_f64(uint8_t v) { return (double)v; }
  static __device__ float as_f32(uint8_t v) { return (float)v; }
  static __device__ long long as_i64(uint8_t v) { return v; }
};

// one element of the file, big-endian, at byte offset i * |BITPIX| / 8
template <int BITPIX, typename S>
__device__ __forceinline__ void put(uint8_t *dst, long i, S v) {
  if (BITPIX == -64) {
    ((unsigned long long *)dst)[i] =
        bswap64((unsigned long long)__double_as_longlong(Elem<S>::as_f64(v)));
  } else if (BITPIX == -32) {
    ((unsigned *)dst)[i] = bswap32(__float_as_uint(Elem<S>::as_f32(v)));
  } else if (BITPIX == 64) {
    ((unsigned long long *)dst)[i] = bswap64((unsigned long long)Elem<S>::as_i64(v));
  } else if (BITPIX == 32) {
    ((unsigned *)dst)[i] = bswap32((unsigned)(int)Elem<S>::as_i64(v));
  } else if (BITPIX == 16) {
    const unsigned short h = (unsigned short)(short)Elem<S>::as_i64(v);
    ((unsigned short *)dst)[i] = (unsigned short)((h >> 8) | (h << 8));
  } else {
    dst[i] = (uint8_t)Elem<S>::as_i64(v);
  }
}

// the file value at element i, as float64 / int64 carrier
template <int BITPIX>
__device__ __forceinline__ double get_f(const uint8_t *src, long i) {
  if (BITPIX == -64)
    return __longlong_as_double((long long)bswap64(((const unsigned long long *)src)[i]));
  return (double)__uint_as_float(bswap32(((const unsigned *)src)[i]));
}
template <int BITPIX>
__device__ __forceinline__ long long get_i(const uint8_t *src, long i) {
  if (BITPIX == 64) return (long long)bswap64(((const unsigned long long *)src)[i]);
  if (BITPIX == 32) return (int)bswap32(((const unsigned *)src)[i]);
  if (BITPIX == 16) {
    const unsigned short h = ((const unsigned short *)src)[i];
    return (short)((h >> 8) | (h << 8));
  }
  return src[i];
}

// four consecutive elements per thread: 16-byte loads of float32 / int32 sources and 16-byte
// stores (two for 8-byte file types); consecutive lanes touch consecutive 16-byte slots
template <int BITPIX, typename S>
__global__ __launch_bounds__(256) void fits_encode_kernel(const S *__restrict__ src, long n,
                                                          uint8_t *__restrict__ dst) {
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      S v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = src[i + e];
#pragma unroll
      for (int e = 0; e < 4; ++e) put<BITPIX, S>(dst, i + e, v[e]);
    } else {
      for (long e = i; e < n; ++e) put<BITPIX, S>(dst, e, src[e]);
    }
  }
}

template <int BITPIX, typename D>
__global__ __launch_bounds__(256) void fits_decode_kernel(const uint8_t *__restrict__ src, long n,
                                                          D *__restrict__ dst) {
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      D v[4];  // all four file elements are read before the first store
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (BITPIX < 0) v[e] = (D)get_f<BITPIX>(src, i + e);
        else v[e] = (D)get_i<BITPIX>(src, i + e);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[i + e] = v[e];
    } else {
      for (long e = i; e < n; ++e) {
        if (BITPIX < 0) dst[e] = (D)get_f<BITPIX>(src, e);
        else dst[e] = (D)get_i<BITPIX>(src, e);
      }
    }
  }
}

template <typename S>
int encode_as(origin_ctx *ctx, const void *d_src, long n, int bitpix, void *d_dst) {
  const int grid = (int)std::min<long>((n + 1023) / 1024, (long)ctx->num_cu * 16);
  const S *s = (const S *)d_src;
  uint8_t *d = (uint8_t *)d_dst;
  hipStream_t st = ctx->stream;
  switch (bitpix) {
    case -64: hipLaunchKernelGGL((fits_encode_kernel<-64, S>), dim3(grid), dim3(256), 0, st, s, n, d); break;
    case -32: hipLaunchKernelGGL((fits_encode_kernel<-32, S>), dim3(grid), dim3(256), 0, st, s, n, d); break;
    case 64: hipLaunchKernelGGL((fits_encode_kernel<64, S>), dim3(grid), dim3(256), 0, st, s, n, d); break;
    case 32: hipLaunchKernelGGL((fits_encode_kernel<32, S>), dim3(grid), dim3(256), 0, st, s, n, d); break;
    case 16: hipLaunchKernelGGL((fits_encode_kernel<16, S>), dim3(grid), dim3(256), 0, st, s, n, d); break;
    case 8: hipLaunchKernelGGL((fits_encode_kernel<8, S>), dim3(grid), dim3(256), 0, st, s, n, d); break;
    default: origin_set_error("BITPIX %d is not a FITS image type", bitpix); return ORIGIN_E_ARG;
  }
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

template <typename D>
int decode_as(origin_ctx *ctx, const void *d_src, int bitpix, long n, void *d_dst) {
  const int grid = (int)std::min<long>((n + 1023) / 1024, (long)ctx->num_cu * 16);
  const uint8_t *s = (const uint8_t *)d_src;
  D *d = (D *)d_dst;
  hipStream_t st = ctx->stream;
  switch (bitpix) {
    case -64: hipLaunchKernelGGL((fits_decode_kernel<-64, D>), dim3(grid), dim3(256), 0, st, s, n, d); break;
    case -32: hipLaunchKernelGGL((fits_decode_kernel<-32, D>), dim3(grid), dim3(256), 0, st, s, n, d); break;
    case 64: hipLaunchKernelGGL((fits_decode_kernel<64, D>), dim3(grid), dim3(256), 0, st, s, n, d); break;
    case 32: hipLaunchKernelGGL((fits_decode_kernel<32, D>), dim3(grid), dim3(256), 0, st, s, n, d); break;
    case 16: hipLaunchKernelGGL((fits_decode_kernel<16, D>), dim3(grid), dim3(256), 0, st, s, n, d); break;
    case 8: hipLaunchKernelGGL((fits_decode_kernel<8, D>), dim3(grid), dim3(256), 0, st, s, n, d); break;
    default: origin_set_error("BITPIX %d is not a FITS image type", bitpix); return ORIGIN_E_ARG;
  }
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

}  // namespace

extern "C" {

int origin_fits_encode(origin_ctx *ctx, const void *d_src, int src_type, long n, int bitpix,
                       void *d_dst) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_src && d_dst && n > 0, "bad arguments");
  ORIGIN_CHECK_ARG(((uintptr_t)d_dst & 7) == 0, "destination must be 8-byte aligned");
  switch (src_type) {
    case T_F32: return encode_as<float>(ctx, d_src, n, bitpix, d_dst);
    case T_U8: return encode_as<uint8_t>(ctx, d_src, n, bitpix, d_dst);
    case T_I32: return encode_as<int>(ctx, d_src, n, bitpix, d_dst);
    case T_F64: return encode_as<double>(ctx, d_src, n, bitpix, d_dst);
  }
  origin_set_error("unknown element type %d", src_type);
  return ORIGIN_E_ARG;
}

int origin_fits_decode(origin_ctx *ctx, const void *d_src, int bitpix, long n, int dst_type,
                       void *d_dst) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_src && d_dst && n > 0, "bad arguments");
  ORIGIN_CHECK_ARG(((uintptr_t)d_src & 7) == 0, "source must be 8-byte aligned");
  switch (dst_type) {
    case T_F32: return decode_as<float>(ctx, d_src, bitpix, n, d_dst);
    case T_U8: return decode_as<uint8_t>(ctx, d_src, bitpix, n, d_dst);
    case T_I32: return decode_as<int>(ctx, d_src, bitpix, n, d_dst);
    case T_F64: return decode_as<double>(ctx, d_src, bitpix, n, d_dst);
  }
  origin_set_error("unknown element type %d", dst_type);
  return ORIGIN_E_ARG;
}


// ---- data unit <-> file descriptor -------------------------------------------------------
// The conversion runs chunk by chunk on the stream into one of two pinned host buffers while
// the calling thread is inside write() / read() for the neighbouring chunk, so the file
// system, the PCIe copy and the kernel overlap.  Sequential I/O at the descriptor's offset.
namespace {

constexpr long IO_CHUNK_BYTES = 64l << 20;

struct IoStage {
  void *h[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  ~IoStage() {
    for (int i = 0; i < 2; ++i) {
      if (h[i]) (void)hipHostFree(h[i]);
      if (ev[i]) (void)hipEventDestroy(ev[i]);
    }
  }
};

int write_all(int fd, const char *p, size_t bytes) {
  while (bytes) {
    const ssize_t w = ::write(fd, p, bytes);
    if (w < 0) {
      if (errno == EINTR) continue;
      origin_set_error("write failed: %s", strerror(errno));
      return ORIGIN_E_STATE;
    }
    p += w;
    bytes -= (size_t)w;
  }
  return ORIGIN_OK;
}

int read_all(int fd, char *p, size_t bytes) {
  while (bytes) {
    const ssize_t r = ::read(fd, p, bytes);
    if (r < 0 && errno == EINTR) continue;
    if (r <= 0) {
      origin_set_error(r == 0 ? "unexpected end of file" : "read failed: %s", strerror(errno));
      return ORIGIN_E_STATE;
    }
    p += r;
    bytes -= (size_t)r;
  }
  return ORIGIN_OK;
}

size_t elem_size(int type) { return type == T_U8 ? 1 : type == T_F64 ? 8 : 4; }

}  // namespace

int origin_fits_write_data(origin_ctx *ctx, const void *d_src, int src_type, long n, int bitpix,
                           int fd) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_src && n > 0 && fd >= 0, "bad arguments");
  ORIGIN_CHECK_ARG(src_type >= T_F32 && src_type <= T_F64, "unknown element type %d", src_type);
  const long width = (bitpix < 0 ? -bitpix : bitpix) / 8;
  ORIGIN_CHECK_ARG(width == 1 || width == 2 || width == 4 || width == 8, "bad BITPIX %d", bitpix);
  const long chunk = std::min(n, IO_CHUNK_BYTES / width);
  void *scr = nullptr;
  int rc = origin_scratch(ctx, (size_t)(2 * chunk * width), &scr);
  if (rc) return rc;
  IoStage st;
  for (int i = 0; i < 2; ++i) {
    ORIGIN_HIP(hipHostMalloc(&st.h[i], (size_t)(chunk * width), hipHostMallocDefault));
    ORIGIN_HIP(hipEventCreateWithFlags(&st.ev[i], hipEventDisableTiming));
  }
  const size_t es = elem_size(src_type);
  const long nchunks = (n + chunk - 1) / chunk;
  for (long k = 0; k <= nchunks; ++k) {
    if (k < nchunks) {  // convert + copy chunk k (asynchronous)
      const long e0 = k * chunk, m = std::min(chunk, n - e0);
      char *dst = (char *)scr + (k & 1) * chunk * width;
      rc = origin_fits_encode(ctx, (const char *)d_src + (size_t)e0 * es, src_type, m, bitpix, dst);
      if (rc) return rc;
      ORIGIN_HIP(hipMemcpyAsync(st.h[k & 1], dst, (size_t)(m * width), hipMemcpyDeviceToHost,
                                ctx->stream));
      ORIGIN_HIP(hipEventRecord(st.ev[k & 1], ctx->stream));
    }
    if (k > 0) {  // write chunk k - 1 while chunk k is on its way
      const long j = k - 1, m = std::min(chunk, n - j * chunk);
      ORIGIN_HIP(hipEventSynchronize(st.ev[j & 1]));
      if ((rc = write_all(fd, (const char *)st.h[j & 1], (size_t)(m * width)))) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
      }
    }
  }
  return ORIGIN_OK;
}

int origin_fits_read_data(origin_ctx *ctx, int fd, int bitpix, long n, int dst_type, void *d_dst) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_dst && n > 0 && fd >= 0, "bad arguments");
  ORIGIN_CHECK_ARG(dst_type >= T_F32 && dst_type <= T_F64, "unknown element type %d", dst_type);
  const long width = (bitpix < 0 ? -bitpix : bitpix) / 8;
  ORIGIN_CHECK_ARG(width == 1 || width == 2 || width == 4 || width == 8, "bad BITPIX %d", bitpix);
  const long chunk = std::min(n, IO_CHUNK_BYTES / width);
  void *scr = nullptr;
  int rc = origin_scratch(ctx, (size_t)(2 * chunk * width), &scr);
  if (rc) return rc;
  IoStage st;
  for (int i = 0; i < 2; ++i) {
    ORIGIN_HIP(hipHostMalloc(&st.h[i], (size_t)(chunk * width), hipHostMallocDefault));
    ORIGIN_HIP(hipEventCreateWithFlags(&st.ev[i], hipEventDisableTiming));
  }
  const size_t es = elem_size(dst_type);
  const long nchunks = (n + chunk - 1) / chunk;
  for (long k = 0; k < nchunks; ++k) {
    const long e0 = k * chunk, m = std::min(chunk, n - e0);
    if (k >= 2) ORIGIN_HIP(hipEventSynchronize(st.ev[k & 1]));  // buffer free again
    if ((rc = read_all(fd, (char *)st.h[k & 1], (size_t)(m * width)))) {
      (void)hipStreamSynchronize(ctx->stream);
      return rc;
    }
    char *src = (char *)scr + (k & 1) * chunk * width;
    ORIGIN_HIP(hipMemcpyAsync(src, st.h[k & 1], (size_t)(m * width), hipMemcpyHostToDevice,
                              ctx->stream));
    ORIGIN_HIP(hipEventRecord(st.ev[k & 1], ctx->stream));
    rc = origin_fits_decode(ctx, src, bitpix, m, dst_type, (char *)d_dst + (size_t)e0 * es);
    if (rc) {
      (void)hipStreamSynchronize(ctx->stream);
      return rc;
    }
  }
  ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
  return ORIGIN_OK;
}

}  // extern "C"
