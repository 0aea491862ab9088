// Thresholding of Detection.run  (reference muse_origin/steps.py:956-974 and det_correl_min,
// :935-939; the consumer of the local-maximum cubes of SURVEY.md 8f row 1).
//
//   z, y, x = np.where(cube > threshold)          # C order: z slowest, x fastest
//   T_GLR   = cube[z, y, x];  profile = cube_profile[z, y, x]
//
// The reference scans three full cubes on the host for what are 10^3..10^5 voxels.  Here the
// cubes stay in HBM and only the detections leave it: an ordered stream compaction in three
// launches --
//   count : one coalesced pass, the number of hits of every 4096-voxel chunk and (integer
//           atomics: exact in any order) of every group of 32 chunks            (HBM bound)
//   scan  : exclusive prefix over the GROUP counts (one block, 64-bit running sum; scanning the
//           323 000 chunk counts of a 3681 x 600 x 600 cube in one block took 0.31 ms of a
//           1.2 ms call, profiles/r02_detect_kernel_stats.csv)
//   emit  : chunks that hold a hit are read again, a lane owning 16 CONSECUTIVE voxels, so that a
//           block-wide prefix over the lanes' counts gives the C-order rank of every hit; the
//           chunk's own offset is its group's plus the counts of the chunks before it in the group
// The comparison is made in float64 like the reference's (its threshold is a Python float, the
// cubes widen exactly); NaN compares false.
#include <algorithm>

#include "common.h"

namespace {

constexpr int WA_EPT = 16;                // voxels per lane
constexpr int WA_CHUNK = 256 * WA_EPT;    // voxels per block
constexpr int WA_GROUP = 32;              // chunks per scanned group

typedef float f32x4w __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int wa_block_sum(int v, int *red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void where_count_kernel(const float *__restrict__ cube, long n,
                                                          double thr, int *__restrict__ counts,
                                                          int *__restrict__ gcounts) {
  __shared__ int red[4];
  const long base = (long)blockIdx.x * WA_CHUNK;
  int c = 0;
  if (base + WA_CHUNK <= n) {  // whole chunk: four coalesced 16-byte loads per lane
    const f32x4w *p = reinterpret_cast<const f32x4w *>(cube + base);
    f32x4w v[WA_EPT / 4];
#pragma unroll
    for (int i = 0; i < WA_EPT / 4; ++i) v[i] = p[i * 256 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < WA_EPT / 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) c += (double)v[i][e] > thr;
  } else {
    for (long i = base + threadIdx.x; i < n; i += 256) c += (double)cube[i] > thr;
  }
  c = wa_block_sum(c, red);
  if (threadIdx.x == 0) {
    counts[blockIdx.x] = c;
    if (c) atomicAdd(&gcounts[blockIdx.x / WA_GROUP], c);
  }
}

// offs[g] = sum of counts[0..g), offs[nblk] = total (here: the group counts)  one block of 1024 lanes
__global__ __launch_bounds__(1024) void where_scan_kernel(const int *__restrict__ counts,
                                                          long nblk, long *__restrict__ offs) {
  __shared__ long wtot[16];
  __shared__ long carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  constexpr int PER = 8;  // consecutive chunks per lane and round
  for (long r0 = 0; r0 < nblk; r0 += 1024 * PER) {
    const long i0 = r0 + (long)threadIdx.x * PER;
    int c[PER];
    long mine = 0;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      c[e] = i0 + e < nblk ? counts[i0 + e] : 0;
      mine += c[e];
    }
    long incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const long v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    long before = carry_s;
    for (int w = 0; w < wave; ++w) before += wtot[w];
    long run = before + incl - mine;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      if (i0 + e < nblk) offs[i0 + e] = run;
      run += c[e];
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = run;  // the last lane's running sum = total so far
    __syncthreads();
  }
  if (threadIdx.x == 0) offs[nblk] = carry_s;
}

__global__ __launch_bounds__(256) void where_emit_kernel(
    const float *__restrict__ cube, const uint8_t *__restrict__ aux, long n, long S, int Nx,
    double thr, const int *__restrict__ counts, const long *__restrict__ offs, long cap,
    int *__restrict__ oz, int *__restrict__ oy, int *__restrict__ ox, float *__restrict__ oval,
    uint8_t *__restrict__ oaux) {
  __shared__ int wsum[4];
  __shared__ int before_s;  // hits of the chunks before this one in its group
  if (counts[blockIdx.x] == 0) return;  // (uniform: the cubes are mostly zeros)
  if (threadIdx.x < 64) {
    const int g0 = (int)(blockIdx.x / WA_GROUP) * WA_GROUP, j = (int)threadIdx.x;
    int v = (j < WA_GROUP && g0 + j < (int)blockIdx.x) ? counts[g0 + j] : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (threadIdx.x == 0) before_s = v;
  }
  const long base = (long)blockIdx.x * WA_CHUNK + (long)threadIdx.x * WA_EPT;
  float v[WA_EPT];
  if (base + WA_EPT <= n) {
    const f32x4w *p = reinterpret_cast<const f32x4w *>(cube + base);
#pragma unroll
    for (int i = 0; i < WA_EPT / 4; ++i) {
      const f32x4w q = p[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * i + e] = q[e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < WA_EPT; ++e) v[e] = base + e < n ? cube[base + e] : -INFINITY;
  }
  unsigned hits = 0;
#pragma unroll
  for (int e = 0; e < WA_EPT; ++e) hits |= (unsigned)((double)v[e] > thr) << e;
  const int mine = __popc(hits);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  long pos = offs[blockIdx.x / WA_GROUP] + before_s + (incl - mine);
  for (int w = 0; w < wave; ++w) pos += wsum[w];
#pragma unroll
  for (int e = 0; e < WA_EPT; ++e) {
    if (!((hits >> e) & 1u)) continue;
    if (pos < cap) {
      const long i = base + e;
      const long z = i / S, r = i - z * S;
      const int y = (int)(r / Nx);
      oz[pos] = (int)z;
      oy[pos] = y;
      ox[pos] = (int)(r - (long)y * Nx);
      if (oval) oval[pos] = v[e];
      if (oaux) oaux[pos] = aux[i];
    }
    ++pos;
  }
}

}  // namespace

extern "C" {

int origin_where_above(origin_ctx *ctx, const float *d_cube, const uint8_t *d_aux, int Nz, int Ny,
                       int Nx, double thr, long cap, int *d_z, int *d_y, int *d_x, float *d_val,
                       uint8_t *d_auxout, long *h_count) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_cube && h_count && Nz > 0 && Ny > 0 && Nx > 0, "bad arguments");
  ORIGIN_CHECK_ARG(cap >= 0 && (cap == 0 || (d_z && d_y && d_x)), "cap > 0 needs d_z, d_y, d_x");
  if (thr != thr) {  // np.where(cube > nan) is empty (step 6 without a purity crossing hands a NaN
    *h_count = 0;    // threshold to step 7, reference steps.py:935-939, and the step carries on)
    return ORIGIN_OK;
  }
  ORIGIN_CHECK_ARG(!d_auxout || d_aux, "d_auxout without d_aux");
  ORIGIN_CHECK_ARG(((uintptr_t)d_cube & 15) == 0, "d_cube must be 16-byte aligned");
  const long S = (long)Ny * Nx, n = (long)Nz * S;
  const long nblk = (n + WA_CHUNK - 1) / WA_CHUNK;
  ORIGIN_CHECK_ARG(nblk < (1L << 31), "cube too large");
  const long ngrp = (nblk + WA_GROUP - 1) / WA_GROUP;
  // [offs: (ngrp + 1) x int64 | group counts: ngrp x int32 | counts: nblk x int32]
  void *scr = nullptr;
  const size_t ob = (size_t)(ngrp + 1) * sizeof(long), gb = (size_t)ngrp * sizeof(int);
  int rc = origin_scratch(ctx, ob + gb + (size_t)nblk * sizeof(int), &scr);
  if (rc) return rc;
  long *d_offs = (long *)scr;
  int *d_gcounts = (int *)((char *)scr + ob);
  int *d_counts = (int *)((char *)scr + ob + gb);
  ORIGIN_HIP(hipMemsetAsync(d_gcounts, 0, gb, ctx->stream));
  {
    ProfScope ps(ctx, K_SMALL);
    hipLaunchKernelGGL(where_count_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, d_cube,
                       n, thr, d_counts, d_gcounts);
    hipLaunchKernelGGL(where_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_gcounts, ngrp,
                       d_offs);
    if (cap > 0)
      hipLaunchKernelGGL(where_emit_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream,
                         d_cube, d_aux, n, S, Nx, thr, d_counts, d_offs, cap, d_z, d_y, d_x, d_val,
                         d_auxout);
  }
  ORIGIN_LAUNCH_CHECK();
  ORIGIN_HIP(hipMemcpyAsync(h_count, d_offs + ngrp, sizeof(long), hipMemcpyDeviceToHost,
                            ctx->stream));
  ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
  return ORIGIN_OK;
}

}  // extern "C"
