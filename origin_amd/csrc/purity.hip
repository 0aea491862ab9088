// Device side of Compute_threshold_purity  (reference muse_origin/lib_origin.py:1391-1479,
// called by ComputePurityThreshold.run, steps.py:874-890).
//
// The reference pulls four full cubes through NumPy to obtain (a) two scalars and a median of
// the per-spaxel maxima for the default threshold list (:1437-1442) and (b) the number of
// voxels above each of ~50 thresholds in cube_local_max and in cube_local_min restricted to
// the background of the segmentation map (:1444-1452).  Both are reductions: here only a
// (Ny, Nx) map or nthr integers leave the GPU.  Counts are integer atomics (exact, order
// independent); comparisons are made in float64 like the reference's.
#include <algorithm>
#include <vector>

#include "common.h"

namespace {

// map[s] = max_z cube[z][s]  (0 where keep[s] == 0: cube_local_min * segmask, :1430)
__global__ __launch_bounds__(256) void zmax_map_kernel(const float *__restrict__ cube,
                                                       const uint8_t *__restrict__ keep, int Nz,
                                                       long S, int zchunk,
                                                       float *__restrict__ part) {
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  const int z0 = blockIdx.y * zchunk, z1 = min(Nz, z0 + zchunk);
  const bool k = !keep || keep[s];
  float m = -INFINITY;
#pragma unroll 4
  for (int z = z0; z < z1; ++z) {
    const float v = cube[(long)z * S + s];
    m = fmaxf(m, k ? v : 0.0f);
  }
  part[(long)blockIdx.y * S + s] = m;
}

__global__ __launch_bounds__(256) void zmax_final_kernel(const float *__restrict__ part, int nzc,
                                                         long S, float *__restrict__ map) {
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  float m = -INFINITY;
  for (int c = 0; c < nzc; ++c) m = fmaxf(m, part[(long)c * S + s]);
  map[s] = m;
}

// hist[b] = number of voxels whose value exceeds exactly the b+1 smallest thresholds
// (thr sorted ascending, nthr <= 1024); n(thr[j]) = sum_{b >= j} hist[b] on the host
__global__ __launch_bounds__(256) void count_above_kernel(const float *__restrict__ cube,
                                                          const uint8_t *__restrict__ keep,
                                                          long total, long S, int nthr,
                                                          const double *__restrict__ thr,
                                                          unsigned long long *__restrict__ hist) {
  __shared__ double sthr[1024];
  __shared__ unsigned shist[1024];
  for (int i = threadIdx.x; i < nthr; i += 256) sthr[i] = thr[i], shist[i] = 0u;
  __syncthreads();
  const double t0 = sthr[0];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const float vf = cube[i];
    if (!((double)vf > t0)) continue;        // the cubes are mostly zeros
    if (keep && !keep[i % S]) continue;      // cube_local_min * segmask
    const double v = (double)vf;
    int lo = 0, hi = nthr;                   // number of thresholds < v
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sthr[mid] < v) lo = mid + 1;
      else hi = mid;
    }
    atomicAdd(&shist[lo - 1], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nthr; i += 256)
    if (shist[i]) atomicAdd(&hist[i], (unsigned long long)shist[i]);
}

}  // namespace

extern "C" {

int origin_zmax_map(origin_ctx *ctx, const float *d_cube, const uint8_t *d_keep, int Nz, long S,
                    float *d_map) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_cube && d_map && Nz > 0 && S > 0, "bad arguments");
  int nzc = (int)(((long)ctx->num_cu * 8 * 256 + S - 1) / S);
  nzc = std::max(1, std::min(nzc, std::min(64, Nz)));
  const int zchunk = cdiv(Nz, nzc);
  nzc = cdiv(Nz, zchunk);
  void *scr = nullptr;
  int rc = origin_scratch(ctx, (size_t)nzc * S * sizeof(float), &scr);
  if (rc) return rc;
  ProfScope ps(ctx, K_SMALL);
  hipLaunchKernelGGL(zmax_map_kernel, dim3(cdiv(S, 256), nzc), dim3(256), 0, ctx->stream, d_cube,
                     d_keep, Nz, S, zchunk, (float *)scr);
  hipLaunchKernelGGL(zmax_final_kernel, dim3(cdiv(S, 256)), dim3(256), 0, ctx->stream,
                     (const float *)scr, nzc, S, d_map);
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

int origin_count_above(origin_ctx *ctx, const float *d_cube, const uint8_t *d_keep, int Nz, long S,
                       int nthr, const double *h_thr, long *h_counts) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_cube && h_thr && h_counts && Nz > 0 && S > 0, "bad arguments");
  ORIGIN_CHECK_ARG(nthr >= 1 && nthr <= 1024, "1..1024 thresholds");
  for (int i = 0; i < nthr; ++i) ORIGIN_CHECK_ARG(h_thr[i] == h_thr[i], "NaN threshold");
  std::vector<int> order(nthr);
  for (int i = 0; i < nthr; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return h_thr[a] < h_thr[b]; });
  std::vector<double> sorted(nthr);
  for (int i = 0; i < nthr; ++i) sorted[i] = h_thr[order[i]];
  void *scr = nullptr;
  const size_t tb = (size_t)nthr * sizeof(double), hb = (size_t)nthr * sizeof(unsigned long long);
  int rc = origin_scratch(ctx, tb + hb, &scr);
  if (rc) return rc;
  double *d_thr = (double *)scr;
  unsigned long long *d_hist = (unsigned long long *)((char *)scr + tb);
  ORIGIN_HIP(hipMemcpyAsync(d_thr, sorted.data(), tb, hipMemcpyHostToDevice, ctx->stream));
  ORIGIN_HIP(hipMemsetAsync(d_hist, 0, hb, ctx->stream));
  {
    ProfScope ps(ctx, K_SMALL);
    hipLaunchKernelGGL(count_above_kernel, dim3(ctx->num_cu * 16), dim3(256), 0, ctx->stream,
                       d_cube, d_keep, (long)Nz * S, S, nthr, d_thr, d_hist);
  }
  ORIGIN_LAUNCH_CHECK();
  std::vector<unsigned long long> hist(nthr);
  ORIGIN_HIP(hipMemcpyAsync(hist.data(), d_hist, hb, hipMemcpyDeviceToHost, ctx->stream));
  ORIGIN_HIP(hipStreamSynchronize(ctx->stream));  // `sorted` / `hist` are host temporaries
  unsigned long long run = 0;
  for (int j = nthr - 1; j >= 0; --j) {  // n(thr_sorted[j]) = sum_{b >= j} hist[b]
    run += hist[j];
    h_counts[order[j]] = (long)run;
  }
  return ORIGIN_OK;
}

}  // extern "C"
