// compute_local_max  (reference muse_origin/lib_origin.py:1220-1256): size^3 maximum
// filter, keep voxels equal to their window maximum and not masked, zero elsewhere; the
// same on -correl_min.  scipy's default border mode 'reflect' duplicates edge samples,
// which for a maximum is the same as clamping the window to the cube.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void local_max_kernel(const float *__restrict__ a,
                                                        const uint8_t *__restrict__ mask, int Nz,
                                                        int Ny, int Nx, int lo, int hi,
                                                        float sign, float *__restrict__ out) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = blockIdx.y * 4 + threadIdx.y;
  const int z = blockIdx.z;
  if (x >= Nx || y >= Ny) return;
  const long S = (long)Ny * Nx;
  const long idx = (long)z * S + (long)y * Nx + x;
  const float v = sign * a[idx];
  float m = v;
  const int z0 = max(0, z - lo), z1 = min(Nz - 1, z + hi);
  const int y0 = max(0, y - lo), y1 = min(Ny - 1, y + hi);
  const int x0 = max(0, x - lo), x1 = min(Nx - 1, x + hi);
  for (int zz = z0; zz <= z1; ++zz)
    for (int yy = y0; yy <= y1; ++yy) {
      const float *row = a + (long)zz * S + (long)yy * Nx;
      for (int xx = x0; xx <= x1; ++xx) m = fmaxf(m, sign * row[xx]);
    }
  const bool keep = (v == m) && !(mask && mask[idx]);
  out[idx] = keep ? m : 0.0f;  // local_max *= local_mask                  (lib :1247)
}

}  // namespace

extern "C" int origin_local_max(origin_ctx *ctx, const float *d_correl,
                                const float *d_correl_min, const uint8_t *d_mask, int Nz,
                                int Ny, int Nx, int size, float *d_local_max,
                                float *d_local_min) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(Nz > 0 && Ny > 0 && Nx > 0 && size >= 1 && size <= 15, "bad shape / size");
  ORIGIN_CHECK_ARG(Nz <= 65535, "Nz too large for the launch grid");
  // scipy maximum_filter: window offsets  -(size//2) .. size-1-(size//2)
  const int lo = size / 2, hi = size - 1 - size / 2;
  dim3 grid(cdiv(Nx, 64), cdiv(Ny, 4), Nz), block(64, 4);
  ProfScope ps(ctx, K_LOCAL_MAX);
  if (d_correl && d_local_max)
    hipLaunchKernelGGL(local_max_kernel, grid, block, 0, ctx->stream, d_correl, d_mask, Nz, Ny, Nx,
                       lo, hi, 1.0f, d_local_max);
  if (d_correl_min && d_local_min)
    hipLaunchKernelGGL(local_max_kernel, grid, block, 0, ctx->stream, d_correl_min, d_mask, Nz, Ny,
                       Nx, lo, hi, -1.0f, d_local_min);
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}
