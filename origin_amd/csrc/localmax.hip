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

// size == 3: a thread owns 4 consecutive rows of one image column and marches z.  Per channel it
// loads the 6 x 3 samples around its outputs (clamped at the field border; the neighbours'
// loads hit the same cache lines), forms the row maxima over x, the 3 x 3 maxima over y, and the
// 3 x 3 x 3 maximum of channel z as the maximum of the three most recent plane maxima kept in
// registers.  No LDS, no barrier: the tile-through-LDS form of round 1 (three block barriers per
// channel) ran at 2.5 TB/s of algorithmic bytes.  Exact: a maximum does not depend on the order
// of its operands.
#ifndef LM_ROWS_N
#define LM_ROWS_N 4
#endif
constexpr int LM_ROWS = LM_ROWS_N;
// NC = 2: correl (local maxima) and correl_min (local maxima of its negative) in one march: the
// mask is read once and the index arithmetic is shared.
template <int NC>
__global__ __launch_bounds__(256) void local_max3_kernel(const float *__restrict__ a0,
                                                         const float *__restrict__ a1,
                                                         const uint8_t *__restrict__ mask, int Nz,
                                                         int Ny, int Nx, int zper, float sign0,
                                                         float *__restrict__ out0,
                                                         float *__restrict__ out1) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int yb = (blockIdx.y * 4 + threadIdx.y) * LM_ROWS;
  if (x >= Nx || yb >= Ny) return;
  const int z0 = blockIdx.z * zper, z1 = min(Nz, z0 + zper);
  const long S = (long)Ny * Nx;
  const int xl = max(x - 1, 0), xr = min(x + 1, Nx - 1);
  long roff[LM_ROWS + 2];  // rows yb - 1 .. yb + LM_ROWS, clamped
#pragma unroll
  for (int r = 0; r < LM_ROWS + 2; ++r) roff[r] = (long)min(max(yb - 1 + r, 0), Ny - 1) * Nx;
  auto plane = [&](const float *a, float sign, int z, float (&p)[LM_ROWS], float (&c)[LM_ROWS]) {
    const float *pz = a + (long)min(max(z, 0), Nz - 1) * S;
    float l[LM_ROWS + 2], m[LM_ROWS + 2], rr[LM_ROWS + 2];
#pragma unroll
    for (int r = 0; r < LM_ROWS + 2; ++r) {
      const float *row = pz + roff[r];
      l[r] = sign * row[xl], m[r] = sign * row[x], rr[r] = sign * row[xr];
    }
    float xm[LM_ROWS + 2];
#pragma unroll
    for (int r = 0; r < LM_ROWS + 2; ++r) xm[r] = fmaxf(fmaxf(l[r], m[r]), rr[r]);
#pragma unroll
    for (int r = 0; r < LM_ROWS; ++r) {
      p[r] = fmaxf(fmaxf(xm[r], xm[r + 1]), xm[r + 2]);
      c[r] = m[r + 1];
    }
  };
  float pa[NC][LM_ROWS], pb[NC][LM_ROWS], pc[NC][LM_ROWS], cb[NC][LM_ROWS], cc[NC][LM_ROWS];
  float dummy[LM_ROWS];
  plane(a0, sign0, z0 - 1, pa[0], dummy);
  plane(a0, sign0, z0, pb[0], cb[0]);
  if constexpr (NC == 2) {
    plane(a1, -1.0f, z0 - 1, pa[1], dummy);
    plane(a1, -1.0f, z0, pb[1], cb[1]);
  }
  for (int z = z0; z < z1; ++z) {
    plane(a0, sign0, z + 1, pc[0], cc[0]);
    if constexpr (NC == 2) plane(a1, -1.0f, z + 1, pc[1], cc[1]);
#pragma unroll
    for (int r = 0; r < LM_ROWS; ++r) {
      const int y = yb + r;
      if (y < Ny) {
        const long idx = (long)z * S + (long)y * Nx + x;
        const bool unmasked = !(mask && mask[idx]);
        const float m0 = fmaxf(fmaxf(pa[0][r], pb[0][r]), pc[0][r]);
        out0[idx] = (cb[0][r] == m0 && unmasked) ? m0 : 0.0f;  // local_max *= local_mask (lib :1247)
        if constexpr (NC == 2) {
          const float m1 = fmaxf(fmaxf(pa[1][r], pb[1][r]), pc[1][r]);
          out1[idx] = (cb[1][r] == m1 && unmasked) ? m1 : 0.0f;
        }
      }
#pragma unroll
      for (int q = 0; q < NC; ++q) pa[q][r] = pb[q][r], pb[q][r] = pc[q][r], cb[q][r] = cc[q][r];
    }
  }
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
  if (size == 3) {  // the reference's default (steps.py:453, :796)
    const long tiles = (long)cdiv(Nx, 64) * cdiv(Ny, 4 * LM_ROWS);
    int nzb = (int)(((long)ctx->num_cu * 32 + tiles - 1) / tiles);  // ~32 blocks per CU
    nzb = nzb < 1 ? 1 : (nzb > Nz ? Nz : nzb);
    const int zper = cdiv(Nz, nzb);
    dim3 g3(cdiv(Nx, 64), cdiv(Ny, 4 * LM_ROWS), cdiv(Nz, zper));
    if (d_correl && d_local_max && d_correl_min && d_local_min)
      hipLaunchKernelGGL(local_max3_kernel<2>, g3, block, 0, ctx->stream, d_correl, d_correl_min,
                         d_mask, Nz, Ny, Nx, zper, 1.0f, d_local_max, d_local_min);
    else if (d_correl && d_local_max)
      hipLaunchKernelGGL(local_max3_kernel<1>, g3, block, 0, ctx->stream, d_correl,
                         (const float *)nullptr, d_mask, Nz, Ny, Nx, zper, 1.0f, d_local_max,
                         (float *)nullptr);
    else if (d_correl_min && d_local_min)
      hipLaunchKernelGGL(local_max3_kernel<1>, g3, block, 0, ctx->stream, d_correl_min,
                         (const float *)nullptr, d_mask, Nz, Ny, Nx, zper, -1.0f, d_local_min,
                         (float *)nullptr);
    ORIGIN_LAUNCH_CHECK();
    return ORIGIN_OK;
  }
  if (d_correl && d_local_max)
    hipLaunchKernelGGL(local_max_kernel, grid, block, 0, ctx->stream, d_correl, d_mask, Nz, Ny, Nx,
                       lo, hi, 1.0f, d_local_max);
  if (d_correl_min && d_local_min)
    hipLaunchKernelGGL(local_max_kernel, grid, block, 0, ctx->stream, d_correl_min, d_mask, Nz, Ny,
                       Nx, lo, hi, -1.0f, d_local_min);
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}
