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

// size == 3: a block owns a 64 x 16 tile of spaxels and marches z.  Per channel the tile (with a
// one-spaxel rim, clamped at the field border) goes through LDS once, every thread forms the 3x3
// spatial maximum of its 4 outputs, and the 3x3x3 maximum of channel z is the maximum of the three
// most recent plane maxima, kept in registers: each plane is read once (+16 % rim) instead of 27
// times.  Exact: a maximum does not depend on the order of its operands.
constexpr int LM_TX = 64, LM_TY = 16;
__global__ __launch_bounds__(256) void local_max3_kernel(const float *__restrict__ a,
                                                         const uint8_t *__restrict__ mask, int Nz,
                                                         int Ny, int Nx, int zper, float sign,
                                                         float *__restrict__ out) {
  __shared__ float tile[LM_TY + 2][LM_TX + 2];
  const int x0 = blockIdx.x * LM_TX, y0 = blockIdx.y * LM_TY;
  const int z0 = blockIdx.z * zper, z1 = min(Nz, z0 + zper);
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 64 + tx;
  const long S = (long)Ny * Nx;
  const int x = x0 + tx;
  auto load_plane = [&](int z) {  // plane z (clamped) -> LDS, sign applied
    const float *pz = a + (long)min(max(z, 0), Nz - 1) * S;
    for (int e = tid; e < (LM_TY + 2) * (LM_TX + 2); e += 256) {
      const int ry = e / (LM_TX + 2), rx = e - ry * (LM_TX + 2);
      const int yy = min(max(y0 - 1 + ry, 0), Ny - 1), xx = min(max(x0 - 1 + rx, 0), Nx - 1);
      tile[ry][rx] = sign * pz[(long)yy * Nx + xx];
    }
  };
  auto plane_max = [&](float (&p)[4], float (&c)[4]) {  // 3x3 maxima and centre values
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ry = ty + 4 * r + 1;  // row in the tile
      float m = -INFINITY;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
        m = fmaxf(m, fmaxf(fmaxf(tile[ry + dy][tx], tile[ry + dy][tx + 1]), tile[ry + dy][tx + 2]));
      p[r] = m;
      c[r] = tile[ry][tx + 1];
    }
  };
  float pa[4], pb[4], pc[4], ca[4], cb[4], cc[4];  // planes z-1, z, z+1
  load_plane(z0 - 1);
  __syncthreads();
  plane_max(pa, ca);
  __syncthreads();
  load_plane(z0);
  __syncthreads();
  plane_max(pb, cb);
  for (int z = z0; z < z1; ++z) {
    __syncthreads();
    load_plane(z + 1);
    __syncthreads();
    plane_max(pc, cc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = y0 + ty + 4 * r;
      if (x < Nx && y < Ny) {
        const long idx = (long)z * S + (long)y * Nx + x;
        const float m = fmaxf(fmaxf(pa[r], pb[r]), pc[r]);
        const bool keep = (cb[r] == m) && !(mask && mask[idx]);
        out[idx] = keep ? m : 0.0f;  // local_max *= local_mask                (lib :1247)
      }
      pa[r] = pb[r], pb[r] = pc[r], ca[r] = cb[r], cb[r] = cc[r];
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
    const long tiles = (long)cdiv(Nx, LM_TX) * cdiv(Ny, LM_TY);
    int nzb = (int)(((long)ctx->num_cu * 16 + tiles - 1) / tiles);
    nzb = nzb < 1 ? 1 : (nzb > Nz ? Nz : nzb);
    const int zper = cdiv(Nz, nzb);
    dim3 g3(cdiv(Nx, LM_TX), cdiv(Ny, LM_TY), cdiv(Nz, zper));
    if (d_correl && d_local_max)
      hipLaunchKernelGGL(local_max3_kernel, g3, block, 0, ctx->stream, d_correl, d_mask, Nz, Ny, Nx,
                         zper, 1.0f, d_local_max);
    if (d_correl_min && d_local_min)
      hipLaunchKernelGGL(local_max3_kernel, g3, block, 0, ctx->stream, d_correl_min, d_mask, Nz, Ny,
                         Nx, zper, -1.0f, d_local_min);
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
