// Spaxel-list moves between (Nz, S) cubes and packed (Nz, n) buffers.
//
// The tiled path (origin_amd/multigpu.py) hands PCA areas -- irregular sets of spaxels, reference
// muse_origin/steps.py:492-569 -- to ranks as wholes, so what one rank needs from another before
// the GLR (lib_origin.py:1027-1043 reads P // 2 spaxels around every output) is a LIST of spaxel
// columns, not a rectangle.  gather packs the listed columns of a cube into a contiguous buffer
// that goes over RCCL (or through the host group), scatter unpacks it on the other side.  Lists
// are sorted in C order by their makers, so neighbouring lanes mostly touch neighbouring spaxels.
#include "common.h"

namespace {

// PACK = true : packed[z][i] = cube[z][idx[i]]      (gather)
// PACK = false: cube[z][idx[i]] = packed[z][i]      (scatter)
template <typename T, bool PACK>
__global__ __launch_bounds__(256) void columns_kernel(T *__restrict__ cube, long S,
                                                      const int *__restrict__ idx, long n, int Nz,
                                                      int zper, T *__restrict__ packed) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long s = idx[i];
  const int z0 = blockIdx.y * zper, z1 = min(Nz, z0 + zper);
#pragma unroll 4
  for (int z = z0; z < z1; ++z) {
    if (PACK) packed[(long)z * n + i] = cube[(long)z * S + s];
    else cube[(long)z * S + s] = packed[(long)z * n + i];
  }
}

template <bool PACK>
int columns_launch(origin_ctx *ctx, void *cube, int Nz, long S, const int *d_idx, long n, int elem,
                   void *packed) {
  if (n == 0 || Nz == 0) return ORIGIN_OK;
  const long bx = (n + 255) / 256;
  int nzb = (int)(((long)ctx->num_cu * 16 + bx - 1) / bx);  // ~16 blocks per CU
  nzb = nzb < 1 ? 1 : (nzb > Nz ? Nz : nzb);
  if (nzb > 65535) nzb = 65535;
  const int zper = cdiv(Nz, nzb);
  dim3 grid((unsigned)bx, cdiv(Nz, zper));
  ProfScope ps(ctx, K_SMALL);
  if (elem == 4)
    hipLaunchKernelGGL((columns_kernel<float, PACK>), grid, dim3(256), 0, ctx->stream,
                       (float *)cube, S, d_idx, n, Nz, zper, (float *)packed);
  else
    hipLaunchKernelGGL((columns_kernel<uint8_t, PACK>), grid, dim3(256), 0, ctx->stream,
                       (uint8_t *)cube, S, d_idx, n, Nz, zper, (uint8_t *)packed);
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

}  // namespace

extern "C" {

int origin_gather_columns(origin_ctx *ctx, const void *d_cube, int Nz, long S, const int *d_idx,
                          long n, int elem, void *d_packed) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(Nz >= 0 && S > 0 && n >= 0 && n < (1L << 31), "bad shape");
  ORIGIN_CHECK_ARG(elem == 1 || elem == 4, "elem must be 1 or 4");
  ORIGIN_CHECK_ARG(n == 0 || (d_cube && d_idx && d_packed), "null pointer");
  return columns_launch<true>(ctx, const_cast<void *>(d_cube), Nz, S, d_idx, n, elem, d_packed);
}

int origin_scatter_columns(origin_ctx *ctx, void *d_cube, int Nz, long S, const int *d_idx, long n,
                           int elem, const void *d_packed) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(Nz >= 0 && S > 0 && n >= 0 && n < (1L << 31), "bad shape");
  ORIGIN_CHECK_ARG(elem == 1 || elem == 4, "elem must be 1 or 4");
  ORIGIN_CHECK_ARG(n == 0 || (d_cube && d_idx && d_packed), "null pointer");
  return columns_launch<false>(ctx, d_cube, Nz, S, d_idx, n, elem, const_cast<void *>(d_packed));
}

}  // extern "C"
