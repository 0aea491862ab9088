// compute_local_max  (reference muse_origin/lib_origin.py:1220-1256): size^3 maximum
// filter, keep voxels equal to their window maximum and not masked, zero elsewhere; the
// same on -correl_min.  scipy's default border mode 'reflect' duplicates edge samples,
// which for a maximum is the same as clamping the window to the cube.
#include <cstdlib>
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

// size == 3, Nx % 4 == 0 (round 3): four consecutive x per lane.  The 4-byte form above issues
// 37 loads and 8 stores per 8 outputs and ran at 3.6 TB/s of algorithmic bytes (17 B per voxel:
// correl 4 + correl_min 4 + mask 1 in, two cubes out) -- bound by instruction issue, not by HBM.
// Here a lane owns a float4 of R rows; the x neighbours of its first / last sample come from the
// adjacent lanes (whole-wave DPP shifts: no LDS, no extra load).  Lanes walk the flattened (row
// group, float4 column) index, so every wave is full whatever Nx is; at a row's ends the window
// is clamped to the row (a duplicate does not change a maximum: scipy's 'reflect').  Per plane and cube: R + 2 float4 loads for 4 R
// outputs.  Bit exact.
__device__ __forceinline__ float wave_from_prev(float v) {  // lane i gets lane i - 1 (lane 0: itself)
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138,
                                                    0xF, 0xF, false));  // wave_shr:1
}
__device__ __forceinline__ float wave_from_next(float v) {  // lane i gets lane i + 1 (lane 63: itself)
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130,
                                                    0xF, 0xF, false));  // wave_shl:1
}

template <int NC, int R>
__global__ __launch_bounds__(256) void local_max3v_kernel(const float *__restrict__ a0,
                                                          const float *__restrict__ a1,
                                                          const uint8_t *__restrict__ mask, int Nz,
                                                          int Ny, int Nx, int zper, float sign0,
                                                          float *__restrict__ out0,
                                                          float *__restrict__ out1) {
  // Waves overlap by two lanes: wave w holds the flattened (row group, float4 column) indices
  // 62 w - 1 .. 62 w + 62; lanes 1..62 produce outputs, lanes 0 and 63 only hand their samples to
  // their neighbours -- no lane ever loads a halo sample (a conditional 4-byte load per row and
  // side cost more than the 3 % of idle lanes: 9.0 against 6.3 ms for the one-sample form).
  const int nx4 = Nx >> 2, ngrp = (Ny + R - 1) / R;
  const long total = (long)ngrp * nx4;
  const int lane = threadIdx.x & 63;
  const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long t_raw = 62 * wv - 1 + lane;
  const bool live = lane >= 1 && lane <= 62 && t_raw < total;
  const long t = min(max(t_raw, 0L), total - 1);
  const int grp = (int)(t / nx4);
  const int x4 = (int)(t - (long)grp * nx4);
  const int yb = grp * R;
  const int z0 = blockIdx.y * zper, z1 = min(Nz, z0 + zper);
  const long S = (long)Ny * Nx;
  // at a row's ends the window is clamped to the row: the lane's own sample
  const bool first = x4 == 0, last = x4 == nx4 - 1;
  long roff[R + 2];  // rows yb - 1 .. yb + R, clamped
#pragma unroll
  for (int r = 0; r < R + 2; ++r) roff[r] = (long)min(max(yb - 1 + r, 0), Ny - 1) * Nx + 4 * x4;
  // 3 x 3 maxima of the lane's R x 4 outputs in plane z -> p, and the plane's own samples -> c
  auto plane = [&](const float *a, float sign, int z, float (&p)[R][4], float (&c)[R][4]) {
    const float *pz = a + (long)min(max(z, 0), Nz - 1) * S;
    float xm[R + 2][4], mid[R + 2][4];
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
      const float *row = pz + roff[r];
      const float4 v = *reinterpret_cast<const float4 *>(row);
      const float s0 = sign * v.x, s1 = sign * v.y, s2 = sign * v.z, s3 = sign * v.w;
      float l = wave_from_prev(s3), rr = wave_from_next(s0);
      if (first) l = s0;
      if (last) rr = s3;
      xm[r][0] = fmaxf(fmaxf(l, s0), s1);
      xm[r][1] = fmaxf(fmaxf(s0, s1), s2);
      xm[r][2] = fmaxf(fmaxf(s1, s2), s3);
      xm[r][3] = fmaxf(fmaxf(s2, s3), rr);
      mid[r][0] = s0, mid[r][1] = s1, mid[r][2] = s2, mid[r][3] = s3;
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        p[r][e] = fmaxf(fmaxf(xm[r][e], xm[r + 1][e]), xm[r + 2][e]);
        c[r][e] = mid[r + 1][e];
      }
  };
  float pa[NC][R][4], pb[NC][R][4], pc[NC][R][4], cb[NC][R][4], cc[NC][R][4];
  float dummy[R][4];
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
    for (int r = 0; r < R; ++r) {
      const int y = yb + r;
      if (live && y < Ny) {
        const long idx = (long)z * S + (long)y * Nx + 4 * x4;
        const unsigned mk = mask ? *reinterpret_cast<const unsigned *>(mask + idx) : 0u;
        float o[NC][4];
#pragma unroll
        for (int q = 0; q < NC; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float m = fmaxf(fmaxf(pa[q][r][e], pb[q][r][e]), pc[q][r][e]);
            const bool unmasked = ((mk >> (8 * e)) & 0xffu) == 0u;
            o[q][e] = (cb[q][r][e] == m && unmasked) ? m : 0.0f;  // local_max *= local_mask (lib :1247)
          }
        *reinterpret_cast<float4 *>(out0 + idx) = make_float4(o[0][0], o[0][1], o[0][2], o[0][3]);
        if constexpr (NC == 2)
          *reinterpret_cast<float4 *>(out1 + idx) = make_float4(o[1][0], o[1][1], o[1][2], o[1][3]);
      }
#pragma unroll
      for (int q = 0; q < NC; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          pa[q][r][e] = pb[q][r][e], pb[q][r][e] = pc[q][r][e], cb[q][r][e] = cc[q][r][e];
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
    auto al16 = [](const void *q) { return ((uintptr_t)q & 15) == 0; };
    const bool vec = (Nx & 3) == 0 && al16(d_correl) && al16(d_correl_min) && al16(d_local_max) &&
                     al16(d_local_min) && ((uintptr_t)d_mask & 3) == 0 && !getenv("ORIGIN_LOCALMAX_SCALAR");
    if (vec) {
      // ORIGIN_LOCALMAX_FORM: 2 = both cubes in one march, four rows per lane (default: 234
      // VGPRs, two waves per SIMD); 0 = both cubes, two rows per lane; 1 = one march per cube,
      // four rows per lane.  Measured at 3681 x 600 x 600 (tools/localmax_time.py): 4.79 / 6.42 /
      // 5.00 ms against 6.27-6.50 ms for the one-sample form -- the row loads a lane shares with
      // the row groups above and below (R + 2 rows for R outputs) are what is left: both forms
      // move ~3.5 TB/s of loads through L2; streaming (non-temporal) stores changed nothing.
      static const int form = getenv("ORIGIN_LOCALMAX_FORM") ? atoi(getenv("ORIGIN_LOCALMAX_FORM")) : 2;
      const bool both = d_correl && d_local_max && d_correl_min && d_local_min;
      auto go = [&](auto kernel, int R, const float *a, const float *b, float sgn, float *oa,
                    float *ob) {
        const long threads = (long)cdiv(Ny, R) * (Nx / 4);
        const long bx = (threads + 4 * 62 - 1) / (4 * 62);  // 62 producing lanes per wave
        int nzc = (int)(((long)ctx->num_cu * 16 + bx - 1) / bx);  // ~16 blocks per CU
        nzc = nzc < 1 ? 1 : (nzc > cdiv(Nz, 32) ? cdiv(Nz, 32) : nzc);
        const int zp = cdiv(Nz, nzc);
        hipLaunchKernelGGL(kernel, dim3((unsigned)bx, cdiv(Nz, zp)), dim3(256), 0, ctx->stream, a, b,
                           d_mask, Nz, Ny, Nx, zp, sgn, oa, ob);
      };
      if (both && form == 0)
        go(local_max3v_kernel<2, 2>, 2, d_correl, d_correl_min, 1.0f, d_local_max, d_local_min);
      else if (both && form == 2)
        go(local_max3v_kernel<2, 4>, 4, d_correl, d_correl_min, 1.0f, d_local_max, d_local_min);
      else {
        if (d_correl && d_local_max)
          go(local_max3v_kernel<1, 4>, 4, d_correl, (const float *)nullptr, 1.0f, d_local_max,
             (float *)nullptr);
        if (d_correl_min && d_local_min)
          go(local_max3v_kernel<1, 4>, 4, d_correl_min, (const float *)nullptr, -1.0f, d_local_min,
             (float *)nullptr);
      }
      ORIGIN_LAUNCH_CHECK();
      return ORIGIN_OK;
    }
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
