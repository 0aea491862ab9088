// GLR spatial stage on the matrix cores  (reference _convolve_fsf, lib_origin.py:1027-1043,
// for one field without weight map:  cube_fsf[z] = corr2(cube[z], psf_z - mean(psf_z)), 'same',
// zeros outside the field).
//
// For one channel and a tile of 32 output columns x0..x0+31 and 32 output rows y0..y0+31
//     out[y0+n][x0+m] = sum_dy sum_i A_dy[m][i] X_dy[i][n]
//     A_dy[m][i] = k[dy][i - m]              (0 <= i - m < P, banded Toeplitz, 32 x 64)
//     X_dy[i][n] = in[y0 + n + dy - c][x0 - c + i]
// i.e. P accumulated [32 x 64] x [64 x 32] products: the x axis of the kernel becomes the K
// dimension of an MFMA, the y axis stays an outer sum whose B operand is the input tile read
// one row lower each time.  As in spectral_mfma_kernel the product runs on
// v_mfma_f32_32x32x16_f16 with a two-term f16 split of both operands (data scaled by a power of
// two per block and channel, taps by 2^12; Ah Bh + Ah Bl + Al Bh, fp32 accumulation).
//
// A block (8 waves, two per SIMD) owns a 128 x 64 region (two of them, stacked in y, sharing
// the channel's fragment table) of one channel at a time and marches z.  Per channel it (1) converts the (128+P-1) x (64+P-1) input tile -- staged in registers
// while the previous channel was computed -- to f16 hi/lo images in LDS (row pitch 19 x 16 B:
// the 16-byte B-fragment reads of a lane group fall on distinct banks), (2) expands the P x P
// taps into the Toeplitz fragment table: per kernel row 8 copies of the zero-padded tap array
// shifted by 0..7 elements, so that the A fragment of any (lane, k-step) is one aligned
// ds_read_b128 (pitch between copies = 64 mod 256 B: conflict free), (3) runs P x 4 k-steps x 3
// MFMAs per wave, four 16-byte LDS reads per 3 MFMAs (LDS sustains two ds_read_b128 per MFMA
// slot and SIMD; MI355X_MICROARCH.md, LDS), no VALU work in the loop, (4) stores the 32 x 32
// tile.  Only eligible shapes come here (P <= 25, P/2 and Nx multiples of 4); everything else
// -- weight maps, several fields, other PSF sizes -- stays on spatial4x4_kernel / spatial_kernel.
#include <algorithm>

#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SM_PMAX = 25;
constexpr int SM_RX = 128, SM_RY = 64;          // region of a block
constexpr int SM_IW = SM_RX + SM_PMAX - 1;      // 152 input columns
constexpr int SM_IH = SM_RY + SM_PMAX - 1;      // 88 input rows
constexpr int SM_PITCH = 304;                   // bytes per image row: 152 f16 = 19 x 16 B (odd)
constexpr int SM_IMG = SM_IH * SM_PITCH + 32;   // one f16 image (hi or lo) + pad (a window of
                                                // the last wave column runs 8 columns past a row)
constexpr int SM_GROUPS = 12;                   // 16-byte groups of a shifted tap copy
constexpr int SM_TAP_LOG2 = 12;
constexpr int SM_NQ = (SM_IW / 4 * SM_IH + 511) / 512;  // staged float4 per thread (7)

// Fragment table (bytes): offset(copy, dy, hl, q) = copy * sm_copy_all(P) + (2 dy + hl) * 192 + 16 q;
// the pitch between the 8 shifted copies is 64 mod 256 (the lanes of a ds_read_b128 group differ
// in copy and q: with this pitch their 16-byte slots are distinct modulo the 64 banks).
__host__ __device__ constexpr int sm_copy_all(int P) {
  const int raw = P * 2 * SM_GROUPS * 16;
  return (raw + 255) / 256 * 256 + 64;
}

template <bool DUMMY>
__global__ __launch_bounds__(512, 1) void spatial_mfma_kernel(const float *__restrict__ A,
                                                              const float *__restrict__ taps,
                                                              int Nz, int Ny, int Nx, int P,
                                                              int zper, int R,
                                                              float *__restrict__ out) {
  extern __shared__ __align__(16) char sm_lds[];
  const int copy_all = sm_copy_all(P);
  char *tab = sm_lds;                           // Toeplitz fragments
  char *img_h = sm_lds + 8 * copy_all;          // f16 hi image of the input tile
  char *img_l = img_h + SM_IMG;                 // f16 lo image
  float *tapf = reinterpret_cast<float *>(img_l + SM_IMG);  // [P*P] taps of this channel
  unsigned *redmax = reinterpret_cast<unsigned *>(tapf + SM_PMAX * SM_PMAX);

  const int c = P / 2, H = P - 1;
  // all 152 columns are staged whatever P (the k-steps of a wave always read a 64-column
  // window; columns beyond x0 + 127 + H only ever meet zero taps but must be finite)
  const int iw4 = SM_IW / 4, ih = SM_RY + H;
  // a block serves R regions stacked in y: the fragment table of a channel is built once for
  // all of them (the PSF differs per channel, not per region)
  const int x0 = blockIdx.x * SM_RX;
  const int yb = blockIdx.y * R * SM_RY;  // y0 of sub-region r is yb + r * SM_RY
  const int nr = min(R, (Ny - yb + SM_RY - 1) / SM_RY);  // sub-regions inside the field
  const long S = (long)Ny * Nx;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wx = wave & 3, wy = wave >> 2;
  const int z0 = blockIdx.z * zper, z1 = min(Nz, z0 + zper);

  // ---- register staging of the next channel's input tile and taps
  float4 stage[SM_NQ];
  float tapreg[2];
  auto load_tile = [&](int z, int y0) {
    const float *Az = A + (long)z * S;
#pragma unroll
    for (int q = 0; q < SM_NQ; ++q) {
      const int e = tid + 512 * q;
      const int ry = e / iw4, cx = e - ry * iw4;
      const int y = y0 - c + ry, x = x0 - c + 4 * cx;  // x % 4 == 0, Nx % 4 == 0
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ry < ih && y >= 0 && y < Ny && x >= 0 && x < Nx)
        v = *reinterpret_cast<const float4 *>(Az + (long)y * Nx + x);
      stage[q] = v;
    }
    const float *kz = taps + (long)z * P * P;
    tapreg[0] = tid < P * P ? kz[tid] : 0.f;
    tapreg[1] = tid + 512 < P * P ? kz[tid + 512] : 0.f;
  };

  // per-lane fragment addresses
  const int E0 = 8 * h - r + 31;
  const char *a_lane = tab + (E0 & 7) * copy_all + (E0 >> 3) * 16;
  const int b_off = (wy * 32 + r) * SM_PITCH + (wx * 32 + 8 * h) * 2;

  // unwritten image rows / pads must read as finite numbers (they only meet zero taps)
  for (int i = tid; i < (2 * SM_IMG) / 16; i += 512)
    reinterpret_cast<uint4 *>(img_h)[i] = make_uint4(0u, 0u, 0u, 0u);
  load_tile(z0, yb);
  for (int z = z0; z < z1; ++z)
  for (int rr = 0; rr < nr; ++rr) {
    const int y0 = yb + rr * SM_RY;
    // ---- (1) scale of this channel's tile: max |x| -> 2^e with max |y| in [2^14, 2^15)
    float m = 0.f;
#pragma unroll
    for (int q = 0; q < SM_NQ; ++q)
      m = fmaxf(fmaxf(m, fmaxf(fabsf(stage[q].x), fabsf(stage[q].y))),
                fmaxf(fabsf(stage[q].z), fabsf(stage[q].w)));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (tid == 0) *redmax = 0u;
    __syncthreads();  // also: every wave is done with the previous channel's LDS images
    if (lane == 0) atomicMax(redmax, __float_as_uint(m));
    if (rr == 0) {  // taps of this channel (they came with the tile of sub-region 0)
      if (tid < P * P) tapf[tid] = tapreg[0];
      if (tid + 512 < P * P) tapf[tid + 512] = tapreg[1];
    }
    __syncthreads();
    const int ex = (int)((*redmax >> 23) & 0xffu);
    const bool tiny = ex < 40 || ex == 255;
    const float scale = __uint_as_float((unsigned)(tiny ? 127 : 268 - ex) << 23);
    const float inv = __uint_as_float((unsigned)(tiny ? 127 - SM_TAP_LOG2 : ex - 14 - SM_TAP_LOG2)
                                      << 23);
    // ---- (2a) f16 hi / lo images of the tile
#pragma unroll
    for (int q = 0; q < SM_NQ; ++q) {
      const int e = tid + 512 * q;
      const int ry = e / iw4, cx = e - ry * iw4;
      if (ry < ih) {
        const float v[4] = {stage[q].x * scale, stage[q].y * scale, stage[q].z * scale,
                            stage[q].w * scale};
        f16x4 vh, vl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const _Float16 yh = (_Float16)v[j];
          vh[j] = yh;
          vl[j] = (_Float16)(v[j] - (float)yh);
        }
        *reinterpret_cast<f16x4 *>(img_h + ry * SM_PITCH + cx * 8) = vh;
        *reinterpret_cast<f16x4 *>(img_l + ry * SM_PITCH + cx * 8) = vl;
      }
    }
    // ---- (2b) Toeplitz fragment table: G_dy[e] = k[dy][e - 31] (0 outside), copies shifted
    // by 0..7 elements, hi and lo; group (dy, hl, copy, q) holds G_dy[8 q + copy .. + 7]
    if (rr == 0) {
      const float tscale = (float)(1 << SM_TAP_LOG2);
      const int ngroups = P * 8 * SM_GROUPS;
      for (int gidx = tid; gidx < ngroups; gidx += 512) {
        const int q = gidx % SM_GROUPS;
        const int cp = (gidx / SM_GROUPS) & 7;
        const int dy = gidx / (SM_GROUPS * 8);
        f16x8 gh, gl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int d = 8 * q + cp + j - 31;
          // branch-free: all eight LDS reads of a group are in flight together
          const float t0 = tapf[dy * P + min(max(d, 0), P - 1)];
          const float g = (d >= 0 && d < P) ? t0 * tscale : 0.f;
          const _Float16 t = (_Float16)g;
          gh[j] = t;
          gl[j] = (_Float16)(g - (float)t);
        }
        char *dst = tab + cp * copy_all + (dy * 2) * (SM_GROUPS * 16) + q * 16;
        *reinterpret_cast<f16x8 *>(dst) = gh;
        *reinterpret_cast<f16x8 *>(dst + SM_GROUPS * 16) = gl;
      }
    }
    __syncthreads();
    // next tile (same channel, next sub-region; else next channel): in flight during the MFMAs
    if (rr + 1 < nr) load_tile(z, y0 + SM_RY);
    else if (z + 1 < z1) load_tile(z + 1, yb);

    // ---- (3) P x 4 k-steps: acc += Ah Bh + Ah Bl + Al Bh
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    {
      // the 16 fragments of kernel row dy+1 are requested before the 12 MFMAs of row dy: a
      // whole row (>= 384 cycles) of matrix work covers the LDS latency of the next one
      const char *ap = a_lane, *bp = img_h + b_off;
      f16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        ah[ks] = *reinterpret_cast<const f16x8 *>(ap + ks * 32);
        al[ks] = *reinterpret_cast<const f16x8 *>(ap + ks * 32 + SM_GROUPS * 16);
        bh[ks] = *reinterpret_cast<const f16x8 *>(bp + ks * 32);
        bl[ks] = *reinterpret_cast<const f16x8 *>(bp + ks * 32 + SM_IMG);
      }
      for (int dy = 0; dy < P; ++dy) {
        const bool last = dy == P - 1;  // the last row re-reads itself (valid addresses)
        const char *an = last ? ap : ap + 2 * SM_GROUPS * 16;
        const char *bn = last ? bp : bp + SM_PITCH;
        f16x8 nah[4], nal[4], nbh[4], nbl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          nah[ks] = *reinterpret_cast<const f16x8 *>(an + ks * 32);
          nal[ks] = *reinterpret_cast<const f16x8 *>(an + ks * 32 + SM_GROUPS * 16);
          nbh[ks] = *reinterpret_cast<const f16x8 *>(bn + ks * 32);
          nbl[ks] = *reinterpret_cast<const f16x8 *>(bn + ks * 32 + SM_IMG);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh[ks], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl[ks], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh[ks], acc, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ah[ks] = nah[ks], al[ks] = nal[ks], bh[ks] = nbh[ks], bl[ks] = nbl[ks];
        ap = an;
        bp = bn;
      }
    }
    // ---- (4) store: lane (r, h) holds row y0 + 32 wy + r, columns x0 + 32 wx + 8 g + 4 h + 0..3
    const int y = y0 + wy * 32 + r;
    if (y < Ny) {
      float *o = out + (long)z * S + (long)y * Nx + x0 + wx * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int x = x0 + wx * 32 + 8 * g + 4 * h;
        if (x < Nx)  // Nx % 4 == 0: a float4 is inside or outside as a whole
          *reinterpret_cast<float4 *>(o + 8 * g) =
              make_float4(acc[4 * g] * inv, acc[4 * g + 1] * inv, acc[4 * g + 2] * inv,
                          acc[4 * g + 3] * inv);
      }
    }
  }
}

}  // namespace

// 1 if this shape can run on spatial_mfma_kernel
int origin_spatial_mfma_ok(int Ny, int Nx, int P) {
  return P >= 1 && P <= SM_PMAX && (P & 1) && ((P / 2) & 3) == 0 && (Nx & 3) == 0 && Ny >= 1;
}

int origin_spatial_mfma_launch(origin_ctx *ctx, const float *A, const float *taps, int Nz, int Ny,
                               int Nx, int P, float *out) {
  const size_t lds = (size_t)8 * sm_copy_all(P) + 2 * SM_IMG + SM_PMAX * SM_PMAX * sizeof(float) + 64;
  static bool attr_done = false;
  if (!attr_done) {
    ORIGIN_HIP(hipFuncSetAttribute((const void *)spatial_mfma_kernel<true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  // R regions stacked in y share a block (the per-channel fragment table is built once for all
  // of them); more z chunks keep the number of blocks up
  const int ry = cdiv(Ny, SM_RY);
  int R = 1;  // largest divisor of the region count up to 10 (no idle sub-region slots)
  for (int c = 2; c <= 10; ++c)
    if (ry % c == 0) R = c;
  const long groups = (long)cdiv(Nx, SM_RX) * cdiv(ry, R);
  int nzb = (int)(((long)ctx->num_cu * 4 + groups - 1) / groups);
  nzb = std::max(1, std::min(nzb, Nz));
  const int zper = cdiv(Nz, nzb);
  dim3 grid(cdiv(Nx, SM_RX), cdiv(ry, R), cdiv(Nz, zper));
  hipLaunchKernelGGL(spatial_mfma_kernel<true>, grid, dim3(512), lds, ctx->stream, A, taps, Nz, Ny,
                     Nx, P, zper, R, out);
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}
