// GLR matched-filter correlation  (SURVEY.md 2.2 rows k8-k12).
//
// Replaces Correlation_GLR_test (reference muse_origin/lib_origin.py:1070-1217, helpers
// _convolve_fsf :1027-1043 and _convolve_profile :1046-1060) and the dense lines of
// ComputeTGLR.run (steps.py:781-793).  The reference evaluates everything with FFTs; the
// kernels below evaluate the same algebra directly (verified against the
// reference to 1e-15 by oracle/gen_golden.py "direct algebra"):
//
//   cube_fsf[z,y,x] = sum_f sum_{dy,dx} k_fz[dy,dx] (w_f cube)[z, y+dy-c, x+dx-c]
//   norm_fsf[z,y,x] = sum_f sum_{dy,dx} k_fz[dy,dx]^2 w_f[y+dy-c, x+dx-c]
//   num_k[z] = sum_j p_k[j] cube_fsf[z + lw_k - j]
//   den_k[z] = sum_j p_k[j]^2 norm_fsf[z + lw_k - j]
//   T_k = num_k / sqrt(den_k)   (den_k <= 0 -> 0);  correl = max_k, correl_min = min_k,
//   profile = first argmax_k
//
// with k = PSF - mean(PSF), c = P/2 and zeros outside the cube.
//
// Normalisation.  With weights=None, norm_fsf[z,y,x] depends on (y,x) only through how the
// PSF window is clipped by the field border: P*P "border classes" (class (c,c) = interior).
// The plan tabulates 1/sqrt(den_k[z]) per class once, so the hot kernel never touches a
// second cube.  With field weights (or fields smaller than the PSF) norm_fsf is a real
// cube, produced by the same stencil kernel, and den_k is convolved next to num_k.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"
#include "glr_tables.h"

struct origin_glr_plan {
  origin_ctx *ctx;
  int Nz, Ny, Nx, nfields, P, K;
  int mode;   // 0 = border-class table (weights=None), 1 = explicit norm cube
  int lwmax;  // largest profile half width
  float *d_k;     // [F][Nz][P][P]  zero-mean PSF
  float *d_k2;    // [F][Nz][P][P]  its square
  float *d_w;     // [F][Ny][Nx] or null
  float *d_taps;  // concatenated profiles (odd lengths; even ones padded with a 0 tap)
  float *d_taps2; // squares
  int *d_tap_off; // [K+1]
  float *d_rden;  // mode 0: [P*P][K][NzP]  1/sqrt(den) per border class (0 for z >= Nz)
  int Kp;         // z stride of d_rden (= NzP)
  int symmetric;   // every prepared profile is exactly symmetric about its centre
  float *d_htaps;  // symmetric case: half profiles h_k[d] = p_k[lw_k + d], d = 0..lw_k
  int *d_htap_off; // [K+1]
  float *d_rows;   // [K+1][RL] rows (lw, p[0..2 lw]) for spectral3_kernel<LWT>
  float *d_rdi;    // mode 0: interior-class slice of d_rden, [K][NzP] (not owned)
  int *d_border;   // mode 0: flat indices of the spaxels whose border class is not interior
  int nborder;
  int lwt;         // template half width chosen for d_rows (8, 16, 24, 29 or 32; 0 = none)
  int NzP;
  uint4 *d_atab;   // matrix-core spectral stage: shifted hi/lo f16 tap copies (glr_tables.h)
  uint4 *d_atab_bf16;  // the same with bf16 taps (precision 2)
  uint4 *d_atab2;      // the squared taps in the same layout (plans with an explicit norm cube)
  int *d_pwide;    // [K] processing order, narrow first: original index | (half width > 16) << 8
  int n_narrow;    // number of narrow profiles (the first n_narrow slots)
  int order_ident; // the processing order is the caller's order (slot = index)
  float *d_rdi_s;  // interior-class 1/sqrt(den) in processing order [slot][NzP]
  // FOLD (glr_spectral_mfma.hip): taps times a_k = 1/sqrt(sum p_k^2), the table 1/(a_k sqrt(den))
  // [P*P][K][NzP], the class factors s [P*P][NzP]; fold_eps = max |1/(a_k sqrt(den)) / s - 1| over
  // the FOLD channels (the tables are dropped when it exceeds MF_FOLD_EPS)
  uint4 *d_atab_fold, *d_atab_bf16_fold;
  float *d_rden_fold, *d_sden;
  float fold_eps;
  // mode 1 (explicit norm cube): 1 once the first run has measured eps on the norm cube (then
  // fold_eps holds it) -- NORMW runs where it is <= MF_FOLD_EPS
  int normw_checked;
  std::vector<int> *h_order;  // processing order on the host (plan creation only)
  int precision;   // 0 = fp32 FMA kernels, 1 = split-f16 MFMA stages, 2 = bf16 MFMA stages
  float *d_normc;  // mode 1: norm_fsf [Nz][Ny][Nx], a constant of the plan (PSFs and weight maps
                   // only), computed by the first run and kept
  int normc_ready;
  size_t bytes;
};

namespace {

// ------------------------------------------------------------------------------------
// spatial stage: out[z] (+)= corr2(A[z] * B, taps[z]),  zero padded, 'same'
// block (64,4): tile 64 x 16 outputs, each thread 4 rows (ty, ty+4, ty+8, ty+12)
// ------------------------------------------------------------------------------------
constexpr int TX = 64, TY = 16;

__global__ __launch_bounds__(256) void spatial_kernel(const float *__restrict__ A,
                                                      const float *__restrict__ B,
                                                      const float *__restrict__ taps, int Ny,
                                                      int Nx, int P, int accumulate,
                                                      float *__restrict__ out) {
  extern __shared__ float tile[];  // [(TY+P-1)][pitch]
  const int c = P / 2;
  const int pitch = TX + P - 1;
  const int rows = TY + P - 1;
  const int z = blockIdx.z;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
  const long S = (long)Ny * Nx;
  const int tid = threadIdx.y * 64 + threadIdx.x;
  for (int i = tid; i < rows * pitch; i += 256) {
    const int ry = i / pitch, rx = i - ry * pitch;
    const int y = y0 + ry - c, x = x0 + rx - c;
    float v = 0.0f;
    if (y >= 0 && y < Ny && x >= 0 && x < Nx) {
      const long p = (long)y * Nx + x;
      v = A ? A[(long)z * S + p] : 1.0f;
      if (B) v *= B[p];
    }
    tile[i] = v;
  }
  __syncthreads();
  const float *kz = taps + (long)z * P * P;  // uniform -> scalar loads
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int tx = threadIdx.x, ty = threadIdx.y;
  for (int dy = 0; dy < P; ++dy) {
    const float *r0 = tile + (ty + dy) * pitch + tx;
    for (int dx = 0; dx < P; ++dx) {
      const float kv = kz[dy * P + dx];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = fmaf(kv, r0[(4 * r) * pitch + dx], acc[r]);
    }
  }
  const int x = x0 + tx;
  if (x < Nx) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = y0 + ty + 4 * r;
      if (y < Ny) {
        const long o = (long)z * S + (long)y * Nx + x;
        out[o] = accumulate ? out[o] + acc[r] : acc[r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// spatial stage, register-tiled: block 256 threads = 16 x 16, every thread owns a 4 x 4
// patch of outputs (tile 64 x 64).  For each of the 4+P-1 input rows of its patch a thread
// reads the row segment it needs once from LDS (ds_read_b128, conflict free because the
// pitch is a multiple of 16 floats) and feeds up to 4 output rows x P taps x 4 columns of
// FMAs from registers; the taps are wave-uniform and come from scalar loads.
// ------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int P, bool VEC, bool HAS_B>
__global__ __launch_bounds__(256) void spatial4x4_kernel(const float *__restrict__ A,
                                                         const float *__restrict__ B,
                                                         const float *__restrict__ taps, int Nz,
                                                         int Ny, int Nx, int zper, int accumulate,
                                                         float *__restrict__ out) {
  constexpr int H = P - 1;
  constexpr int W = 64 + H;                  // tile width in floats (multiple of 4: P odd)
  constexpr int W4 = (W + 3) / 4;            // float4 per tile row
  constexpr int PITCH = (W + 15) / 16 * 16;  // multiple of 16 floats: conflict-free b128 reads
  constexpr int ROWS = 64 + H;
  constexpr int NV = (4 + H + 3) / 4;        // float4 per row segment a thread consumes
  constexpr int RPT = 256 / W4;               // tile rows staged per pass by the block
  constexpr int NQ = (ROWS + RPT - 1) / RPT;  // staged float4 per thread
  static_assert(60 + 4 * NV <= PITCH && 4 * W4 <= PITCH, "row segment exceeds the LDS pitch");
  __shared__ __attribute__((aligned(16))) float tile[ROWS * PITCH];
  constexpr int c = P / 2;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
  const long S = (long)Ny * Nx;
  const int tid = threadIdx.x;
  const int z0 = blockIdx.z * zper, z1 = min(Nz, z0 + zper);

  // Register staging of the next plane's tile (issue early, write to LDS late).  Thread
  // (sr, sc4) stages the float4 column sc4 of tile rows sr, sr+RPT, sr+2 RPT, ...
  const int sr = tid / W4, sc4 = tid - sr * W4;
  const bool stager = sr < RPT;
  const int sx = x0 - c + 4 * sc4;  // first field column of the staged float4
  // VEC: Nx % 4 == 0 and sx % 4 == 0, so a float4 is either fully inside or fully outside
  const bool xin = sx >= 0 && sx + 3 < Nx;
  float4 stage[NQ];
  auto load_tile = [&](int z) {
    const float *Az = A + (long)z * S;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int ry = sr + RPT * j;
      const int y = y0 - c + ry;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (stager && ry < ROWS && y >= 0 && y < Ny) {
        const long p = (long)y * Nx + sx;
        if constexpr (VEC) {
          if (xin) {
            v = *reinterpret_cast<const float4 *>(Az + p);
            if constexpr (HAS_B) {
              const float4 w = *reinterpret_cast<const float4 *>(B + p);
              v.x *= w.x, v.y *= w.y, v.z *= w.z, v.w *= w.w;
            }
          }
        } else {
          float e[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int xx = sx + t;
            float u = 0.f;
            if (xx >= 0 && xx < Nx) {
              u = Az[p + t];
              if constexpr (HAS_B) u *= B[p + t];
            }
            e[t] = u;
          }
          v = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
      stage[j] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int ry = sr + RPT * j;
      if (stager && ry < ROWS)
        *reinterpret_cast<float4 *>(tile + ry * PITCH + 4 * sc4) = stage[j];
    }
  };

  const int tx = tid & 15, ty = tid >> 4;
  load_tile(z0);
  for (int z = z0; z < z1; ++z) {
    __syncthreads();  // every wave is done reading the previous tile
    store_tile();
    __syncthreads();
    if (z + 1 < z1) load_tile(z + 1);  // in flight while this plane is computed
    const float *kz = taps + (long)z * P * P;  // uniform -> scalar loads
    // Full-rate fp32 on gfx950 needs v_pk_fma_f32, whose 64-bit operands are even-aligned
    // register pairs: keep the row segment twice, as pairs starting at even (rowE) and at
    // odd (rowO) columns, so that every (column, column+1) pair is a ready-made operand.
    f32x2 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a][0] = acc[a][1] = (f32x2){0.f, 0.f};
#pragma unroll 1
    for (int i = 0; i < 4 + H; ++i) {
      const float *rbase = tile + (4 * ty + i) * PITCH + 4 * tx;
      const float4 *rp = reinterpret_cast<const float4 *>(rbase);
      f32x2 rowE[2 * NV], rowO[2 * NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const float4 v = rp[q];
        rowE[2 * q] = (f32x2){v.x, v.y};
        rowE[2 * q + 1] = (f32x2){v.z, v.w};
      }
      // odd-aligned pairs (2q+1, 2q+2) are assembled from neighbouring even pairs in
      // registers (one v_pk_mov_b32 each): reading them from LDS with ds_read2_b32 costs
      // 8-way bank conflicts and made the LDS, not the VALU, the bottleneck
#pragma unroll
      for (int q = 0; q + 1 < 2 * NV; ++q)
        rowO[q] = __builtin_shufflevector(rowE[q], rowE[q + 1], 1, 2);
      rowO[2 * NV - 1] = (f32x2){rowE[2 * NV - 1].y, 0.0f};
      // output rows are processed in pairs so that four independent accumulators are in
      // flight (a packed FMA then never waits for the previous one on the same register)
      auto single = [&](int ry) {
        const float *kr = kz + (i - ry) * P;
#pragma unroll
        for (int dx = 0; dx < P; ++dx) {
          const float kv = kr[dx];
          const f32x2 k2 = (f32x2){kv, kv};
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int j = 2 * h + dx;  // first column of the pair
            const f32x2 in = (j & 1) ? rowO[j >> 1] : rowE[j >> 1];
            acc[ry][h] = __builtin_elementwise_fma(k2, in, acc[ry][h]);
          }
        }
      };
#pragma unroll
      for (int rp = 0; rp < 4; rp += 2) {
        const int dya = i - rp, dyb = i - rp - 1;
        const bool va = dya >= 0 && dya < P, vb = dyb >= 0 && dyb < P;  // wave-uniform
        if (va && vb) {
          const float *ka = kz + dya * P, *kb = kz + dyb * P;
#pragma unroll
          for (int dx = 0; dx < P; ++dx) {
            const f32x2 a2 = (f32x2){ka[dx], ka[dx]}, b2 = (f32x2){kb[dx], kb[dx]};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int j = 2 * h + dx;
              const f32x2 in = (j & 1) ? rowO[j >> 1] : rowE[j >> 1];
              acc[rp][h] = __builtin_elementwise_fma(a2, in, acc[rp][h]);
              acc[rp + 1][h] = __builtin_elementwise_fma(b2, in, acc[rp + 1][h]);
            }
          }
        } else {
          if (va) single(rp);
          if (vb) single(rp + 1);
        }
      }
    }
    const int xo = x0 + 4 * tx;
#pragma unroll
    for (int ry = 0; ry < 4; ++ry) {
      const int y = y0 + 4 * ty + ry;
      if (y >= Ny) continue;
      float *o = out + (long)z * S + (long)y * Nx + xo;
      if (VEC && xo + 3 < Nx) {
        float4 v = make_float4(acc[ry][0].x, acc[ry][0].y, acc[ry][1].x, acc[ry][1].y);
        if (accumulate) {
          const float4 old = *reinterpret_cast<float4 *>(o);
          v.x += old.x, v.y += old.y, v.z += old.z, v.w += old.w;
        }
        *reinterpret_cast<float4 *>(o) = v;
      } else {
        const float r4[4] = {acc[ry][0].x, acc[ry][0].y, acc[ry][1].x, acc[ry][1].y};
#pragma unroll
        for (int rx = 0; rx < 4; ++rx)
          if (xo + rx < Nx) o[rx] = accumulate ? o[rx] + r4[rx] : r4[rx];
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// border-class normalisation tables (mode 0)
// ------------------------------------------------------------------------------------
// normcls[z][cy][cx] = sum over the in-field part of the window of k_z^2
__global__ __launch_bounds__(256) void norm_classes_kernel(const float *__restrict__ k2, int Nz,
                                                           int P, double *__restrict__ ncls) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = (long)Nz * P * P;
  if (i >= n) return;
  const int z = (int)(i / (P * P));
  const int cls = (int)(i - (long)z * P * P);
  const int cy = cls / P, cx = cls - cy * P;
  const int c = P / 2;
  // class id t <-> window clipped to dy in [max(0,c-t), min(P-1, P-1+c-t)]
  const int dy0 = max(0, c - cy), dy1 = min(P - 1, P - 1 + c - cy);
  const int dx0 = max(0, c - cx), dx1 = min(P - 1, P - 1 + c - cx);
  const float *kz = k2 + (long)z * P * P;
  double acc = 0.0;
  for (int dy = dy0; dy <= dy1; ++dy)
    for (int dx = dx0; dx <= dx1; ++dx) acc += (double)kz[dy * P + dx];
  ncls[i] = acc;
}

// eps of the FOLD form on an explicit norm cube (NORMW): over every voxel of the channels
// [zf0, zf1) and every profile, |sqrt(den_k / (norm sum_j p_k[j]^2)) - 1| with den_k the true
// smoothed norm (lib_origin.py:1055: conv of norm_fsf with p_k^2); a spaxel no field covers has
// norm = 0 and den = 0 in both forms.  norm: channel 0 of the padded cube.  A thread takes
// NE_ZT consecutive channels of one spaxel (their common window in registers), the squared taps
// sit in LDS as dense 65-slot rows (slot u = channel offset u - 32); float bits through atomicMax
// (values >= 0).  Runs once per plan, in its first run: ~0.1 s at 3681 x 600 x 600.
constexpr int NE_ZT = 4;
__global__ __launch_bounds__(256) void normw_eps_kernel(const float *__restrict__ norm,
                                                        const float *__restrict__ taps2,
                                                        const int *__restrict__ tap_off, int K,
                                                        long S, int zf0, int zf1,
                                                        unsigned *__restrict__ eps_bits) {
  __shared__ float tt[MF_MAX_K][65];
  __shared__ float ts2[MF_MAX_K];
  for (int i = threadIdx.x; i < K * 65; i += 256) {
    const int k = i / 65, u = i - 65 * k;
    const int o = tap_off[k], L = tap_off[k + 1] - o, lw = (L - 1) / 2;
    const int j = 32 + lw - u;  // window slot u = channel z - 32 + u = z + lw - j
    tt[k][u] = (j >= 0 && j < L) ? taps2[o + j] : 0.0f;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    float a = 0.0f;
    for (int u = 0; u < 65; ++u) a += tt[threadIdx.x][u];
    ts2[threadIdx.x] = a;
  }
  __syncthreads();
  const long sp = (long)blockIdx.x * 256 + threadIdx.x;
  const int z = zf0 + NE_ZT * (int)blockIdx.y;
  float eps = 0.0f;
  if (sp < S) {
    float w[64 + NE_ZT];  // channels z - 32 .. z + NE_ZT + 31 (the pads of the cube cover the ends)
#pragma unroll
    for (int j = 0; j < 64 + NE_ZT; ++j) w[j] = norm[(long)(z - 32 + j) * S + sp];
    for (int k = 0; k < K; ++k) {
      float den[NE_ZT];
#pragma unroll
      for (int c = 0; c < NE_ZT; ++c) den[c] = 0.0f;
#pragma unroll
      for (int u = 0; u < 65; ++u) {
        const float t = tt[k][u];
#pragma unroll
        for (int c = 0; c < NE_ZT; ++c) den[c] += t * w[u + c];
      }
#pragma unroll
      for (int c = 0; c < NE_ZT; ++c) {
        if (z + c >= zf1) continue;
        const float ref = w[32 + c] * ts2[k];
        const float e = ref > 0.0f ? fabsf(sqrtf(den[c] / ref) - 1.0f)
                                   : (den[c] > 0.0f ? INFINITY : 0.0f);
        eps = fmaxf(eps, e);
      }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) eps = fmaxf(eps, __shfl_xor(eps, o));
  if ((threadIdx.x & 63) == 0 && eps > 0.0f) atomicMax(eps_bits, __float_as_uint(eps));
}

// FOLD tables of the matrix-core spectral stage (glr_spectral_mfma.hip): rden_fold = rden / a_k,
// s[cls][z] = the middle of its range over k, eps = the largest half width of that range relative
// to s over the FOLD channels [zf0, zf1) (float bits, atomicMax: every value is >= 0)
__global__ __launch_bounds__(256) void fold_tables_kernel(const float *__restrict__ rden,
                                                          const float *__restrict__ ainv, int K,
                                                          int PP, int NzP, int zf0, int zf1,
                                                          float *__restrict__ rden_fold,
                                                          float *__restrict__ sden,
                                                          unsigned *__restrict__ eps_bits) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)PP * NzP) return;
  const int cls = (int)(i / NzP), z = (int)(i % NzP);
  float lo = INFINITY, hi = 0.0f;
  for (int k = 0; k < K; ++k) {
    const long j = ((long)cls * K + k) * NzP + z;
    const float v = rden[j] * ainv[k];
    rden_fold[j] = v;
    lo = fminf(lo, v), hi = fmaxf(hi, v);
  }
  sden[i] = 0.5f * (lo + hi);
  if (z >= zf0 && z < zf1) {
    const float eps = lo > 0.0f ? (hi - lo) / (hi + lo) : INFINITY;
    atomicMax(eps_bits, __float_as_uint(eps));
  }
}

// rden[cls][k][z] = 1/sqrt(sum_j p_k[j]^2 normcls[z + lw - j][cls])   (0 if den <= 0 or z >= Nz)
// z is the fastest axis (stride NzP, a multiple of 32): a lane of the matrix-core kernel
// fetches the four consecutive channels of an accumulator group with one 16-byte load.
__global__ __launch_bounds__(256) void rden_kernel(const double *__restrict__ ncls,
                                                   const float *__restrict__ taps2,
                                                   const int *__restrict__ tap_off, int K, int Nz,
                                                   int PP, int NzP, float *__restrict__ rden) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = (long)PP * K * NzP;
  if (i >= n) return;
  const int z = (int)(i % NzP);
  const int k = (int)((i / NzP) % K);
  const int cls = (int)(i / ((long)NzP * K));
  if (z >= Nz) {
    rden[i] = 0.0f;
    return;
  }
  const int off = tap_off[k], L = tap_off[k + 1] - off, lw = (L - 1) / 2;
  double den = 0.0;
  for (int j = 0; j < L; ++j) {
    const int zz = z + lw - j;
    if (zz >= 0 && zz < Nz) den += (double)taps2[off + j] * ncls[(long)zz * PP + cls];
  }
  rden[i] = den > 0.0 ? (float)(1.0 / sqrt(den)) : 0.0f;
}

// ------------------------------------------------------------------------------------
// spectral stage.  One lane per spaxel, marching z with a register window of the last
// 2*LWMAX+1 channels; the taps of a profile are wave-uniform (scalar loads), the window
// index of every FMA is a compile-time constant (switch on the half width).
// ------------------------------------------------------------------------------------
template <int LWMAX, int LW>
__device__ __forceinline__ float conv_lw(const float (&w)[2 * LWMAX + 1],
                                         const float *__restrict__ p) {
  float acc = 0.0f;
#pragma unroll
  for (int j = 0; j <= 2 * LW; ++j) acc = fmaf(p[j], w[LWMAX + LW - j], acc);
  return acc;
}

#define CASE_LW(N)                   \
  case N:                            \
    if constexpr (N <= LWMAX) return conv_lw<LWMAX, (N <= LWMAX ? N : 0)>(w, p); \
    break;

template <int LWMAX>
__device__ __forceinline__ float conv_sel(const float (&w)[2 * LWMAX + 1],
                                          const float *__restrict__ p, int lw) {
  switch (lw) {
    CASE_LW(0) CASE_LW(1) CASE_LW(2) CASE_LW(3) CASE_LW(4) CASE_LW(5) CASE_LW(6) CASE_LW(7)
    CASE_LW(8) CASE_LW(9) CASE_LW(10) CASE_LW(11) CASE_LW(12) CASE_LW(13) CASE_LW(14)
    CASE_LW(15) CASE_LW(16) CASE_LW(17) CASE_LW(18) CASE_LW(19) CASE_LW(20) CASE_LW(21)
    CASE_LW(22) CASE_LW(23) CASE_LW(24) CASE_LW(25) CASE_LW(26) CASE_LW(27) CASE_LW(28)
    CASE_LW(29) CASE_LW(30) CASE_LW(31) CASE_LW(32)
    default:
      break;
  }
  return 0.0f;
}
#undef CASE_LW

__device__ __forceinline__ int border_class(int t, int N, int P) {
  const int c = P / 2;
  return t < c ? t : (t > N - 1 - c ? P - 1 - (N - 1 - t) : c);
}

template <int LWMAX, bool GENERAL>
__global__ __launch_bounds__(256) void spectral_kernel(
    const float *__restrict__ fsf, const float *__restrict__ norm,
    const float *__restrict__ rden, const float *__restrict__ taps,
    const float *__restrict__ taps2, const int *__restrict__ tap_off, int K, int Kp, int Nz,
    int Ny, int Nx, int P, int zchunk, const uint8_t *__restrict__ mask,
    float *__restrict__ correl,
    uint8_t *__restrict__ profile, float *__restrict__ correl_min,
    float *__restrict__ part_max, float *__restrict__ part_min,
    const int *__restrict__ list, int nlist) {
  // with `list` the kernel only (re)computes the listed spaxels (border fix-up pass)
  constexpr int W = 2 * LWMAX + 1;
  const long S = (long)Ny * Nx;
  const long i0 = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = list ? i0 < nlist : i0 < S;
  const long s = list ? (long)list[live ? i0 : 0] : i0;
  const long sc = live ? s : S - 1;
  const int z0 = blockIdx.y * zchunk;
  const int z1 = min(Nz, z0 + zchunk);

  const float *rd = nullptr;
  if constexpr (!GENERAL) {
    const int y = (int)(sc / Nx), x = (int)(sc - (long)y * Nx);
    const int cls = border_class(y, Ny, P) * P + border_class(x, Nx, P);
    rd = rden + (long)cls * K * Kp;  // Kp: z stride of the table
  }

  float w[W];
  float wn[GENERAL ? W : 1];
#pragma unroll
  for (int i = 0; i < W; ++i) {
    const int zz = z0 - LWMAX + i;
    const bool in = zz >= 0 && zz < Nz;
    w[i] = in ? fsf[(long)zz * S + sc] : 0.0f;
    if constexpr (GENERAL) wn[i] = in ? norm[(long)zz * S + sc] : 0.0f;
  }

  float vmax = -INFINITY, vmin = INFINITY;
  for (int z = z0; z < z1; ++z) {
    float best = -INFINITY, worst = INFINITY;
    int bk = 0;
    for (int k = 0; k < K; ++k) {
      const int off = tap_off[k];
      const int lw = (tap_off[k + 1] - off - 1) >> 1;
      const float num = conv_sel<LWMAX>(w, taps + off, lw);
      float T;
      if constexpr (GENERAL) {
        const float den = conv_sel<LWMAX>(wn, taps2 + off, lw);
        T = den > 0.0f ? num / sqrtf(den) : 0.0f;  // den <= 0 -> inf -> T = 0  (lib :1057)
      } else {
        T = num * rd[(long)k * Kp + z];
      }
      if (T > best) {  // strict '>' : first maximum wins                      (lib :1210)
        best = T;
        bk = k;
      }
      worst = fminf(worst, T);
    }
    const long idx = (long)z * S + sc;
    if (mask && mask[idx]) {  // correl[mask] = 0 ; profile[mask] = 0   (steps.py:781,788)
      best = 0.0f;
      bk = 0;
    }
    if (live) {
      correl[idx] = best;
      profile[idx] = (uint8_t)bk;
      correl_min[idx] = worst;
    }
    vmax = fmaxf(vmax, best);
    vmin = fminf(vmin, worst);
    // slide the window by one channel
#pragma unroll
    for (int i = 0; i < W - 1; ++i) {
      w[i] = w[i + 1];
      if constexpr (GENERAL) wn[i] = wn[i + 1];
    }
    const int zn = z + 1 + LWMAX;
    const bool in = zn < Nz;
    w[W - 1] = in ? fsf[(long)zn * S + sc] : 0.0f;
    if constexpr (GENERAL) wn[W - 1] = in ? norm[(long)zn * S + sc] : 0.0f;
  }
  if (live && part_max) {
    part_max[(long)blockIdx.y * S + s] = vmax;
    part_min[(long)blockIdx.y * S + s] = vmin;
  }
}

// ------------------------------------------------------------------------------------
// spectral stage, packed and z-blocked.  One lane owns TWO adjacent spaxels (float2 loads,
// v_pk_fma_f32: the only way to the full fp32 rate on gfx950) and produces SPEC_ZC
// consecutive channels per step from a register window of 2*LWMAX + SPEC_ZC float2.
// Profiles are the OUTER loop of a step: the constants of profile k come as one fixed-length
// row  [lw_k (int bits), p_k[0], ..., p_k[2 lw_k], 0 ...]  fetched by a few wide scalar loads
// and then feed SPEC_ZC * (2 lw_k + 1) packed FMAs, so the scalar-load latency is amortised
// over hundreds of cycles of arithmetic.  The kernel normalises EVERY spaxel with the
// interior-class 1/sqrt(den) (wave-uniform, scalar loads); the spaxels within P/2 of the field
// border, whose normalisation differs, are recomputed afterwards by spectral_kernel on the
// plan's border list (a few percent of the field).
// ------------------------------------------------------------------------------------
constexpr int SPEC_ZC = 4;

template <int LWMAX, int LW>
__device__ __forceinline__ void conv3_one(const f32x2 (&w)[2 * LWMAX + SPEC_ZC],
                                          const float *__restrict__ taps,  // wave-uniform
                                          f32x2 (&num)[SPEC_ZC]) {
  constexpr int NT = 2 * LW + 1;         // taps of this profile
  constexpr int NCH = (NT + 15) / 16;    // chunks of 16 scalars
#pragma unroll
  for (int o = 0; o < SPEC_ZC; ++o) num[o] = (f32x2){0.f, 0.f};
  // Taps are consumed in chunks of 16 scalars from two alternating SGPR sets: the chunk
  // c+1 is requested right after chunk c has arrived and before the 64 packed FMAs of chunk c
  // are issued, so the scalar-load latency hides behind them.  Scalar loads return out of
  // order, hence the explicit lgkmcnt(0) / sched_barrier fences that pin this order.
  float ta[16], tb[16];
  auto load = [&](float (&t)[16], int c) {
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = taps[16 * c + i];
  };
  auto fmas = [&](const float (&t)[16], int c, int i0, int i1) {
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      const int j = 16 * c + i;
      if (j < NT) {
        const f32x2 pj = (f32x2){t[i], t[i]};
#pragma unroll
        for (int o = 0; o < SPEC_ZC; ++o)
          num[o] = __builtin_elementwise_fma(pj, w[LWMAX + o + LW - j], num[o]);
      }
    }
  };
  // hipcc's own waitcnt insertion puts lgkmcnt(0) in front of the first use of a chunk, so
  // the first tap of chunk c is consumed BEFORE chunk c+1 is requested: the wait then covers
  // only chunk c, and the request for c+1 flies during the remaining 15 x SPEC_ZC FMAs.
  load(ta, 0);
#pragma unroll
  for (int c = 0; c < NCH; c += 2) {
    fmas(ta, c, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < NCH) load(tb, c + 1);
    __builtin_amdgcn_sched_barrier(0);
    fmas(ta, c, 1, 16);
    if (c + 1 < NCH) {
      __builtin_amdgcn_sched_barrier(0);
      fmas(tb, c + 1, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 2 < NCH) load(ta, c + 2);
      __builtin_amdgcn_sched_barrier(0);
      fmas(tb, c + 1, 1, 16);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

#define CASE3(N)                                                              \
  case N:                                                                     \
    if constexpr (N <= LWMAX) conv3_one<LWMAX, (N <= LWMAX ? N : 0)>(w, taps, num); \
    break;

template <int LWMAX>
__device__ __forceinline__ void conv3_sel(const f32x2 (&w)[2 * LWMAX + SPEC_ZC],
                                          const float *__restrict__ taps, int lw,
                                          f32x2 (&num)[SPEC_ZC]) {
  switch (lw) {
    CASE3(0) CASE3(1) CASE3(2) CASE3(3) CASE3(4) CASE3(5) CASE3(6) CASE3(7) CASE3(8) CASE3(9)
    CASE3(10) CASE3(11) CASE3(12) CASE3(13) CASE3(14) CASE3(15) CASE3(16) CASE3(17) CASE3(18)
    CASE3(19) CASE3(20) CASE3(21) CASE3(22) CASE3(23) CASE3(24) CASE3(25) CASE3(26) CASE3(27)
    CASE3(28) CASE3(29) CASE3(30) CASE3(31) CASE3(32)
    default:
#pragma unroll
      for (int o = 0; o < SPEC_ZC; ++o) num[o] = (f32x2){0.f, 0.f};
      break;
  }
}
#undef CASE3

template <int LWMAX>
__global__ __launch_bounds__(256) void spectral3_kernel(
    const float *__restrict__ fsf, const float *__restrict__ rdi, int NzP,
    const float *__restrict__ rows, int K, int Nz, int Ny, int Nx, int zchunk,
    const uint8_t *__restrict__ mask, float *__restrict__ correl,
    uint8_t *__restrict__ profile, float *__restrict__ correl_min, float *__restrict__ part_max,
    float *__restrict__ part_min) {
  constexpr int ZC = SPEC_ZC;
  constexpr int RL = (2 * LWMAX + 1 + 15) / 16 * 16 + 16;
  constexpr int W = 2 * LWMAX + ZC;
  const long S = (long)Ny * Nx;  // even
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = 2 * t < S;
  const long s0 = live ? 2 * t : S - 2;
  const int z0 = blockIdx.y * zchunk;  // multiple of ZC
  const int z1 = min(Nz, z0 + zchunk);

  auto load2 = [&](int zz) -> f32x2 {
    if (zz < 0 || zz >= Nz) return (f32x2){0.f, 0.f};
    return *reinterpret_cast<const f32x2 *>(fsf + (long)zz * S + s0);
  };

  f32x2 w[W];
#pragma unroll
  for (int i = 0; i < W; ++i) w[i] = load2(z0 - LWMAX + i);

  f32x2 vmax = (f32x2){-INFINITY, -INFINITY}, vmin = (f32x2){INFINITY, INFINITY};
  for (int zb = z0; zb < z1; zb += ZC) {
    // request the planes that enter the window at the end of this step now: the loads have
    // the whole step (K profiles) to land
    f32x2 incoming[ZC];
#pragma unroll
    for (int o = 0; o < ZC; ++o) incoming[o] = load2(zb + ZC + LWMAX + o);
    unsigned short mk[ZC];  // mask bytes of the two spaxels, also requested a step ahead
#pragma unroll
    for (int o = 0; o < ZC; ++o)
      mk[o] = (mask && zb + o < z1)
                  ? *reinterpret_cast<const unsigned short *>(mask + (long)(zb + o) * S + s0)
                  : (unsigned short)0;
    f32x2 best[ZC], worst[ZC];
    int bk0[ZC], bk1[ZC];
#pragma unroll
    for (int o = 0; o < ZC; ++o) {
      best[o] = (f32x2){-INFINITY, -INFINITY};
      worst[o] = (f32x2){INFINITY, INFINITY};
      bk0[o] = bk1[o] = 0;
    }
    int lw_next = __float_as_int(rows[0]);
    for (int k = 0; k < K; ++k) {
      const float *rk = rows + (long)k * RL;  // wave-uniform -> wide scalar loads
      const int lw = lw_next;
      lw_next = __float_as_int(rk[RL]);  // rows has K+1 entries; used by the next iteration
      float rdu[ZC];
      {
        const float *ru = rdi + (long)k * NzP + zb;  // NzP >= Nz + ZC: no bound check
#pragma unroll
        for (int o = 0; o < ZC; ++o) rdu[o] = ru[o];
      }
      f32x2 num[ZC];
      conv3_sel<LWMAX>(w, rk + 1, lw, num);
#pragma unroll
      for (int o = 0; o < ZC; ++o) {
        const f32x2 T = num[o] * (f32x2){rdu[o], rdu[o]};
        // strict '>' : the first maximum wins                             (lib :1210)
        if (T.x > best[o].x) best[o].x = T.x, bk0[o] = k;
        if (T.y > best[o].y) best[o].y = T.y, bk1[o] = k;
        worst[o].x = fminf(worst[o].x, T.x);
        worst[o].y = fminf(worst[o].y, T.y);
      }
    }
#pragma unroll
    for (int o = 0; o < ZC; ++o) {
      const int zz = zb + o;
      if (zz < z1) {
        const long idx = (long)zz * S + s0;
        f32x2 b = best[o];
        int k0 = bk0[o], k1 = bk1[o];
        // correl[mask] = 0 ; profile[mask] = 0                        (steps.py:781,788)
        if (mk[o] & 0x00ff) b.x = 0.0f, k0 = 0;
        if (mk[o] & 0xff00) b.y = 0.0f, k1 = 0;
        if (live) {
          *reinterpret_cast<f32x2 *>(correl + idx) = b;
          *reinterpret_cast<f32x2 *>(correl_min + idx) = worst[o];
          *reinterpret_cast<unsigned short *>(profile + idx) = (unsigned short)(k0 | (k1 << 8));
        }
        vmax.x = fmaxf(vmax.x, b.x), vmax.y = fmaxf(vmax.y, b.y);
        vmin.x = fminf(vmin.x, worst[o].x), vmin.y = fminf(vmin.y, worst[o].y);
      }
    }
    // slide the window by ZC channels
#pragma unroll
    for (int i = 0; i < W - ZC; ++i) w[i] = w[i + ZC];
#pragma unroll
    for (int o = 0; o < ZC; ++o) w[W - ZC + o] = incoming[o];
  }
  if (live && part_max) {
    *reinterpret_cast<f32x2 *>(part_max + (long)blockIdx.y * S + s0) = vmax;
    *reinterpret_cast<f32x2 *>(part_min + (long)blockIdx.y * S + s0) = vmin;
  }
}

// ------------------------------------------------------------------------------------
// spectral stage on the matrix cores: operand layout (the kernel is glr_spectral_mfma.hip).
//
// num_k[z] = sum_j p_k[j] x[z + lw_k - j] is a banded Toeplitz product: for a tile of 32 output
// channels z0..z0+31 and the 96-channel window x[z0-32 .. z0+63],
//     num_k[z0+m, s] = sum_{i=0}^{95} A_k[m][i] X[i][s],   A_k[m][i] = p_k[m + lw_k + 32 - i]
// (zero outside the band), i.e. a [32 x 96] x [96 x N] GEMM per profile whose B operand -- the
// data -- is shared by all K profiles.  fp32 MFMA runs at the vector rate, so the product is
// evaluated with v_mfma_f32_32x32x16_f16 (16x that rate) on a two-term split of both
// operands: y = x * 2^e (e per tile, from the tile's max |x|, so any input range is safe and
// scaling the cube by a power of two scales the result exactly), y = yh + yl with yh = f16(y),
// yl = f16(y - yh) -- 22 significant bits -- likewise the taps, and
//     num = Ah Bh + Ah Bl + Al Bh      (the dropped Al Bl term is 2^-22 relative)
// accumulated in fp32 by the matrix core.  Error vs float64: ~3e-7 of sum |p x|, the same
// order as the fp32 FMA chain of spectral3_kernel (~1e-7); origin_glr_plan_set_precision
// selects that kernel instead.
//
// A wave owns 32 consecutive spaxels (one 32-column B tile, fragments loaded straight from
// global memory: lane (r, h) holds X[16 ks + 8 h + j][r], 128-byte segments per half wave) and
// marches z in tiles of 32.  The A fragments of a Toeplitz matrix are 8 consecutive entries of
// one padded tap array G_k[e] = p_k[lw_k + 63 - e] starting at e = 16 ks + 8 h - m + 31; LDS
// holds, per profile, 8 copies of G_k shifted by 0..7 elements (hi and lo halves, 320-byte
// copies: conflict-free for the lane groups of ds_read_b128) so that every fragment is ONE
// aligned ds_read_b128 at a per-lane base plus an immediate offset.  Profiles whose half width
// is <= 16 only touch window blocks 1..4 (4 of the 6 k-steps).
// Normalisation: 1/sqrt(den) of the lane's border class, exact for every spaxel (no fix-up
// pass behind this kernel).
// ------------------------------------------------------------------------------------
// fallback for profiles wider than the register window: plain loops over global memory
template <bool GENERAL>
__global__ __launch_bounds__(256) void spectral_generic_kernel(
    const float *__restrict__ fsf, const float *__restrict__ norm,
    const float *__restrict__ rden, const float *__restrict__ taps,
    const float *__restrict__ taps2, const int *__restrict__ tap_off, int K, int Kp, int Nz,
    int Ny, int Nx, int P, int zchunk, const uint8_t *__restrict__ mask,
    float *__restrict__ correl,
    uint8_t *__restrict__ profile, float *__restrict__ correl_min,
    float *__restrict__ part_max, float *__restrict__ part_min,
    const int *__restrict__ list, int nlist) {
  const long S = (long)Ny * Nx;
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  (void)list;
  (void)nlist;
  if (s >= S) return;
  const int z0 = blockIdx.y * zchunk;
  const int z1 = min(Nz, z0 + zchunk);
  const float *rd = nullptr;
  if constexpr (!GENERAL) {
    const int y = (int)(s / Nx), x = (int)(s - (long)y * Nx);
    rd = rden + (long)(border_class(y, Ny, P) * P + border_class(x, Nx, P)) * K * Kp;
  }
  float vmax = -INFINITY, vmin = INFINITY;
  for (int z = z0; z < z1; ++z) {
    float best = -INFINITY, worst = INFINITY;
    int bk = 0;
    for (int k = 0; k < K; ++k) {
      const int off = tap_off[k], L = tap_off[k + 1] - off, lw = (L - 1) >> 1;
      float num = 0.0f, den = 0.0f;
      for (int j = 0; j < L; ++j) {
        const int zz = z + lw - j;
        if (zz >= 0 && zz < Nz) {
          num = fmaf(taps[off + j], fsf[(long)zz * S + s], num);
          if constexpr (GENERAL) den = fmaf(taps2[off + j], norm[(long)zz * S + s], den);
        }
      }
      float T;
      if constexpr (GENERAL)
        T = den > 0.0f ? num / sqrtf(den) : 0.0f;
      else
        T = num * rd[(long)k * Kp + z];
      if (T > best) {
        best = T;
        bk = k;
      }
      worst = fminf(worst, T);
    }
    const long idx = (long)z * S + s;
    if (mask && mask[idx]) {
      best = 0.0f;
      bk = 0;
    }
    correl[idx] = best;
    profile[idx] = (uint8_t)bk;
    correl_min[idx] = worst;
    vmax = fmaxf(vmax, best);
    vmin = fminf(vmin, worst);
  }
  if (part_max) {
    part_max[(long)blockIdx.y * S + s] = vmax;
    part_min[(long)blockIdx.y * S + s] = vmin;
  }
}

__global__ __launch_bounds__(256) void maxmap_final_kernel(const float *__restrict__ part_max,
                                                           const float *__restrict__ part_min,
                                                           int nzc, long S,
                                                           float *__restrict__ maxmap,
                                                           float *__restrict__ minmap) {
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  float a = -INFINITY, b = INFINITY;
  for (int k = 0; k < nzc; ++k) {
    a = fmaxf(a, part_max[(long)k * S + s]);
    b = fminf(b, part_min[(long)k * S + s]);
  }
  if (maxmap) maxmap[s] = a;
  if (minmap) minmap[s] = b;
}

// maxmap / minmap of the listed spaxels straight from the final cubes (border fix-up).
// Lanes run over list entries (border rows are contiguous in memory), z is cut in slices
// whose partial extrema are merged with ordered-int atomics (max/min are order independent,
// so the result is deterministic).
__device__ __forceinline__ void atomic_max_f(float *addr, float v) {
  if (v >= 0.0f)
    atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
  else
    atomicMin(reinterpret_cast<unsigned *>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_min_f(float *addr, float v) {
  if (v >= 0.0f)
    atomicMin(reinterpret_cast<int *>(addr), __float_as_int(v));
  else
    atomicMax(reinterpret_cast<unsigned *>(addr), __float_as_uint(v));
}

__global__ __launch_bounds__(256) void list_maps_init_kernel(const int *__restrict__ list, int nlist,
                                                             float *__restrict__ maxmap,
                                                             float *__restrict__ minmap) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nlist) return;
  if (maxmap) maxmap[list[i]] = -INFINITY;
  if (minmap) minmap[list[i]] = INFINITY;
}

__global__ __launch_bounds__(256) void list_maps_kernel(const float *__restrict__ correl,
                                                        const float *__restrict__ correl_min,
                                                        int Nz, long S, int zper,
                                                        const int *__restrict__ list, int nlist,
                                                        float *__restrict__ maxmap,
                                                        float *__restrict__ minmap) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nlist) return;
  const long s = list[i];
  const int z0 = blockIdx.y * zper, z1 = min(Nz, z0 + zper);
  float a = -INFINITY, b = INFINITY;
  for (int z = z0; z < z1; ++z) {
    a = fmaxf(a, correl[(long)z * S + s]);
    b = fminf(b, correl_min[(long)z * S + s]);
  }
  if (maxmap) atomic_max_f(maxmap + s, a);
  if (minmap) atomic_min_f(minmap + s, b);
}

int spectral_zchunks(origin_ctx *ctx, long S, int Nz, int lwmax) {
  const long blocks = (S + 255) / 256;
  long want = ((long)ctx->num_cu * 8 + blocks - 1) / blocks;
  // the window warm-up reads 2*lwmax extra channels per chunk: keep chunks >= 4 windows
  const long maxc = std::max(1L, (long)Nz / (8L * lwmax + 8));
  if (want > maxc) want = maxc;
  if (want < 1) want = 1;
  return (int)want;
}

template <typename T>
int upload(origin_ctx *ctx, const std::vector<T> &h, T **d, size_t *bytes) {
  void *p = nullptr;
  const size_t n = std::max<size_t>(h.size(), 1) * sizeof(T);
  ORIGIN_HIP(hipMalloc(&p, n));
  *d = (T *)p;
  *bytes += n;
  if (!h.empty()) {
    ORIGIN_HIP(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice,
                              ctx->stream));
    ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
  }
  return ORIGIN_OK;
}

}  // namespace

extern "C" {

int origin_glr_plan_destroy(origin_glr_plan *plan) {
  if (!plan) return ORIGIN_OK;
  (void)hipSetDevice(plan->ctx->device);
  (void)hipStreamSynchronize(plan->ctx->stream);
  for (void *p : {(void *)plan->d_k, (void *)plan->d_k2, (void *)plan->d_w, (void *)plan->d_taps,
                  (void *)plan->d_taps2, (void *)plan->d_tap_off, (void *)plan->d_rden,
                  (void *)plan->d_htaps, (void *)plan->d_htap_off, (void *)plan->d_rows,
                  (void *)plan->d_border, (void *)plan->d_atab, (void *)plan->d_atab_bf16,
                  (void *)plan->d_pwide, (void *)plan->d_rdi_s, (void *)plan->d_normc,
                  (void *)plan->d_atab2, (void *)plan->d_atab_fold, (void *)plan->d_atab_bf16_fold,
                  (void *)plan->d_rden_fold, (void *)plan->d_sden})
    if (p) (void)hipFree(p);
  delete plan->h_order;
  delete plan;
  return ORIGIN_OK;
}

int origin_glr_plan_create(origin_ctx *ctx, int Nz, int Ny, int Nx, int nfields, int P,
                           const double *h_psf, const double *h_weights, int K,
                           const double *h_taps, const int *h_tap_off,
                           origin_glr_plan **out) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(out, "out is null");
  *out = nullptr;
  ORIGIN_CHECK_ARG(Nz > 0 && Ny > 0 && Nx > 0, "bad cube shape (%d,%d,%d)", Nz, Ny, Nx);
  ORIGIN_CHECK_ARG(nfields >= 1 && h_psf, "need at least one PSF");
  ORIGIN_CHECK_ARG(P >= 1 && (P & 1) && P <= 63, "PSF size %d unsupported (odd, <= 63)", P);
  ORIGIN_CHECK_ARG(K >= 1 && K <= 255 && h_taps && h_tap_off,
                   "need 1..255 profiles (profile index is uint8)");
  ORIGIN_CHECK_ARG(h_weights || nfields == 1, "several fields need weight maps");
  for (int k = 0; k < K; ++k)
    ORIGIN_CHECK_ARG(h_tap_off[k + 1] > h_tap_off[k], "profile %d is empty", k);

  origin_glr_plan *pl = new origin_glr_plan();
  pl->fold_eps = INFINITY;
  memset(pl, 0, sizeof(*pl));
  pl->ctx = ctx;
  pl->Nz = Nz, pl->Ny = Ny, pl->Nx = Nx, pl->nfields = nfields, pl->P = P, pl->K = K;
  pl->mode = (h_weights == nullptr && Ny >= P && Nx >= P) ? 0 : 1;
  int rc = ORIGIN_OK;
#define TRY(x)                       \
  do {                               \
    rc = (x);                        \
    if (rc != ORIGIN_OK) {           \
      origin_glr_plan_destroy(pl);   \
      return rc;                     \
    }                                \
  } while (0)

  // zero-mean PSF per channel and field: psf -= psf.mean()      (lib_origin.py:1033-1034)
  const size_t PP = (size_t)P * P;
  std::vector<float> k((size_t)nfields * Nz * PP), k2(k.size());
  for (size_t i = 0; i < (size_t)nfields * Nz; ++i) {
    const double *src = h_psf + i * PP;
    double m = 0.0;
    for (size_t j = 0; j < PP; ++j) m += src[j];
    m /= (double)PP;
    for (size_t j = 0; j < PP; ++j) {
      const double v = src[j] - m;
      k[i * PP + j] = (float)v;
      k2[i * PP + j] = (float)(v * v);  // psf **= 2                            (lib :1040)
    }
  }
  TRY(upload(ctx, k, &pl->d_k, &pl->bytes));
  TRY(upload(ctx, k2, &pl->d_k2, &pl->bytes));
  if (h_weights) {
    std::vector<float> w((size_t)nfields * Ny * Nx);
    for (size_t i = 0; i < w.size(); ++i) w[i] = (float)h_weights[i];
    TRY(upload(ctx, w, &pl->d_w, &pl->bytes));
  }
  // profiles: every kernel centres a profile of length 2*lw+1 on tap lw.  The reference centres
  // on startind = (L-1)//2 (lib :1179-1181), which for an even L is L/2 - 1: an even profile
  // therefore gets a LEADING zero tap, p' = [0, p_0 .. p_{L-1}], lw' = L/2, so that
  // sum_j p'[j] x[z + lw' - j] = sum_j p[j] x[z + L/2 - 1 - j]  (a trailing zero would shift the
  // output by one channel)
  std::vector<float> taps, taps2;
  std::vector<int> off(K + 1, 0);
  int lwmax = 0;
  for (int kk = 0; kk < K; ++kk) {
    const int L = h_tap_off[kk + 1] - h_tap_off[kk];
    if (!(L & 1)) {
      taps.push_back(0.f);
      taps2.push_back(0.f);
    }
    for (int j = 0; j < L; ++j) {
      const double v = h_taps[h_tap_off[kk] + j];
      taps.push_back((float)v);
      taps2.push_back((float)(v * v));
    }
    off[kk + 1] = (int)taps.size();
    lwmax = std::max(lwmax, (off[kk + 1] - off[kk] - 1) / 2);
  }
  pl->lwmax = lwmax;
  pl->Kp = 0;  // set with NzP below
  // exactly symmetric profiles (the Gaussian dictionaries are) allow p[c+d] (w[c+d] + w[c-d])
  std::vector<float> htaps;
  std::vector<int> hoff(K + 1, 0);
  pl->symmetric = 1;
  for (int kk = 0; kk < K; ++kk) {
    const int L = off[kk + 1] - off[kk], lw = (L - 1) / 2;
    for (int d = 0; d <= lw; ++d) {
      if (taps[off[kk] + lw + d] != taps[off[kk] + lw - d]) pl->symmetric = 0;
      htaps.push_back(taps[off[kk] + lw + d]);
    }
    hoff[kk + 1] = (int)htaps.size();
  }
  for (int i = 0; i < 64; ++i) htaps.push_back(0.f);
  TRY(upload(ctx, htaps, &pl->d_htaps, &pl->bytes));
  TRY(upload(ctx, hoff, &pl->d_htap_off, &pl->bytes));
  pl->lwt = 0;
  pl->NzP = (Nz + 31) / 32 * 32 + 32;  // spectral_mfma_kernel reads whole 32-channel tiles
  pl->Kp = pl->NzP;
  if (lwmax <= 32) {
    const int lwt = lwmax <= 8 ? 8 : lwmax <= 16 ? 16 : lwmax <= 24 ? 24 : lwmax <= 29 ? 29 : 32;
    const int RL = (2 * lwt + 1 + 15) / 16 * 16 + 16;  // [lw | taps padded to 16s]
    std::vector<float> rows((size_t)(K + 2) * RL, 0.f);
    for (int kk = 0; kk < K; ++kk) {
      const int L = off[kk + 1] - off[kk], lw = (L - 1) / 2;
      memcpy(&rows[(size_t)kk * RL], &lw, sizeof(int));
      for (int j = 0; j < L; ++j) rows[(size_t)kk * RL + 1 + j] = taps[off[kk] + j];
    }
    TRY(upload(ctx, rows, &pl->d_rows, &pl->bytes));
    pl->lwt = lwt;
  }
  std::vector<double> fold_a;  // a_k (filled with the tap tables below)
  // matrix-core spectral stage: padded tap arrays G_k[e] = p_k[lw_k + 63 - e], 8 copies shifted
  // by 0..7 elements (glr_tables.h), profiles in processing order (narrow ones -- half width
  // <= 16: window blocks 1..4 -- first, so that the kernel's profile pairs are narrow/narrow,
  // at most one narrow/wide, wide/wide); f16 hi + lo (times 2^MF_TAP_SCALE_LOG2) and bf16
  if (lwmax <= 32 && K <= MF_MAX_K && (pl->mode == 0 || h_weights)) {
    std::vector<int> order(K), pinfo(K, 0);
    for (int kk = 0; kk < K; ++kk) order[kk] = kk;
    auto lw_of = [&](int kk) { return (off[kk + 1] - off[kk] - 1) / 2; };
    std::stable_sort(order.begin(), order.end(),
                     [&](int a, int b) { return (lw_of(a) > 16) < (lw_of(b) > 16); });
    std::vector<_Float16> at((size_t)K * MF_PROF_BYTES / 2, (_Float16)0.0f);
    std::vector<_Float16> at2(pl->mode == 1 ? at.size() : 0, (_Float16)0.0f);
    std::vector<unsigned short> ab((size_t)K * MF_PROF_BYTES / 2, 0);
    // FOLD: the same tables with the taps times a_k = 1/sqrt(sum p_k^2) (plans with an explicit
    // norm cube use them too: NORMW, glr_spectral_mfma.hip)
    std::vector<_Float16> atf(mf_fold_fits(K) ? at.size() : 0, (_Float16)0.0f);
    std::vector<unsigned short> abf(mf_fold_fits(K) ? ab.size() : 0, 0);
    fold_a.assign(K, 1.0);
    for (int kk = 0; kk < K; ++kk) {
      double s2 = 0.0;
      for (int j = off[kk]; j < off[kk + 1]; ++j) s2 += (double)taps[j] * taps[j];
      if (s2 > 0.0) fold_a[kk] = 1.0 / std::sqrt(s2);
    }
    const float tscale = (float)(1 << MF_TAP_SCALE_LOG2);
    auto to_bf16 = [](float v) -> unsigned short {  // round to nearest even
      unsigned u;
      memcpy(&u, &v, 4);
      if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
      return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    };
    for (int slot = 0; slot < K; ++slot) {
      const int kk = order[slot];
      const int L = off[kk + 1] - off[kk], lw = (L - 1) / 2;
      pinfo[slot] = kk | ((lw > 16) << 8);
      for (int c = 0; c < 8; ++c)
        for (int q = 0; q < MF_GROUPS; ++q)
          for (int j = 0; j < 8; ++j) {
            const int e = 8 * q + c + j, ti = lw + 63 - e;
            const float t = (ti >= 0 && ti < L) ? taps[off[kk] + ti] : 0.0f;
            const float g = t * tscale;
            const _Float16 gh = (_Float16)g;
            const _Float16 gl = (_Float16)(g - (float)gh);
            const size_t base = (size_t)slot * (MF_PROF_BYTES / 2) +
                                (size_t)c * (MF_COPY_BYTES / 2) + (size_t)q * 8 + j;
            at[base] = gh;
            at[base + 8 * (MF_COPY_BYTES / 2)] = gl;
            ab[base] = to_bf16(t);
            if (!atf.empty()) {
              const float tf = (float)((double)t * fold_a[kk]), gf = tf * tscale;
              const _Float16 gfh = (_Float16)gf;
              atf[base] = gfh;
              atf[base + 8 * (MF_COPY_BYTES / 2)] = (_Float16)(gf - (float)gfh);
              abf[base] = to_bf16(tf);
            }
            if (!at2.empty()) {  // squared taps (float32 squares, as d_taps2) for the denominator
              const float g2 = (t * t) * tscale;
              const _Float16 g2h = (_Float16)g2;
              at2[base] = g2h;
              at2[base + 8 * (MF_COPY_BYTES / 2)] = (_Float16)(g2 - (float)g2h);
            }
          }
    }
    TRY(upload(ctx, at, (_Float16 **)&pl->d_atab, &pl->bytes));
    TRY(upload(ctx, ab, (unsigned short **)&pl->d_atab_bf16, &pl->bytes));
    if (!at2.empty()) TRY(upload(ctx, at2, (_Float16 **)&pl->d_atab2, &pl->bytes));
    if (!atf.empty()) {
      TRY(upload(ctx, atf, (_Float16 **)&pl->d_atab_fold, &pl->bytes));
      TRY(upload(ctx, abf, (unsigned short **)&pl->d_atab_bf16_fold, &pl->bytes));
    }
    TRY(upload(ctx, pinfo, &pl->d_pwide, &pl->bytes));
    pl->n_narrow = 0;
    for (int slot = 0; slot < K; ++slot) pl->n_narrow += (pinfo[slot] >> 8) == 0;
    pl->h_order = new std::vector<int>(order);
    pl->order_ident = 1;
    for (int slot = 0; slot < K; ++slot) pl->order_ident &= order[slot] == slot;
    pl->precision = getenv("ORIGIN_GLR_FP32") ? 0 : 1;
  }
  // a mosaic of weighted fields: its spatial stage runs on the matrix cores too (per-field
  // accumulation, glr_spatial_mfma.hip); its spectral stage convolves the norm cube next to the
  // data and stays in fp32
  if (h_weights && pl->precision == 0 && origin_spatial_mfma_ok(Ny, Nx, P) && !getenv("ORIGIN_GLR_FP32"))
    pl->precision = 1;
  // scalar loads may read a few taps past the end of a profile row: pad
  for (int i = 0; i < 64; ++i) {
    taps.push_back(0.f);
    taps2.push_back(0.f);
  }
  TRY(upload(ctx, taps, &pl->d_taps, &pl->bytes));
  TRY(upload(ctx, taps2, &pl->d_taps2, &pl->bytes));
  TRY(upload(ctx, off, &pl->d_tap_off, &pl->bytes));

  if (pl->mode == 1) {
    // the norm cube of a weighted plan (padded like cube_fsf: MF_PAD_FRONT zero channels in front,
    // MF_PAD_BACK behind) is a constant of the plan: allocated HERE, where the callers' memory
    // checks run and plan->bytes is read, and filled by the first run
    const size_t padded = ((size_t)Nz + MF_PAD_FRONT + MF_PAD_BACK) * (size_t)Ny * Nx;
    hipError_t e = hipMalloc((void **)&pl->d_normc, padded * sizeof(float));
    if (e != hipSuccess) {
      origin_set_error("norm cube of the weighted plan (%zu bytes): %s", padded * sizeof(float),
                       hipGetErrorString(e));
      origin_glr_plan_destroy(pl);
      return e == hipErrorOutOfMemory ? ORIGIN_E_NOMEM : ORIGIN_E_HIP;
    }
    e = hipMemsetAsync(pl->d_normc, 0, padded * sizeof(float), ctx->stream);
    if (e != hipSuccess) {
      origin_set_error("norm cube memset: %s", hipGetErrorString(e));
      origin_glr_plan_destroy(pl);
      return ORIGIN_E_HIP;
    }
    pl->bytes += padded * sizeof(float);
    pl->normc_ready = 0;
  }
  if (pl->mode == 0) {
    std::vector<int> border;
    const int c = P / 2;
    for (int y = 0; y < Ny; ++y)
      for (int x = 0; x < Nx; ++x)
        if (y < c || y > Ny - 1 - c || x < c || x > Nx - 1 - c) border.push_back(y * Nx + x);
    pl->nborder = (int)border.size();
    TRY(upload(ctx, border, &pl->d_border, &pl->bytes));
    double *ncls = nullptr;
    const size_t ncls_n = (size_t)Nz * PP;
    hipError_t e = hipMalloc((void **)&ncls, ncls_n * sizeof(double));
    if (e != hipSuccess) {
      origin_set_error("hipMalloc(norm classes): %s", hipGetErrorString(e));
      origin_glr_plan_destroy(pl);
      return ORIGIN_E_NOMEM;
    }
    const size_t rn = PP * (size_t)K * pl->NzP;
    e = hipMalloc((void **)&pl->d_rden, rn * sizeof(float));
    if (e != hipSuccess) {
      (void)hipFree(ncls);
      origin_set_error("hipMalloc(rden, %zu bytes): %s", rn * sizeof(float), hipGetErrorString(e));
      origin_glr_plan_destroy(pl);
      return ORIGIN_E_NOMEM;
    }
    pl->bytes += rn * sizeof(float);
    ProfScope ps(ctx, K_GLR_TABLES);
    hipLaunchKernelGGL(norm_classes_kernel, dim3(cdiv((long)ncls_n, 256)), dim3(256), 0,
                       ctx->stream, pl->d_k2, Nz, P, ncls);
    hipLaunchKernelGGL(rden_kernel, dim3(cdiv((long)rn, 256)), dim3(256), 0, ctx->stream, ncls,
                       pl->d_taps2, pl->d_tap_off, K, Nz, (int)PP, pl->NzP, pl->d_rden);
    e = hipGetLastError();
    // the interior class is a slice of the table
    pl->d_rdi = pl->d_rden + (size_t)((P / 2) * P + P / 2) * K * pl->NzP;
    if (e == hipSuccess && pl->h_order) {  // the same slice in the kernel's processing order
      e = hipMalloc((void **)&pl->d_rdi_s, (size_t)K * pl->NzP * sizeof(float));
      pl->bytes += (size_t)K * pl->NzP * sizeof(float);
      for (int slot = 0; slot < K && e == hipSuccess; ++slot)
        e = hipMemcpyAsync(pl->d_rdi_s + (size_t)slot * pl->NzP,
                           pl->d_rdi + (size_t)(*pl->h_order)[slot] * pl->NzP,
                           pl->NzP * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
    }
    // FOLD tables and their eps test
    int zf0, zf1;
    mf_fold_range(Nz, &zf0, &zf1);
    float *d_ainv = nullptr;
    unsigned *d_eps = nullptr;
    unsigned eps_bits = 0x7f800000u;
    if (e == hipSuccess && pl->d_atab_fold && zf1 > zf0 && mf_fold_fits(K)) {
      std::vector<float> ainv(K);
      for (int kk = 0; kk < K; ++kk) ainv[kk] = (float)(1.0 / fold_a[kk]);
      e = hipMalloc((void **)&d_ainv, K * sizeof(float) + sizeof(unsigned));
      if (e == hipSuccess) {
        d_eps = (unsigned *)(d_ainv + K);
        e = hipMemcpyAsync(d_ainv, ainv.data(), K * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // (ainv is a local)
      if (e == hipSuccess) e = hipMemsetAsync(d_eps, 0, sizeof(unsigned), ctx->stream);
      if (e == hipSuccess) e = hipMalloc((void **)&pl->d_rden_fold, rn * sizeof(float));
      if (e == hipSuccess) e = hipMalloc((void **)&pl->d_sden, PP * (size_t)pl->NzP * sizeof(float));
      if (e == hipSuccess) {
        hipLaunchKernelGGL(fold_tables_kernel, dim3(cdiv((long)(PP * pl->NzP), 256)), dim3(256), 0,
                           ctx->stream, pl->d_rden, d_ainv, K, (int)PP, pl->NzP, zf0, zf1,
                           pl->d_rden_fold, pl->d_sden, d_eps);
        e = hipGetLastError();
      }
      if (e == hipSuccess)
        e = hipMemcpyAsync(&eps_bits, d_eps, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(ncls);
    if (d_ainv) (void)hipFree(d_ainv);
    if (e != hipSuccess) {
      origin_set_error("rden kernels: %s", hipGetErrorString(e));
      origin_glr_plan_destroy(pl);
      return ORIGIN_E_HIP;
    }
    memcpy(&pl->fold_eps, &eps_bits, sizeof(float));
    if (pl->d_rden_fold && pl->fold_eps <= MF_FOLD_EPS) {
      pl->bytes += (rn + PP * (size_t)pl->NzP) * sizeof(float);
    } else {  // no FOLD for this plan: drop its tables
      for (void **q : {(void **)&pl->d_atab_fold, (void **)&pl->d_atab_bf16_fold,
                       (void **)&pl->d_rden_fold, (void **)&pl->d_sden}) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
      }
    }
  }
#undef TRY
  *out = pl;
  return ORIGIN_OK;
}

int origin_glr_plan_set_precision(origin_glr_plan *plan, int precision) {
  ORIGIN_CHECK_ARG(plan && precision >= 0 && precision <= 2, "precision must be 0, 1 or 2");
  // the matrix-core stages need the plan's tap tables (weights=None, half widths <= 32)
  // (mode 1 without weight maps -- a field smaller than the PSF -- stays on the fp32 kernels)
  const bool eligible = plan->mode == 0 ? plan->d_atab != nullptr
                                        : plan->d_w && (plan->d_atab2 ||
                                                        origin_spatial_mfma_ok(plan->Ny, plan->Nx, plan->P));
  plan->precision = eligible ? precision : 0;
  return ORIGIN_OK;
}

int origin_glr_plan_get_precision(origin_glr_plan *plan, int *precision) {
  ORIGIN_CHECK_ARG(plan && precision, "null argument");
  *precision = plan->precision;
  return ORIGIN_OK;
}

int origin_glr_plan_fold_eps(origin_glr_plan *plan, float *eps, int *active) {
  ORIGIN_CHECK_ARG(plan && eps && active, "null argument");
  *eps = plan->fold_eps;
  // (a plan with a norm cube measures eps in its first run: +inf and inactive before that)
  *active = (plan->mode == 0 ? plan->d_rden_fold != nullptr
                             : plan->normw_checked && plan->d_atab_fold != nullptr &&
                                   plan->fold_eps <= MF_FOLD_EPS) &&
            !getenv("ORIGIN_GLR_NO_FOLD");
  return ORIGIN_OK;
}

int origin_glr_plan_mfma_count(origin_glr_plan *plan, long *spatial, long *spectral) {
  ORIGIN_CHECK_ARG(plan && spatial && spectral, "null argument");
  const origin_glr_plan *pl = plan;
  *spatial = *spectral = 0;
  // the same conditions as origin_glr_run
  if (pl->precision >= 1 && pl->mode == 0 && pl->nfields == 1 &&
      origin_spatial_mfma_ok(pl->Ny, pl->Nx, pl->P))
    *spatial = origin_spatial_mfma_count(pl->precision == 2 ? 1 : 3, pl->Nz, pl->Ny, pl->Nx, pl->P);
  if (pl->precision >= 1 && pl->mode == 0 && pl->d_atab && pl->d_rdi)
    *spectral = origin_spectral_mfma_count(pl->ctx->num_cu, pl->precision == 2 ? 1 : 3, pl->K,
                                           pl->n_narrow, pl->Nz, pl->Ny, pl->Nx);
  return ORIGIN_OK;
}

int origin_glr_mfma_count_model(int num_cu, int terms, int K, int n_narrow, int Nz, int Ny, int Nx,
                                int P, long *spatial, long *spectral) {
  ORIGIN_CHECK_ARG(spatial && spectral && num_cu > 0 && (terms == 1 || terms == 3) && K > 0 &&
                       n_narrow >= 0 && n_narrow <= K && Nz > 0 && Ny > 0 && Nx > 0 && P > 0,
                   "bad arguments");
  *spatial = origin_spatial_mfma_ok(Ny, Nx, P) ? origin_spatial_mfma_count(terms, Nz, Ny, Nx, P) : 0;
  *spectral = origin_spectral_mfma_count(num_cu, terms, K, n_narrow, Nz, Ny, Nx);
  return ORIGIN_OK;
}

int origin_glr_plan_bytes(origin_glr_plan *plan, size_t *bytes) {
  ORIGIN_CHECK_ARG(plan && bytes, "null argument");
  *bytes = plan->bytes;
  return ORIGIN_OK;
}

int origin_glr_work_elems(origin_glr_plan *plan, size_t *elems) {
  ORIGIN_CHECK_ARG(plan && elems, "null argument");
  const size_t cube = (size_t)plan->Nz * plan->Ny * plan->Nx;
  const size_t S = (size_t)plan->Ny * plan->Nx;
  // zero pad + cube_fsf + zero pad + maxmap/minmap partials (<= 64 chunks each); the norm cube
  // of mode 1 belongs to the plan
  (void)cube;
  *elems = cube + 2 * 64 * S + (MF_PAD_FRONT + MF_PAD_BACK) * S;
  return ORIGIN_OK;
}

// ---- a GLR run in row bands (plans whose two stages run the table kernels on the matrix cores)
static bool glr_rows_ok(const origin_glr_plan *pl) {
  return pl->mode == 0 && pl->nfields == 1 && !pl->d_w && pl->precision >= 1 && pl->d_atab &&
         pl->d_rdi && origin_spatial_mfma_ok(pl->Ny, pl->Nx, pl->P);
}

int origin_glr_rows_supported(origin_glr_plan *plan, int *ok) {
  ORIGIN_CHECK_ARG(plan && ok, "null argument");
  *ok = glr_rows_ok(plan) ? 1 : 0;
  return ORIGIN_OK;
}

int origin_glr_run_rows(origin_ctx *ctx, origin_glr_plan *pl, const float *d_cube,
                        const uint8_t *d_mask, float *d_work, float *d_correl, uint8_t *d_profile,
                        float *d_correl_min, int y0, int y1, int flags) {
  return origin_glr_run_rect(ctx, pl, d_cube, d_mask, d_work, d_correl, d_profile, d_correl_min, y0,
                             y1, 0, pl ? pl->Nx : 0, flags);
}

int origin_glr_run_rect(origin_ctx *ctx, origin_glr_plan *pl, const float *d_cube,
                        const uint8_t *d_mask, float *d_work, float *d_correl, uint8_t *d_profile,
                        float *d_correl_min, int y0, int y1, int x0, int x1, int flags) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(pl && pl->ctx == ctx, "plan does not belong to this context");
  ORIGIN_CHECK_ARG(d_cube && d_work && d_correl && d_profile && d_correl_min, "null pointer");
  const int Nz = pl->Nz, Ny = pl->Ny, Nx = pl->Nx, P = pl->P, K = pl->K;
  ORIGIN_CHECK_ARG(y0 >= 0 && y0 < y1 && y1 <= Ny && y0 % 64 == 0 && (y1 % 64 == 0 || y1 == Ny),
                   "row band must start at a multiple of 64 and end at one or at Ny");
  ORIGIN_CHECK_ARG(x0 >= 0 && x0 < x1 && x1 <= Nx && x0 % 64 == 0 && (x1 % 64 == 0 || x1 == Nx),
                   "column range must start at a multiple of 64 and end at one or at Nx");
  const bool whole_rows = x0 == 0 && x1 == Nx;  // (then the waves are those of a run over the field)
  if (!glr_rows_ok(pl)) {
    origin_set_error("origin_glr_run_rows: the plan's stages do not run the matrix-core table kernels");
    return ORIGIN_E_STATE;
  }
  const long S = (long)Ny * Nx;
  const size_t cube = (size_t)Nz * S;
  float *fsf = d_work + (size_t)MF_PAD_FRONT * S;
  float *part = fsf + cube + (size_t)MF_PAD_BACK * S;
  const bool side = (flags & ORIGIN_GLR_SIDE) != 0;
  if (flags & ORIGIN_GLR_FIRST) {  // the zero channels around cube_fsf (main stream: before any band)
    ORIGIN_HIP(hipMemsetAsync(d_work, 0, (size_t)MF_PAD_FRONT * S * sizeof(float), ctx->stream));
    ORIGIN_HIP(hipMemsetAsync(fsf + cube, 0, (size_t)MF_PAD_BACK * S * sizeof(float), ctx->stream));
  }
  // the launch functions enqueue on ctx->stream: the side stream takes its place for this band
  if (side) {
    if (int rc = origin_side_begin(ctx)) return rc;
    std::swap(ctx->stream, ctx->side_stream);
  }
  int rc = ORIGIN_OK;
  {
    ProfScope ps(ctx, K_GLR_SPATIAL);
    rc = origin_spatial_mfma_launch(ctx, pl->precision == 2 ? 1 : 3, d_cube, nullptr, pl->d_k, Nz, Ny,
                                    Nx, P, 0, fsf, y0 / 64, cdiv(y1 - y0, 64), x0 / 64,
                                    cdiv(x1 - x0, 64));
  }
  if (rc == ORIGIN_OK) {
    ProfScope ps(ctx, K_GLR_SPECTRAL);
    int nzc = 0;
    float *pmax = nullptr, *pmin = nullptr;
    rc = origin_spectral_mfma_launch(
        ctx, pl->precision == 2 ? 1 : 3, fsf, pl->d_rden, pl->d_rdi_s, pl->NzP,
        pl->precision == 2 ? pl->d_atab_bf16 : pl->d_atab, pl->d_pwide, K, pl->n_narrow, Nz, Ny, Nx,
        P, d_mask, d_correl, d_profile, d_correl_min, part, true, &nzc, &pmax, &pmin,
        pl->precision == 2 ? pl->d_atab_bf16_fold : pl->d_atab_fold, pl->d_rden_fold, pl->d_sden,
        pl->order_ident, (long)y0 * Nx, (long)(y1 - y0) * Nx, nullptr, 0, whole_rows ? 0 : x0,
        whole_rows ? 0 : x1);
  }
  if (side) {
    std::swap(ctx->stream, ctx->side_stream);
    if (rc == ORIGIN_OK) rc = origin_side_end(ctx);
  }
  return rc;
}

int origin_glr_run_finish(origin_ctx *ctx, origin_glr_plan *pl, float *d_work, float *d_maxmap,
                          float *d_minmap) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(pl && pl->ctx == ctx && d_work, "bad argument");
  if (int rc = origin_side_join(ctx)) return rc;  // bands on the side stream
  if (!d_maxmap && !d_minmap) return ORIGIN_OK;
  const long S = (long)pl->Ny * pl->Nx;
  float *fsf = d_work + (size_t)MF_PAD_FRONT * S;
  float *part = fsf + (size_t)pl->Nz * S + (size_t)MF_PAD_BACK * S;
  const int nzc = origin_spectral_mfma_chunks(ctx->num_cu, pl->Nz, pl->Ny, pl->Nx);
  ProfScope ps(ctx, K_SMALL);
  hipLaunchKernelGGL(maxmap_final_kernel, dim3(cdiv(S, 256)), dim3(256), 0, ctx->stream, part,
                     part + (size_t)nzc * S, nzc, S, d_maxmap, d_minmap);
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

int origin_glr_run(origin_ctx *ctx, origin_glr_plan *pl, const float *d_cube,
                   const uint8_t *d_mask, float *d_work, float *d_correl,
                   uint8_t *d_profile, float *d_correl_min, float *d_maxmap,
                   float *d_minmap) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(pl && pl->ctx == ctx, "plan does not belong to this context");
  ORIGIN_CHECK_ARG(d_cube && d_work && d_correl && d_profile && d_correl_min, "null pointer");
  const int Nz = pl->Nz, Ny = pl->Ny, Nx = pl->Nx, P = pl->P, K = pl->K;
  const long S = (long)Ny * Nx;
  const size_t cube = (size_t)Nz * S;
  // [pad | cube_fsf | pad | partial maps]; the pads are zero channels
  float *fsf = d_work + (size_t)MF_PAD_FRONT * S;
  float *part = fsf + cube + (size_t)MF_PAD_BACK * S;
  // mode 1: norm_fsf depends on the PSFs and the weight maps only -- the first run computes it
  // into a cube the plan keeps
  // (padded like cube_fsf: MF_PAD_FRONT zero channels in front, MF_PAD_BACK behind)
  if (pl->mode == 1 && !pl->d_normc) {
    const size_t padded = cube + (size_t)(MF_PAD_FRONT + MF_PAD_BACK) * S;
    ORIGIN_HIP(hipMalloc((void **)&pl->d_normc, padded * sizeof(float)));
    ORIGIN_HIP(hipMemsetAsync(pl->d_normc, 0, padded * sizeof(float), ctx->stream));
    pl->bytes += padded * sizeof(float);
    pl->normc_ready = 0;
  }
  float *norm = pl->mode == 1 ? pl->d_normc + (size_t)MF_PAD_FRONT * S : nullptr;
  ORIGIN_HIP(hipMemsetAsync(d_work, 0, (size_t)MF_PAD_FRONT * S * sizeof(float), ctx->stream));
  ORIGIN_HIP(hipMemsetAsync(fsf + cube, 0, (size_t)MF_PAD_BACK * S * sizeof(float), ctx->stream));

  // ---- spatial stage
  auto spatial = [&](const float *A, const float *B, const float *taps, int acc, float *dst) {
    // each block marches `zper` channels of one 64x64 tile, prefetching the next plane
    const long tiles = (long)cdiv(Nx, 64) * cdiv(Ny, 64);
    int nzb = (int)(((long)ctx->num_cu * 16 + tiles - 1) / tiles);
    nzb = std::max(1, std::min(nzb, Nz));
    const int zper = cdiv(Nz, nzb);
    dim3 g4(cdiv(Nx, 64), cdiv(Ny, 64), cdiv(Nz, zper));
    const bool vec = (Nx & 3) == 0 && ((P / 2) & 3) == 0;
#define LAUNCH_SP2(PP, VV, BB)                                                                  \
  hipLaunchKernelGGL((spatial4x4_kernel<PP, VV, BB>), g4, dim3(256), 0, ctx->stream, A, B, taps, \
                     Nz, Ny, Nx, zper, acc, dst)
#define LAUNCH_SP(PP)                                \
  if (vec && B) LAUNCH_SP2(PP, true, true);          \
  else if (vec) LAUNCH_SP2(PP, true, false);         \
  else if (B) LAUNCH_SP2(PP, false, true);           \
  else LAUNCH_SP2(PP, false, false)
    switch (A ? P : 0) {  // A == NULL (norm of the weights) takes the generic kernel
      case 7:
        LAUNCH_SP(7);
        break;
      case 9:
        LAUNCH_SP(9);
        break;
      case 25:
        LAUNCH_SP(25);
        break;
      default: {  // any other odd PSF size: generic LDS-tiled kernel
        dim3 sgrid(cdiv(Nx, TX), cdiv(Ny, TY), Nz), sblock(64, 4);
        const size_t lds = (size_t)(TY + P - 1) * (TX + P - 1) * sizeof(float);
        hipLaunchKernelGGL(spatial_kernel, sgrid, sblock, lds, ctx->stream, A, B, taps, Ny, Nx, P,
                           acc, dst);
      }
    }
#undef LAUNCH_SP
#undef LAUNCH_SP2
  };
  for (int f = 0; f < pl->nfields; ++f) {
    ProfScope ps(ctx, K_GLR_SPATIAL);
    const float *kf = pl->d_k + (size_t)f * Nz * P * P;
    const float *wf = pl->d_w ? pl->d_w + (size_t)f * S : nullptr;
    const bool sp_mfma = pl->precision >= 1 && origin_spatial_mfma_ok(Ny, Nx, P) &&
                         ((pl->mode == 0 && !wf && pl->nfields == 1) || wf);
    if (sp_mfma) {
      // matrix cores, two-term f16 split (glr_spatial_mfma.hip); weighted fields accumulate
      int rc = origin_spatial_mfma_launch(ctx, pl->precision == 2 ? 1 : 3, d_cube, wf, kf, Nz, Ny,
                                          Nx, P, wf && f > 0, fsf);
      if (rc) return rc;
    } else {
      spatial(d_cube, wf, kf, f > 0, fsf);
    }
    if (pl->mode == 1 && !pl->normc_ready) {
      const float *k2f = pl->d_k2 + (size_t)f * Nz * P * P;
      spatial(nullptr, wf, k2f, f > 0, norm);
    }
  }
  if (pl->mode == 1) pl->normc_ready = 1;
  // NORMW: the first run measures eps of the FOLD form on the norm cube it has just made
  if (pl->mode == 1 && !pl->normw_checked && pl->d_atab_fold) {
    pl->normw_checked = 1;
    pl->fold_eps = INFINITY;
    int zf0, zf1;
    mf_fold_range(Nz, &zf0, &zf1);
    if (zf1 > zf0 && pl->lwmax <= 32) {
      void *scr = nullptr;
      if (int rc = origin_scratch(ctx, 256, &scr)) return rc;
      unsigned *d_eps = (unsigned *)scr, bits = 0x7f800000u;
      ORIGIN_HIP(hipMemsetAsync(d_eps, 0, sizeof(unsigned), ctx->stream));
      hipLaunchKernelGGL(normw_eps_kernel,
                         dim3((unsigned)cdiv(S, 256), (unsigned)cdiv(zf1 - zf0, NE_ZT)), dim3(256), 0,
                         ctx->stream, norm, pl->d_taps2, pl->d_tap_off, K, S, zf0, zf1, d_eps);
      ORIGIN_LAUNCH_CHECK();
      ORIGIN_HIP(hipMemcpyAsync(&bits, d_eps, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
      ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
      memcpy(&pl->fold_eps, &bits, sizeof(float));
    }
  }
  ORIGIN_LAUNCH_CHECK();

  // ---- spectral stage
  const bool want_maps = d_maxmap || d_minmap;
  int nzc = spectral_zchunks(ctx, S, Nz, std::max(pl->lwmax, 1));
  if (nzc > 64) nzc = 64;
  int zchunk = cdiv(Nz, nzc);
  zchunk = (zchunk + SPEC_ZC - 1) / SPEC_ZC * SPEC_ZC;  // the packed kernel steps SPEC_ZC channels
  nzc = cdiv(Nz, zchunk);
  float *pmax = want_maps ? part : nullptr;
  float *pmin = want_maps ? part + (size_t)nzc * S : nullptr;
  dim3 grid(cdiv(S, 256), nzc), block(256);
#define LAUNCH(KERNEL)                                                                        \
  hipLaunchKernelGGL(KERNEL, grid, block, 0, ctx->stream, fsf, norm, pl->d_rden, pl->d_taps, \
                     pl->d_taps2, pl->d_tap_off, K, pl->Kp, Nz, Ny, Nx, P, zchunk, d_mask,   \
                     d_correl,                                                              \
                     d_profile, d_correl_min, pmax, pmin, (const int *)nullptr, 0)
  const bool gen = pl->mode == 1;
  bool border_fix = false;
  {
  ProfScope ps(ctx, K_GLR_SPECTRAL);
  const bool mfma = !gen && pl->precision >= 1 && pl->d_atab && pl->d_rdi;
  // explicit norm cube: second Toeplitz product on the matrix cores (f16 split only: precision 1)
  const bool mfma_norm = gen && pl->precision == 1 && pl->d_atab && pl->d_atab2 &&
                         K <= origin_spectral_norm_mfma_max_k();
  const bool packed = !mfma && !gen && (S & 1) == 0 && pl->lwt;
  // NORMW: the FOLD form of the table kernel on the norm cube where the plan's eps allows it, the
  // two-product kernel for the 32 channels at either end of the cube
  int zf0 = 0, zf1 = 0;
  mf_fold_range(Nz, &zf0, &zf1);
  const int end_rows = cdiv(zf0, 32) + cdiv(Nz - zf1, 32);
  // (bf16 plans too: one bf16 MFMA per product between the ends, the ends on the f16 split)
  const bool norm_tables = gen && pl->precision >= 1 && pl->d_atab && pl->d_atab2 &&
                           K <= origin_spectral_norm_mfma_max_k();
  const bool normw = norm_tables && pl->normw_checked && pl->fold_eps <= MF_FOLD_EPS &&
                     pl->d_atab_fold && zf1 > zf0 && !getenv("ORIGIN_GLR_NO_FOLD") &&
                     origin_spectral_mfma_chunks(ctx->num_cu, Nz, Ny, Nx) + end_rows <= 64;
  if (normw) {
    const int rows = origin_spectral_mfma_chunks(ctx->num_cu, Nz, Ny, Nx) + end_rows;
    const bool b16 = pl->precision == 2;
    int rc = origin_spectral_mfma_launch(
        ctx, b16 ? 1 : 3, fsf, nullptr, nullptr, pl->NzP, b16 ? pl->d_atab_bf16 : pl->d_atab,
        pl->d_pwide, K, pl->n_narrow, Nz, Ny, Nx, P, d_mask, d_correl, d_profile, d_correl_min, part,
        want_maps, &nzc, &pmax, &pmin, b16 ? pl->d_atab_bf16_fold : pl->d_atab_fold, nullptr,
        nullptr, pl->order_ident, 0, 0, norm, rows);
    if (rc) return rc;
    int got = 0;
    rc = origin_spectral_norm_mfma_launch_ends(ctx, fsf, norm, pl->d_atab, pl->d_atab2, pl->d_pwide, K,
                                               Nz, Ny, Nx, d_mask, d_correl, d_profile,
                                               d_correl_min, pmax, pmin, zf0, zf1, nzc, &got);
    if (rc) return rc;
    nzc += got;
  } else if (mfma_norm) {
    int rc = origin_spectral_norm_mfma_launch(ctx, fsf, norm, pl->d_atab, pl->d_atab2, pl->d_pwide, K,
                                              Nz, Ny, Nx, d_mask, d_correl, d_profile, d_correl_min,
                                              part, want_maps, &nzc, &pmax, &pmin);
    if (rc) return rc;
  } else if (mfma) {
    int rc = origin_spectral_mfma_launch(
        ctx, pl->precision == 2 ? 1 : 3, fsf, pl->d_rden, pl->d_rdi_s, pl->NzP,
        pl->precision == 2 ? pl->d_atab_bf16 : pl->d_atab, pl->d_pwide, K, pl->n_narrow, Nz, Ny, Nx,
        P, d_mask,
        d_correl, d_profile, d_correl_min, part, want_maps, &nzc, &pmax, &pmin,
        pl->precision == 2 ? pl->d_atab_bf16_fold : pl->d_atab_fold, pl->d_rden_fold, pl->d_sden,
        pl->order_ident);
    if (rc) return rc;
  } else if (packed) {
    // packed path: one lane = two adjacent spaxels, SPEC_ZC channels per step
    dim3 g2(cdiv(S / 2, 256), nzc);
#define LAUNCH3(LW)                                                                            \
  hipLaunchKernelGGL((spectral3_kernel<LW>), g2, block, 0, ctx->stream, fsf, pl->d_rdi, pl->NzP, \
                     pl->d_rows, K, Nz, Ny, Nx, zchunk, d_mask, d_correl, d_profile,           \
                     d_correl_min, pmax, pmin)
    switch (pl->lwt) {
      case 8: LAUNCH3(8); break;
      case 16: LAUNCH3(16); break;
      case 24: LAUNCH3(24); break;
      case 29: LAUNCH3(29); break;
      default: LAUNCH3(32); break;
    }
#undef LAUNCH3
  } else if (pl->lwmax <= 8) {
    if (gen) LAUNCH((spectral_kernel<8, true>)); else LAUNCH((spectral_kernel<8, false>));
  } else if (pl->lwmax <= 16) {
    if (gen) LAUNCH((spectral_kernel<16, true>)); else LAUNCH((spectral_kernel<16, false>));
  } else if (pl->lwmax <= 32) {
    if (gen) LAUNCH((spectral_kernel<32, true>)); else LAUNCH((spectral_kernel<32, false>));
  } else {
    if (gen) LAUNCH((spectral_generic_kernel<true>)); else LAUNCH((spectral_generic_kernel<false>));
  }
  if (packed && pl->nborder > 0) {  // border spaxels: exact per-class normalisation
    ps.next(K_GLR_BORDER);
    // few spaxels: cut z finer so that the pass still fills the chip (its maps are redone
    // from the final cubes below, so it writes no partials)
    const long bb = cdiv(pl->nborder, 256);
    int nzb = (int)(((long)ctx->num_cu * 12 + bb - 1) / bb);
    nzb = std::max(1, std::min(nzb, Nz / (4 * std::max(pl->lwmax, 1) + 4)));
    nzb = std::max(nzb, 1);
    const int zcb = cdiv(Nz, nzb);
    dim3 gb((unsigned)bb, cdiv(Nz, zcb));
    border_fix = true;
#define LAUNCHB(LW)                                                                            \
hipLaunchKernelGGL((spectral_kernel<LW, false>), gb, block, 0, ctx->stream, fsf, norm,       \
                   pl->d_rden, pl->d_taps, pl->d_taps2, pl->d_tap_off, K, pl->Kp, Nz, Ny, Nx, \
                   P, zcb, d_mask, d_correl, d_profile, d_correl_min, (float *)nullptr,      \
                   (float *)nullptr, pl->d_border, pl->nborder)
    if (pl->lwmax <= 8) LAUNCHB(8); else if (pl->lwmax <= 16) LAUNCHB(16); else LAUNCHB(32);
#undef LAUNCHB
  }
  }
#undef LAUNCH
  ORIGIN_LAUNCH_CHECK();
  if (want_maps) {
    ProfScope ps(ctx, K_SMALL);
    hipLaunchKernelGGL(maxmap_final_kernel, dim3(cdiv(S, 256)), dim3(256), 0, ctx->stream, pmax,
                       pmin, nzc, S, d_maxmap, d_minmap);
    if (border_fix) {
      const int zper = 64;
      hipLaunchKernelGGL(list_maps_init_kernel, dim3(cdiv(pl->nborder, 256)), dim3(256), 0,
                         ctx->stream, pl->d_border, pl->nborder, d_maxmap, d_minmap);
      hipLaunchKernelGGL(list_maps_kernel, dim3(cdiv(pl->nborder, 256), cdiv(Nz, zper)), dim3(256),
                         0, ctx->stream, d_correl, d_correl_min, Nz, S, zper, pl->d_border,
                         pl->nborder, d_maxmap, d_minmap);
    }
    ORIGIN_LAUNCH_CHECK();
  }
  return ORIGIN_OK;
}

}  // extern "C"
