// GLR spectral stage on the matrix cores for plans with an EXPLICIT norm cube: mosaics of weighted
// fields (reference muse_origin/lib_origin.py:1046-1060 with weights, :1134-1147), where
//
//   T_k[z, s] = num_k / sqrt(den_k),  num_k = sum_j p_k[j]   fsf [z + lw_k - j, s]
//                                      den_k = sum_j p_k[j]^2 norm[z + lw_k - j, s]   (den <= 0 -> T = 0)
//
// and the denominator is no longer a function of (profile, channel, border class) that a table
// could hold (glr_spectral_mfma.hip) but a second banded Toeplitz product per profile, on the norm
// cube the plan keeps (csrc/glr.hip).  Same tiling as the table kernel -- a wave owns 32 spaxels,
// marches z in 32-channel tiles = two 16-channel halves, the 32 MFMA rows hold a PAIR of profiles
// -- with a second accumulator for den: taps^2 (second tap table) against the window of the norm
// cube, both on the two-term f16 split (Ah Bh + Ah Bl + Al Bh, power-of-two tile scales; the norm
// window's scale exponent is kept even so that its square root is a power of two as well).  The
// epilogue multiplies num by v_rsq_f32(den) (0 where den <= 0) before the max / min / arg-max
// bookkeeping of the table kernel.
//
// Two tap tables per profile do not fit LDS for 20 profiles (2 x 4.5 KiB each), so the profiles
// go in TWO launches of up to 13; the second one merges with what the first left in the output
// cubes (exact: it rescales the stored maximum back by the tile's power-of-two factor, rebuilds
// its arg-max key and compares keys).  Mask glue and the per-chunk maxima / minima of the maps
// belong to the last launch.
//
// Round 3: where the norm cube is smooth along z -- den_k[z, s] = norm[z, s] sum_j p_k[j]^2 up to
// a relative eps the plan measures in its first run -- the table kernel's FOLD form takes over
// between the cube's ends (glr_spectral_mfma.hip, NORMW: one Toeplitz product, rsq(norm) of each
// voxel behind the profile loop; 28.6 -> 10.4 ms for two fields at 3681 x 600 x 600) and this
// kernel keeps the 32 channels at either end (origin_spectral_norm_mfma_launch_ends) and the plans
// whose eps is too large.
//
// This is the straightforward form: no software pipelining of the epilogue under the next pair's
// MFMAs, the whole 96-channel window of both cubes converted per tile, two waves per SIMD.  It
// replaces an fp32 kernel that took 110 ms for two fields at 3681 x 600 x 600.
#include <algorithm>

#include "common.h"
#include "glr_tables.h"

namespace {

constexpr int NW = 8;         // waves per block
constexpr int NM_MAX_K = 13;  // profiles per launch: 13 * 2 * 4608 B = 117 KiB of LDS

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));

template <int I, int N, typename F>
__device__ __forceinline__ void nm_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    nm_for<I + 1, N>(f);
  }
}

template <bool FIRST>
__device__ __forceinline__ void nm_mma(f32x16 &acc, const u32x4v &a, const u32x4v &b) {
  if constexpr (FIRST)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

struct NmState {
  float best[8], worst[8], key[8];
};

// Epilogue item o of a profile pair: output o of the lane is accumulator register o (profile a)
// and o + 8 (profile b) of num and den.  (The reciprocal square roots are issued first and used
// three instructions later.)
template <int O>
__device__ __forceinline__ void nm_epi_item(const f32x16 &num, const f32x16 &den, NmState &st,
                                            unsigned maskv, int ca, int cb) {
  float R0, R1, T0, T1, K0, K1;
  asm volatile(
      "v_rsq_f32 %0, %10\n\t"
      "v_rsq_f32 %1, %12\n\t"
      "v_cmp_lt_f32 vcc, 0, %10\n\t"
      "s_nop 0\n\t"
      "v_cndmask_b32 %0, 0, %0, vcc\n\t"
      "v_cmp_lt_f32 vcc, 0, %12\n\t"
      "s_nop 0\n\t"
      "v_cndmask_b32 %1, 0, %1, vcc\n\t"
      "v_mul_f32 %2, %9, %0\n\t"
      "v_mul_f32 %3, %11, %1\n\t"
      "v_and_or_b32 %4, %2, %13, %14\n\t"
      "v_and_or_b32 %5, %3, %13, %15\n\t"
      "v_max3_f32 %6, %6, %4, %5\n\t"
      "v_max3_f32 %7, %7, %2, %3\n\t"
      "v_min3_f32 %8, %8, %2, %3"
      : "=&v"(R0), "=&v"(R1), "=&v"(T0), "=&v"(T1), "=&v"(K0), "=&v"(K1), "+v"(st.key[O]),
        "+v"(st.best[O]), "+v"(st.worst[O])
      : "v"(num[O]), "v"(den[O]), "v"(num[O + 8]), "v"(den[O + 8]), "v"(maskv), "s"(ca), "s"(cb)
      : "vcc");
}

// power-of-two scale of a window: exponent field se of the scale (scale = 2^(se - 127)), max |y|
// in [2^14, 2^15) -- or [2^13, 2^14) when EVEN asks for an even power
template <bool EVEN>
__device__ __forceinline__ int nm_scale_exp(float m) {
  const int ex = (int)((__float_as_uint(m) >> 23) & 0xffu);
  const bool tiny = ex < 40 || ex == 255;  // zero / denormal-small / non-finite: no scaling
  int se = tiny ? 127 : 268 - ex;
  if (EVEN && ((se - 127) & 1)) --se;
  return se;
}

__device__ __forceinline__ void nm_convert(const float (&x)[6][8], float scale, u32x4v (&oh)[6],
                                           u32x4v (&ol)[6]) {
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) {
    f16x2v hh[4], ll[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float y = x[ks][j] * scale;
      const _Float16 yh = (_Float16)y;
      hh[j >> 1][j & 1] = yh;
      ll[j >> 1][j & 1] = (_Float16)(y - (float)yh);
    }
    u32x4v h4, l4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h4[j] = __builtin_bit_cast(unsigned, hh[j]);
      l4[j] = __builtin_bit_cast(unsigned, ll[j]);
    }
    oh[ks] = h4, ol[ks] = l4;
  }
}

// MERGE: the output cubes hold the result of an earlier launch over other profiles
template <bool MERGE>
__global__ __launch_bounds__(64 * NW, 1) void spectral_norm_mfma_kernel(
    const float *__restrict__ fsf, const float *__restrict__ norm, const uint4 *__restrict__ atab,
    const uint4 *__restrict__ atab2, const int *__restrict__ pinfo, int K, int Nz, int Ny, int Nx,
    int zchunk, const uint8_t *__restrict__ mask, float *__restrict__ correl,
    uint8_t *__restrict__ profile, float *__restrict__ correl_min, float *__restrict__ part_max,
    float *__restrict__ part_min, int zstart, int zstop, int prow0) {
  // (zstart, zstop: the channels of this launch, chunk c = [zstart + c zchunk, ...); prow0: row of
  // the partial maps that chunk 0 writes -- the FOLD form of the table kernel leaves the 32
  // channels at either end of the cube to this kernel)
  extern __shared__ __align__(16) char nm_lds[];
  {
    const int nvec = K * (MF_PROF_BYTES / 16);
    for (int i = threadIdx.x; i < nvec; i += 64 * NW) {
      reinterpret_cast<uint4 *>(nm_lds)[i] = atab[i];
      reinterpret_cast<uint4 *>(nm_lds)[nvec + i] = atab2[i];
    }
  }
  __syncthreads();
  const long S = (long)Ny * Nx;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const long s_base = ((long)blockIdx.x * NW + wv) * 32;
  if (s_base >= S) return;  // whole wave; no barrier follows
  const int zc0 = zstart + blockIdx.y * zchunk, zc1 = min(zstop, zc0 + zchunk);
  const int E0 = 8 * h - (r & 15) + 31;
  const char *a_lane = nm_lds + (E0 & 7) * MF_COPY_BYTES + (E0 >> 3) * 16;
  const char *a2_lane = a_lane + K * MF_PROF_BYTES;
  const bool second = (lane & 16) != 0;  // this lane's A rows belong to the pair's profile b
  const bool sv = s_base + r < S;
  const bool all_valid = s_base + 32 <= S;
  const long sc = sv ? s_base + r : S - 1;
  const int rr = (int)(sc - s_base);
  const int NP = (K + 1) / 2;
  float vmax = -INFINITY, vmin = INFINITY;
  unsigned maskv = 0xffffffe0u;
  asm volatile("" : "+v"(maskv));

  const long off_in = (long)(8 * h) * S + rr;  // window rows 16 ks + 8 h + j
  const long off_out = (long)(4 * h) * S + rr;  // outputs (o&3) + 8 (o>>2) + 4 h
  auto load_window = [&](const float *cube, int z0, float (&w)[6][8]) {
    const float *p = cube + (long)(z0 - 32) * S + s_base + off_in;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) w[ks][j] = p[(long)(16 * ks + j) * S];
  };
  auto wave_max_abs = [&](const float (&w)[6][8]) {
    float m = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(w[ks][j]));
    if (!all_valid) m = sv ? m : 0.0f;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
  };

  for (int z0 = zc0; z0 < zc1; z0 += 32) {
    u32x4v bh[6], bl[6], nh[6], nl[6];
    float inv, rinv;
    {
      float x[6][8];
      load_window(fsf, z0, x);
      const int se1 = nm_scale_exp<false>(wave_max_abs(x));
      nm_convert(x, __uint_as_float((unsigned)se1 << 23), bh, bl);
      load_window(norm, z0, x);
      const int se2 = nm_scale_exp<true>(wave_max_abs(x));
      nm_convert(x, __uint_as_float((unsigned)se2 << 23), nh, nl);
      // T = num 2^-(se1-127) 2^-TS / sqrt(den 2^-(se2-127) 2^-TS),  TS = MF_TAP_SCALE_LOG2 (even)
      const int e = (127 - se1) - MF_TAP_SCALE_LOG2 + (se2 - 127) / 2 + MF_TAP_SCALE_LOG2 / 2;
      inv = __uint_as_float((unsigned)(127 + e) << 23);
      rinv = __uint_as_float((unsigned)(127 - e) << 23);
    }
    nm_for<0, 2>([&](auto hc) {
      constexpr int HALF = decltype(hc)::value;
      const int zh = z0 + 16 * HALF;
      if (zh >= zc1) return;  // (uniform)
      NmState st;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        st.best[i] = -INFINITY, st.worst[i] = INFINITY, st.key[i] = -INFINITY;
      for (int p = 0; p < NP; ++p) {
        const int sa = 2 * p, sb = min(2 * p + 1, K - 1);  // (odd K: the last profile twice)
        const int ia = pinfo[sa], ib = pinfo[sb];
        const bool wide = ((ia | ib) >> 8) != 0;
        const int slot = second ? sb : sa;
        const char *ak = a_lane + slot * MF_PROF_BYTES, *a2k = a2_lane + slot * MF_PROF_BYTES;
        constexpr int LO = 8 * MF_COPY_BYTES;
        f32x16 num, den;
        // window blocks HALF + 1 .. HALF + 3 always, HALF + 0 and HALF + 4 for wide pairs
        nm_for<0, 3>([&](auto ic) {
          constexpr int g = decltype(ic)::value, ks = 1 + g;
          const u32x4v ah = *reinterpret_cast<const u32x4v *>(ak + ks * 32);
          const u32x4v al = *reinterpret_cast<const u32x4v *>(ak + ks * 32 + LO);
          const u32x4v qh = *reinterpret_cast<const u32x4v *>(a2k + ks * 32);
          const u32x4v ql = *reinterpret_cast<const u32x4v *>(a2k + ks * 32 + LO);
          nm_mma<g == 0>(num, ah, bh[HALF + ks]);
          nm_mma<false>(num, ah, bl[HALF + ks]);
          nm_mma<false>(num, al, bh[HALF + ks]);
          nm_mma<g == 0>(den, qh, nh[HALF + ks]);
          nm_mma<false>(den, qh, nl[HALF + ks]);
          nm_mma<false>(den, ql, nh[HALF + ks]);
        });
        if (wide) {
#pragma unroll
          for (int e4 = 0; e4 < 2; ++e4) {
            const int ks = 4 * e4;
            const u32x4v ah = *reinterpret_cast<const u32x4v *>(ak + ks * 32);
            const u32x4v al = *reinterpret_cast<const u32x4v *>(ak + ks * 32 + LO);
            const u32x4v qh = *reinterpret_cast<const u32x4v *>(a2k + ks * 32);
            const u32x4v ql = *reinterpret_cast<const u32x4v *>(a2k + ks * 32 + LO);
            nm_mma<false>(num, ah, bh[HALF + ks]);
            nm_mma<false>(num, ah, bl[HALF + ks]);
            nm_mma<false>(num, al, bh[HALF + ks]);
            nm_mma<false>(den, qh, nh[HALF + ks]);
            nm_mma<false>(den, qh, nl[HALF + ks]);
            nm_mma<false>(den, ql, nh[HALF + ks]);
          }
        }
        unsigned mv = maskv;
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(mv));  // the last MFMA's result -> VALU readers
        const int ca = 31 - (ia & 0xff), cb = 31 - (ib & 0xff);
        nm_for<0, 8>([&](auto kc) { nm_epi_item<decltype(kc)::value>(num, den, st, mv, ca, cb); });
      }

      // store (merge with an earlier launch; mask glue steps.py:781,788 on the last one)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int zu = zh + (i & 3) + 8 * (i >> 2);  // uniform part of the channel
        const int z = zu + 4 * h;
        if (z < zc1) {
          const long o = (long)zu * S + s_base + off_out;
          float b = st.best[i], w = st.worst[i];
          float key = st.key[i];
          int kk = 31 - (int)(__float_as_uint(key) & 31u);
          if (MERGE && sv) {
            const float pb = correl[o] * rinv, pw = correl_min[o] * rinv;  // exact: powers of two
            const int pk = profile[o];
            const float pkey =
                __uint_as_float((__float_as_uint(pb) & 0xffffffe0u) | (unsigned)(31 - pk));
            // (the earlier launch stored index 0 for an all-equal spaxel: its key is then below
            // any real one only if this launch's values are larger; equal values keep index 0)
            if (pkey > key || (pb == b && pk < kk)) kk = pk;
            b = fmaxf(b, pb);
            w = fminf(w, pw);
          }
          // profiles run narrow-first, not in index order: when every T is the same number (a
          // spaxel of zeros) the first maximum is index 0 (np.argmax semantics, lib :1210)
          if (b == w) kk = 0;
          b *= inv, w *= inv;
          if (mask && sv && mask[o]) b = 0.0f, kk = 0;
          if (sv) {
            correl[o] = b;
            correl_min[o] = w;
            profile[o] = (uint8_t)kk;
          }
          vmax = fmaxf(vmax, b);
          vmin = fminf(vmin, w);
        }
      }
    });
  }
  if (part_max) {
    const float a = fmaxf(vmax, __shfl_xor(vmax, 32));
    const float b = fminf(vmin, __shfl_xor(vmin, 32));
    if (h == 0 && sv) {
      part_max[(long)(prow0 + blockIdx.y) * S + sc] = a;
      part_min[(long)(prow0 + blockIdx.y) * S + sc] = b;
    }
  }
}

}  // namespace

// the launches over the halves of the profile list for the channels [zstart, zstop) in chunks of
// zcm (grid.y = nchunks); partial maps from row prow0 on
static int nm_passes(origin_ctx *ctx, const float *fsf, const float *norm, const uint4 *atab,
                     const uint4 *atab2, const int *pinfo, int K, int Nz, int Ny, int Nx,
                     const uint8_t *mask, float *correl, uint8_t *profile, float *correl_min,
                     float *pmax, float *pmin, long bx, int nchunks, int zcm, int zstart, int zstop,
                     int prow0) {
  if (K > 2 * NM_MAX_K) {
    origin_set_error("spectral norm MFMA kernel: %d profiles (at most %d)", K, 2 * NM_MAX_K);
    return ORIGIN_E_ARG;
  }
  const int npass = K > NM_MAX_K ? 2 : 1;
  const int K0 = npass == 2 ? (K / 2 + 1) / 2 * 2 : K;  // an even count first: whole pairs
  static OriginPerDeviceOnce attr_once;
  const int dyn = NM_MAX_K * 2 * MF_PROF_BYTES;
  ORIGIN_ONCE_PER_DEVICE(ctx, attr_once,
                         ORIGIN_HIP(hipFuncSetAttribute(
                             (const void *)spectral_norm_mfma_kernel<false>,
                             hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
                         ORIGIN_HIP(hipFuncSetAttribute(
                             (const void *)spectral_norm_mfma_kernel<true>,
                             hipFuncAttributeMaxDynamicSharedMemorySize, dyn)));
  for (int pass = 0; pass < npass; ++pass) {
    const int s0 = pass == 0 ? 0 : K0, Kl = pass == 0 ? K0 : K - K0;
    const bool last = pass == npass - 1;
    const size_t lds = (size_t)Kl * 2 * MF_PROF_BYTES;
    const uint4 *ta = atab + (size_t)s0 * (MF_PROF_BYTES / 16);
    const uint4 *tb = atab2 + (size_t)s0 * (MF_PROF_BYTES / 16);
    const uint8_t *mk = last ? mask : nullptr;
    float *qmax = last ? pmax : nullptr, *qmin = last ? pmin : nullptr;
    dim3 grid((unsigned)bx, nchunks), block(64 * NW);
    if (pass == 0)
      hipLaunchKernelGGL(spectral_norm_mfma_kernel<false>, grid, block, lds, ctx->stream, fsf, norm,
                         ta, tb, pinfo + s0, Kl, Nz, Ny, Nx, zcm, mk, correl, profile, correl_min,
                         qmax, qmin, zstart, zstop, prow0);
    else
      hipLaunchKernelGGL(spectral_norm_mfma_kernel<true>, grid, block, lds, ctx->stream, fsf, norm,
                         ta, tb, pinfo + s0, Kl, Nz, Ny, Nx, zcm, mk, correl, profile, correl_min,
                         qmax, qmin, zstart, zstop, prow0);
    ORIGIN_LAUNCH_CHECK();
  }
  return ORIGIN_OK;
}

int origin_spectral_norm_mfma_max_k() { return 2 * NM_MAX_K; }

// fsf and norm: cubes padded with MF_PAD_FRONT zero channels in front and MF_PAD_BACK behind (the
// pointers are to channel 0).  atab / atab2: tap and tap^2 tables in processing order (glr.hip),
// pinfo the profile of each slot.  Two launches over the halves of the profile list.
int origin_spectral_norm_mfma_launch(origin_ctx *ctx, const float *fsf, const float *norm,
                                     const uint4 *atab, const uint4 *atab2, const int *pinfo, int K,
                                     int Nz, int Ny, int Nx, const uint8_t *mask, float *correl,
                                     uint8_t *profile, float *correl_min, float *part,
                                     bool want_maps, int *nzc_out, float **pmax_out,
                                     float **pmin_out) {
  const long S = (long)Ny * Nx;
  const long bx = cdiv(S, 32 * NW);
  const int ncu = std::max(1, ctx->num_cu);
  int nzm = 1;
  double best_eff = 0.0;
  for (int n = 1; n <= std::min(64, std::max(1, cdiv(Nz, 64))); ++n) {
    const int zc = (cdiv(Nz, n) + 31) / 32 * 32;
    const long blocks = bx * cdiv(Nz, zc);
    const long rounds = (blocks + ncu - 1) / ncu;
    const double eff = (double)bx * Nz / ((double)rounds * ncu * (zc + 32));
    if (eff > best_eff * 1.0001) best_eff = eff, nzm = n;
  }
  const int zcm = (cdiv(Nz, nzm) + 31) / 32 * 32;
  nzm = cdiv(Nz, zcm);
  float *pmax = want_maps ? part : nullptr;
  float *pmin = want_maps ? part + (size_t)nzm * S : nullptr;
  int rc = nm_passes(ctx, fsf, norm, atab, atab2, pinfo, K, Nz, Ny, Nx, mask, correl, profile,
                     correl_min, pmax, pmin, bx, nzm, zcm, 0, Nz, 0);
  if (rc) return rc;
  *nzc_out = nzm;
  *pmax_out = pmax;
  *pmin_out = pmin;
  return ORIGIN_OK;
}

// The 32-channel tiles at either end of the cube only -- [0, zf0) and [zf1, Nz) -- for plans whose
// other channels run the FOLD form of the table kernel (glr_spectral_mfma.hip, NORMW).  Partial
// maps go to rows prow0 (front) and prow0 + 1 .. (back, chunks of 32).
int origin_spectral_norm_mfma_launch_ends(origin_ctx *ctx, const float *fsf, const float *norm,
                                          const uint4 *atab, const uint4 *atab2, const int *pinfo,
                                          int K, int Nz, int Ny, int Nx, const uint8_t *mask,
                                          float *correl, uint8_t *profile, float *correl_min,
                                          float *pmax, float *pmin, int zf0, int zf1, int prow0,
                                          int *rows_out) {
  const long S = (long)Ny * Nx;
  const long bx = cdiv(S, 32 * NW);
  int rows = 0;
  if (zf0 > 0) {
    const int n = cdiv(zf0, 32);
    int rc = nm_passes(ctx, fsf, norm, atab, atab2, pinfo, K, Nz, Ny, Nx, mask, correl, profile,
                       correl_min, pmax, pmin, bx, n, 32, 0, zf0, prow0);
    if (rc) return rc;
    rows += n;
  }
  if (zf1 < Nz) {
    const int n = cdiv(Nz - zf1, 32);
    int rc = nm_passes(ctx, fsf, norm, atab, atab2, pinfo, K, Nz, Ny, Nx, mask, correl, profile,
                       correl_min, pmax, pmin, bx, n, 32, zf1, Nz, prow0 + rows);
    if (rc) return rc;
    rows += n;
  }
  *rows_out = rows;
  return ORIGIN_OK;
}
