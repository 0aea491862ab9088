// GLR spectral stage on the matrix cores (reference muse_origin/lib_origin.py:1046-1060 and
// :1185-1212: T_k = conv_z(cube_fsf, p_k) / sqrt(conv_z(norm_fsf, p_k^2)), correl = max_k,
// profile = first argmax_k, correl_min = min_k; ComputeTGLR.run's mask glue steps.py:781,788).
//
//   num_k[z] = sum_j p_k[j] x[z + lw_k - j]   for 32 output channels z0..z0+31 and the window
//   x[z0-32 .. z0+63] is a banded Toeplitz product  [32 x 96] . [96 x N]  per profile whose data
//   operand is shared by all K profiles (csrc/glr.hip has the derivation and the table layout).
//
// TERMS = 3: v_mfma_f32_32x32x16_f16 on a two-term f16 split of data and taps (Ah Bh + Ah Bl +
//            Al Bh, 22 significant bits, power-of-two tile scale): fp32-class results.
// TERMS = 1: v_mfma_f32_32x32x16_bf16, one MFMA per product, operands rounded to bf16 (no scaling
//            needed): BASELINE config 4's "bf16 GLR" (SURVEY 8c bf16 tolerances).
//
// What round 1's kernel lost and this one does differently (tools/mfma_valu_overlap.hip, measured
// on MI355X: one wave can issue ~5 plain VALU instructions per MFMA for free, v_pk_mul_f32 is an
// anti-lever -- 2 per MFMA stretch the MFMA interval from 32 to 57 cycles):
//  * the profile loop is software-pipelined: the MFMAs of profile k are interleaved, by hand,
//    with the epilogue of profile k-1 (asm volatile statements keep their order): 4.4 (wide
//    profiles, 18 MFMAs) to 6.7 (narrow, 12 MFMAs) VALU instructions per MFMA gap, two
//    accumulators in ping-pong; no s_nop between MFMA results and their readers (the epilogue
//    of a profile starts two MFMAs after the profile's last MFMA);
//  * the epilogue is five plain instructions per output and profile (v_mul, v_cmp_gt,
//    v_cndmask, v_max, v_min: strict '>' keeps the FIRST maximum, lib_origin.py:1210), no packed
//    math.  (A paired variant -- v_max3 / v_min3 over two profiles and the arg-max through a key
//    in the low mantissa bits, 3.5 instructions -- needs four accumulators and 32 registers of
//    1/sqrt(den): with the 48 registers of B fragments it does not fit the 256 registers a wave
//    has at two waves per SIMD, and hipcc splits the file 128/128 as soon as an "a" operand
//    appears, so everything here lives in VGPRs.)
//  * the power-of-two unscaling is applied to the final max / min (32 multiplies per tile), not
//    to every 1/sqrt(den) value;
#include <algorithm>

#include "common.h"
#include "glr_tables.h"

namespace {

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));

template <int I, int N, typename F>
__device__ __forceinline__ void sm_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sm_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ int sm_border_class(int t, int N, int P) {
  const int c = P / 2;
  return t < c ? t : (t > N - 1 - c ? P - 1 - (N - 1 - t) : c);
}

// one MFMA: acc (+)= A . B, A fragment in VGPRs, B fragment in AGPRs
template <int TERMS, bool FIRST>
__device__ __forceinline__ void sm_mma(f32x16 &acc, const u32x4v &a, const u32x4v &b) {
  if constexpr (TERMS == 3) {
    if constexpr (FIRST)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
    else
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  } else {
    if constexpr (FIRST)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
    else
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  }
}

// running state of a tile: first maximum, its profile index, minimum, per accumulator register
struct SmState {
  float best[16], worst[16];
  int bk[16];
};

// Item r of the epilogue of one profile: the five instructions of accumulator register r, as ONE
// asm statement (between separate statements hipcc pads every def -> use pair with an s_nop, 32
// extra issue slots per profile; inside a statement the hardware interlocks are all it takes).
template <int R>
__device__ __forceinline__ void sm_epi_item(const f32x16 &a, const f32x4v (&f)[4], SmState &st,
                                            int k) {
  constexpr int g = R >> 2, q = R & 3;
  float T;
  asm volatile(
      "v_mul_f32 %0, %4, %5\n\t"
      "v_cmp_gt_f32 vcc, %0, %1\n\t"  // strict '>': the first maximum wins      (lib :1210)
      "v_cndmask_b32 %2, %2, %6, vcc\n\t"
      "v_max_f32 %1, %1, %0\n\t"
      "v_min_f32 %3, %3, %0"
      : "=&v"(T), "+v"(st.best[R]), "+v"(st.bk[R]), "+v"(st.worst[R])
      : "v"(a[R]), "v"(f[g][q]), "v"(k)
      : "vcc");
}

constexpr int SM_EPI_ITEMS = 16;

// first epilogue item of MFMA gap j of a stage with NM MFMAs: nothing in gaps 0 and 1 (the
// previous profile's last MFMA must have written its result), the rest spread evenly
constexpr int sm_epi_start(int j, int NM) {
  return j < 2 ? 0 : (j >= NM ? SM_EPI_ITEMS : (SM_EPI_ITEMS * (j - 2)) / (NM - 2));
}

// One pipeline stage: the MFMAs of one profile (A fragments at ak) into acc, interleaved with
// the epilogue of the previous profile (its accumulator pacc, its 1/sqrt(den) values pf, its
// index pk) when EPI.  WIDE: the profile uses all six window blocks (half width > 16), else
// blocks 1..4.
template <int TERMS, bool WIDE, bool EPI>
__device__ __forceinline__ void sm_stage(const char *ak, const u32x4v (&bh)[6],
                                         const u32x4v (&bl)[6], f32x16 &acc, const f32x16 &pacc,
                                         const f32x4v (&pf)[4], SmState &st, int pk) {
  constexpr int LO = 8 * MF_COPY_BYTES;
  constexpr int KS0 = WIDE ? 0 : 1, NK = WIDE ? 6 : 4;
  constexpr int NM = TERMS * NK;
  int pkv = pk;
  asm volatile("" : "+v"(pkv));  // the profile index in a VGPR (v_cndmask source)
  u32x4v ah = *reinterpret_cast<const u32x4v *>(ak + KS0 * 32), al = ah;
  if constexpr (TERMS == 3) al = *reinterpret_cast<const u32x4v *>(ak + KS0 * 32 + LO);
  sm_for<0, NK>([&](auto ic) {
    constexpr int g = decltype(ic)::value, ks = KS0 + g;
    u32x4v nh = ah, nl = al;
    if constexpr (g + 1 < NK) {  // A fragments are requested one block ahead
      nh = *reinterpret_cast<const u32x4v *>(ak + (ks + 1) * 32);
      if constexpr (TERMS == 3) nl = *reinterpret_cast<const u32x4v *>(ak + (ks + 1) * 32 + LO);
    }
    auto gap = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (EPI) {
        sm_for<sm_epi_start(j, NM), sm_epi_start(j + 1, NM)>([&](auto kc) {
          sm_epi_item<decltype(kc)::value>(pacc, pf, st, pkv);
        });
      }
    };
    sm_mma<TERMS, g == 0>(acc, ah, bh[ks]);
    gap(std::integral_constant<int, TERMS * g>{});
    if constexpr (TERMS == 3) {
      sm_mma<TERMS, false>(acc, ah, bl[ks]);
      gap(std::integral_constant<int, TERMS * g + 1>{});
      sm_mma<TERMS, false>(acc, al, bh[ks]);
      gap(std::integral_constant<int, TERMS * g + 2>{});
    }
    ah = nh, al = nl;
  });
}

// the epilogue of the last profile, nothing to overlap it with
__device__ __forceinline__ void sm_drain(const f32x16 &pacc, const f32x4v (&pf)[4], SmState &st,
                                         int pk) {
  int pkv = pk;
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(pkv));  // the last MFMA's result -> VALU readers
  sm_for<0, SM_EPI_ITEMS>([&](auto kc) {
    sm_epi_item<decltype(kc)::value>(pacc, pf, st, pkv);
  });
}

// Everything a wave does: march z in tiles of 32 channels.  BW: the wave touches the field border
// (each lane reads the 1/sqrt(den) values of its own border class from global memory; interior
// waves share the interior class through an LDS table).  A template parameter, not a run-time
// test inside the loop: with a test the two kinds of pointer merge into flat loads.
// MIX / KODD / LASTN: the profile list (narrow ones first) has an odd number of narrow profiles
// followed by a wide one (-> one mixed pair) / an odd length (-> one stage alone at the end) /
// that last one is narrow.  Compile-time too: the tile loop is then loops of ONE stage variant
// each with nothing conditional between them (with run-time choices hipcc shuffles the 48 state
// registers at every merge: 49 v_mov per stage).
template <int TERMS, bool BW, bool MIX, bool KODD, bool LASTN>
__device__ __forceinline__ void sm_tiles(
    const float *__restrict__ fsf, const float *__restrict__ rdb, const float *__restrict__ rdi_s,
    int NzP, const int *__restrict__ pinfo, int K, int nN, int Nz, long S, long s_base, int rr,
    bool sv, int h, int lane, const char *a_lane, char *rd_wave, int zc0, int zc1,
    const uint8_t *__restrict__ mask, float *__restrict__ correl, uint8_t *__restrict__ profile,
    float *__restrict__ correl_min, float &vmax, float &vmin) {
  const char *rd_lane = rd_wave + 16 * h;  // channels 4h..4h+3 of each group of 8
  // Addresses are a wave-uniform base (SGPR pair) plus ONE 32-bit lane offset: the 48 window
  // rows, 48 stores and 16 mask bytes of a tile would otherwise hold a 64-bit pointer each.
  // (The fsf work cube is padded with 32 zero channels in front and 96 behind, origin_glr_run:
  // every window is read without a bounds test.)
  const int off_in = (int)(8 * h * S) + rr;   // window rows 16 ks + 8 h + j
  const int off_out = (int)(4 * h * S) + rr;  // outputs (i&3) + 8 (i>>2) + 4 h
  const int off_rd = (lane >> 5) * NzP + (lane & 31);

  for (int z0 = zc0; z0 < zc1; z0 += 32) {
    // ---- 1/sqrt(den)[slot][z0 .. z0+31] of the interior class -> this wave's LDS table (raw:
    // the power-of-two unscaling is applied to the final max / min).  rdi_s is in processing
    // order [slot][NzP]: element i = lane + 64 q is slot (lane>>5) + 2q, channel lane & 31
    if constexpr (!BW) {
      float rv[MF_MAX_K / 2];
      const float *ub = rdi_s + z0;
#pragma unroll
      for (int q = 0; q < MF_MAX_K / 2; ++q)
        rv[q] = (lane >> 5) + 2 * q < K ? (ub + (long)(2 * q) * NzP)[off_rd] : 0.0f;
#pragma unroll
      for (int q = 0; q < MF_MAX_K / 2; ++q)
        if ((lane >> 5) + 2 * q < K) reinterpret_cast<float *>(rd_wave)[lane + 64 * q] = rv[q];
    }
    // ---- window X[z0-32 .. z0+63] in B-fragment order: lane (r, h) holds rows 16 ks + 8 h + j
    float x[6][8];
    {
      const float *ub = fsf + (long)(z0 - 32) * S + s_base;  // uniform
#pragma unroll
      for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[ks][j] = (ub + (long)(16 * ks + j) * S)[off_in];
    }
    u32x4v bh[6], bl[6];
    float inv = 1.0f;
    if constexpr (TERMS == 3) {
      // power-of-two scale of this tile: max |y| in [2^14, 2^15)
      float m = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(sv ? x[ks][j] : 0.0f));
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      const int ex = (int)((__float_as_uint(m) >> 23) & 0xffu);
      const bool tiny = ex < 40 || ex == 255;  // zero / denormal-small / non-finite: no scaling
      const float scale = __uint_as_float((unsigned)(tiny ? 127 : 268 - ex) << 23);
      // 2^-(e + MF_TAP_SCALE_LOG2): undoes both scalings, exactly
      inv = __uint_as_float((unsigned)(tiny ? 127 - MF_TAP_SCALE_LOG2
                                            : ex - 14 - MF_TAP_SCALE_LOG2) << 23);
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
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bh[ks][j] = __builtin_bit_cast(unsigned, hh[j]);
          bl[ks][j] = __builtin_bit_cast(unsigned, ll[j]);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
          bf2 t;  // round to nearest even (v_cvt_pk_bf16_f32)
          t[0] = (__bf16)x[ks][2 * j];
          t[1] = (__bf16)x[ks][2 * j + 1];
          bh[ks][j] = __builtin_bit_cast(unsigned, t);
          bl[ks][j] = 0u;
        }
    }

    SmState st;
#pragma unroll
    for (int i = 0; i < 16; ++i) st.best[i] = -INFINITY, st.worst[i] = INFINITY, st.bk[i] = 0;

    // ---- profiles, software pipelined; two accumulators in ping-pong: the stage of slot s
    // writes X (s even) or Y (s odd) while the epilogue of slot s-1 reads the other.  The slots
    // are cut into straight-line loops of one stage variant each (narrow pairs, at most one
    // mixed pair, wide pairs): a run-time choice of the variant inside ONE loop makes the
    // compiler shuffle the 48 state registers at every merge.
    f32x16 accX, accY;
    f32x4v fp[4];
    int kp = 0;
    auto load_f = [&](int slot) {
      // 1/sqrt(den) of this profile for the lane's 16 channels: requested at the end of its
      // stage (the epilogue that read the previous values is done), used from the third MFMA
      // gap of the next stage
      kp = pinfo[slot] & 0xff;
      if constexpr (BW) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          fp[g] = *reinterpret_cast<const f32x4v *>(rdb + (long)kp * NzP + z0 + 8 * g);
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          fp[g] = *reinterpret_cast<const f32x4v *>(rd_lane + slot * MF_RD_BYTES + 32 * g);
      }
    };
    auto ak_of = [&](int slot) { return a_lane + slot * MF_PROF_BYTES; };
    // The first stage runs the same code as every other: its "previous profile" is neutral --
    // 1/sqrt(den) = NaN makes T = NaN, which v_cmp_gt rejects and v_max / v_min ignore.
#pragma unroll
    for (int g = 0; g < 4; ++g) fp[g] = (f32x4v){NAN, NAN, NAN, NAN};
    asm volatile("s_nop 1" : "=v"(accY));  // (accY: any bits; B fragment writes -> first MFMA)
    int s = 0;
    for (; s + 1 < nN; s += 2) {  // narrow, narrow
      sm_stage<TERMS, false, true>(ak_of(s), bh, bl, accX, accY, fp, st, kp);
      load_f(s);
      sm_stage<TERMS, false, true>(ak_of(s + 1), bh, bl, accY, accX, fp, st, kp);
      load_f(s + 1);
    }
    if constexpr (MIX) {  // narrow, wide
      sm_stage<TERMS, false, true>(ak_of(s), bh, bl, accX, accY, fp, st, kp);
      load_f(s);
      sm_stage<TERMS, true, true>(ak_of(s + 1), bh, bl, accY, accX, fp, st, kp);
      load_f(s + 1);
      s += 2;
    }
    for (; s + 1 < K; s += 2) {  // wide, wide
      sm_stage<TERMS, true, true>(ak_of(s), bh, bl, accX, accY, fp, st, kp);
      load_f(s);
      sm_stage<TERMS, true, true>(ak_of(s + 1), bh, bl, accY, accX, fp, st, kp);
      load_f(s + 1);
    }
    if constexpr (KODD) {  // the last profile alone (wide unless every profile is narrow)
      if constexpr (LASTN) sm_stage<TERMS, false, true>(ak_of(s), bh, bl, accX, accY, fp, st, kp);
      else sm_stage<TERMS, true, true>(ak_of(s), bh, bl, accX, accY, fp, st, kp);
      load_f(s);
      sm_drain(accX, fp, st, kp);
    } else {
      sm_drain(accY, fp, st, kp);
    }

    // ---- store, mask glue (steps.py:781,788)
    unsigned char mk[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) mk[i] = 0;
    if (mask) {  // branch-free inside: every load is issued before the first is awaited
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        // channel min(zu + 4 h, Nz - 1): uniform base, the lane part clamped at the cube's end
        const int zu = min(z0 + (i & 3) + 8 * (i >> 2), Nz - 1);
        const int hs = (int)(min(4, Nz - 1 - zu) * S);
        mk[i] = (mask + (long)zu * S + s_base)[rr + (h ? hs : 0)];
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int zu = z0 + (i & 3) + 8 * (i >> 2);  // uniform part of the channel
      if (zu + 4 * h < zc1) {
        const long ubase = (long)zu * S + s_base;
        float b = st.best[i] * inv;
        const float w = st.worst[i] * inv;
        int kk = st.bk[i];
        // profiles run narrow-first, not in index order: when every T is the same number (a
        // spaxel of zeros) the first maximum is index 0 (np.argmax semantics, lib :1210)
        if (st.best[i] == st.worst[i]) kk = 0;
        if (mk[i]) b = 0.0f, kk = 0;
        if (sv) {
          (correl + ubase)[off_out] = b;
          (correl_min + ubase)[off_out] = w;
          (profile + ubase)[off_out] = (uint8_t)kk;
        }
        vmax = fmaxf(vmax, b);
        vmin = fminf(vmin, w);
      }
    }
  }
}

template <int TERMS, int VARIANT>
__global__ __launch_bounds__(64 * MF_WAVES, 1) void spectral_mfma2_kernel(
    const float *__restrict__ fsf, const float *__restrict__ rden,
    const float *__restrict__ rdi_s, int NzP, const uint4 *__restrict__ atab,
    const int *__restrict__ pinfo, int K, int nN, int Nz, int Ny, int Nx, int P, int zchunk,
    const uint8_t *__restrict__ mask, float *__restrict__ correl, uint8_t *__restrict__ profile,
    float *__restrict__ correl_min, float *__restrict__ part_max, float *__restrict__ part_min) {
  extern __shared__ __align__(16) char sm_lds[];
  {
    const int nvec = K * (MF_PROF_BYTES / 16);
    for (int i = threadIdx.x; i < nvec; i += 64 * MF_WAVES)
      reinterpret_cast<uint4 *>(sm_lds)[i] = atab[i];
  }
  __syncthreads();
  const long S = (long)Ny * Nx;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, in an SGPR
  const int r = lane & 31, h = lane >> 5;
  const long s_base = ((long)blockIdx.x * MF_WAVES + wv) * 32;
  if (s_base >= S) return;  // whole wave; no barrier follows
  const int zc0 = blockIdx.y * zchunk, zc1 = min(Nz, zc0 + zchunk);
  const int E0 = 8 * h - r + 31;
  const char *a_lane = sm_lds + (E0 & 7) * MF_COPY_BYTES + (E0 >> 3) * 16;
  // this wave's [K][32] table of 1/sqrt(den) for the current tile (behind the tap copies), in
  // the order the profiles are processed
  char *rd_wave = sm_lds + K * MF_PROF_BYTES + wv * K * MF_RD_BYTES;
  const bool sv = s_base + r < S;
  const long sc = sv ? s_base + r : S - 1;
  const int rr = (int)(sc - s_base);  // r, clamped for lanes past the field
  float vmax = -INFINITY, vmin = INFINITY;
  // normalisation class of this lane's spaxel (how the field border clips the PSF window)
  const int ccls = (P / 2) * P + P / 2;
  int cls;
  {
    const int y = (int)(sc / Nx), xx = (int)(sc - (long)y * Nx);
    cls = sm_border_class(y, Ny, P) * P + sm_border_class(xx, Nx, P);
  }
  const float *rdb = rden + (long)cls * K * NzP + 4 * h;
  // VARIANT: bit 0 = mixed narrow/wide pair, bit 1 = odd profile count, bit 2 = that last
  // profile is narrow
  constexpr bool MIX = (VARIANT & 1) != 0, KODD = (VARIANT & 2) != 0, LASTN = (VARIANT & 4) != 0;
  if (__any(cls != ccls))
    sm_tiles<TERMS, true, MIX, KODD, LASTN>(fsf, rdb, rdi_s, NzP, pinfo, K, nN, Nz, S, s_base, rr,
                                            sv, h, lane, a_lane, rd_wave, zc0, zc1, mask, correl,
                                            profile, correl_min, vmax, vmin);
  else
    sm_tiles<TERMS, false, MIX, KODD, LASTN>(fsf, rdb, rdi_s, NzP, pinfo, K, nN, Nz, S, s_base, rr,
                                             sv, h, lane, a_lane, rd_wave, zc0, zc1, mask, correl,
                                             profile, correl_min, vmax, vmin);
  if (part_max) {
    const float a = fmaxf(vmax, __shfl_xor(vmax, 32));
    const float b = fminf(vmin, __shfl_xor(vmin, 32));
    if (h == 0 && sv) {
      part_max[(long)blockIdx.y * S + sc] = a;
      part_min[(long)blockIdx.y * S + sc] = b;
    }
  }
}

}  // namespace

// Launch: a wave = 32 spaxels x 32-channel tiles; z chunks sized to give every CU several blocks
// (one 8-wave block per CU at a time: K * (5 KiB + 8 * 128 B) of LDS).  Returns the number of z
// chunks (rows of part_max / part_min) in *nzc.
int origin_spectral_mfma_launch(origin_ctx *ctx, int terms, const float *fsf, const float *rden,
                                const float *rdi_s, int NzP, const uint4 *atab, const int *pinfo, int K, int nN, int Nz,
                                int Ny, int Nx, int P, const uint8_t *mask, float *correl,
                                uint8_t *profile, float *correl_min, float *part, bool want_maps,
                                int *nzc_out, float **pmax_out, float **pmin_out) {
  const long S = (long)Ny * Nx;
  const long bx = cdiv(S, 32 * MF_WAVES);
  int nzm = (int)(((long)ctx->num_cu * 8 + bx - 1) / bx);
  nzm = std::max(1, std::min(nzm, std::min(64, cdiv(Nz, 64))));
  const int zcm = (cdiv(Nz, nzm) + 31) / 32 * 32;
  nzm = cdiv(Nz, zcm);
  float *pmax = want_maps ? part : nullptr;
  float *pmin = want_maps ? part + (size_t)nzm * S : nullptr;
  const size_t lds = (size_t)K * (MF_PROF_BYTES + MF_WAVES * MF_RD_BYTES);
  // stage sequence: narrow pairs, [narrow + wide], wide pairs, [one alone]
  const bool mix = (nN & 1) && nN < K;
  const bool kodd = ((K - (mix ? nN + 1 : nN)) & 1) != 0 || (!mix && (nN & 1));
  const bool lastn = kodd && nN == K;
  const int variant = (mix ? 1 : 0) | (kodd ? 2 : 0) | (lastn ? 4 : 0);
  const void *fn = nullptr;
#define SM_PICK(T, V) \
  if (terms == T && variant == V) fn = (const void *)spectral_mfma2_kernel<T, V>
  SM_PICK(3, 0); SM_PICK(3, 1); SM_PICK(3, 2); SM_PICK(3, 3); SM_PICK(3, 6);
  SM_PICK(1, 0); SM_PICK(1, 1); SM_PICK(1, 2); SM_PICK(1, 3); SM_PICK(1, 6);
#undef SM_PICK
  if (!fn) {
    origin_set_error("spectral MFMA kernel: no variant %d for %d terms", variant, terms);
    return ORIGIN_E_STATE;
  }
  ORIGIN_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 MF_MAX_K * (MF_PROF_BYTES + MF_WAVES * MF_RD_BYTES)));
  const float *a_fsf = fsf, *a_rden = rden, *a_rdi = rdi_s;
  const uint4 *a_atab = atab;
  const int *a_pinfo = pinfo;
  const uint8_t *a_mask = mask;
  int a_NzP = NzP, a_K = K, a_nN = nN, a_Nz = Nz, a_Ny = Ny, a_Nx = Nx, a_P = P, a_zcm = zcm;
  void *args[] = {&a_fsf, &a_rden, &a_rdi, &a_NzP, &a_atab, &a_pinfo, &a_K, &a_nN, &a_Nz, &a_Ny,
                  &a_Nx, &a_P, &a_zcm, &a_mask, &correl, &profile, &correl_min, &pmax, &pmin};
  ORIGIN_HIP(hipLaunchKernel(fn, dim3((unsigned)bx, nzm), dim3(64 * MF_WAVES), args, lds,
                             ctx->stream));
  ORIGIN_LAUNCH_CHECK();
  *nzc_out = nzm;
  *pmax_out = pmax;
  *pmin_out = pmin;
  return ORIGIN_OK;
}
