// GLR spectral stage on the matrix cores (reference muse_origin/lib_origin.py:1046-1060 and
// :1185-1212: T_k = conv_z(cube_fsf, p_k) / sqrt(conv_z(norm_fsf, p_k^2)), correl = max_k,
// profile = first argmax_k, correl_min = min_k; ComputeTGLR.run's mask glue steps.py:781,788).
//
//   num_k[z] = sum_j p_k[j] x[z + lw_k - j]   is a banded Toeplitz product per profile whose
//   data operand is shared by all K profiles (csrc/glr.hip has the derivation and the layout of
//   the tap table).  Here a tile is 16 output channels z0..z0+15 and the 32 rows of the MFMA hold
//   a PAIR of profiles: rows 0-15 profile a, rows 16-31 profile b, against the window
//   x[z0-32 .. z0+47] (five 16-channel blocks; pairs of narrow profiles -- half width <= 16 --
//   only need blocks 1..3).  Against 32-channel tiles of one profile this
//     * trims the band: 3 or 5 k-steps per 16 outputs and pair instead of 4 or 6 per 32 outputs
//       and profile -- 240 instead of 306 MFMAs per 32 channels for Dico_FWHM_2_12;
//     * puts T_a and T_b of the same output into ONE lane (accumulator registers o and o + 8), so
//       that the running maximum / minimum over the pair is one v_max3 / v_min3 -- 3.5 instead of
//       5 VALU instructions per output and profile -- without a second accumulator;
//     * halves every per-tile register set (16 accumulator, 24 state, 16 normalisation, 40 B
//       registers): ~150 registers, two waves per SIMD with room to spare, where the paired
//       epilogue on 32-row tiles did not fit.
//
// TERMS = 3: v_mfma_f32_32x32x16_f16 on a two-term f16 split of data and taps (Ah Bh + Ah Bl +
//            Al Bh, 22 significant bits, power-of-two tile scale): fp32-class results.
// TERMS = 1: v_mfma_f32_32x32x16_bf16, one MFMA per product, operands rounded to bf16 (no scaling
//            needed): BASELINE config 4's "bf16 GLR" (SURVEY 8c bf16 tolerances).
//
// Measured on MI355X (tools/mfma_valu_overlap.hip; profiles/r02_glr_pmc.json): one wave issues ~5
// plain VALU instructions per MFMA for free, v_pk_mul_f32 is an anti-lever (2 per MFMA stretch
// the MFMA interval from 32 to 57 cycles), and the first pipelined version of this kernel was
// VALU-issue bound (2500 VALU + 306 MFMA instructions per 32 x 32 outputs at 4.1 cycles each =
// 11 600 cycles against 9 800 of matrix work).  So:
//  * the pair loop is software-pipelined by hand: the MFMAs of pair p are interleaved with the
//    epilogue of pair p-1 (asm volatile statements keep their order; an epilogue item is ONE
//    statement of seven instructions: between separate statements hipcc pads def -> use pairs
//    with s_nop), two accumulators in ping-pong, no s_nop between MFMA results and their
//    readers (an epilogue starts two MFMAs after the pair's last MFMA);
//  * arg-max through a KEY: T with its 5 low mantissa bits replaced by 31 - k; one v_max3 over
//    the keys carries the index of the first maximum.  The key only decides between profiles
//    whose T differ by less than 2^-18 relative (such a pair can come out with the other index:
//    counted in the arg-max mismatch rate the tests bound, <= 1e-4); correl / correl_min are
//    exact maxima / minima (v_max3 / v_min3 on the unmodified T);
//  * the power-of-two unscaling is applied to the final max / min, not to every 1/sqrt(den);
//  * FOLD (round 3): 1/sqrt(den_k[z]) = a_k s(z) (1 + eps_k(z)) with a_k = 1/sqrt(sum_j p_k[j]^2),
//    s a factor of the lane's border class alone and |eps| a few 1e-7 wherever the PSF varies
//    smoothly with the channel and the profile's support lies inside the cube (den_k is norm_fsf
//    smoothed by p_k^2: what depends on k is the curvature of norm_fsf times the profile's
//    variance).  The plan measures eps on its own tables (fold_tables_kernel, glr.hip: s = the
//    middle of the range over k, eps = half its relative width; 5.4e-7 for the benchmark's Moffat
//    PSF and dictionary); where it is <= MF_FOLD_EPS = 2e-6 the taps carry a_k and the pair loop
//    compares the bare accumulators -- 5 instead of 7 VALU instructions per output and pair, no
//    table of 1/sqrt(den) in LDS, no values of it in registers -- and max and min take s(z) once,
//    behind the loop: T carries a relative error <= eps (the f16 split's own is ~3e-7 of the
//    window's largest product; making the maximum exact with its profile's own 1/sqrt(den) was
//    measured: a gather of eight values per half tile from global memory or from a per-tile LDS
//    table costs 1.8 - 2.3 ms, more than the two multiplies had).  The 32 channels at each end of
//    the cube (profile support cut: eps ~ 0.2) run the exact pair loop on the same folded taps
//    with the table 1/(a_k sqrt(den)); plans that fail the test, or ORIGIN_GLR_NO_FOLD=1, run the
//    exact form everywhere.  What the freed registers buy: the next tile's two new window blocks
//    are requested a tile ahead through LDS (global_load_lds_dword: sixteen staging rows per wave
//    where the 1/sqrt(den) tables were) and a stage's first A fragments by the stage before.
//    11.2 -> 9.5 ms at 3681 x 600 x 600 (the exact form itself 11.2 -> 10.9 with the pointer
//    stepping that came along).
//  * everything lives in VGPRs: hipcc splits the register file 128 / 128 as soon as an "a"
//    operand appears in inline asm;
//  * the window's f16 fragments are carried from tile to tile: consecutive 32-channel tiles share
//    four of their six 16-channel blocks, so a tile loads and converts two blocks instead of
//    six (kept fragments are multiplied by the ratio of the tile scales, a power of two within
//    2^+-4, exact in f16; a larger step re-converts the window): 12.4 -> 11.5 ms.
#include <algorithm>

#include "common.h"
#include "glr_tables.h"

namespace {

// Measured at 3681 x 600 x 600: two waves per SIMD with the next tile's window prefetched into
// registers (240 VGPRs) 12.6 ms, three waves per SIMD without it (168 VGPRs) 11.7 ms.
constexpr bool SM_PREFETCH = MF_WAVES <= 8;

// -DSM_TIMING: clock64 stamps of block (3, 1), every wave, tiles 2..9 of the chunk's FOLD range
// (tools/sm_phase_times.py; a stamp is an s_memtime round trip: read the numbers as shares)
#ifdef SM_TIMING
__device__ long long sm_wave[4 * MF_WAVES * 4];  // blocks (3..6, 1): per wave start, end, columns taken
__device__ long long sm_tim[8 * MF_WAVES * 8];
#define SM_STAMP(k)                                                                       \
  if (blockIdx.x == 3 && blockIdx.y == 1 && lane == 0 && sm_tile >= 2 && sm_tile < 10)    \
  sm_tim[((sm_tile - 2) * MF_WAVES + (threadIdx.x >> 6)) * 8 + (k)] = clock64()
#else
#define SM_STAMP(k)
#endif

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

// running state of a 16-channel tile: exact maximum / minimum and the arg-max key of each of the
// lane's 8 outputs
struct SmState {
  float best[8], worst[8], key[8];
};

// Epilogue item o of a profile pair: output o of the lane is accumulator register o (profile a)
// and o + 8 (profile b).
// EPI_FIRST (FOLD, round 4): the epilogue of a half tile's FIRST pair writes the state instead of
// updating it -- the same five instructions with -inf / +inf (scalar operands) in the state's
// place, so that the state needs no initialisation (24 v_mov per half tile).
constexpr int EPI_UPDATE = 0, EPI_FIRST = 1, EPI_NONE = 2;
template <int O, bool FOLD, int MODE = EPI_UPDATE>
__device__ __forceinline__ void sm_epi_item(const f32x16 &acc, const f32x4v (&fa)[2],
                                            const f32x4v (&fb)[2], SmState &st, unsigned maskv,
                                            int ca, int cb) {
  constexpr int g = O >> 2, q = O & 3;
  float T0, T1, K0, K1;
  if constexpr (MODE == EPI_NONE) return;
  if constexpr (FOLD && MODE == EPI_FIRST) {
    // (a VOP3 instruction reads ONE scalar register: the and-ors take ca / cb, the extrema the
    // infinities; NaN accumulators are ignored by v_max3 / v_min3 as in the update form)
    const int ninf = 0xff800000, pinf = 0x7f800000;
    asm volatile(
        "v_and_or_b32 %0, %5, %7, %8\n\t"
        "v_and_or_b32 %1, %6, %7, %9\n\t"
        "v_max3_f32 %2, %10, %0, %1\n\t"
        "v_max3_f32 %3, %10, %5, %6\n\t"
        "v_min3_f32 %4, %11, %5, %6"
        : "=&v"(K0), "=&v"(K1), "=&v"(st.key[O]), "=&v"(st.best[O]), "=&v"(st.worst[O])
        : "v"(acc[O]), "v"(acc[O + 8]), "v"(maskv), "s"(ca), "s"(cb), "s"(ninf), "s"(pinf));
    return;
  }
  if constexpr (FOLD) {  // the taps carry a_k: the accumulators are compared as they are
    asm volatile(
        "v_and_or_b32 %0, %5, %7, %8\n\t"
        "v_and_or_b32 %1, %6, %7, %9\n\t"
        "v_max3_f32 %2, %2, %0, %1\n\t"
        "v_max3_f32 %3, %3, %5, %6\n\t"
        "v_min3_f32 %4, %4, %5, %6"
        : "=&v"(K0), "=&v"(K1), "+v"(st.key[O]), "+v"(st.best[O]), "+v"(st.worst[O])
        : "v"(acc[O]), "v"(acc[O + 8]), "v"(maskv), "s"(ca), "s"(cb));
    return;
  }
  asm volatile(
      "v_mul_f32 %0, %7, %8\n\t"
      "v_mul_f32 %1, %9, %10\n\t"
      "v_and_or_b32 %2, %0, %11, %12\n\t"
      "v_and_or_b32 %3, %1, %11, %13\n\t"
      "v_max3_f32 %4, %4, %2, %3\n\t"
      "v_max3_f32 %5, %5, %0, %1\n\t"
      "v_min3_f32 %6, %6, %0, %1"
      : "=&v"(T0), "=&v"(T1), "=&v"(K0), "=&v"(K1), "+v"(st.key[O]), "+v"(st.best[O]),
        "+v"(st.worst[O])
      : "v"(acc[O]), "v"(fa[g][q]), "v"(acc[O + 8]), "v"(fb[g][q]), "v"(maskv), "s"(ca), "s"(cb));
}

constexpr int SM_EPI_ITEMS = 8;

// first epilogue item of MFMA gap j of a stage with NM MFMAs: nothing in gaps 0 and 1 (the
// previous pair's last MFMA must have written its result), the rest spread evenly
constexpr int sm_epi_start(int j, int NM) {
  return j < 2 ? 0 : (j >= NM ? SM_EPI_ITEMS : (SM_EPI_ITEMS * (j - 2)) / (NM - 2));
}

// One pipeline stage: the MFMAs of one profile pair (A fragments at ak: the lane's own profile
// of the pair) into acc, interleaved with the epilogue of the previous pair (its accumulator
// pacc, its 1/sqrt(den) values pfa / pfb, its keys' index parts pca / pcb).  Window blocks
// OFF+1 .. OFF+3 always; blocks OFF+0 and OFF+4 after them when the pair is wide (a wave-uniform
// branch around six bare MFMAs: one stage body for every pair, so the pair loop is ONE loop and
// the state registers never move; the epilogue sits in the gaps of the nine mandatory MFMAs).
// OFF: 0 for the first 16 channels of a 32-channel tile, 1 for the second (window one block on).
// FOLD (the registers of the 1/sqrt(den) values are free): the fragments of block 1 arrive in
// fh / fl -- requested by the stage before, behind its own last request -- and leave as the next
// stage's (ak_next), so that no stage opens with a wait for LDS.
template <int TERMS, int OFF, bool FOLD, int MODE = EPI_UPDATE>
__device__ __forceinline__ void sm_stage(const char *ak, const char *ak_next, u32x4v &fh,
                                         u32x4v &fl, bool wide, const u32x4v (&bh)[6],
                                         const u32x4v (&bl)[6], f32x16 &acc, const f32x16 &pacc,
                                         const f32x4v (&pfa)[2], const f32x4v (&pfb)[2],
                                         SmState &st, unsigned maskv, int pca, int pcb) {
  constexpr int LO = 8 * MF_COPY_BYTES;
  constexpr int NM = TERMS * 3;
  // A fragments of window blocks 1, 2, 3 (mandatory), requested one block ahead
  u32x4v ah, al;
  if constexpr (FOLD) {
    ah = fh, al = fl;
  } else {
    ah = *reinterpret_cast<const u32x4v *>(ak + 32), al = ah;
    if constexpr (TERMS == 3) al = *reinterpret_cast<const u32x4v *>(ak + 32 + LO);
  }
  sm_for<0, 3>([&](auto ic) {
    constexpr int g = decltype(ic)::value, ks = 1 + g;
    u32x4v nh = ah, nl = al;
    if constexpr (g + 1 < 3) {
      nh = *reinterpret_cast<const u32x4v *>(ak + (ks + 1) * 32);
      if constexpr (TERMS == 3) nl = *reinterpret_cast<const u32x4v *>(ak + (ks + 1) * 32 + LO);
    } else if constexpr (FOLD) {
      nh = *reinterpret_cast<const u32x4v *>(ak_next + 32), nl = nh;
      if constexpr (TERMS == 3) nl = *reinterpret_cast<const u32x4v *>(ak_next + 32 + LO);
    }
    auto gap = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      sm_for<sm_epi_start(j, NM), sm_epi_start(j + 1, NM)>([&](auto kc) {
        sm_epi_item<decltype(kc)::value, FOLD, MODE>(pacc, pfa, pfb, st, maskv, pca, pcb);
      });
    };
    sm_mma<TERMS, g == 0>(acc, ah, bh[OFF + ks]);
    gap(std::integral_constant<int, TERMS * g>{});
    if constexpr (TERMS == 3) {
      sm_mma<TERMS, false>(acc, ah, bl[OFF + ks]);
      gap(std::integral_constant<int, TERMS * g + 1>{});
      sm_mma<TERMS, false>(acc, al, bh[OFF + ks]);
      gap(std::integral_constant<int, TERMS * g + 2>{});
    }
    ah = nh, al = nl;
  });
  if constexpr (FOLD) fh = ah, fl = al;
  if (wide) {  // window blocks 0 and 4
    const u32x4v a0h = *reinterpret_cast<const u32x4v *>(ak);
    const u32x4v a4h = *reinterpret_cast<const u32x4v *>(ak + 4 * 32);
    if constexpr (TERMS == 3) {
      const u32x4v a0l = *reinterpret_cast<const u32x4v *>(ak + LO);
      const u32x4v a4l = *reinterpret_cast<const u32x4v *>(ak + 4 * 32 + LO);
      sm_mma<TERMS, false>(acc, a0h, bh[OFF + 0]);
      sm_mma<TERMS, false>(acc, a0h, bl[OFF + 0]);
      sm_mma<TERMS, false>(acc, a0l, bh[OFF + 0]);
      sm_mma<TERMS, false>(acc, a4h, bh[OFF + 4]);
      sm_mma<TERMS, false>(acc, a4h, bl[OFF + 4]);
      sm_mma<TERMS, false>(acc, a4l, bh[OFF + 4]);
    } else {
      sm_mma<TERMS, false>(acc, a0h, bh[OFF + 0]);
      sm_mma<TERMS, false>(acc, a4h, bh[OFF + 4]);
    }
  }
}

// the epilogue of the last pair, nothing to overlap it with
template <bool FOLD>
__device__ __forceinline__ void sm_drain(const f32x16 &pacc, const f32x4v (&pfa)[2],
                                         const f32x4v (&pfb)[2], SmState &st, unsigned maskv,
                                         int pca, int pcb) {
  unsigned mv = maskv;
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(mv));  // the last MFMA's result -> VALU readers
  sm_for<0, SM_EPI_ITEMS>([&](auto kc) {
    sm_epi_item<decltype(kc)::value, FOLD>(pacc, pfa, pfb, st, mv, pca, pcb);
  });
}

// Everything a wave does: march z in tiles of 32 channels -- one load + conversion of the
// 96-channel window x[z0-32 .. z0+63] (six blocks), then the two 16-channel halves, each with its
// own pass over the profile pairs (window blocks 0..4 and 1..5).  BW: the wave touches the field
// border (each lane reads the 1/sqrt(den) values of its own border class from global memory;
// interior waves share the interior class through an LDS table).  A template parameter, not a
// run-time test inside the loop: with a test the two kinds of pointer merge into flat loads.
// PODD: the number of profile pairs is odd (the two accumulators alternate per pair; the pair
// loop is unrolled by two, and an odd count leaves one stage behind it).
template <int TERMS, bool BW, bool PODD, bool FOLD, bool IDENT = false, bool NORMW = false>
__device__ __forceinline__ void sm_tiles(
    const float *__restrict__ sdl, const float *__restrict__ fsf, const float *__restrict__ rdb, const float *__restrict__ rdi_s,
    int NzP, const int *__restrict__ pinfo, int K, int NP, int Nz, long S, long s_base, int rr,
    bool sv, bool all_valid, int h, int lane, const char *a_lane, char *rd_wave, int zc0, int zc1,
    const uint8_t *__restrict__ mask, float *__restrict__ correl, uint8_t *__restrict__ profile,
    float *__restrict__ correl_min, float &vmax, float &vmin, int nN, char *stage = nullptr) {
  // (NORMW, with FOLD: plans with an explicit norm cube -- weighted mosaics.  sdl is that cube
  // (channel 0): den_k[z, s] = a_k^-2 norm[z, s] (1 + eps)^-2 away from the cube's ends, so the
  // factor behind the loop is rsq(norm) of the lane's own voxels instead of a class table's.)
  // (IDENT: the processing order is the caller's order -- slot = index, wide from slot nN on: no
  // look-up per pair, and with no scalar load outstanding the waits for LDS are counted ones)
  // (stage: FOLD, the wave's 16 staging rows of 256 bytes in LDS)
  const char *rd_lane = rd_wave + 16 * h;  // channels 4h..4h+3 of each group of 8
  const bool second = (lane & 16) != 0;    // this lane's A rows belong to the pair's profile b
  // Addresses are a wave-uniform base (SGPR pair) plus ONE 32-bit lane offset: the window rows,
  // stores and mask bytes of a tile would otherwise hold a 64-bit pointer each.  (The fsf work
  // cube is padded with 32 zero channels in front and 64 behind, origin_glr_run: every window
  // is read without a bounds test.)
  // (unsigned BYTE offsets: global_load / global_store take an SGPR base plus a zero-extended
  // 32-bit VGPR byte offset -- with element offsets hipcc adds base and offset on the VALU)
  const unsigned off_in = 4u * ((unsigned)(8 * h * S) + rr);   // window rows 16 ks + 8 h + j
  const unsigned off_out = (unsigned)(4 * h * S) + rr;         // outputs (o&3) + 8 (o>>2) + 4 h
  const unsigned off_out4 = 4u * off_out;
  const unsigned off_rd = 4u * (unsigned)((lane >> 5) * NzP + (lane & 31));
  // (the empty asm keeps the zero-extension of the offset in the basic block of the access:
  // hoisted out of the loop it becomes a 64-bit VGPR pair, and the instruction selector no longer
  // sees "SGPR base + zext(VGPR)" -- one v_lshl_add_u64 per load)
  auto ldf = [](const float *base, unsigned boff) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + boff);
  };
  auto stf = [](float *base, unsigned boff, float v) {
    *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + boff) = v;
  };
#define SM_PIN(v) asm volatile("" : "+v"(v))  // "defined here": once per straight-line block
  unsigned maskv = 0xffffffe0u;
  asm volatile("" : "+v"(maskv));  // in a VGPR: a VOP3 instruction takes one SGPR, no literal

  auto load_window = [&](int z0, float (&w)[6][8]) {
    const float *ub = fsf + (long)(z0 - 32) * S + s_base;  // uniform, one row on per load
    unsigned oin = off_in;
    SM_PIN(oin);
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        w[ks][j] = ldf(ub, oin);
        ub += S;
      }
      ub += 8 * S;
    }
  };
  u32x4v bh[6], bl[6];  // the window's fragments, carried from tile to tile
  auto load_new_blocks = [&](int z0n, float (&xb)[2][8]) {  // channels z0n+32 .. z0n+63
    const float *ub = fsf + (long)(z0n + 32) * S + s_base;
    unsigned oin = off_in;
    SM_PIN(oin);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xb[ks][j] = ldf(ub, oin);
        ub += S;
      }
      ub += 8 * S;
    }
  };
  // FOLD: the next tile's two new blocks.  A tile's global loads are awaited right where they
  // are issued -- the pair loop leaves no registers to keep sixteen values in flight (three waves
  // per SIMD: 168 VGPRs) -- so they go through LDS instead: global_load_lds_dword writes lane l's
  // dword to M0 + 4 l without touching a VGPR.  Sixteen of them (one 256-byte staging row each, in
  // the space the exact form's 1/sqrt(den) tables take) are issued when a tile starts and read
  // when the next one does.  By then they have landed: the second half's stores waited for its
  // s(z) loads, which were issued after the sixteen, and loads complete in order; the s_waitcnt
  // in front of the reads says the same in the counter's terms (a half that stores without tests
  // issues exactly 24 stores behind those loads).
  const unsigned stage_lds = (unsigned)(uintptr_t)stage;  // LDS byte address (uniform)
  auto stage_new_blocks = [&](int z0n) {  // channels z0n+32 .. z0n+63
    const float *ub = fsf + (long)(z0n + 32) * S + s_base;
    unsigned m = __builtin_amdgcn_readfirstlane(stage_lds);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %1\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dword %2, %3\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "s"(m), "v"(off_in), "s"(ub)
            : "memory");
        ub += S;
        m += 256;
      }
      ub += 8 * S;
    }
  };
  auto take_staged_blocks = [&](float (&xb)[2][8]) {
    if (all_valid) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float *sp = reinterpret_cast<const float *>(stage) + lane;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) xb[ks][j] = sp[64 * (8 * ks + j)];
    asm volatile("" ::: "memory");
  };
  int bmx[6] = {0, 0, 0, 0, 0, 0};  // max |x| of each window block (float bits, wave-uniform)
  int se_cur = 127;                 // exponent field of the scale the fragments are stored under
  int sm_tile = -1;
  (void)sm_tile;
  for (int z0 = zc0; z0 < zc1; z0 += 32) {
    ++sm_tile;
    SM_STAMP(0);
    // ---- 1/sqrt(den)[slot][z0 .. z0+31] of the interior class -> this wave's LDS table (raw:
    // the power-of-two unscaling is applied to the final max / min).  rdi_s is in processing
    // order [slot][NzP]: element i = lane + 64 q is slot (lane>>5) + 2q, channel lane & 31
    if constexpr (!BW && !FOLD) {
      float rv[MF_MAX_K / 2];
      const float *ub = rdi_s + z0;
      unsigned ord = off_rd;
      SM_PIN(ord);
#pragma unroll
      for (int q = 0; q < MF_MAX_K / 2; ++q) {
        rv[q] = (lane >> 5) + 2 * q < K ? ldf(ub, ord) : 0.0f;
        ub += 2 * NzP;
      }
#pragma unroll
      for (int q = 0; q < MF_MAX_K / 2; ++q)
        if ((lane >> 5) + 2 * q < K) reinterpret_cast<float *>(rd_wave)[lane + 64 * q] = rv[q];
    }
    // ---- window X[z0-32 .. z0+63] in B-fragment order: lane (n, h) holds rows 16 ks + 8 h + j,
    // as f16 hi / lo fragments (bh, bl) under the tile's power-of-two scale.  Consecutive tiles
    // overlap by four of the six 16-channel blocks: those fragments are KEPT (moved two blocks
    // down, multiplied by the ratio of the two scales when the window maximum crossed a power of
    // two -- exact in f16 short of underflow) and only the two new blocks are loaded and
    // converted.  The first tile of a chunk, and a tile whose scale differs from its
    // predecessor's by more than 2^4 (or is not finite), loads and converts all six blocks:
    // fragments kept across a large jump would have lost their low bits under the old scale.
    float inv = 1.0f;
    bool full = z0 == zc0;
    auto scale_exp = [](float m, int &se, bool &finite) {
      const int ex = (int)((__float_as_uint(m) >> 23) & 0xffu);
      const bool tiny = ex < 40 || ex == 255;  // zero / denormal-small / non-finite: no scaling
      finite = ex != 255;
      se = tiny ? 127 : 268 - ex;  // scale = 2^(se - 127): max |y| in [2^14, 2^15)
    };
    auto wave_max = [&](float m) {
      if (!all_valid) m = sv ? m : 0.0f;  // lanes past the end of the field hold a copy
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      return __builtin_amdgcn_readfirstlane(__float_as_int(m));  // (>= 0: integer order = float order)
    };
    auto convert_block = [&](const float (&xb)[8], float scale, u32x4v &oh, u32x4v &ol) {
      if constexpr (TERMS == 3) {
        f16x2v hh[4], ll[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float y = xb[j] * scale;
          const _Float16 yh = (_Float16)y;
          hh[j >> 1][j & 1] = yh;
          ll[j >> 1][j & 1] = (_Float16)(y - (float)yh);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          oh[j] = __builtin_bit_cast(unsigned, hh[j]);
          ol[j] = __builtin_bit_cast(unsigned, ll[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
          bf2 t;  // round to nearest even (v_cvt_pk_bf16_f32)
          t[0] = (__bf16)xb[2 * j];
          t[1] = (__bf16)xb[2 * j + 1];
          oh[j] = __builtin_bit_cast(unsigned, t);
          ol[j] = 0u;
        }
      }
    };
    if (!full) {
      // the two new blocks: window rows 64 .. 95 = channels z0+32 .. z0+63.  (Requesting them
      // one tile ahead, in front of the previous tile's stores, was measured: 13.9 ms against
      // 11.4 -- sixteen registers carried around the tile loop cost more than the wait.)
      float xb[2][8];
      if constexpr (FOLD) take_staged_blocks(xb);
      else load_new_blocks(z0, xb);
      int se = 127;
      if constexpr (TERMS == 3) {
        // (one v_max3 with |.| modifiers per two values: fmaxf(m, fabsf(x)) is a canonicalising
        // maximum per value)
        float m4 = 0.0f, m5 = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m4) : "v"(xb[0][j]), "v"(xb[0][j + 1]));
          asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m5) : "v"(xb[1][j]), "v"(xb[1][j + 1]));
        }
        bmx[0] = bmx[2], bmx[1] = bmx[3], bmx[2] = bmx[4], bmx[3] = bmx[5];
        bmx[4] = wave_max(m4), bmx[5] = wave_max(m5);
        int mm = max(max(max(bmx[0], bmx[1]), max(bmx[2], bmx[3])), max(bmx[4], bmx[5]));
        bool finite;
        scale_exp(__int_as_float(mm), se, finite);
        const int d = se - se_cur;
        if (!finite || d > 4 || d < -4) full = true;  // (uniform)
        else if (d != 0) {
          // kept fragments: times 2^d, eight halves of a fragment register quad at a time
          // (element-wise updates of the unsigned vectors were mis-compiled by hipcc 7.2: element
          // 1 came out as element 0 times the factor and was copied to elements 2 and 3)
          typedef _Float16 h8 __attribute__((ext_vector_type(8)));
          const _Float16 f1 = __builtin_bit_cast(_Float16, (unsigned short)((15 + d) << 10));
          const h8 fac = {f1, f1, f1, f1, f1, f1, f1, f1};
#pragma unroll
          for (int ks = 2; ks < 6; ++ks) {
            bh[ks] = __builtin_bit_cast(u32x4v, __builtin_bit_cast(h8, bh[ks]) * fac);
            bl[ks] = __builtin_bit_cast(u32x4v, __builtin_bit_cast(h8, bl[ks]) * fac);
          }
        }
      }
      if (!full) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bh[ks] = bh[ks + 2], bl[ks] = bl[ks + 2];
        const float scale = __uint_as_float((unsigned)se << 23);
        convert_block(xb[0], scale, bh[4], bl[4]);
        convert_block(xb[1], scale, bh[5], bl[5]);
        se_cur = se;
      }
    }
    if (full) {
      float x[6][8];
      load_window(z0, x);
      int se = 127;
      if constexpr (TERMS == 3) {
        int mm = 0;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
          float m = 0.0f;
#pragma unroll
          for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(x[ks][j]));
          bmx[ks] = wave_max(m);
          mm = max(mm, bmx[ks]);
        }
        bool finite;
        scale_exp(__int_as_float(mm), se, finite);
      }
      const float scale = __uint_as_float((unsigned)se << 23);
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) convert_block(x[ks], scale, bh[ks], bl[ks]);
      se_cur = se;
    }
    if constexpr (TERMS == 3) {
      // 2^-(e + MF_TAP_SCALE_LOG2): undoes both scalings, exactly  (scale = 2^(se_cur - 127))
      inv = __uint_as_float((unsigned)(254 - se_cur - MF_TAP_SCALE_LOG2) << 23);
    }

    SM_STAMP(1);
    // FOLD: request the next tile's two new blocks now; they land in the wave's staging rows
    // while this tile's pairs run and are picked up behind the second half's pairs
    if constexpr (FOLD) {
      if (z0 + 32 < zc1) stage_new_blocks(z0 + 32);  // (uniform)
    }
    // ---- the two 16-channel halves
    sm_for<0, 2>([&](auto hc) {
      constexpr int HALF = decltype(hc)::value;
      const int zh = z0 + 16 * HALF;  // first output channel of this half
      if (zh >= zc1) return;  // (uniform)
      // mask bytes of this half's outputs (steps.py:781,788): requested now, used after the pairs.
      // Output i of the lane is channel zh + (i&3) + 8 (i>>2) + 4 h: uniform row pointers
      const bool inside = zh + 16 <= zc1;  // (uniform) every channel of the half exists
      unsigned char mk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) mk[i] = 0;
      if (mask) {  // branch-free inside: every load is issued before the first is awaited
        if (inside) {
          const uint8_t *mp = mask + (long)zh * S + s_base;
          unsigned om = off_out;
          SM_PIN(om);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            mk[i] = mp[om];
            mp += (i & 3) == 3 ? 5 * S : S;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            // channel min(zu + 4 h, Nz - 1): the lane part clamped at the cube's end
            const int zu = min(zh + (i & 3) + 8 * (i >> 2), Nz - 1);
            const unsigned hs = (unsigned)(min(4, Nz - 1 - zu) * S);
            mk[i] = (mask + (long)zu * S + s_base)[(unsigned)rr + (h ? hs : 0u)];
          }
        }
      }
      // FOLD: s(z) of the lane's border class for its eight channels (two 16-byte loads, requested
      // in front of the pairs like the mask bytes; zero beyond Nz like the 1/sqrt(den) table)
      f32x4v sg[2] = {(f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}};
      if constexpr (FOLD && NORMW) {
        // norm[z, s] at the lane's eight outputs (the addresses of its stores)
        const float *qp = sdl + (long)zh * S + s_base;
        unsigned on4 = off_out4;
        SM_PIN(on4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          sg[i >> 2][i & 3] = ldf(qp, on4);
          qp += (i & 3) == 3 ? 5 * S : S;
        }
      } else if constexpr (FOLD) {
#pragma unroll
        for (int g = 0; g < 2; ++g) sg[g] = *reinterpret_cast<const f32x4v *>(sdl + zh + 8 * g);
      }
      SmState st;
      if constexpr (!FOLD) {   // (FOLD: the first pair's epilogue writes the state, EPI_FIRST)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          st.best[i] = -INFINITY, st.worst[i] = INFINITY, st.key[i] = -INFINITY;
      }

      // profile pairs, software pipelined; two accumulators in ping-pong: a stage writes one
      // while the epilogue of the pair before reads the other.  The first stage runs the same
      // code as every other: its "previous pair" is neutral -- 1/sqrt(den) = NaN makes T and its
      // key NaN, which v_max3 / v_min3 ignore.
      f32x16 accX, accY;
      f32x4v fa[2], fb[2];
      int ca = 0, cb = 0;
#pragma unroll
      for (int g = 0; g < 2; ++g) fa[g] = fb[g] = (f32x4v){NAN, NAN, NAN, NAN};
      // (accY: any bits -- FOLD: the first stage has no epilogue, EPI_NONE; exact form: its neutral
      // "previous pair" comes from the NaN 1/sqrt(den) values.  B fragment writes -> first MFMA)
      asm volatile("s_nop 1" : "=v"(accY));
      // the lane's own profile of the pair: slot 2p (rows 0-15) or 2p+1 (rows 16-31; an odd K's
      // last profile sits in LDS twice): one pointer, a constant step per pair
      const char *akp = a_lane + (second ? MF_PROF_BYTES : 0);
      u32x4v fh, fl;
      if constexpr (FOLD) {
        fh = *reinterpret_cast<const u32x4v *>(akp + 32), fl = fh;
        if constexpr (TERMS == 3) fl = *reinterpret_cast<const u32x4v *>(akp + 32 + 8 * MF_COPY_BYTES);
      }
      auto run = [&](int p, f32x16 &wacc, const f32x16 &racc, auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;
        const int sa = 2 * p, sb = min(2 * p + 1, K - 1);  // (odd K: the last profile twice)
        int ia, ib;
        if constexpr (IDENT) {
          ia = sa, ib = sb | (sb >= nN ? 0x100 : 0);
        } else {
          ia = pinfo[sa], ib = pinfo[sb];
        }
        const char *ak = akp;
        akp += 2 * MF_PROF_BYTES;
        sm_stage<TERMS, HALF, FOLD, MODE>(ak, akp, fh, fl, ((ia | ib) >> 8) != 0, bh, bl, wacc,
                                          racc, fa, fb, st, maskv, ca, cb);
        // 1/sqrt(den) of this pair for the lane's 8 channels: requested now (the epilogue that
        // read the previous values is done), used from the third MFMA gap of the next stage
        const int ka = ia & 0xff, kb = ib & 0xff;
        if constexpr (FOLD) {
          // (no per-pair normalisation)
        } else if constexpr (BW) {
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            fa[g] = *reinterpret_cast<const f32x4v *>(rdb + (long)ka * NzP + zh + 8 * g);
            fb[g] = *reinterpret_cast<const f32x4v *>(rdb + (long)kb * NzP + zh + 8 * g);
          }
        } else {
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            fa[g] = *reinterpret_cast<const f32x4v *>(rd_lane + sa * MF_RD_BYTES + 64 * HALF + 32 * g);
            fb[g] = *reinterpret_cast<const f32x4v *>(rd_lane + sb * MF_RD_BYTES + 64 * HALF + 32 * g);
          }
        }
        ca = 31 - ka, cb = 31 - kb;
      };
      using UpdC = std::integral_constant<int, EPI_UPDATE>;
      int p = 0;
      if constexpr (FOLD) {
        // (the launch gives FOLD plans at least two pairs.)  Stage 0 has no pair before it: no
        // epilogue in its gaps, no neutral accumulator to set up; stage 1 carries the epilogue that
        // WRITES the state: per half tile 16 + 24 register moves and eight wasted epilogue items
        // (40 VALU instructions) less than with one stage body for all pairs
        run(0, accX, accY, std::integral_constant<int, EPI_NONE>{});
        run(1, accY, accX, std::integral_constant<int, EPI_FIRST>{});
        p = 2;
      }
      for (; p + 1 < NP; p += 2) {
        run(p, accX, accY, UpdC{});
        run(p + 1, accY, accX, UpdC{});
      }
      SM_STAMP(2 + 3 * HALF);
      if constexpr (PODD) {
        run(p, accX, accY, UpdC{});
        sm_drain<FOLD>(accX, fa, fb, st, maskv, ca, cb);
      } else {
        sm_drain<FOLD>(accY, fa, fb, st, maskv, ca, cb);
      }
      SM_STAMP(3 + 3 * HALF);

      // store (mask glue: steps.py:781,788)
      auto store = [&](auto straight_c) {
        constexpr bool STRAIGHT = decltype(straight_c)::value;  // nothing to test per output
        float *cp = correl + (long)zh * S + s_base, *np = correl_min + (long)zh * S + s_base;
        uint8_t *pp = profile + (long)zh * S + s_base;
        unsigned oo = off_out, oo4 = off_out4;
        if constexpr (STRAIGHT) {
          SM_PIN(oo);
          SM_PIN(oo4);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int zu = zh + (i & 3) + 8 * (i >> 2);  // uniform part of the channel
          if (STRAIGHT || zu + 4 * h < zc1) {
            // (FOLD: the class factor s(z) of the lane's channel times the power of two that
            // undoes the tile scale -- one product per output instead of two, the same bits)
            float f = inv;
            if constexpr (FOLD && NORMW) {  // (norm <= 0: no field covers the spaxel, T = 0)
              const float nv = sg[i >> 2][i & 3];
              f = nv > 0.0f ? __builtin_amdgcn_rsqf(nv) * inv : 0.0f;
            } else if constexpr (FOLD) {
              f = sg[i >> 2][i & 3] * inv;
            }
            float b = st.best[i] * f;
            float w = st.worst[i] * f;
            int kk = 31 - (int)(__float_as_uint(st.key[i]) & 31u);
            // profiles run narrow-first, not in index order: when every T is the same number (a
            // spaxel of zeros) the first maximum is index 0 (np.argmax semantics, lib :1210)
            if (st.best[i] == st.worst[i]) kk = 0;
            if (mk[i]) b = 0.0f, kk = 0;
            if (STRAIGHT || sv) {
              stf(cp, oo4, b);
              stf(np, oo4, w);
              pp[oo] = (uint8_t)kk;
            }
            vmax = fmaxf(vmax, b);
            vmin = fminf(vmin, w);
          }
          const long step = (i & 3) == 3 ? 5 * S : S;
          cp += step, np += step, pp += step;
        }
      };
      if (inside && all_valid) store(std::true_type{});
      else store(std::false_type{});
      SM_STAMP(4 + 3 * HALF);
    });
  }
}

// VARIANT bit 0: odd number of profile pairs; bit 2 (with FOLD): IDENT; bit 3 (with FOLD): NORMW,
// sden is the plan's norm cube and the tiles outside [zf0, zf1) are left to
// glr_spectral_norm_mfma.hip; bit 1: FOLD -- atab holds the taps times a_k, rden
// the table 1/(a_k sqrt(den)), sden the class factors s; tiles in [zf0, zf1) (multiples of 32)
// run the FOLD pair loop, the tiles at the cube's ends the exact one (with each lane reading its
// own class from global memory, interior or not: the block's LDS holds no 1/sqrt(den) table).
template <int TERMS, int VARIANT>
__global__ __launch_bounds__(64 * MF_WAVES, 1) void spectral_mfma2_kernel(
    const float *__restrict__ fsf, const float *__restrict__ rden,
    const float *__restrict__ rdi_s, int NzP, const uint4 *__restrict__ atab,
    const int *__restrict__ pinfo, int K, int NP, int Nz, int Ny, int Nx, int P, int zchunk, const uint8_t *__restrict__ mask, float *__restrict__ correl,
    uint8_t *__restrict__ profile, float *__restrict__ correl_min, float *__restrict__ part_max,
    float *__restrict__ part_min, const float *__restrict__ sden, int zf0, int zf1, int nN,
    long s_first, long s_end_launch, int rx0, int rx1, int cols_per_block) {
  // (s_first, s_end: the spaxels of this launch, in multiples of 32 from the field's first -- a run
  // may be split into row bands; waves hold the same 32 spaxels as in a launch over the field.
  // rx1 > 0: a RECTANGLE instead -- the rows s_first / Nx .. s_end / Nx, columns rx0 .. rx1 - 1 of
  // each; a wave holds 32 consecutive columns of one row)
  // A block owns cols_per_block COLUMNS (32 spaxels x the chunk's channels) and its waves PULL them
  // from a counter in LDS (round 4).  With one column per wave -- the form of rounds 2-3 -- the
  // in-kernel clock stamps (-DSM_TIMING, profiles/r04_sm_phase_times.txt) showed the three waves
  // of a SIMD taking 23 k / 32 k / 49 k cycles per tile: the issue arbiter serves the oldest wave
  // first, the first wave of a SIMD is done with its nine tiles when the third has done four, and
  // its slot stays empty until the block ends (the block holds the CU's LDS) -- 2.1 waves per SIMD
  // on average, one alone at the end.  Pulled columns go to whoever is free.  (It evens out the
  // waves and leaves the kernel's time where it was: see the launch.)
  extern __shared__ __align__(16) char sm_lds[];
  __shared__ int sm_next_col;
  const int Kp = K + (K & 1);  // slots in LDS: an odd K's last profile twice (its pair partner)
  {
    constexpr int PV = MF_PROF_BYTES / 16;
    const int nvec = Kp * PV;
    for (int i = threadIdx.x; i < nvec; i += 64 * MF_WAVES)
      reinterpret_cast<uint4 *>(sm_lds)[i] = atab[i < K * PV ? i : i - PV];
    if (threadIdx.x == 0) sm_next_col = MF_WAVES;  // (the first MF_WAVES columns: one per wave)
  }
  __syncthreads();
  const long S = (long)Ny * Nx;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, in an SGPR
  const int r = lane & 31, h = lane >> 5;
  const int zc0 = blockIdx.y * zchunk, zc1 = min(Nz, zc0 + zchunk);
#ifdef SM_TIMING
  const bool sm_rec = blockIdx.x >= 3 && blockIdx.x < 7 && blockIdx.y == 1 && lane == 0;
  long long sm_ncol = 0;
  long long sm_wall0 = 0;
  if (sm_rec) {
    sm_wave[((blockIdx.x - 3) * MF_WAVES + wv) * 4 + 0] = clock64();
    sm_wall0 = wall_clock64();   // (the constant 100 MHz counter: the ratio gives the shader clock)
  }
#endif
  for (int col = wv; col < cols_per_block;) {
  long s_end = s_end_launch;
  const long w_col = (long)blockIdx.x * cols_per_block + col;
  long s_base = s_first + w_col * 32;
  bool have = true;
  if (rx1 > 0) {
    const int wpr = (rx1 - rx0 + 31) / 32;  // waves per row
    const long row = s_first / Nx + w_col / wpr;
    have = row < s_end / Nx;
    s_base = row * Nx + rx0 + 32 * (int)(w_col % wpr);
    s_end = row * Nx + rx1;
  }
  if (!have || s_base >= s_end) break;  // (columns are handed out in order: nothing behind this one)
  // A rows: lane r is output channel zi = r & 15 of the pair's profile r >> 4; its fragment of
  // window block b starts at G[31 - zi + 8 h + 16 b]
  const int E0 = 8 * h - (r & 15) + 31;
  const char *a_lane = sm_lds + (E0 & 7) * MF_COPY_BYTES + (E0 >> 3) * 16;
  // this wave's [K][32] table of 1/sqrt(den) for the current tile (behind the tap copies), in
  // the order the profiles are processed
  char *rd_wave = sm_lds + Kp * MF_PROF_BYTES + wv * K * MF_RD_BYTES;
  const bool sv = s_base + r < s_end;
  const bool all_valid = s_base + 32 <= s_end;
  const long sc = sv ? s_base + r : s_end - 1;
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
  constexpr bool PODD = (VARIANT & 1) != 0;  // odd number of profile pairs
  constexpr bool FOLD = (VARIANT & 2) != 0;
  constexpr bool IDENT = (VARIANT & 4) != 0;
  constexpr bool NORMW = (VARIANT & 8) != 0;
  if constexpr (FOLD) {
    const float *sdl = NORMW ? sden : sden + (long)cls * NzP + 4 * h;
    const int f0 = max(zc0, zf0), f1 = min(zc1, zf1);
    if (f0 < f1)
      sm_tiles<TERMS, false, PODD, true, IDENT, NORMW>(sdl, fsf, rdb, rden + (long)ccls * K * NzP, NzP, pinfo, K,
                                         NP, Nz, S, s_base, rr, sv, all_valid, h, lane, a_lane,
                                         rd_wave, f0, f1, mask, correl, profile, correl_min, vmax,
                                         vmin, nN,
                                         sm_lds + Kp * MF_PROF_BYTES + wv * MF_STAGE_BYTES);
    for (int e = 0; e < (NORMW ? 0 : 2); ++e) {  // the chunk's tiles in front of / behind the FOLD range
      const int a = e ? max(zc0, zf1) : zc0, b = e ? zc1 : min(zc1, zf0);
      if (a < b)
        sm_tiles<TERMS, true, PODD, false>(nullptr, fsf, rdb, rdi_s, NzP, pinfo, K, NP, Nz, S,
                                           s_base, rr, sv, all_valid, h, lane, a_lane, rd_wave, a, b,
                                           mask, correl, profile, correl_min, vmax, vmin, nN);
    }
  } else if (__any(cls != ccls)) {
    sm_tiles<TERMS, true, PODD, false>(nullptr, fsf, rdb, rdi_s, NzP, pinfo, K, NP, Nz, S, s_base,
                                       rr, sv, all_valid, h, lane, a_lane, rd_wave, zc0, zc1, mask,
                                       correl, profile, correl_min, vmax, vmin, nN);
  } else {
    sm_tiles<TERMS, false, PODD, false>(nullptr, fsf, rdb, rdi_s, NzP, pinfo, K, NP, Nz, S, s_base,
                                        rr, sv, all_valid, h, lane, a_lane, rd_wave, zc0, zc1, mask,
                                        correl, profile, correl_min, vmax, vmin, nN);
  }
  if (part_max) {
    const float a = fmaxf(vmax, __shfl_xor(vmax, 32));
    const float b = fminf(vmin, __shfl_xor(vmin, 32));
    if (h == 0 && sv) {
      part_max[(long)blockIdx.y * S + sc] = a;
      part_min[(long)blockIdx.y * S + sc] = b;
    }
  }
  // the next column nobody has taken yet (lane 0 asks, the wave follows)
  int nxt = 0;
  if (lane == 0) nxt = atomicAdd(&sm_next_col, 1);
  col = __builtin_amdgcn_readfirstlane(nxt);
#ifdef SM_TIMING
  ++sm_ncol;
#endif
  }  // columns of this block
#ifdef SM_TIMING
  if (sm_rec) {
    sm_wave[((blockIdx.x - 3) * MF_WAVES + wv) * 4 + 1] = clock64();
    sm_wave[((blockIdx.x - 3) * MF_WAVES + wv) * 4 + 2] = sm_ncol;
    sm_wave[((blockIdx.x - 3) * MF_WAVES + wv) * 4 + 3] = wall_clock64() - sm_wall0;
  }
#endif
}

}  // namespace

// Grid of the spectral kernel: bx blocks of MF_WAVES waves (32 spaxels each) x nzm chunks of zcm
// channels.  One block per CU at a time (LDS): the number of z chunks is picked so that the
// blocks fill whole rounds of the chip -- useful channel slots / (rounds x CUs x (chunk length +
// ~one tile of start-up per block)); chunks are whole 32-channel tiles, at most 64 of them
// (partial maps).  Shared by the launch and by origin_spectral_mfma_count.
static void sm_geometry(int num_cu, int Nz, long S, long *bx_out, int *nzm_out, int *zcm_out) {
  const long bx = cdiv(S, 32 * MF_WAVES);
  const int ncu = std::max(1, num_cu);
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
  *bx_out = bx;
  *nzm_out = cdiv(Nz, zcm);
  *zcm_out = zcm;
}

// MFMA instructions one launch issues (what SQ_INSTS_MFMA counts): every wave that holds a
// spaxel walks all 32-channel tiles of its chunk; a tile is two 16-channel halves, a half runs
// every profile pair (slots 2p, 2p+1 of the processing order, narrow profiles first) with 3
// k-steps (both narrow: window blocks 1..3) or 5, `terms` MFMAs per k-step.
// z chunks (rows of the partial maps) of a launch
int origin_spectral_mfma_chunks(int num_cu, int Nz, int Ny, int Nx) {
  long bx;
  int nzm, zcm;
  sm_geometry(num_cu, Nz, (long)Ny * Nx, &bx, &nzm, &zcm);
  return nzm;
}

long origin_spectral_mfma_count(int num_cu, int terms, int K, int n_narrow, int Nz, int Ny,
                                int Nx) {
  const long S = (long)Ny * Nx;
  long bx;
  int nzm, zcm;
  sm_geometry(num_cu, Nz, S, &bx, &nzm, &zcm);
  long tiles = 0;  // 32-channel tiles per wave, over all chunks
  for (int c = 0; c < nzm; ++c) tiles += cdiv(std::min(zcm, Nz - c * zcm), 32);
  long ksteps = 0;
  for (int p = 0; p < (K + 1) / 2; ++p) {
    const int sb = std::min(2 * p + 1, K - 1);
    ksteps += sb >= n_narrow ? 5 : 3;  // (slot sb is the wider of the pair)
  }
  const long waves = cdiv(S, 32);
  return waves * tiles * 2 * ksteps * terms;
}

// Launch: a wave = 32 spaxels x 32-channel tiles (two 16-channel halves); z chunks sized to give every CU several blocks
// (one block per CU at a time: K * (4.5 KiB + MF_WAVES * 128 B) of LDS).  Returns the number of z
// chunks (rows of part_max / part_min) in *nzc.  nN: number of narrow profiles (the first nN
// slots of the processing order).
int origin_spectral_mfma_launch(origin_ctx *ctx, int terms, const float *fsf, const float *rden,
                                const float *rdi_s, int NzP, const uint4 *atab, const int *pinfo,
                                int K, int nN, int Nz, int Ny, int Nx, int P, const uint8_t *mask,
                                float *correl, uint8_t *profile, float *correl_min, float *part,
                                bool want_maps, int *nzc_out, float **pmax_out, float **pmin_out,
                                const uint4 *atab_fold, const float *rden_fold, const float *sden,
                                int ident, long s_first, long s_count, const float *normc,
                                int part_rows, int rx0, int rx1) {
  const long S = (long)Ny * Nx;
  long bx;
  int nzm, zcm;
  // (the z chunks are those of a launch over the whole field, whatever part of it this one takes:
  // the partial maps of all parts of a run share their layout)
  sm_geometry(ctx->num_cu, Nz, S, &bx, &nzm, &zcm);
  if (s_count <= 0) s_first = 0, s_count = S;
  if ((rx1 <= 0 && s_first % 32 != 0) || s_first < 0 || s_first + s_count > S) {
    origin_set_error("spectral MFMA kernel: bad spaxel range");
    return ORIGIN_E_ARG;
  }
  // columns (32 spaxels x one z chunk) per block: MF_WAVES x SM_COLS_FACTOR, pulled by the waves
  // (measured at 3681 x 600 x 600 with 1 / 2 / 3 / 4 / 6 columns per wave: 9.59 / 9.80 / 9.72 / 9.80 /
  // 11.1 ms -- evening out the waves does not move the kernel: while a SIMD's first wave is gone
  // the other two run that much faster.  One column per wave, as in rounds 2-3, stays the default)
  static const int cols_factor = getenv("ORIGIN_GLR_SPECTRAL_COLS") ? std::max(1, atoi(getenv("ORIGIN_GLR_SPECTRAL_COLS"))) : 1;
  int cols_per_block = MF_WAVES * cols_factor;
  long ncols = cdiv(s_count, 32);
  long s_end = s_first + s_count;
  if (rx1 > 0) {  // rectangle: whole rows s_first / Nx .. , columns rx0 .. rx1 - 1
    if (s_first % Nx != 0 || s_count % Nx != 0 || rx0 < 0 || rx0 >= rx1 || rx1 > Nx) {
      origin_set_error("spectral MFMA kernel: bad rectangle");
      return ORIGIN_E_ARG;
    }
    ncols = (s_count / Nx) * (long)cdiv(rx1 - rx0, 32);
  }
  // (small launches -- narrow row bands, rectangles: keep every CU busy before sharing columns)
  while (cols_per_block > MF_WAVES && cdiv(ncols, cols_per_block) * (long)nzm < 2L * ctx->num_cu)
    cols_per_block -= MF_WAVES;
  bx = cdiv(ncols, cols_per_block);
  // (part_rows: rows of each partial map when other launches add theirs behind this one's)
  float *pmax = want_maps ? part : nullptr;
  float *pmin = want_maps ? part + (size_t)std::max(nzm, part_rows) * S : nullptr;
  const int Kp = K + (K & 1);
  // (+ one profile of slack: the last stage requests the fragments of a pair that is not there)
  const size_t lds = std::max((size_t)Kp * MF_PROF_BYTES + (size_t)K * MF_WAVES * MF_RD_BYTES,
                              (size_t)(Kp + 1) * MF_PROF_BYTES);
  // pairs of slots (2p, 2p+1); an odd last profile pairs with itself
  const int NP = (K + 1) / 2;
  auto pick = [&](int fold) -> const void * {
    const int variant = (NP & 1) | (fold << 1) | ((fold && ident) << 2) | ((fold && normc) << 3);
#define SM_PICK(T, V) \
  if (terms == T && variant == V) return (const void *)spectral_mfma2_kernel<T, V>
    SM_PICK(3, 0); SM_PICK(3, 1); SM_PICK(3, 2); SM_PICK(3, 3); SM_PICK(3, 6); SM_PICK(3, 7);
    SM_PICK(1, 0); SM_PICK(1, 1); SM_PICK(1, 2); SM_PICK(1, 3); SM_PICK(1, 6); SM_PICK(1, 7);
    SM_PICK(3, 10); SM_PICK(3, 11); SM_PICK(3, 14); SM_PICK(3, 15);
    SM_PICK(1, 10); SM_PICK(1, 11); SM_PICK(1, 14); SM_PICK(1, 15);
#undef SM_PICK
    return nullptr;
  };
  // FOLD for the tiles whose profile supports lie inside the cube (plans whose eps test passed
  // bring the folded tables), the exact form for the tiles at the ends
  int zf0 = 0, zf1 = 0;
  if (atab_fold && ((rden_fold && sden) || normc) && !getenv("ORIGIN_GLR_NO_FOLD")) {
    mf_fold_range(Nz, &zf0, &zf1);
  }
  // (LDS: the tap copies and the staging rows -- K <= 24 with twelve waves)
  const size_t lds_fold = (size_t)Kp * MF_PROF_BYTES + (size_t)MF_WAVES * MF_STAGE_BYTES;
  if (!mf_fold_fits(K) || NP < 2) zf0 = zf1 = 0;  // (the FOLD pair loop peels its first two stages)
  const int fold = zf1 > zf0;
  const void *fn = pick(fold);
  if (!fn) {
    origin_set_error("spectral MFMA kernel: no variant for %d terms", terms);
    return ORIGIN_E_STATE;
  }
  ORIGIN_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 MF_MAX_K * (MF_PROF_BYTES + MF_WAVES * MF_RD_BYTES)));
  const uint4 *a_atab = fold ? atab_fold : atab;
  const float *a_rden = fold ? rden_fold : rden;
  if (fold && normc) sden = normc;  // (NORMW: the kernel's sden argument is the norm cube)
  if (normc && !fold) {
    origin_set_error("spectral MFMA kernel: the norm-cube form needs the FOLD range");
    return ORIGIN_E_STATE;
  }
  int a_NP = NP, a_zcm = zcm;
  void *args[] = {&fsf, &a_rden, &rdi_s, &NzP, &a_atab, &pinfo, &K, &a_NP, &Nz, &Ny, &Nx, &P, &a_zcm,
                  &mask, &correl, &profile, &correl_min, &pmax, &pmin, &sden, &zf0, &zf1, &nN, &s_first, &s_end, &rx0, &rx1,
                  &cols_per_block};
  ORIGIN_HIP(hipLaunchKernel(fn, dim3((unsigned)bx, (unsigned)nzm), dim3(64 * MF_WAVES), args,
                             fold ? std::max(lds, lds_fold) : lds, ctx->stream));
  ORIGIN_LAUNCH_CHECK();
  *nzc_out = nzm;
  *pmax_out = pmax;
  *pmin_out = pmin;
  return ORIGIN_OK;
}

#ifdef SM_TIMING
extern "C" int origin_debug_sm_timing(long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sm_tim), sizeof(long long) * 8 * MF_WAVES * 8);
}
extern "C" int origin_debug_sm_waves(long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sm_wave), sizeof(long long) * 4 * MF_WAVES * 4);
}
#endif
