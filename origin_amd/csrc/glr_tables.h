// Layout of the LDS tap table of the matrix-core spectral GLR stage, shared by the plan builder
// (glr.hip) and the kernel (glr_spectral_mfma.hip).
//
// Per profile (in the order the kernel processes them: narrow ones first), the padded tap array
// G_k[e] = p_k[lw_k + 63 - e], e = 0 .. 8 MF_GROUPS + 6, as 8 copies shifted by 0..7 elements --
// so that the 8 consecutive entries a lane needs for an A fragment are ONE aligned 16-byte read
// -- first the "hi" halves of all 8 copies, then the "lo" halves (two-term f16 split, taps times
// 2^MF_TAP_SCALE_LOG2; the bf16 table holds the rounded taps in the hi half and zeros in lo).
#pragma once

constexpr int MF_TAP_SCALE_LOG2 = 12;                 // f16 taps are stored times 2^12
constexpr int MF_GROUPS = 18;                         // 16-byte groups per shifted copy (14 used)
// 288 = 256 + 32: the 16 lanes a ds_read_b128 serves at once are the 16 output channels of a
// tile -- 8 copies x 2 adjacent groups -- and 2 copy + group covers the 16 bank quads once
constexpr int MF_COPY_BYTES = MF_GROUPS * 16;
constexpr int MF_PROF_BYTES = 2 * 8 * MF_COPY_BYTES;  // hi copies, then lo copies
constexpr int MF_RD_BYTES = 32 * 4;                   // per wave and profile: 32 channels of 1/sqrt(den)
constexpr int MF_WAVES = 12;                          // waves per block (three per SIMD: 168 VGPRs)
constexpr int MF_MAX_K = 26;                          // 26 * (4608 + 12 * 128) B = 156 KiB of LDS
// zero channels in front of / behind the cube_fsf work cube: the matrix-core spectral kernel reads
// its 96-channel windows [z0 - 32, z0 + 63] without bounds tests
constexpr int MF_PAD_FRONT = 32, MF_PAD_BACK = 64;

// FOLD (glr_spectral_mfma.hip): the channels [zf0, zf1) whose tiles compare bare accumulators --
// every profile of the matrix-core path (half width <= 32) has its support inside the cube there
constexpr int MF_FOLD_MARGIN = 32;
inline void mf_fold_range(int Nz, int *zf0, int *zf1) {
  *zf0 = MF_FOLD_MARGIN;
  *zf1 = (Nz - MF_FOLD_MARGIN) / 32 * 32;
  if (*zf1 <= *zf0) *zf0 = *zf1 = 0;
}
// a wave's staging rows for the next tile's new window blocks (16 rows x 64 lanes x 4 bytes), in
// the LDS behind the tap copies (an odd K's last profile is held twice): K <= 24 with twelve waves
constexpr int MF_STAGE_BYTES = 16 * 256;
inline bool mf_fold_fits(int K) {
  return (size_t)(K + (K & 1)) * MF_PROF_BYTES + (size_t)MF_WAVES * MF_STAGE_BYTES <=
         (size_t)MF_MAX_K * (MF_PROF_BYTES + MF_WAVES * MF_RD_BYTES);
}
// largest |1/(a_k sqrt(den_k)) / s - 1| a plan may show over [zf0, zf1) and still run FOLD
constexpr float MF_FOLD_EPS = 2e-6f;

// atab_fold / rden_fold / sden: the folded tables (nullptr: the exact form everywhere)
int origin_spectral_mfma_launch(origin_ctx *ctx, int terms, const float *fsf, const float *rden,
                                const float *rdi_s, int NzP, const uint4 *atab, const int *pinfo, int K,
                                int nN, int Nz, int Ny, int Nx, int P, const uint8_t *mask, float *correl,
                                uint8_t *profile, float *correl_min, float *part, bool want_maps,
                                int *nzc_out, float **pmax_out, float **pmin_out,
                                const uint4 *atab_fold, const float *rden_fold, const float *sden,
                                int ident, long s_first = 0, long s_count = 0,
                                const float *normc = nullptr, int part_rows = 0, int rx0 = 0,
                                int rx1 = 0);
// glr_spectral_norm_mfma.hip: the end tiles [0, zf0) and [zf1, Nz) of a plan with a norm cube
int origin_spectral_norm_mfma_launch_ends(origin_ctx *ctx, const float *fsf, const float *norm,
                                          const uint4 *atab, const uint4 *atab2, const int *pinfo,
                                          int K, int Nz, int Ny, int Nx, const uint8_t *mask,
                                          float *correl, uint8_t *profile, float *correl_min,
                                          float *pmax, float *pmin, int zf0, int zf1, int prow0,
                                          int *rows_out);

long origin_spectral_mfma_count(int num_cu, int terms, int K, int n_narrow, int Nz, int Ny,
                                int Nx);
int origin_spectral_mfma_chunks(int num_cu, int Nz, int Ny, int Nx);

// glr_spectral_norm_mfma.hip: the same stage for plans with an explicit norm cube (weighted fields)
int origin_spectral_norm_mfma_max_k();
int origin_spectral_norm_mfma_launch(origin_ctx *ctx, const float *fsf, const float *norm,
                                     const uint4 *atab, const uint4 *atab2, const int *pinfo, int K,
                                     int Nz, int Ny, int Nx, const uint8_t *mask, float *correl,
                                     uint8_t *profile, float *correl_min, float *part,
                                     bool want_maps, int *nzc_out, float **pmax_out,
                                     float **pmin_out);
