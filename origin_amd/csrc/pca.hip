// Greedy PCA building blocks  (SURVEY.md 2.2 rows k5-k7).
//
// One iteration of Compute_GreedyPCA (reference muse_origin/lib_origin.py:899-949) for a
// batch of areas, on a cube that stays in place in HBM as (Nz, S), S = Ny*Nx:
//
//   b   = mean of the background spectra                                  (lib :917)
//   Xp  = X_nuis - b (b^T X_nuis)          [un-normalised projection]     (lib :920-923)
//   u   = leading left singular vector of Xp                              (lib :940)
//   F  -= u (u^T F) over the whole area; test = mean_z F^2                (lib :943-946)
//
// The reference calls ARPACK svds(k=1, tol=0).  Here u comes from the Gram matrix
// G = Xp^T Xp (float64 MFMA, the only matrix-shaped contraction of the path): its leading
// eigenvector v gives u = Xp v / |Xp v|.  SURVEY.md section 7 (hard part 1) shows the loop is
// threshold driven, so the eigen-solve must be *converged*; everything that feeds it is
// float64, only the cube itself is float32.
//
// Lists (spaxel indices of the nuisance / background / area members) are concatenated per
// area with int64 offsets; a block never straddles two areas, so the per-area vectors
// (b, u) are wave-uniform and read with scalar loads.
#include "common.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ------------------------------------------------------------------------------------
// b_a[z] = mean_{i in bg_a} F[z, bg[i]]        grid (ceil(Nz/4), na), block (64,4)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bmean_kernel(const float *__restrict__ F, int Nz, long S,
                                                    const int *__restrict__ bg,
                                                    const long *__restrict__ bg_off,
                                                    double *__restrict__ b) {
  const int a = blockIdx.y;
  const int z = blockIdx.x * 4 + threadIdx.y;
  if (z >= Nz) return;
  const long o0 = bg_off[a], o1 = bg_off[a + 1];
  const float *row = F + (long)z * S;
  double acc = 0.0;
  for (long i = o0 + threadIdx.x; i < o1; i += 64) acc += (double)row[bg[i]];
  acc = wave_sum_d(acc);
  if (threadIdx.x == 0) b[(long)a * Nz + z] = acc / (double)(o1 - o0);
}

// ------------------------------------------------------------------------------------
// gather the nuisance columns into X (float64, [Nz][ld]) and c_j = b^T X_j
// grid (ceil(ld/64), na), block (64 columns, 16 waves over z)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gather_xp_kernel(
    const float *__restrict__ F, int Nz, long S, const int *__restrict__ nuis,
    const long *__restrict__ nuis_off, const double *__restrict__ b, double *__restrict__ Xp,
    const long *__restrict__ xp_off, const int *__restrict__ ld_, double *__restrict__ cvec,
    const long *__restrict__ c_off) {
  __shared__ double red[16][64];
  const int a = blockIdx.y;
  const int ld = ld_[a];
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (blockIdx.x * 64 >= ld) return;  // whole block out of range (uniform)
  const long n = nuis_off[a + 1] - nuis_off[a];
  const bool live = j < n;       // real nuisance column
  const bool inld = j < ld;      // padded column (stored as zeros)
  const long col = live ? (long)nuis[nuis_off[a] + j] : 0;
  const double *ba = b + (long)a * Nz;
  double *X = Xp + xp_off[a];
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
  double acc = 0.0;
  for (int z = w; z < Nz; z += 16) {
    const double v = live ? (double)F[(long)z * S + col] : 0.0;
    if (inld) X[(long)z * ld + j] = v;
    acc = fma(ba[z], v, acc);
  }
  red[threadIdx.y][threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.y == 0 && inld) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
    cvec[c_off[a] + j] = t;
  }
}

// Xp[z][j] -= b[z] c[j]       grid (ceil(Nz/16), na), block 256 (lanes over columns)
__global__ __launch_bounds__(256) void project_xp_kernel(const double *__restrict__ b, int Nz,
                                                         double *__restrict__ Xp,
                                                         const long *__restrict__ xp_off,
                                                         const int *__restrict__ ld_,
                                                         const double *__restrict__ cvec,
                                                         const long *__restrict__ c_off) {
  const int a = blockIdx.y;
  const int ld = ld_[a];
  double *X = Xp + xp_off[a];
  const double *c = cvec + c_off[a];
  const double *ba = b + (long)a * Nz;
  const int z0 = blockIdx.x * 16, z1 = min(Nz, z0 + 16);
  for (int z = z0; z < z1; ++z) {
    const double bz = ba[z];
    for (int j = threadIdx.x; j < ld; j += 256) X[(long)z * ld + j] = fma(-bz, c[j], X[(long)z * ld + j]);
  }
}

// ------------------------------------------------------------------------------------
// G = Xp^T Xp with v_mfma_f64_16x16x4_f64.
// One wave computes a 32x32 tile (2x2 MFMA tiles) of the upper triangle for one K-slice of
// the channels; slabs are summed in fixed order by gram_reduce_kernel and mirrored.
//   A operand (16x4): lane l holds A[i = l&15][k = l>>4] = Xp[k0 + (l>>4)][i0 + (l&15)]
//   B operand (4x16): lane l holds B[k = l>>4][j = l&15] = Xp[k0 + (l>>4)][j0 + (l&15)]
//   D (16x16): lane l, reg r holds D[row = (l>>4) + 4r][col = l&15]
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void gram_kernel(const double *__restrict__ Xp,
                                                  const long *__restrict__ xp_off,
                                                  const int *__restrict__ ld_,
                                                  const int *__restrict__ tile_i,
                                                  const int *__restrict__ tile_j,
                                                  const int *__restrict__ tile_a, int Nz,
                                                  int ksplit, double *__restrict__ slab,
                                                  const long *__restrict__ g_off, long slab_stride) {
  const int t = blockIdx.x;
  const int a = tile_a[t];
  const int ld = ld_[a];
  const int i0 = tile_i[t] * 32, j0 = tile_j[t] * 32;
  const int ks = blockIdx.y;
  const int zper = ((Nz + ksplit - 1) / ksplit + 3) & ~3;
  const int z0 = ks * zper, z1 = min(Nz, z0 + zper);
  const double *X = Xp + xp_off[a];
  const int lane = threadIdx.x;
  const int r16 = lane & 15, kq = lane >> 4;
  // columns beyond ld (ld is a multiple of 16, tiles are 32 wide) are clamped and zeroed
  const bool ia0 = i0 + r16 < ld, ia1 = i0 + 16 + r16 < ld;
  const bool jb0 = j0 + r16 < ld, jb1 = j0 + 16 + r16 < ld;
  double4_t acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
  for (int z = z0; z < z1; z += 4) {
    const int zz = z + kq;
    const bool zin = zz < z1;
    const double *row = X + (long)(zin ? zz : z0) * ld;
    const double a0 = (zin && ia0) ? row[i0 + r16] : 0.0;
    const double a1 = (zin && ia1) ? row[i0 + 16 + r16] : 0.0;
    const double b0 = (zin && jb0) ? row[j0 + r16] : 0.0;
    const double b1 = (zin && jb1) ? row[j0 + 16 + r16] : 0.0;
    acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc00, 0, 0, 0);
    acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc01, 0, 0, 0);
    acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc10, 0, 0, 0);
    acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc11, 0, 0, 0);
  }
  double *G = slab + (long)ks * slab_stride + g_off[a];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = kq + 4 * r;
    const int gi0 = i0 + row, gi1 = i0 + 16 + row;
    const int gj0 = j0 + r16, gj1 = j0 + 16 + r16;
    if (gi0 < ld && gj0 < ld) G[(long)gi0 * ld + gj0] = acc00[r];
    if (gi0 < ld && gj1 < ld) G[(long)gi0 * ld + gj1] = acc01[r];
    if (gi1 < ld && gj0 < ld) G[(long)gi1 * ld + gj0] = acc10[r];
    if (gi1 < ld && gj1 < ld) G[(long)gi1 * ld + gj1] = acc11[r];
  }
}

// G[i][j] = sum_ks slab[ks][i][j] for tile (ti <= tj), mirrored into the lower triangle
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double *__restrict__ slab,
                                                          long slab_stride, int ksplit,
                                                          const int *__restrict__ ld_,
                                                          const int *__restrict__ tile_i,
                                                          const int *__restrict__ tile_j,
                                                          const int *__restrict__ tile_a,
                                                          double *__restrict__ G,
                                                          const long *__restrict__ g_off) {
  const int t = blockIdx.x;
  const int a = tile_a[t];
  const int ld = ld_[a];
  const int i0 = tile_i[t] * 32, j0 = tile_j[t] * 32;
  double *Ga = G + g_off[a];
  for (int e = threadIdx.x; e < 1024; e += 256) {
    const int i = i0 + (e >> 5), j = j0 + (e & 31);
    if (i >= ld || j >= ld) continue;
    double acc = 0.0;
    for (int ks = 0; ks < ksplit; ++ks) acc += slab[(long)ks * slab_stride + g_off[a] + (long)i * ld + j];
    Ga[(long)i * ld + j] = acc;
    Ga[(long)j * ld + i] = acc;
  }
}

// ------------------------------------------------------------------------------------
// u = Xp v, then normalised.     grid (ceil(Nz/4), na), block (64,4)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xv_kernel(const double *__restrict__ Xp,
                                                 const long *__restrict__ xp_off,
                                                 const int *__restrict__ ld_,
                                                 const int *__restrict__ n_, int Nz,
                                                 const double *__restrict__ v,
                                                 const long *__restrict__ v_off,
                                                 double *__restrict__ u) {
  const int a = blockIdx.y;
  const int z = blockIdx.x * 4 + threadIdx.y;
  if (z >= Nz) return;
  const int ld = ld_[a], n = n_[a];
  const double *row = Xp + xp_off[a] + (long)z * ld;
  const double *va = v + v_off[a];
  double acc = 0.0;
  for (int j = threadIdx.x; j < n; j += 64) acc = fma(row[j], va[j], acc);
  acc = wave_sum_d(acc);
  if (threadIdx.x == 0) u[(long)a * Nz + z] = acc;
}

__global__ __launch_bounds__(1024) void normalize_kernel(double *__restrict__ u, int Nz) {
  __shared__ double red[16];
  __shared__ double inv;
  double *ua = u + (long)blockIdx.x * Nz;
  double acc = 0.0;
  for (int z = threadIdx.x; z < Nz; z += 1024) acc = fma(ua[z], ua[z], acc);
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += red[i];
    inv = t > 0.0 ? 1.0 / sqrt(t) : 0.0;
  }
  __syncthreads();
  for (int z = threadIdx.x; z < Nz; z += 1024) ua[z] *= inv;
}

// ------------------------------------------------------------------------------------
// deflation of whole areas.
//   dot    : cpart[zs][i] = sum_{z in slice zs} u_a[z] F[z, spx[i]]
//   update : c_i = sum_zs cpart ; F[z, spx[i]] -= u_a[z] c_i ; o2part[zs][i] = sum F_new^2
//   final  : test[spx[i]] = sum_zs o2part / Nz
// grid (ceil(max_ns/256), ZS, na); block 256 lanes over the area's spaxel list
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void deflate_dot_kernel(const float *__restrict__ F, int Nz,
                                                          long S, const int *__restrict__ spx,
                                                          const long *__restrict__ spx_off,
                                                          const double *__restrict__ u, int zper,
                                                          double *__restrict__ cpart,
                                                          long ntot) {
  const int a = blockIdx.z;
  const long o0 = spx_off[a], o1 = spx_off[a + 1];
  const long i = o0 + (long)blockIdx.x * 256 + threadIdx.x;
  if (o0 + (long)blockIdx.x * 256 >= o1) return;
  const bool live = i < o1;
  const long col = spx[live ? i : o1 - 1];
  const double *ua = u + (long)a * Nz;
  const int z0 = blockIdx.y * zper, z1 = min(Nz, z0 + zper);
  double acc = 0.0;
#pragma unroll 4
  for (int z = z0; z < z1; ++z) acc = fma(ua[z], (double)F[(long)z * S + col], acc);
  if (live) cpart[(long)blockIdx.y * ntot + i] = acc;
}

__global__ __launch_bounds__(256) void deflate_update_kernel(float *__restrict__ F, int Nz, long S,
                                                             const int *__restrict__ spx,
                                                             const long *__restrict__ spx_off,
                                                             const double *__restrict__ u,
                                                             int zper, int nzs,
                                                             const double *__restrict__ cpart,
                                                             double *__restrict__ o2part,
                                                             long ntot) {
  const int a = blockIdx.z;
  const long o0 = spx_off[a], o1 = spx_off[a + 1];
  const long i = o0 + (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= o1) return;
  const long col = spx[i];
  const double *ua = u + (long)a * Nz;
  double c = 0.0;
  for (int k = 0; k < nzs; ++k) c += cpart[(long)k * ntot + i];
  const int z0 = blockIdx.y * zper, z1 = min(Nz, z0 + zper);
  double acc = 0.0;
#pragma unroll 4
  for (int z = z0; z < z1; ++z) {
    const long idx = (long)z * S + col;
    const float nv = (float)fma(-ua[z], c, (double)F[idx]);
    F[idx] = nv;
    acc = fma((double)nv, (double)nv, acc);
  }
  o2part[(long)blockIdx.y * ntot + i] = acc;
}

__global__ __launch_bounds__(256) void deflate_final_kernel(const int *__restrict__ spx, long ntot,
                                                            int nzs, int Nz,
                                                            const double *__restrict__ o2part,
                                                            double *__restrict__ test) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= ntot) return;
  double acc = 0.0;
  for (int k = 0; k < nzs; ++k) acc += o2part[(long)k * ntot + i];
  test[spx[i]] = acc / (double)Nz;
}

}  // namespace

// host-side view of small descriptor arrays that live on the device: the batched kernels
// need a few of them on the host to size grids, so the binding passes both.
extern "C" {

int origin_pca_bmean(origin_ctx *ctx, const float *d_F, int Nz, long S, const int *d_bg,
                     const long *d_bg_off, int na, double *d_b) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_F && d_bg && d_bg_off && d_b && Nz > 0 && S > 0 && na > 0, "bad arguments");
  ProfScope ps(ctx, K_PCA_BMEAN);
  hipLaunchKernelGGL(bmean_kernel, dim3(cdiv(Nz, 4), na), dim3(64, 4), 0, ctx->stream, d_F, Nz, S,
                     d_bg, d_bg_off, d_b);
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

int origin_pca_build_xp(origin_ctx *ctx, const float *d_F, int Nz, long S, const int *d_nuis,
                        const long *d_nuis_off, int na, int ldmax, const double *d_b,
                        double *d_Xp, const long *d_xp_off, const int *d_ld, double *d_c,
                        const long *d_c_off) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_F && d_nuis && d_nuis_off && d_b && d_Xp && d_xp_off && d_ld && d_c &&
                       d_c_off && Nz > 0 && S > 0 && na > 0 && ldmax > 0,
                   "bad arguments");
  {
    ProfScope ps(ctx, K_PCA_GATHER);
    hipLaunchKernelGGL(gather_xp_kernel, dim3(cdiv(ldmax, 64), na), dim3(64, 16), 0, ctx->stream,
                       d_F, Nz, S, d_nuis, d_nuis_off, d_b, d_Xp, d_xp_off, d_ld, d_c, d_c_off);
  }
  {
    ProfScope ps(ctx, K_PCA_PROJECT);
    hipLaunchKernelGGL(project_xp_kernel, dim3(cdiv(Nz, 16), na), dim3(256), 0, ctx->stream, d_b,
                       Nz, d_Xp, d_xp_off, d_ld, d_c, d_c_off);
  }
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

int origin_pca_gram(origin_ctx *ctx, const double *d_Xp, const long *d_xp_off,
                    const int *d_ld, int Nz, int ntiles, const int *d_tile_i,
                    const int *d_tile_j, const int *d_tile_a, long g_total, double *d_G,
                    const long *d_g_off) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_Xp && d_xp_off && d_ld && d_tile_i && d_tile_j && d_tile_a && d_G &&
                       d_g_off && Nz > 0 && ntiles > 0 && g_total > 0,
                   "bad arguments");
  // K-split so that small problems still put >= ~8 waves on every CU
  int ksplit = (int)(((long)ctx->num_cu * 8 + ntiles - 1) / ntiles);
  if (ksplit < 1) ksplit = 1;
  if (ksplit > 32) ksplit = 32;
  if (ksplit > Nz / 64) ksplit = Nz / 64 > 0 ? Nz / 64 : 1;
  void *scr = nullptr;
  int rc = origin_scratch(ctx, (size_t)ksplit * g_total * sizeof(double), &scr);
  if (rc) return rc;
  ProfScope ps(ctx, K_PCA_GRAM);
  hipLaunchKernelGGL(gram_kernel, dim3(ntiles, ksplit), dim3(64), 0, ctx->stream, d_Xp, d_xp_off,
                     d_ld, d_tile_i, d_tile_j, d_tile_a, Nz, ksplit, (double *)scr, d_g_off,
                     g_total);
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(ntiles), dim3(256), 0, ctx->stream,
                     (const double *)scr, g_total, ksplit, d_ld, d_tile_i, d_tile_j, d_tile_a, d_G,
                     d_g_off);
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

int origin_pca_uvec(origin_ctx *ctx, const double *d_Xp, const long *d_xp_off, const int *d_ld,
                    const int *d_n, int na, int Nz, const double *d_v, const long *d_v_off,
                    double *d_u) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_Xp && d_xp_off && d_ld && d_n && d_v && d_v_off && d_u && na > 0 && Nz > 0,
                   "bad arguments");
  ProfScope ps(ctx, K_PCA_UVEC);
  hipLaunchKernelGGL(xv_kernel, dim3(cdiv(Nz, 4), na), dim3(64, 4), 0, ctx->stream, d_Xp, d_xp_off,
                     d_ld, d_n, Nz, d_v, d_v_off, d_u);
  hipLaunchKernelGGL(normalize_kernel, dim3(na), dim3(1024), 0, ctx->stream, d_u, Nz);
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

int origin_pca_deflate(origin_ctx *ctx, float *d_F, int Nz, long S, const int *d_spx,
                       const long *d_spx_off, int na, long ntot, int nsmax, const double *d_u,
                       double *d_test) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_F && d_spx && d_spx_off && d_u && d_test && Nz > 0 && S > 0 && na > 0 &&
                       ntot > 0 && nsmax > 0,
                   "bad arguments");
  const long blocks = (long)cdiv(nsmax, 256) * na;
  int nzs = (int)(((long)ctx->num_cu * 8 + blocks - 1) / blocks);
  if (nzs < 1) nzs = 1;
  if (nzs > 32) nzs = 32;
  if (nzs > Nz) nzs = Nz;
  const int zper = cdiv(Nz, nzs);
  nzs = cdiv(Nz, zper);
  void *scr = nullptr;
  int rc = origin_scratch(ctx, (size_t)2 * nzs * ntot * sizeof(double), &scr);
  if (rc) return rc;
  double *cpart = (double *)scr;
  double *o2part = cpart + (size_t)nzs * ntot;
  dim3 grid(cdiv(nsmax, 256), nzs, na);
  {
    ProfScope ps(ctx, K_PCA_DEFLATE_DOT);
    hipLaunchKernelGGL(deflate_dot_kernel, grid, dim3(256), 0, ctx->stream, d_F, Nz, S, d_spx,
                       d_spx_off, d_u, zper, cpart, ntot);
  }
  {
    ProfScope ps(ctx, K_PCA_DEFLATE_UPDATE);
    hipLaunchKernelGGL(deflate_update_kernel, grid, dim3(256), 0, ctx->stream, d_F, Nz, S, d_spx,
                       d_spx_off, d_u, zper, nzs, cpart, o2part, ntot);
  }
  ProfScope ps(ctx, K_SMALL);
  hipLaunchKernelGGL(deflate_final_kernel, dim3(cdiv(ntot, 256)), dim3(256), 0, ctx->stream, d_spx,
                     ntot, nzs, Nz, o2part, d_test);
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

}  // extern "C"
