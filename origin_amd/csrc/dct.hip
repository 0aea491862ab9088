// DCT continuum fit + standardisation + O2 test  (SURVEY.md 2.2 rows k1-k3).
//
// Replaces dct_residual (reference muse_origin/lib_origin.py:150-240) and the dense
// lines of Preprocessing.run (steps.py:431-450, :463-465) and O2test (lib :957-974).
//
// Data layout: cubes are (Nz, S) with S = Ny*Nx spaxels, x fastest.  A wavefront owns 64
// consecutive spaxels and marches along z, so every global access of the march is one
// fully coalesced 256-B line; all per-spaxel state lives in VGPRs.  Quantities that are
// uniform across a wave (the cosine table row of the current channel) are fetched with
// scalar loads.
//
// Algebra.  With theta_z = (z + 1/2) pi / Nz the DCT-II atoms of DCTMAT (lib :127-146)
// are D[z][a] = s_a sqrt(2/Nz) cos(a theta_z), s_0 = 1/sqrt 2.  Because
// cos(a t) cos(b t) = (cos((a-b) t) + cos((a+b) t)) / 2, the 66 distinct entries of the
// weighted Gram matrix D^T S^-1 D collapse onto 2*order+1 moments
//      M_k = sum_z w_z cos(k theta_z),          w_z = 1 / var_z,
// and the continuum  D (D^T S^-1 D)^-1 D^T S^-1 s  (lib :233-235) becomes
//      cont[z] = sum_a y_a cos(a theta_z),   H y = 2 Rw,
//      H_ab = M_|a-b| + M_(a+b),   Rw_a = sum_z w_z s_z cos(a theta_z),
// (the s_a factors cancel).  The unweighted fit D D^T s (lib :191-194, :237) is
//      y_0 = R0_0 / Nz,  y_a = 2 R0_a / Nz,   R0_a = sum_z s_z cos(a theta_z).
// So one z-march accumulates (2*order+1) + 2*(order+1) float64 moments per spaxel instead
// of 77 Gram/RHS entries, then solves the (order+1)^2 SPD system by an in-register
// Cholesky.  Everything is accumulated in float64: the fit is HBM-bound (9 B/voxel
// against 43 DFMA/voxel), so exact-ish arithmetic is free.
#include <cmath>
#include <vector>

#include "common.h"
#include <type_traits>

namespace {

constexpr int kMaxOrder = 12;

// ------------------------------------------------------------------------------------
// cosine table  ctab[z][k] = cos(k theta_z), k = 0 .. 2*order     (float64, device)
// ------------------------------------------------------------------------------------
int make_ctab(origin_ctx *ctx, int Nz, int order, double **d_tab) {
  if (ctx->ctab && ctx->ctab_nz == Nz && ctx->ctab_order == order) {
    *d_tab = ctx->ctab;
    return ORIGIN_OK;
  }
  const int NK = 2 * order + 1;
  std::vector<double> h((size_t)Nz * NK);
  for (int z = 0; z < Nz; ++z)
    for (int k = 0; k < NK; ++k) h[(size_t)z * NK + k] = std::cos((z + 0.5) * (M_PI / Nz) * k);
  if (ctx->ctab) {
    ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
    // (a continuum pass on the auxiliary stream may still be reading the table)
    if (ctx->aux_stream && ctx->aux_pending) ORIGIN_HIP(hipStreamSynchronize(ctx->aux_stream));
    ORIGIN_HIP(hipFree(ctx->ctab));
    ctx->ctab = nullptr;
  }
  void *p = nullptr;
  ORIGIN_HIP(hipMalloc(&p, h.size() * sizeof(double)));
  hipError_t e = hipMemcpyAsync(p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice,
                                ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    hipFree(p);
    origin_set_error("ctab upload: %s", hipGetErrorString(e));
    return ORIGIN_E_HIP;
  }
  ctx->ctab = (double *)p;
  ctx->ctab_nz = Nz;
  ctx->ctab_order = order;
  *d_tab = ctx->ctab;
  return ORIGIN_OK;
}

struct CtabGuard {  // the table is cached in the context; nothing to release per call
  double *p = nullptr;
  explicit CtabGuard(origin_ctx *) {}
};

// ------------------------------------------------------------------------------------
// pass 1: moments + solve.  block = (64 lanes, ZS waves); wave w marches z = w, w+ZS, ...
// The weighted fit needs the 2*ORDER+1 moments M_k and the ORDER+1 weighted projections Rw;
// the unweighted projections R0 = D^T s are only used by spaxels that fall back to the plain
// DCT (a masked voxel, lib :226/:237; singular weights; dct_approx): they are left to a second
// march (dct_r0_kernel) over just the 64-spaxel groups that have such a spaxel -- none on a clean
// field -- which takes a quarter of the float64 FMAs out of this kernel.
// mom rows: [0, NK) M, [NK, NK+NA) Rw, NK+NA: any masked voxel.
// ------------------------------------------------------------------------------------
// How the rows reach the wave.  A row of 64 spaxels is 256 B per array and the next row is a whole
// plane away; read this way a plain march reaches 5.7-6.1 TB/s (tools/rowmarch_probe.hip), so the
// pattern is not the limit.  hipcc used to place the three loads of a row right before their use,
// one row at a time: MOM_BATCH rows are now requested back-to-back before the first is used, and
// the rows of a batch are kept apart by scheduling barriers so that each waits for its own loads
// only.  What bounds the kernel after that is the float64 arithmetic (49 VALU per row and wave,
// 32 of them FMAs whose table operand arrives by scalar loads that a row has to wait for) at four
// waves per SIMD: measured 3.2 ms against 2.0 ms for the VALU work alone and 1.9 ms for the
// bytes.  Tried and measured slower or equal: an LDS-DMA ring (`global_load_lds_*` eight rows
// ahead: M0 / SALU overhead per row), batches of 2, 4 and 16 rows, the table row pinned in the
// scalar cache (-4 %: the scalar loads are not the limit on their own).
__device__ __forceinline__ double wave_sum_d64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

#ifndef MOM_BATCH
#define MOM_BATCH 8
#endif
constexpr int MOM_PITCH = 68;  // floats per row of the wave's transpose buffer (FOLD)
constexpr int MOM_TR_FLOATS = 16 * MOM_PITCH;

// FOLD: the per-channel sums of pass 2 (sum over the unmasked spaxels of raw[z, .]) are taken
// here, where raw and mask are in registers anyway: the 64 values of a row (0 for masked and dead
// lanes) go to a wave-private LDS buffer, and every 16 rows the buffer is read transposed
// (lane = row & 15, quarter = lane >> 4), summed in float64 in a fixed order and written to
// part[group][z].  That replaces the 5 B/voxel plane pass by a 1/8 B/voxel partial array.
template <int ORDER, bool FOLD>
__global__ __launch_bounds__(512) void dct_moments_kernel(const float *__restrict__ raw,
                                                          const float *__restrict__ var,
                                                          const uint8_t *__restrict__ mask,
                                                          const double *__restrict__ ctab,
                                                          int Nz, long S, int zchunk,
                                                          double *__restrict__ mom,
                                                          double *__restrict__ part) {
  constexpr int NA = ORDER + 1;
  constexpr int NK = 2 * ORDER + 1;
  constexpr int NACC = NK + NA;
  extern __shared__ double lds[];  // [ZS][MOM_TR_FLOATS] floats (FOLD), then [ZS-1][NACC+1][64] doubles

  const int lane = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int ZS = blockDim.y;
  const long s = (long)blockIdx.x * 64 + lane;
  const bool live = s < S;
  const long sc = live ? s : S - 1;  // clamp: dead lanes redo the last spaxel, never store
  float *tr = reinterpret_cast<float *>(lds) + wave * MOM_TR_FLOATS;
  double *xw = lds + (FOLD ? ZS * MOM_TR_FLOATS / 2 : 0);  // cross-wave area

  double M[NK], Rw[NA];
#pragma unroll
  for (int k = 0; k < NK; ++k) M[k] = 0.0;
#pragma unroll
  for (int a = 0; a < NA; ++a) Rw[a] = 0.0;
  int anymask = 0;
  // MIRROR PAIRS.  theta_z = (z + 1/2) pi / Nz, so channel z' = Nz - 1 - z has cos(k theta_z') =
  // (-1)^k cos(k theta_z): the pair (z, z') shares ONE table row, and with e = w_z + w_z',
  // o = w_z - w_z' the 21 + 11 moment updates of the two channels are 32 FMAs instead of 64 (even
  // k take e, odd k take o) -- the kernel was bound by float64 issue.  A wave owns the pairs
  // p = pc0 + wave + i ZS of its chunk (blockIdx.y: a chunk of pairs; partial moments of the
  // chunks are summed by the solve kernel, and the split gives several times more blocks than the
  // chip holds at once, so that the last round of blocks is a small part of the run).  With odd
  // Nz the middle channel is its own mirror image and is added by the last chunk's first wave.
  const int npair = Nz / 2;
  const int pc0 = blockIdx.y * zchunk, pc1 = min(npair, pc0 + zchunk);
  int nrows = 0;  // rows buffered so far: pair i of the wave fills slots 2 i (front), 2 i + 1 (back)

  // buffered rows [i0, i0 + cnt) -> part[group][z].  Every lane ends with the sum of its row: the
  // four lanes of a row store the same value (no branch for a full buffer: a block boundary here
  // lets the optimiser sink the FMA chains of the rows before below it)
  auto flush = [&](int i0, auto cnt_c) {
    const int cnt = cnt_c;
    const int rrow = lane & 15, seg = lane >> 4;
    const float4 *p = reinterpret_cast<const float4 *>(tr + rrow * MOM_PITCH + seg * 16);
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 f = p[q];
      acc += (double)f.x;
      acc += (double)f.y;
      acc += (double)f.z;
      acc += (double)f.w;
    }
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    const int pr = pc0 + wave + ((i0 + rrow) >> 1) * ZS;  // the pair of this buffered row
    const int zz = (rrow & 1) ? Nz - 1 - pr : pr;
    double *dst = part + (long)blockIdx.x * Nz + zz;
    if (cnt == 16)
      *dst = acc;
    else if (rrow < cnt)
      *dst = acc;
  };
  // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: the weight is a float32
  // quantity either way.  var = inf (masked) -> weight 0
  auto pair = [&](int p, float rf, float vf, int mf, float rb, float vb, int mb) {
    anymask |= mf | mb;
    const double *ct = ctab + (long)p * NK;  // wave-uniform -> scalar loads
    const double wf = (double)__builtin_amdgcn_rcpf(vf), wb = (double)__builtin_amdgcn_rcpf(vb);
    const double xf = wf * (double)rf, xb = wb * (double)rb;
    const double we = wf + wb, wo = wf - wb, xe = xf + xb, xo = xf - xb;
    if constexpr (FOLD) {
      tr[(nrows & 15) * MOM_PITCH + lane] = (mf || !live) ? 0.0f : rf;
      tr[((nrows + 1) & 15) * MOM_PITCH + lane] = (mb || !live) ? 0.0f : rb;
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) M[k] = fma((k & 1) ? wo : we, ct[k], M[k]);
#pragma unroll
    for (int a = 0; a < NA; ++a) Rw[a] = fma((a & 1) ? xo : xe, ct[a], Rw[a]);
    nrows += 2;
  };

  constexpr int B = MOM_BATCH / 2;  // pairs per batch: MOM_BATCH rows requested back to back
  static_assert(8 % B == 0, "a transpose buffer (8 pairs) is a whole number of batches");
  auto batch = [&](int p) {
    float rf[B], vf[B], rb[B], vb[B];
    int mf[B], mb[B];
#pragma unroll
    for (int u = 0; u < B; ++u) {
      const int pp = p + u * ZS;
      const long i0 = (long)pp * S + sc, i1 = (long)(Nz - 1 - pp) * S + sc;
      rf[u] = raw[i0], vf[u] = var[i0], mf[u] = mask[i0];
      rb[u] = raw[i1], vb[u] = var[i1], mb[u] = mask[i1];
    }
#pragma unroll
    for (int u = 0; u < B; ++u) {
      __builtin_amdgcn_sched_barrier(0);  // pairs in order: pair u waits for ITS six loads only
      pair(p + u * ZS, rf[u], vf[u], mf[u], rb[u], vb[u], mb[u]);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  int p = pc0 + wave;
  for (; p + 7 * ZS < pc1; p += 8 * ZS) {  // trips of 8 pairs = 16 rows = one transpose buffer
#pragma unroll
    for (int b = 0; b < 8 / B; ++b) batch(p + b * B * ZS);
    if constexpr (FOLD) flush(nrows - 16, std::integral_constant<int, 16>{});
  }
  for (; p < pc1; p += ZS) {  // the last, incomplete trip, pair by pair
    const long i0 = (long)p * S + sc, i1 = (long)(Nz - 1 - p) * S + sc;
    pair(p, raw[i0], var[i0], mask[i0], raw[i1], var[i1], mask[i1]);
  }
  if constexpr (FOLD) {
    if (nrows & 15) flush(nrows & ~15, nrows & 15);
  }
  if ((Nz & 1) && blockIdx.y == gridDim.y - 1 && wave == 0) {  // the middle channel
    const int zm = npair;
    const long idx = (long)zm * S + sc;
    const float r = raw[idx], v = var[idx];
    const int mk = mask[idx];
    anymask |= mk;
    const double *ct = ctab + (long)zm * NK;
    const double wd = (double)__builtin_amdgcn_rcpf(v);
    const double wr = wd * (double)r;
#pragma unroll
    for (int k = 0; k < NK; ++k) M[k] = fma(wd, ct[k], M[k]);
#pragma unroll
    for (int a = 0; a < NA; ++a) Rw[a] = fma(wr, ct[a], Rw[a]);
    if constexpr (FOLD) {
      const double sum = wave_sum_d64((mk || !live) ? 0.0 : (double)r);
      if (lane == 0) part[(long)blockIdx.x * Nz + zm] = sum;
    }
  }

  // cross-wave reduction (fixed order -> deterministic)
  if (ZS > 1) {
    if (wave > 0) {
      double *dst = xw + (long)(wave - 1) * (NACC + 1) * 64 + lane;
#pragma unroll
      for (int k = 0; k < NK; ++k) dst[k * 64] = M[k];
#pragma unroll
      for (int a = 0; a < NA; ++a) dst[(NK + a) * 64] = Rw[a];
      dst[NACC * 64] = (double)anymask;
    }
    __syncthreads();
    if (wave > 0) return;
    for (int w = 1; w < ZS; ++w) {
      const double *src = xw + (long)(w - 1) * (NACC + 1) * 64 + lane;
#pragma unroll
      for (int k = 0; k < NK; ++k) M[k] += src[k * 64];
#pragma unroll
      for (int a = 0; a < NA; ++a) Rw[a] += src[(NK + a) * 64];
      anymask |= (int)src[NACC * 64];
    }
  }

  // moments of this spaxel -> global (the solve runs in a second, register-hungry kernel so
  // that this streaming kernel keeps a high occupancy)
  if (live) {
    double *mc = mom + (long)blockIdx.y * (NACC + 1) * S;
#pragma unroll
    for (int k = 0; k < NK; ++k) mc[(long)k * S + s] = M[k];
#pragma unroll
    for (int a = 0; a < NA; ++a) mc[(long)(NK + a) * S + s] = Rw[a];
    mc[(long)NACC * S + s] = (double)anymask;
  }
}

// per spaxel: H y = 2 Rw by Cholesky (valid spaxels).  Spaxels that take the plain DCT
// coefficients instead are flagged in need[] for dct_r0_kernel.  approx: every spaxel (mom is
// not read).
template <int ORDER>
__global__ __launch_bounds__(256) void dct_solve_kernel(const double *__restrict__ mom, int nzc,
                                                        long S, int approx,
                                                        double *__restrict__ coef,
                                                        uint8_t *__restrict__ need) {
  constexpr int NA = ORDER + 1;
  constexpr int NK = 2 * ORDER + 1;
  constexpr int NACC = NK + NA;
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  if (approx) {
    need[s] = 1;
    return;
  }
  double M[NK], Rw[NA];
#pragma unroll
  for (int k = 0; k < NK; ++k) M[k] = mom[(long)k * S + s];
#pragma unroll
  for (int a = 0; a < NA; ++a) Rw[a] = mom[(long)(NK + a) * S + s];
  int anymask = mom[(long)NACC * S + s] != 0.0;
  for (int c = 1; c < nzc; ++c) {  // the channel chunks of the moments pass, in order
    const double *mc = mom + (long)c * (NACC + 1) * S;
#pragma unroll
    for (int k = 0; k < NK; ++k) M[k] += mc[(long)k * S + s];
#pragma unroll
    for (int a = 0; a < NA; ++a) Rw[a] += mc[(long)(NK + a) * S + s];
    anymask |= mc[(long)NACC * S + s] != 0.0;
  }
  double y[NA];
  bool weighted = !anymask;  // valid = ~any(mask, axis=0)   (lib :226)
  if (weighted) {
    // H = L L^T, in place (lower triangle), fully unrolled -> registers
    double L[NA][NA];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) L[a][b] = M[a - b] + M[a + b];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      double d = L[j][j];
#pragma unroll
      for (int k = 0; k < j; ++k) d = fma(-L[j][k], L[j][k], d);
      ok = ok && (d > 0.0);
      const double inv = 1.0 / sqrt(d);
      L[j][j] = d * inv;
#pragma unroll
      for (int i = j + 1; i < NA; ++i) {
        double t = L[i][j];
#pragma unroll
        for (int k = 0; k < j; ++k) t = fma(-L[i][k], L[j][k], t);
        L[i][j] = t * inv;
      }
    }
    // L q = 2 Rw ;  L^T y = q
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      double t = 2.0 * Rw[i];
#pragma unroll
      for (int k = 0; k < i; ++k) t = fma(-L[i][k], y[k], t);
      y[i] = t / L[i][i];
    }
#pragma unroll
    for (int i = NA - 1; i >= 0; --i) {
      double t = y[i];
#pragma unroll
      for (int k = i + 1; k < NA; ++k) t = fma(-L[k][i], y[k], t);
      y[i] = t / L[i][i];
    }
    weighted = ok;  // singular weights (reference: LinAlgError) -> unweighted fit
  }
  need[s] = !weighted;
  if (weighted) {
#pragma unroll
    for (int a = 0; a < NA; ++a) coef[(long)a * S + s] = y[a];
  }
}

// plain DCT coefficients y = D^T s (with the 1/Nz, 2/Nz scaling) of the flagged spaxels:
// same block shape and z march as dct_moments_kernel; a block none of whose 64 spaxels is
// flagged leaves at once.
template <int ORDER>
__global__ __launch_bounds__(512) void dct_r0_kernel(const float *__restrict__ raw,
                                                     const uint8_t *__restrict__ need,
                                                     const double *__restrict__ ctab, int Nz,
                                                     long S, double *__restrict__ coef) {
  constexpr int NA = ORDER + 1;
  constexpr int NK = 2 * ORDER + 1;
  extern __shared__ double lds[];  // [ZS-1][NA][64]
  const int lane = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int ZS = blockDim.y;
  const long s = (long)blockIdx.x * 64 + lane;
  const bool live = s < S;
  const long sc = live ? s : S - 1;
  const bool flagged = live && need[sc];
  if (!__any(flagged)) return;  // the same 64 spaxels in every wave of the block: uniform
  double R0[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) R0[a] = 0.0;
#pragma unroll 4
  for (int z = wave; z < Nz; z += ZS) {
    const double rd = (double)raw[(long)z * S + sc];
    const double *ct = ctab + (long)z * NK;
#pragma unroll
    for (int a = 0; a < NA; ++a) R0[a] = fma(rd, ct[a], R0[a]);
  }
  if (ZS > 1) {
    if (wave > 0) {
      double *dst = lds + (long)(wave - 1) * NA * 64 + lane;
#pragma unroll
      for (int a = 0; a < NA; ++a) dst[a * 64] = R0[a];
    }
    __syncthreads();
    if (wave > 0) return;
    for (int w = 1; w < ZS; ++w) {
      const double *src = lds + (long)(w - 1) * NA * 64 + lane;
#pragma unroll
      for (int a = 0; a < NA; ++a) R0[a] += src[a * 64];
    }
  }
  if (flagged) {
    const double inl = 1.0 / (double)Nz;
    coef[s] = R0[0] * inl;
#pragma unroll
    for (int a = 1; a < NA; ++a) coef[(long)a * S + s] = 2.0 * R0[a] * inl;
  }
}

// cont[z, s] = sum_a coef[a][s] cos(a theta_z)
template <int ORDER>
__device__ __forceinline__ double eval_cont(const double (&c)[ORDER + 1], const double *ct) {
  double acc = c[0] * ct[0];
#pragma unroll
  for (int a = 1; a <= ORDER; ++a) acc = fma(c[a], ct[a], acc);
  return acc;
}

// The continuum of a MIRROR PAIR of channels (z, Nz - 1 - z) from the table row of z:
// cos(a theta) is even about the middle of the cube for even a and odd for odd a, so with
// E = sum_{a even} c_a cos(a theta_z), O = sum_{a odd} c_a cos(a theta_z) the two values are
// E + O and E - O -- one table row and ORDER + 1 FMAs for two voxels.
// 1 / sqrt(var) for the two quotients of the standardisation (steps.py:439-440): v_rsq_f32 (1 ulp;
// inf -> 0, 0 -> inf, negative -> NaN as the reference's float64 expression gives).  The IEEE
// sqrtf + division it replaces cost ~20 VALU instructions per voxel of a kernel that issues ~65.
__device__ __forceinline__ float dct_inv_std(float v) { return __builtin_amdgcn_rsqf(v); }

// element of a row whose base is wave-uniform at a 32-bit byte offset of the lane
template <typename T>
__device__ __forceinline__ T &dct_at(T *row, unsigned byte_off) {
  return *reinterpret_cast<T *>(reinterpret_cast<char *>(const_cast<std::remove_const_t<T> *>(row)) + byte_off);
}

template <int ORDER>
__device__ __forceinline__ void eval_cont_pair(const double (&c)[ORDER + 1], const double *ct,
                                               double &front, double &back) {
  double E = c[0] * ct[0], O = 0.0;
  if constexpr (ORDER >= 1) O = c[1] * ct[1];
#pragma unroll
  for (int a = 2; a <= ORDER; ++a) {
    if (a & 1)
      O = fma(c[a], ct[a], O);
    else
      E = fma(c[a], ct[a], E);
  }
  front = E + O;
  back = E - O;
}

template <int ORDER>
__global__ __launch_bounds__(256) void dct_continuum_kernel(const double *__restrict__ coef,
                                                            const double *__restrict__ ctab,
                                                            int Nz, long S, int zchunk,
                                                            float *__restrict__ cont) {
  constexpr int NA = ORDER + 1;
  constexpr int NK = 2 * ORDER + 1;
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  double c[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) c[a] = coef[(long)a * S + s];
  const int z0 = blockIdx.y * zchunk;
  const int z1 = min(Nz, z0 + zchunk);
  for (int z = z0; z < z1; ++z)
    cont[(long)z * S + s] = (float)eval_cont<ORDER>(c, ctab + (long)z * NK);
}

// ------------------------------------------------------------------------------------
// pass 2: per-channel sum / count of (raw - cont) over unmasked spaxels.
//   sum_s (raw - cont) = sum_{s unmasked} raw[z,s]
//                        - sum_a cos(a theta_z) (Ctot_a - sum_{s masked at z} coef[a][s])
// so the plane reduction only touches raw + mask (5 B/voxel); coefficients are gathered
// for masked voxels only.  One block reduces `SPB` spaxels of one channel.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int ORDER, bool VEC>
__global__ __launch_bounds__(256) void dct_plane_sums_kernel(
    const float *__restrict__ raw, const uint8_t *__restrict__ mask,
    const double *__restrict__ coef, const double *__restrict__ ctab, long S, int spb,
    double *__restrict__ part /* [Nz][nchunk][2] */) {
  constexpr int NA = ORDER + 1;
  constexpr int NK = 2 * ORDER + 1;
  __shared__ double red[2][4];
  const int z = blockIdx.y;
  const long s0 = (long)blockIdx.x * spb;
  const long s1 = min(S, s0 + spb);
  const float *r = raw + (long)z * S;
  const uint8_t *m = mask + (long)z * S;
  const double *ct = ctab + (long)z * NK;
  double sum = 0.0, cnt = 0.0;
  auto one = [&](long s, float v, uint8_t mk) {
    if (mk) {
      // masked voxel: remove its continuum from the "all spaxels" term
      double c = coef[s] * ct[0];
#pragma unroll
      for (int a = 1; a < NA; ++a) c = fma(coef[(long)a * S + s], ct[a], c);
      sum += c;
    } else {
      sum += (double)v;
      cnt += 1.0;
    }
  };
  if constexpr (VEC) {  // S % 4 == 0, spb % 4 == 0: 16-byte loads of raw, 4-byte loads of mask
    for (long s = s0 + 4 * threadIdx.x; s < s1; s += 1024) {
      const float4 v = *reinterpret_cast<const float4 *>(r + s);
      const uchar4 mk = *reinterpret_cast<const uchar4 *>(m + s);
      one(s, v.x, mk.x);
      one(s + 1, v.y, mk.y);
      one(s + 2, v.z, mk.z);
      one(s + 3, v.w, mk.w);
    }
  } else {
    for (long s = s0 + threadIdx.x; s < s1; s += 256) one(s, r[s], m[s]);
  }
  sum = wave_sum(sum);
  cnt = wave_sum(cnt);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][wave] = sum;
    red[1][wave] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double *p = part + ((long)z * gridDim.x + blockIdx.x) * 2;
    p[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    p[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// Ctot_a = sum_s coef[a][s]: one block per a, fixed-order tree
__global__ __launch_bounds__(1024) void coef_total_kernel(const double *__restrict__ coef, long S,
                                                          double *__restrict__ ctot) {
  __shared__ double red[16];
  const int a = blockIdx.x;
  double acc = 0.0;
  for (long s = threadIdx.x; s < S; s += 1024) acc += coef[(long)a * S + s];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += red[i];
    ctot[a] = t;
  }
}

template <int ORDER>
__global__ __launch_bounds__(256) void dct_zsum_final_kernel(const double *__restrict__ part,
                                                             const double *__restrict__ ctot,
                                                             const double *__restrict__ ctab,
                                                             int Nz, int nchunk,
                                                             double *__restrict__ zsum,
                                                             double *__restrict__ zcnt) {
  constexpr int NA = ORDER + 1;
  constexpr int NK = 2 * ORDER + 1;
  const int z = blockIdx.x * 256 + threadIdx.x;
  if (z >= Nz) return;
  double sum = 0.0, cnt = 0.0;
  for (int c = 0; c < nchunk; ++c) {
    sum += part[((long)z * nchunk + c) * 2];
    cnt += part[((long)z * nchunk + c) * 2 + 1];
  }
  const double *ct = ctab + (long)z * NK;
  double call = 0.0;
#pragma unroll
  for (int a = 0; a < NA; ++a) call = fma(ctot[a], ct[a], call);
  zsum[z] = sum - call;
  zcnt[z] = cnt;
}

// ---- folded form of pass 2 (origin_dct_fit_sums): dct_moments_kernel<., true> has left
// part[group][z] = sum of raw over the group's unmasked spaxels.  Masked voxels: the groups
// that have any (`need` of the solve kernel: a spaxel with a masked voxel, or singular weights)
// are visited once more, mask bytes only, 64 channels per wave:
//   part[group][z] += sum_{masked lanes} cont[z, s]      (they are in "all spaxels" Ctot)
//   nmask[group][z] = masked lanes
template <int ORDER>
__global__ __launch_bounds__(64) void dct_masked_corr_kernel(const uint8_t *__restrict__ mask,
                                                            const uint8_t *__restrict__ flagged,
                                                            const double *__restrict__ coef,
                                                            const double *__restrict__ ctab,
                                                            int Nz, long S,
                                                            double *__restrict__ part,
                                                            uint8_t *__restrict__ nmask) {
  constexpr int NA = ORDER + 1;
  constexpr int NK = 2 * ORDER + 1;
  const int lane = threadIdx.x;
  const long s = (long)blockIdx.x * 64 + lane;
  const bool live = s < S;
  const long sc = live ? s : S - 1;
  if (!__any(live && flagged[sc] != 0)) return;
  double c[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) c[a] = coef[(long)a * S + sc];
  const int z0 = blockIdx.y * 64;
  const int z1 = min(Nz, z0 + 64);
  for (int zb = z0; zb < z1; zb += 16) {
    int mk[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) mk[j] = mask[(long)min(zb + j, Nz - 1) * S + sc];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int z = zb + j;
      const bool m = live && mk[j] && z < z1;
      const unsigned long long b = __ballot(m);
      if (b == 0) continue;  // (uniform)
      double v = m ? eval_cont<ORDER>(c, ctab + (long)z * NK) : 0.0;
      v = wave_sum(v);
      if (lane == 0) {
        part[(long)blockIdx.x * Nz + z] += v;
        nmask[(long)blockIdx.x * Nz + z] = (uint8_t)__popcll(b);
      }
    }
  }
}

// sum over the groups, in two fixed-order stages: slab[c][z] = sum of GSLAB groups
constexpr int GSLAB = 64;
__global__ __launch_bounds__(256) void dct_part_reduce_kernel(const double *__restrict__ part,
                                                             const uint8_t *__restrict__ nmask,
                                                             int Nz, int ngroups,
                                                             double *__restrict__ slab) {
  __shared__ double red[2][4][64];
  const int zl = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int z = blockIdx.x * 64 + zl;
  const int g0 = blockIdx.y * GSLAB, g1 = min(ngroups, g0 + GSLAB);
  double sum = 0.0, cnt = 0.0;
  if (z < Nz)
    for (int g = g0 + ph; g < g1; g += 4) {
      sum += part[(long)g * Nz + z];
      cnt += (double)nmask[(long)g * Nz + z];
    }
  red[0][ph][zl] = sum, red[1][ph][zl] = cnt;
  __syncthreads();
  if (ph == 0 && z < Nz) {
    double *o = slab + ((long)blockIdx.y * Nz + z) * 2;
    o[0] = (red[0][0][zl] + red[0][1][zl]) + (red[0][2][zl] + red[0][3][zl]);
    o[1] = (red[1][0][zl] + red[1][1][zl]) + (red[1][2][zl] + red[1][3][zl]);
  }
}

template <int ORDER>
__global__ __launch_bounds__(256) void dct_zsum_fold_final_kernel(
    const double *__restrict__ slab, int nslab, const double *__restrict__ ctot,
    const double *__restrict__ ctab, int Nz, long S, double *__restrict__ zsum,
    double *__restrict__ zcnt) {
  constexpr int NA = ORDER + 1;
  constexpr int NK = 2 * ORDER + 1;
  const int z = blockIdx.x * 256 + threadIdx.x;
  if (z >= Nz) return;
  double sum = 0.0, nm = 0.0;
  for (int c = 0; c < nslab; ++c) {
    sum += slab[((long)c * Nz + z) * 2];
    nm += slab[((long)c * Nz + z) * 2 + 1];
  }
  const double *ct = ctab + (long)z * NK;
  double call = 0.0;
#pragma unroll
  for (int a = 0; a < NA; ++a) call = fma(ctot[a], ct[a], call);
  zsum[z] = sum - call;
  zcnt[z] = (double)S - nm;
}

// ------------------------------------------------------------------------------------
// pass 3: standardise.  grid (spaxel blocks, z chunks); per-spaxel sums go to partials.
// ------------------------------------------------------------------------------------
template <int ORDER>
__global__ __launch_bounds__(256) void dct_standardize_kernel(
    const float *__restrict__ raw, const float *__restrict__ var,
    const uint8_t *__restrict__ mask, const double *__restrict__ coef,
    const double *__restrict__ ctab, const double *__restrict__ zmean, int Nz, long S, int zchunk,
    float *__restrict__ cube_std, float *__restrict__ cont_dct,
    double *__restrict__ part /* [nzc][3][S] or null */) {
  constexpr int NA = ORDER + 1;
  constexpr int NK = 2 * ORDER + 1;
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  double c[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) c[a] = coef[(long)a * S + s];
  // blockIdx.y: a chunk of mirror pairs of channels (eval_cont_pair); the middle channel of an
  // odd Nz goes with the last chunk
  const int npair = Nz / 2;
  const int p0 = blockIdx.y * zchunk, p1 = min(npair, p0 + zchunk);
  double a_std = 0.0, a_dct = 0.0, a_o2 = 0.0;
  // addresses: the channel's row base is wave-uniform (scalar registers), the lane adds a 32-bit
  // offset -- no 64-bit vector arithmetic per access
  const unsigned so4 = (unsigned)s * 4u, so1 = (unsigned)s;
  auto voxel = [&](long row, double cont, double mean) {
    const float r = dct_at(raw + row, so4);
    const float v = dct_at(var + row, so4);
    const bool mk = dct_at(mask + row, so1) != 0;
    // std = sqrt(var) (steps.py:439); both quotients share one reciprocal
    const float rs = dct_inv_std(v);
    const float t = (float)(((double)r - cont) - mean);  // nanmean over unmasked spaxels (:442)
    const float o = mk ? 0.0f : t * rs;     // data[mask] = 0                (steps.py:446)
    const float cd = (float)cont * rs;      // cont_dct /= std ; astype(f32) (:440, :463)
    dct_at(cube_std + row, so4) = o;
    if (cont_dct) dct_at(cont_dct + row, so4) = cd;
    a_std += (double)o;
    a_dct += (double)cd;
    a_o2 = fma((double)o, (double)o, a_o2);
  };
#pragma unroll 2
  for (int p = p0; p < p1; ++p) {
    const int zb = Nz - 1 - p;
    double cf, cb;
    eval_cont_pair<ORDER>(c, ctab + (long)p * NK, cf, cb);
    voxel((long)p * S, cf, zmean[p]);
    voxel((long)zb * S, cb, zmean[zb]);
  }
  if ((Nz & 1) && blockIdx.y == gridDim.y - 1)
    voxel((long)npair * S, eval_cont<ORDER>(c, ctab + (long)npair * NK), zmean[npair]);
  if (part) {
    double *p = part + (long)blockIdx.y * 3 * S + s;
    p[0] = a_std;
    p[S] = a_dct;
    p[2 * S] = a_o2;
  }
}

// cont_dct = cont / sqrt(var) alone (steps.py:440, :463) with the partial sums of its mean over
// z: the half of dct_standardize_kernel nothing downstream of the O2 map waits for.  Run as a
// pass of its own it overlaps the host's threshold fit (origin_dct_cont_std).  Same expressions
// as the fused kernel: identical values.
template <int ORDER>
__global__ __launch_bounds__(256) void dct_cont_std_kernel(const float *__restrict__ var,
                                                           const double *__restrict__ coef,
                                                           const double *__restrict__ ctab, int Nz,
                                                           long S, int zchunk,
                                                           float *__restrict__ cont_dct,
                                                           double *__restrict__ part /* [nzc][S] */) {
  constexpr int NA = ORDER + 1;
  constexpr int NK = 2 * ORDER + 1;
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  double c[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) c[a] = coef[(long)a * S + s];
  const int npair = Nz / 2;  // (mirror pairs, as dct_standardize_kernel: identical values)
  const int p0 = blockIdx.y * zchunk, p1 = min(npair, p0 + zchunk);
  double a_dct = 0.0;
  const unsigned so4 = (unsigned)s * 4u;
  auto voxel = [&](long row, double cont) {
    const float rs = dct_inv_std(dct_at(var + row, so4));
    const float cd = (float)cont * rs;
    dct_at(cont_dct + row, so4) = cd;
    a_dct += (double)cd;
  };
#pragma unroll 2
  for (int p = p0; p < p1; ++p) {
    double cf, cb;
    eval_cont_pair<ORDER>(c, ctab + (long)p * NK, cf, cb);
    voxel((long)p * S, cf);
    voxel((long)(Nz - 1 - p) * S, cb);
  }
  if ((Nz & 1) && blockIdx.y == gridDim.y - 1)
    voxel((long)npair * S, eval_cont<ORDER>(c, ctab + (long)npair * NK));
  if (part) part[(long)blockIdx.y * S + s] = a_dct;
}

__global__ __launch_bounds__(256) void cont_image_final_kernel(const double *__restrict__ part,
                                                               int nzc, long S, int Nz,
                                                               float *__restrict__ ima_dct) {
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  double b = 0.0;
  for (int k = 0; k < nzc; ++k) b += part[(long)k * S + s];
  ima_dct[s] = (float)(b * (1.0 / (double)Nz));
}

// the per-channel mean is wave-uniform: divide once per channel, not once per voxel
__global__ __launch_bounds__(256) void zmean_kernel(const double *__restrict__ zsum,
                                                    const double *__restrict__ zcnt, int Nz,
                                                    double *__restrict__ zmean) {
  const int z = blockIdx.x * 256 + threadIdx.x;
  if (z < Nz) zmean[z] = zsum[z] / zcnt[z];
}

__global__ __launch_bounds__(256) void std_images_final_kernel(const double *__restrict__ part,
                                                               int nzc, long S, int Nz,
                                                               float *__restrict__ ima_std,
                                                               float *__restrict__ ima_dct,
                                                               double *__restrict__ o2) {
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  double a = 0.0, b = 0.0, c = 0.0;
  for (int k = 0; k < nzc; ++k) {
    const double *p = part + (long)k * 3 * S + s;
    a += p[0];
    b += p[S];
    c += p[2 * S];
  }
  const double inv = 1.0 / (double)Nz;
  if (ima_std) ima_std[s] = (float)(a * inv);
  if (ima_dct) ima_dct[s] = (float)(b * inv);
  if (o2) o2[s] = c * inv;
}

// ------------------------------------------------------------------------------------
// O2 test on any cube
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void o2_partial_kernel(const float *__restrict__ cube, int Nz,
                                                         long S, int zchunk,
                                                         double *__restrict__ part) {
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  const int z0 = blockIdx.y * zchunk;
  const int z1 = min(Nz, z0 + zchunk);
  double acc = 0.0;
#pragma unroll 4
  for (int z = z0; z < z1; ++z) {
    const double v = (double)cube[(long)z * S + s];
    acc = fma(v, v, acc);
  }
  part[(long)blockIdx.y * S + s] = acc;
}

__global__ __launch_bounds__(256) void o2_final_kernel(const double *__restrict__ part, int nzc,
                                                       long S, int Nz, double *__restrict__ out) {
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  double acc = 0.0;
  for (int k = 0; k < nzc; ++k) acc += part[(long)k * S + s];
  out[s] = acc / (double)Nz;
}

int pick_zchunks(origin_ctx *ctx, long S, int Nz) {
  // aim for >= 32 blocks of 256 threads per CU: several rounds of blocks, so that the last,
  // partly filled round is a small part of the pass (3681 x 600 x 600: 2 chunks 3.93 ms, 4-16
  // chunks 3.53-3.55 ms, 32 chunks 3.65 ms)
  const long blocks = (S + 255) / 256;
  long want = ((long)ctx->num_cu * 32 + blocks - 1) / blocks;
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  if (want > Nz) want = Nz;
  return (int)want;
}

#define DISPATCH_ORDER(order, CALL)              \
  switch (order) {                               \
    case 1: { CALL(1); } break;                  \
    case 2: { CALL(2); } break;                  \
    case 3: { CALL(3); } break;                  \
    case 4: { CALL(4); } break;                  \
    case 5: { CALL(5); } break;                  \
    case 6: { CALL(6); } break;                  \
    case 7: { CALL(7); } break;                  \
    case 8: { CALL(8); } break;                  \
    case 9: { CALL(9); } break;                  \
    case 10: { CALL(10); } break;                \
    case 11: { CALL(11); } break;                \
    case 12: { CALL(12); } break;                \
    default:                                     \
      origin_set_error("dct order %d unsupported (1..%d)", order, kMaxOrder); \
      return ORIGIN_E_ARG;                       \
  }

int check_dims(int Nz, int Ny, int Nx, int order) {
  ORIGIN_CHECK_ARG(Nz > 0 && Ny > 0 && Nx > 0, "bad cube shape (%d,%d,%d)", Nz, Ny, Nx);
  ORIGIN_CHECK_ARG(order >= 1 && order <= kMaxOrder, "dct order %d unsupported (1..%d)", order,
                   kMaxOrder);
  ORIGIN_CHECK_ARG(order + 1 <= Nz, "dct order %d needs at least %d channels", order, order + 1);
  return ORIGIN_OK;
}

}  // namespace

extern "C" {

// fit (+ the per-channel residual sums when d_zsum is given: origin_dct_fit_sums)
static int dct_fit_impl(origin_ctx *ctx, const float *d_raw, const float *d_var,
                        const uint8_t *d_mask, int Nz, int Ny, int Nx, int order, int approx,
                        double *d_coef, double *d_zsum, double *d_zcnt) {
  ORIGIN_USE(ctx);
  int rc = check_dims(Nz, Ny, Nx, order);
  if (rc) return rc;
  ORIGIN_CHECK_ARG(d_raw && d_var && d_mask && d_coef, "null pointer");
  const long S = (long)Ny * Nx;
  CtabGuard tab(ctx);
  rc = make_ctab(ctx, Nz, order, &tab.p);
  if (rc) return rc;
  const bool fold = d_zsum != nullptr;
  // z-split so that small fields still fill the chip: waves = S/64 * ZS >= ~4 per SIMD
  const long waves = (S + 63) / 64;
  int ZS = 1;
  while (ZS < 8 && waves * ZS < (long)ctx->num_cu * 16) ZS *= 2;
  const int NACC = (2 * order + 1) + (order + 1);
  while (ZS > 1 && (size_t)(ZS - 1) * (NACC + 1) * 64 * sizeof(double) > 64 * 1024) ZS /= 2;
  const size_t lds = (fold ? (size_t)ZS * MOM_TR_FLOATS * sizeof(float) : 0) +
                     (size_t)(ZS - 1) * (NACC + 1) * 64 * sizeof(double);
  const size_t lds_r0 = (size_t)(ZS - 1) * (order + 1) * 64 * sizeof(double);
  // scratch: mom | need | part [groups][Nz] | slab [nslab][Nz][2] | ctot | nmask [groups][Nz]
  const int nslab = cdiv((int)waves, GSLAB);
  // channel chunks of the moments pass: ~10 rounds of blocks (4 waves per SIMD), chunks >= 256
  // rows (measured at 3681 x 600 x 600: 2 chunks 3.55 ms, 4: 3.47, 8: 3.21, 12: 3.17)
  int nzc = (int)((10L * ctx->num_cu * 16 + waves * ZS - 1) / (waves * ZS));
  nzc = std::max(1, std::min(std::min(nzc, 16), Nz / 256));
  // (chunks of mirror PAIRS of channels: whole trips of 8 pairs = 16 rows of every wave)
  const int npair = std::max(1, Nz / 2);
  const int zchunk = cdiv(cdiv(npair, nzc), 8 * ZS) * 8 * ZS;
  nzc = cdiv(npair, zchunk);
  dim3 grid((unsigned)waves), gridm((unsigned)waves, nzc), block(64, ZS);
  const size_t mom_bytes = (size_t)nzc * (NACC + 1) * S * sizeof(double);
  const size_t need_bytes = ((size_t)S + 7) & ~(size_t)7;
  const size_t part_bytes = fold ? (size_t)waves * Nz * sizeof(double) : 0;
  const size_t slab_bytes = fold ? (size_t)nslab * Nz * 2 * sizeof(double) : 0;
  const size_t nm_bytes = fold ? (size_t)waves * Nz : 0;
  void *scr = nullptr;
  rc = origin_scratch(ctx, mom_bytes + need_bytes + part_bytes + slab_bytes + 64 * sizeof(double) +
                               nm_bytes, &scr);
  if (rc) return rc;
  double *mom = (double *)scr;
  uint8_t *need = (uint8_t *)scr + mom_bytes;
  double *part = (double *)((char *)scr + mom_bytes + need_bytes);
  double *slab = (double *)((char *)part + part_bytes);
  double *ctot = (double *)((char *)slab + slab_bytes);
  uint8_t *nmask = (uint8_t *)(ctot + 64);
  {
    ProfScope ps(ctx, K_DCT_FIT);
    if (fold) ORIGIN_HIP(hipMemsetAsync(nmask, 0, nm_bytes, ctx->stream));
#define CALL(O)                                                                                  \
  if (fold)                                                                                      \
    hipLaunchKernelGGL((dct_moments_kernel<O, true>), gridm, block, lds, ctx->stream, d_raw,     \
                       d_var, d_mask, tab.p, Nz, S, zchunk, mom, part);                          \
  else if (!approx)                                                                              \
    hipLaunchKernelGGL((dct_moments_kernel<O, false>), gridm, block, lds, ctx->stream, d_raw,    \
                       d_var, d_mask, tab.p, Nz, S, zchunk, mom, (double *)nullptr);             \
  hipLaunchKernelGGL(dct_solve_kernel<O>, dim3(cdiv(S, 256)), dim3(256), 0, ctx->stream, mom,    \
                     nzc, S, approx, d_coef, need);                                              \
  hipLaunchKernelGGL(dct_r0_kernel<O>, grid, block, lds_r0, ctx->stream, d_raw, need, tab.p,     \
                     Nz, S, d_coef)
    DISPATCH_ORDER(order, CALL)
#undef CALL
    ORIGIN_LAUNCH_CHECK();
  }
  if (!fold) return ORIGIN_OK;
  ProfScope ps(ctx, K_DCT_SUMS);
  hipLaunchKernelGGL(coef_total_kernel, dim3(order + 1), dim3(1024), 0, ctx->stream, d_coef, S,
                     ctot);
#define CALL(O)                                                                                  \
  hipLaunchKernelGGL(dct_masked_corr_kernel<O>, dim3((unsigned)waves, cdiv(Nz, 64)), dim3(64), 0, \
                     ctx->stream, d_mask, need, d_coef, tab.p, Nz, S, part, nmask);              \
  hipLaunchKernelGGL(dct_part_reduce_kernel, dim3(cdiv(Nz, 64), nslab), dim3(256), 0,            \
                     ctx->stream, part, nmask, Nz, (int)waves, slab);                            \
  hipLaunchKernelGGL(dct_zsum_fold_final_kernel<O>, dim3(cdiv(Nz, 256)), dim3(256), 0,           \
                     ctx->stream, slab, nslab, ctot, tab.p, Nz, S, d_zsum, d_zcnt)
  DISPATCH_ORDER(order, CALL)
#undef CALL
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

int origin_dct_fit(origin_ctx *ctx, const float *d_raw, const float *d_var,
                   const uint8_t *d_mask, int Nz, int Ny, int Nx, int order, int approx,
                   double *d_coef) {
  return dct_fit_impl(ctx, d_raw, d_var, d_mask, Nz, Ny, Nx, order, approx, d_coef, nullptr,
                      nullptr);
}

int origin_dct_fit_sums(origin_ctx *ctx, const float *d_raw, const float *d_var,
                        const uint8_t *d_mask, int Nz, int Ny, int Nx, int order, int approx,
                        double *d_coef, double *d_zsum, double *d_zcnt) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_zsum && d_zcnt, "null pointer");
  if (approx) {  // no moments pass to fold the sums into: the two calls, one after the other
    int rc = dct_fit_impl(ctx, d_raw, d_var, d_mask, Nz, Ny, Nx, order, approx, d_coef, nullptr,
                          nullptr);
    if (rc) return rc;
    return origin_dct_resid_sums(ctx, d_raw, d_mask, d_coef, Nz, Ny, Nx, order, d_zsum, d_zcnt);
  }
  return dct_fit_impl(ctx, d_raw, d_var, d_mask, Nz, Ny, Nx, order, 0, d_coef, d_zsum, d_zcnt);
}

int origin_dct_continuum(origin_ctx *ctx, const double *d_coef, int Nz, int Ny, int Nx,
                         int order, float *d_cont) {
  ORIGIN_USE(ctx);
  int rc = check_dims(Nz, Ny, Nx, order);
  if (rc) return rc;
  ORIGIN_CHECK_ARG(d_coef && d_cont, "null pointer");
  const long S = (long)Ny * Nx;
  CtabGuard tab(ctx);
  rc = make_ctab(ctx, Nz, order, &tab.p);
  if (rc) return rc;
  const int nzc = pick_zchunks(ctx, S, Nz);
  const int zchunk = cdiv(Nz, nzc);
  dim3 grid(cdiv(S, 256), cdiv(Nz, zchunk));
  ProfScope ps(ctx, K_DCT_CONTINUUM);
#define CALL(O)                                                                                \
  hipLaunchKernelGGL(dct_continuum_kernel<O>, grid, dim3(256), 0, ctx->stream, d_coef, tab.p, \
                     Nz, S, zchunk, d_cont)
  DISPATCH_ORDER(order, CALL)
#undef CALL
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

int origin_dct_resid_sums(origin_ctx *ctx, const float *d_raw, const uint8_t *d_mask,
                          const double *d_coef, int Nz, int Ny, int Nx, int order,
                          double *d_zsum, double *d_zcnt) {
  ORIGIN_USE(ctx);
  int rc = check_dims(Nz, Ny, Nx, order);
  if (rc) return rc;
  ORIGIN_CHECK_ARG(d_raw && d_mask && d_coef && d_zsum && d_zcnt, "null pointer");
  const long S = (long)Ny * Nx;
  CtabGuard tab(ctx);
  rc = make_ctab(ctx, Nz, order, &tab.p);
  if (rc) return rc;
  const int spb = 8192;
  const int nchunk = cdiv(S, spb);
  const bool vec = (S & 3) == 0;
  void *scr = nullptr;
  const size_t part_bytes = (size_t)Nz * nchunk * 2 * sizeof(double);
  rc = origin_scratch(ctx, part_bytes + 64 * sizeof(double), &scr);
  if (rc) return rc;
  double *part = (double *)scr;
  double *ctot = (double *)((char *)scr + part_bytes);
  ProfScope ps(ctx, K_DCT_SUMS);
  hipLaunchKernelGGL(coef_total_kernel, dim3(order + 1), dim3(1024), 0, ctx->stream, d_coef, S,
                     ctot);
  dim3 grid(nchunk, Nz);
#define CALL(O)                                                                                \
  if (vec)                                                                                     \
    hipLaunchKernelGGL((dct_plane_sums_kernel<O, true>), grid, dim3(256), 0, ctx->stream, d_raw, \
                       d_mask, d_coef, tab.p, S, spb, part);                                   \
  else                                                                                         \
    hipLaunchKernelGGL((dct_plane_sums_kernel<O, false>), grid, dim3(256), 0, ctx->stream,     \
                       d_raw, d_mask, d_coef, tab.p, S, spb, part);                            \
  hipLaunchKernelGGL(dct_zsum_final_kernel<O>, dim3(cdiv(Nz, 256)), dim3(256), 0, ctx->stream, \
                     part, ctot, tab.p, Nz, nchunk, d_zsum, d_zcnt)
  DISPATCH_ORDER(order, CALL)
#undef CALL
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

int origin_dct_standardize(origin_ctx *ctx, const float *d_raw, const float *d_var,
                           const uint8_t *d_mask, const double *d_coef,
                           const double *d_zsum, const double *d_zcnt, int Nz, int Ny,
                           int Nx, int order, float *d_cube_std, float *d_cont_dct,
                           float *d_ima_std, float *d_ima_dct, double *d_o2) {
  ORIGIN_USE(ctx);
  int rc = check_dims(Nz, Ny, Nx, order);
  if (rc) return rc;
  ORIGIN_CHECK_ARG(d_raw && d_var && d_mask && d_coef && d_zsum && d_zcnt && d_cube_std,
                   "null pointer");
  const long S = (long)Ny * Nx;
  CtabGuard tab(ctx);
  rc = make_ctab(ctx, Nz, order, &tab.p);
  if (rc) return rc;
  const int nzc0 = pick_zchunks(ctx, S, Nz);
  const int npair = std::max(1, Nz / 2);  // the kernel's chunks are of mirror PAIRS of channels
  const int zchunk = cdiv(npair, nzc0);
  const int nzc = cdiv(npair, zchunk);
  const bool want = d_ima_std || d_ima_dct || d_o2;
  double *part = nullptr, *zmean = nullptr;
  {
    void *scr = nullptr;
    const size_t pbytes = want ? (size_t)nzc * 3 * S * sizeof(double) : 0;
    rc = origin_scratch(ctx, pbytes + (size_t)Nz * sizeof(double), &scr);
    if (rc) return rc;
    if (want) part = (double *)scr;
    zmean = (double *)((char *)scr + pbytes);
  }
  dim3 grid(cdiv(S, 256), nzc);
  ProfScope ps(ctx, K_DCT_STANDARDIZE);
  hipLaunchKernelGGL(zmean_kernel, dim3(cdiv(Nz, 256)), dim3(256), 0, ctx->stream, d_zsum, d_zcnt,
                     Nz, zmean);
#define CALL(O)                                                                               \
  hipLaunchKernelGGL(dct_standardize_kernel<O>, grid, dim3(256), 0, ctx->stream, d_raw, d_var, \
                     d_mask, d_coef, tab.p, zmean, Nz, S, zchunk, d_cube_std, d_cont_dct, part)
  DISPATCH_ORDER(order, CALL)
#undef CALL
  ORIGIN_LAUNCH_CHECK();
  if (want) {
    hipLaunchKernelGGL(std_images_final_kernel, dim3(cdiv(S, 256)), dim3(256), 0, ctx->stream,
                       part, nzc, S, Nz, d_ima_std, d_ima_dct, d_o2);
    ORIGIN_LAUNCH_CHECK();
  }
  return ORIGIN_OK;
}

static int dct_cont_std_on(origin_ctx *ctx, bool aux, const float *d_var, const double *d_coef,
                           int Nz, int Ny, int Nx, int order, float *d_cont_dct,
                           float *d_ima_dct) {
  ORIGIN_USE(ctx);
  int rc = check_dims(Nz, Ny, Nx, order);
  if (rc) return rc;
  ORIGIN_CHECK_ARG(d_var && d_coef && d_cont_dct, "null pointer");
  const long S = (long)Ny * Nx;
  CtabGuard tab(ctx);
  rc = make_ctab(ctx, Nz, order, &tab.p);
  if (rc) return rc;
  const int nzc0 = pick_zchunks(ctx, S, Nz);
  const int npair = std::max(1, Nz / 2);  // (chunks of mirror pairs of channels)
  const int zchunk = cdiv(npair, nzc0);
  const int nzc = cdiv(npair, zchunk);
  double *part = nullptr;
  if (d_ima_dct) {
    void *scr = nullptr;  // (the aux stream has a scratch of its own: the PCA uses the main one)
    rc = aux ? origin_aux_scratch(ctx, (size_t)nzc * S * sizeof(double), &scr)
             : origin_scratch(ctx, (size_t)nzc * S * sizeof(double), &scr);
    if (rc) return rc;
    part = (double *)scr;
  }
  dim3 grid(cdiv(S, 256), nzc);
  if (aux && (rc = origin_aux_begin(ctx))) return rc;
  hipStream_t st = aux ? ctx->aux_stream : ctx->stream;
  {
    ProfScope ps(ctx, K_DCT_CONTINUUM, aux ? 3 : 1);  // (events of the main stream: sync form only)
#define CALL(O)                                                                              \
  hipLaunchKernelGGL(dct_cont_std_kernel<O>, grid, dim3(256), 0, st, d_var, d_coef, tab.p, Nz, S, \
                     zchunk, d_cont_dct, part)
    DISPATCH_ORDER(order, CALL)
#undef CALL
    ORIGIN_LAUNCH_CHECK();
    if (d_ima_dct) {
      hipLaunchKernelGGL(cont_image_final_kernel, dim3(cdiv(S, 256)), dim3(256), 0, st, part, nzc,
                         S, Nz, d_ima_dct);
      ORIGIN_LAUNCH_CHECK();
    }
  }
  if (aux && (rc = origin_aux_end(ctx))) return rc;
  return ORIGIN_OK;
}

int origin_dct_cont_std(origin_ctx *ctx, const float *d_var, const double *d_coef, int Nz, int Ny,
                        int Nx, int order, float *d_cont_dct, float *d_ima_dct) {
  return dct_cont_std_on(ctx, false, d_var, d_coef, Nz, Ny, Nx, order, d_cont_dct, d_ima_dct);
}

int origin_dct_cont_std_async(origin_ctx *ctx, const float *d_var, const double *d_coef, int Nz,
                              int Ny, int Nx, int order, float *d_cont_dct, float *d_ima_dct) {
  return dct_cont_std_on(ctx, true, d_var, d_coef, Nz, Ny, Nx, order, d_cont_dct, d_ima_dct);
}

int origin_o2(origin_ctx *ctx, const float *d_cube, int Nz, long S, double *d_out) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(Nz > 0 && S > 0 && d_cube && d_out, "bad arguments");
  const int nzc0 = pick_zchunks(ctx, S, Nz);
  const int zchunk = cdiv(Nz, nzc0);
  const int nzc = cdiv(Nz, zchunk);
  void *scr = nullptr;
  int rc = origin_scratch(ctx, (size_t)nzc * S * sizeof(double), &scr);
  if (rc) return rc;
  ProfScope ps(ctx, K_O2);
  hipLaunchKernelGGL(o2_partial_kernel, dim3(cdiv(S, 256), nzc), dim3(256), 0, ctx->stream,
                     d_cube, Nz, S, zchunk, (double *)scr);
  hipLaunchKernelGGL(o2_final_kernel, dim3(cdiv(S, 256)), dim3(256), 0, ctx->stream,
                     (const double *)scr, nzc, S, Nz, d_out);
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

}  // extern "C"
