// Greedy PCA  (SURVEY.md 2.2 rows k5-k7): the whole loop of Compute_GreedyPCA
// (reference muse_origin/lib_origin.py:848-954) for all areas of a cube that stays in
// place in HBM as (Nz, S), S = Ny*Nx.
//
// Per iteration and per area (lib :899-949):
//   nuisance set   pypx = {test > thr}; mapO2[pypx] += 1; itermax guard        (:889,:901-905)
//   background     the nb = 1 + int(n_bg / Noise_population) lowest-O2 spectra  (:908-917)
//   b   = mean of the background spectra                                       (:917)
//   Xp  = X_nuis - b (b^T X_nuis)            [un-normalised projection]        (:920-923)
//   u   = leading left singular vector of Xp (ARPACK svds(k=1, tol=0))         (:940)
//   F  -= u (u^T F) over the whole area; test = mean_z F^2                     (:943-946)
//
// Coefficient form.  The cube is NOT rewritten every iteration.  With X the input (cube_std)
// the deflated cube after t iterations is  F_t = X - sum_{k<t} u_k c_k^T,  c_k = F_{k-1}^T u_k,
// so the library keeps the vectors U (per area) and the coefficient rows C (per spaxel) and
// evaluates columns of F_t on the fly where an iteration needs them (background mean, nuisance
// block).  One iteration then reads the area once (c_t = X^T u_t - C^T (U^T u_t)) instead of
// reading it twice and writing it once, and the O2 test follows from Pythagoras:
// |F_t|^2 = |F_{t-1}|^2 - c_t^2 exactly (u_t has unit norm).  The cube itself is produced by
// one final pass F = X - U C (single float32 rounding instead of one per iteration).
//
// All areas advance in lock step (they are independent, lib :806-819), every kernel is
// batched over the areas still iterating, and the control flow (which spaxels are
// nuisances, which background spectra feed the mean, when an area stops) runs on the device:
// per iteration the host reads back two ints per area to size the launches.
//
// u comes from the Gram matrix G = Xp^T Xp (float64 MFMA, the only matrix-shaped
// contraction of the path): its leading eigenvector v (device Lanczos with full
// re-orthogonalisation, restarted until the Ritz residual is at rounding level) gives
// u = Xp v / |Xp v|.  SURVEY.md section 7 (hard part 1): the loop is threshold driven, so the
// eigen-solve must be converged; everything that feeds it is float64, only the cube is f32.
//
// A block never straddles two areas, so per-area vectors (b, u) are wave-uniform.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>

#include "common.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

// Sum of a double over the 64 lanes of a wave, returned in every lane.  Within each row of
// 16 lanes the exchange uses DPP moves (quad_perm / row_ror: a few cycles each) instead of
// ds_bpermute; the four row sums are then combined through v_readlane.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_d(double v) {
  v += dpp_mov_d<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov_d<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov_d<0x124>(v);  // row_ror:4
  v += dpp_mov_d<0x128>(v);  // row_ror:8  -> every lane holds the sum of its row of 16
  const int lo = __double2loint(v), hi = __double2hiint(v);
  double t = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    t += __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * r),
                          __builtin_amdgcn_readlane(lo, 16 * r));
  return t;
}

// descriptor fields of the per-iteration work list (int64 [DF_COUNT][nw])
enum {
  DF_AREA = 0, DF_LIST0, DF_N, DF_NB, DF_LD, DF_XP, DF_C, DF_G, DF_Q, DF_NS, DF_CBASE,
  DF_T,    // vectors already removed from this area (rows of C / columns of U in use)
  DF_CN,   // offset of the gathered coefficient block Cn[T][ld] of this area
  DF_FBOLD,  // offset of this area's nuisance block of the previous iteration (-1: none)
  DF_LDOLD,  // its row stride
  DF_NOLD,   // its number of columns
  DF_SVALID,  // 1: the running column sum of the area's background set is usable (bmean_kernel)
  DF_COUNT
};
constexpr int PCA_CAP = 64;  // vectors kept per area before the cube is flushed (F = X - U C)
#define DSC(f, k) (D[(long)(f) * nw + (k)])
// The host builds an iteration's work list from the nuisance counts of the PREVIOUS selection
// (upper bounds: the counts only shrink) and enqueues the whole chain behind the selection that
// then writes the real counts into DF_N / DF_NB on the device (select_publish).  An area whose
// selection came out with fewer than two nuisance spaxels has finished (lib :899 / :927): every
// kernel of the chain leaves its slot alone.
#define PCA_SLOT_DONE(k) (DSC(DF_N, k) < 2)

// ------------------------------------------------------------------------------------
// selection: nuisance list, background list, iteration bookkeeping.  One block per area.
// ------------------------------------------------------------------------------------
struct BlockScan {
  int *wtot;  // LDS [16]
  __device__ __forceinline__ int exclusive(bool f, int &total) {
    const unsigned long long b = __ballot(f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rank = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();  // previous users of wtot are done
    if (lane == 0) wtot[wave] = __popcll(b);
    __syncthreads();
    int pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int c = wtot[w];
      pre += (w < wave) ? c : 0;
      tot += c;
    }
    total = tot;
    return pre + rank;
  }
};

__device__ __forceinline__ void pca_select_body(
    const int *__restrict__ spx, const long *__restrict__ spx_off, const double *__restrict__ test,
    const double *__restrict__ thr_, double noise_pop, int itermax, int *__restrict__ active,
    int *__restrict__ nbiter, int *__restrict__ nstop, int *__restrict__ mapO2,
    int *__restrict__ nuis, int *__restrict__ bg, int *__restrict__ nuis_pos,
    int *__restrict__ bg_pos, int *__restrict__ n_out, int *__restrict__ nb_out, int lds_cap) {
  __shared__ int wtot[16];
  __shared__ int hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_remaining, s_ncand;
  extern __shared__ double tcache[];  // the area's O2 values (when they fit: lds_cap > 0)
  const int a = blockIdx.x;
  const int tid = threadIdx.x;
  const long o0 = spx_off[a];
  const int ns = (int)(spx_off[a + 1] - o0);
  if (!active[a]) {
    if (tid == 0) n_out[a] = 0, nb_out[a] = 0;
    return;
  }
  const double thr = thr_[a];
  BlockScan scan{wtot};
  // the selection makes ~11 passes over the area's O2 values: keep them (and the spaxel list)
  // in LDS.  Filling is two batches of independent loads (index, then value): every dependent
  // global read costs an L2 round trip of ~0.7 us.
  const bool cached = ns <= lds_cap;
  int *scache = reinterpret_cast<int *>(tcache + lds_cap);
  if (cached) {
    for (int c0 = 0; c0 < ns; c0 += 8 * 1024) {
      int sp[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = c0 + tid + 1024 * e;
        sp[e] = i < ns ? spx[o0 + i] : -1;
      }
      double tv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) tv[e] = sp[e] >= 0 ? test[sp[e]] : 0.0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = c0 + tid + 1024 * e;
        if (i < ns) tcache[i] = tv[e], scache[i] = sp[e];
      }
    }
    __syncthreads();
  }
  auto tval = [&](int i) -> double { return cached ? tcache[i] : test[spx[o0 + i]]; };
  auto sval = [&](long i) -> int { return cached ? scache[i] : spx[o0 + i]; };

  // ---- pass 1: nuisance compaction in index order (np.where(test > thr)), candidates count
  int n = 0, ncand = 0;
  for (int c0 = 0; c0 < ns; c0 += 1024) {
    const int i = c0 + tid;
    const bool valid = i < ns;
    const double t = valid ? tval(i) : 0.0;
    const bool isn = valid && (t > thr);
    const int sp = isn ? sval(i) : 0;
    const bool cand = valid && (t > 0.0) && (t <= thr);
    int tot;
    const int r = scan.exclusive(isn, tot);
    if (isn) {
      nuis[o0 + n + r] = sp;
      nuis_pos[o0 + n + r] = (int)(o0 + i);
      mapO2[sp] += 1;  // mapO2[pypx] += 1                                       (lib :901)
    }
    n += tot;
    int tc;
    scan.exclusive(cand, tc);
    ncand += tc;
  }
  if (n == 0) {  // while len(pypx) > 0                                           (lib :899)
    if (tid == 0) active[a] = 0, n_out[a] = 0, nb_out[a] = 0;
    return;
  }
  __syncthreads();
  if (tid == 0) {
    const int it = nbiter[a] + 1;  // nbiter += 1                                 (lib :900)
    nbiter[a] = it;
    s_ncand = it;
  }
  __syncthreads();
  if (s_ncand > itermax) {  // if nbiter > itermax: nstop += 1; break          (lib :902-905)
    if (tid == 0) {
      atomicAdd(nstop, 1);
      active[a] = 0;
      n_out[a] = 0;
      nb_out[a] = 0;
    }
    return;
  }
  // nb = 1 + int(len(nind) / Noise_population), clipped by the slice [:nb]     (lib :914-917)
  int nb = 1 + (int)floor((double)ncand / noise_pop);
  if (nb > ncand) nb = ncand;

  if (nb > 0) {
    // ---- radix select of the nb-th smallest candidate (keys: bits of positive doubles)
    if (tid == 0) s_prefix = 0ull, s_remaining = nb - 1;
    unsigned long long maskbits = 0ull;
    for (int pass = 7; pass >= 0; --pass) {
      const int shift = pass * 8;
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      const unsigned long long prefix = s_prefix;
      for (int i = tid; i < ns; i += 1024) {
        const double t = tval(i);
        if ((t > 0.0) && (t <= thr)) {
          const unsigned long long key = (unsigned long long)__double_as_longlong(t);
          if ((key & maskbits) == prefix) atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
        }
      }
      __syncthreads();
      if (tid < 64) {  // wave 0: bucket holding rank `remaining` via a 64-lane prefix scan
        const int rem0 = s_remaining;
        const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2],
                  h3 = hist[4 * tid + 3];
        const int mine = h0 + h1 + h2 + h3;
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int v = __shfl_up(incl, off, 64);
          if ((int)tid >= off) incl += v;
        }
        const int excl = incl - mine;
        if (rem0 >= excl && rem0 < incl) {  // exactly one lane
          int rem = rem0 - excl, b = 4 * tid;
          if (rem >= h0) {
            rem -= h0, ++b;
            if (rem >= h1) {
              rem -= h1, ++b;
              if (rem >= h2) rem -= h2, ++b;
            }
          }
          s_remaining = rem;
          s_prefix = prefix | ((unsigned long long)b << shift);
        }
      }
      maskbits |= 255ull << shift;
      __syncthreads();
    }
    const unsigned long long tau = s_prefix;
    const int need_equal = s_remaining + 1;
    // ---- emit the background columns.  The reference indexes the *filtered* vector
    // test[test > 0] and uses those indices on the unfiltered columns (lib :908-917): the
    // column of a selected element is its rank among the elements with test > 0.
    int nfilt = 0, neq = 0, nemit = 0;
    for (int c0 = 0; c0 < ns; c0 += 1024) {
      const int i = c0 + tid;
      const bool valid = i < ns;
      const double t = valid ? tval(i) : 0.0;
      const bool pos = valid && (t > 0.0);
      const bool cand = pos && (t <= thr);
      const unsigned long long key = (unsigned long long)__double_as_longlong(t);
      int tf, te, tm;
      const int rf = scan.exclusive(pos, tf);
      const bool eq = cand && key == tau;
      const int re = scan.exclusive(eq, te);
      const bool emit = cand && (key < tau || (eq && (neq + re) < need_equal));
      const int rm = scan.exclusive(emit, tm);
      if (emit) {
        bg[o0 + nemit + rm] = sval(nfilt + rf);
        bg_pos[o0 + nemit + rm] = (int)(o0 + nfilt + rf);
      }
      nfilt += tf;
      neq += te;
      nemit += tm;
    }
  }
  if (tid == 0) {
    nb_out[a] = nb;
    if (n == 1) {  // if x_red.shape[1] == 1: break                              (lib :927-928)
      active[a] = 0;
      n_out[a] = 0;
    } else {
      n_out[a] = n;
    }
  }
}

// Same selection for areas of at most 1024 * SEL_EPT spaxels (every 100 x 100 area), built for
// the latency of a one-block kernel: thread t owns the SEL_EPT consecutive list entries
// [t*E, (t+1)*E) in registers, so index order is (thread, local) order and each compaction is ONE
// block scan of per-thread counts instead of one scan per 1024-entry chunk (3 scans + a few
// radix passes instead of ~50 scans).  Results are identical to pca_select_kernel.
constexpr int SEL_EPT = 12;

__device__ __forceinline__ unsigned long long block_scan_u64(unsigned long long v,
                                                             unsigned long long *wsum,
                                                             unsigned long long &total) {
  // inclusive scan inside the wave, exclusive across waves; returns the exclusive prefix
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  __syncthreads();  // previous users of wsum are done
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  unsigned long long pre = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const unsigned long long c = wsum[w];
    pre += (w < wave) ? c : 0ull;
    tot += c;
  }
  total = tot;
  return pre + incl - v;
}

__device__ __forceinline__ void pca_select_fast_body(
    const int *__restrict__ spx, const long *__restrict__ spx_off, const double *__restrict__ test,
    const double *__restrict__ thr_, double noise_pop, int itermax, int *__restrict__ active,
    int *__restrict__ nbiter, int *__restrict__ nstop, int *__restrict__ mapO2,
    int *__restrict__ nuis, int *__restrict__ bg, int *__restrict__ nuis_pos,
    int *__restrict__ bg_pos, int *__restrict__ n_out, int *__restrict__ nb_out,
    uint8_t *__restrict__ inB, int *__restrict__ dlist, int *__restrict__ ndiff) {
  __shared__ unsigned long long wsum[16], wmax[16];
  __shared__ unsigned bmap[1024 * SEL_EPT / 32];  // list positions emitted as background
  __shared__ int hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_remaining, s_it, s_done;
  extern __shared__ int scache[];  // spaxel index of every list entry (for the filtered-index quirk)
  const int a = blockIdx.x;
  const int tid = threadIdx.x;
  const long o0 = spx_off[a];
  const int ns = (int)(spx_off[a + 1] - o0);
  if (!active[a]) {
    if (tid == 0) n_out[a] = 0, nb_out[a] = 0;
    return;
  }
  const double thr = thr_[a];
  const int E = (ns + 1023) / 1024;  // entries per thread (<= SEL_EPT)
  const int i0 = tid * E;
  // two batches of independent loads: indices, then values
  int sp[SEL_EPT];
  double tv[SEL_EPT];
#pragma unroll
  for (int e = 0; e < SEL_EPT; ++e) sp[e] = (e < E && i0 + e < ns) ? spx[o0 + i0 + e] : -1;
#pragma unroll
  for (int e = 0; e < SEL_EPT; ++e) tv[e] = sp[e] >= 0 ? test[sp[e]] : 0.0;
#pragma unroll
  for (int e = 0; e < SEL_EPT; ++e)
    if (sp[e] >= 0) scache[i0 + e] = sp[e];
  // membership of the previous iteration's background set (see the end of this function)
  unsigned oldmask = 0;
  if (inB) {
    uint8_t ob[SEL_EPT];
#pragma unroll
    for (int e = 0; e < SEL_EPT; ++e) ob[e] = sp[e] >= 0 ? inB[o0 + i0 + e] : (uint8_t)0;
#pragma unroll
    for (int e = 0; e < SEL_EPT; ++e) oldmask |= (unsigned)(ob[e] != 0) << e;
    for (int i = tid; i < 1024 * SEL_EPT / 32; i += 1024) bmap[i] = 0u;
  }

  // ---- counts: nuisance (t > thr), candidates (0 < t <= thr), positive (t > 0)
  unsigned long long cnt = 0;  // [nuis | cand | pos] x 20 bits
#pragma unroll
  for (int e = 0; e < SEL_EPT; ++e) {
    const bool v = sp[e] >= 0;
    const bool isn = v && tv[e] > thr, pos = v && tv[e] > 0.0, cand = pos && tv[e] <= thr;
    cnt += ((unsigned long long)isn << 40) + ((unsigned long long)cand << 20) + (unsigned long long)pos;
  }
  unsigned long long tot;
  const unsigned long long pre = block_scan_u64(cnt, wsum, tot);
  const int n = (int)(tot >> 40), ncand = (int)((tot >> 20) & 0xfffff);
  if (n == 0) {  // while len(pypx) > 0                                           (lib :899)
    if (tid == 0) active[a] = 0, n_out[a] = 0, nb_out[a] = 0;
    return;
  }
  {  // nuisance compaction in index order (np.where(test > thr)); mapO2[pypx] += 1   (lib :901, before the itermax test)
    int rn = (int)(pre >> 40);
#pragma unroll
    for (int e = 0; e < SEL_EPT; ++e)
      if (sp[e] >= 0 && tv[e] > thr) {
        nuis[o0 + rn] = sp[e];
        nuis_pos[o0 + rn] = (int)(o0 + i0 + e);
        mapO2[sp[e]] += 1;
        ++rn;
      }
  }
  if (tid == 0) {
    const int it = nbiter[a] + 1;  // nbiter += 1                                 (lib :900)
    nbiter[a] = it;
    s_it = it;
  }
  __syncthreads();
  if (s_it > itermax) {  // if nbiter > itermax: nstop += 1; break          (lib :902-905)
    if (tid == 0) {
      atomicAdd(nstop, 1);
      active[a] = 0;
      n_out[a] = 0;
      nb_out[a] = 0;
    }
    return;
  }
  // nb = 1 + int(len(nind) / Noise_population), clipped by the slice [:nb]     (lib :914-917)
  int nb = 1 + (int)floor((double)ncand / noise_pop);
  if (nb > ncand) nb = ncand;

  if (nb > 0) {
    // ---- radix select of the nb-th smallest candidate (keys: bits of positive doubles).
    // The O2 values of an area sit in a narrow range around 1, so the leading bytes of all
    // keys coincide and a histogram over them is one LDS counter hit by every thread.  The
    // digits are therefore taken from key - min(key), most significant bit of the RANGE first
    // (8 bits per pass), and the passes stop as soon as the bucket holding the wanted rank has
    // a single element -- typically after two or three passes instead of eight.
    unsigned long long kmin = ~0ull, kmax = 0ull;
#pragma unroll
    for (int e = 0; e < SEL_EPT; ++e)
      if (sp[e] >= 0 && tv[e] > 0.0 && tv[e] <= thr) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(tv[e]);
        kmin = key < kmin ? key : kmin;
        kmax = key > kmax ? key : kmax;
      }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long o1 = __shfl_xor(kmin, off, 64), o2 = __shfl_xor(kmax, off, 64);
      kmin = o1 < kmin ? o1 : kmin;
      kmax = o2 > kmax ? o2 : kmax;
    }
    __syncthreads();  // previous users of wsum are done
    if ((tid & 63) == 0) wsum[tid >> 6] = kmin, wmax[tid >> 6] = kmax;
    if (tid == 0) s_prefix = 0ull, s_remaining = nb - 1, s_done = 0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      kmin = wsum[w] < kmin ? wsum[w] : kmin;
      kmax = wmax[w] > kmax ? wmax[w] : kmax;
    }
    const unsigned long long range = kmax - kmin;  // nb > 0: there is at least one candidate
    int hi_bit = range ? 64 - __clzll((long long)range) : 0;  // digits below this bit
    while (hi_bit > 0) {
      const int lo_bit = hi_bit > 8 ? hi_bit - 8 : 0;
      const unsigned long long above = hi_bit >= 64 ? 0ull : ~0ull << hi_bit;  // decided bits
      const unsigned dmask = (1u << (hi_bit - lo_bit)) - 1u;
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      const unsigned long long prefix = s_prefix;
#pragma unroll
      for (int e = 0; e < SEL_EPT; ++e)
        if (sp[e] >= 0 && tv[e] > 0.0 && tv[e] <= thr) {
          const unsigned long long key =
              (unsigned long long)__double_as_longlong(tv[e]) - kmin;
          if ((key & above) == prefix) atomicAdd(&hist[(int)((unsigned)(key >> lo_bit) & dmask)], 1);
        }
      __syncthreads();
      if (tid < 64) {  // wave 0: bucket holding rank `remaining` via a 64-lane prefix scan
        const int rem0 = s_remaining;
        const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2],
                  h3 = hist[4 * tid + 3];
        const int mine = h0 + h1 + h2 + h3;
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int v = __shfl_up(incl, off, 64);
          if ((int)tid >= off) incl += v;
        }
        const int excl = incl - mine;
        if (rem0 >= excl && rem0 < incl) {  // exactly one lane
          int rem = rem0 - excl, b = 4 * tid, hb = h0;
          if (rem >= h0) {
            rem -= h0, ++b, hb = h1;
            if (rem >= h1) {
              rem -= h1, ++b, hb = h2;
              if (rem >= h2) rem -= h2, ++b, hb = h3;
            }
          }
          s_remaining = rem;
          s_prefix = prefix | ((unsigned long long)b << lo_bit);
          s_done = hb == 1;  // the bucket's only element is the answer (then rem == 0)
        }
      }
      __syncthreads();
      hi_bit = lo_bit;
      if (s_done) break;
    }
    if (hi_bit > 0) {  // stopped early: the one candidate matching the decided bits
      const unsigned long long above = ~0ull << hi_bit, prefix = s_prefix;
      __syncthreads();  // everyone has read s_prefix
#pragma unroll
      for (int e = 0; e < SEL_EPT; ++e)
        if (sp[e] >= 0 && tv[e] > 0.0 && tv[e] <= thr) {
          const unsigned long long key =
              (unsigned long long)__double_as_longlong(tv[e]) - kmin;
          if ((key & above) == prefix) s_prefix = key;
        }
      __syncthreads();
    }
    const unsigned long long tau = s_prefix + kmin;
    const int need_equal = s_remaining + 1;
    // ---- emit the background columns.  The reference indexes the *filtered* vector
    // test[test > 0] and uses those indices on the unfiltered columns (lib :908-917): the
    // column of a selected element is its rank among the elements with test > 0.
    unsigned long long ce = 0;  // [eq | below] x 20 bits
#pragma unroll
    for (int e = 0; e < SEL_EPT; ++e)
      if (sp[e] >= 0 && tv[e] > 0.0 && tv[e] <= thr) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(tv[e]);
        ce += ((unsigned long long)(key == tau) << 20) + (unsigned long long)(key < tau);
      }
    unsigned long long tot2;
    const unsigned long long pre2 = block_scan_u64(ce, wsum, tot2);
    // emitted before this thread: all smaller keys before it + the first equal ones (index order)
    int eq_before = (int)(pre2 >> 20);
    int emit_before = (int)(pre2 & 0xfffff) + min(eq_before, need_equal);
    int rp = (int)(pre & 0xfffff);  // rank among the positive elements
#pragma unroll
    for (int e = 0; e < SEL_EPT; ++e) {
      if (sp[e] < 0 || !(tv[e] > 0.0)) continue;
      if (tv[e] <= thr) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(tv[e]);
        const bool eq = key == tau;
        const bool emit = key < tau || (eq && eq_before < need_equal);
        if (emit) {
          bg[o0 + emit_before] = scache[rp];
          bg_pos[o0 + emit_before] = (int)(o0 + rp);
          if (inB) atomicOr(&bmap[rp >> 5], 1u << (rp & 31));
          ++emit_before;
        }
        eq_before += eq;
      }
      ++rp;
    }
  }
  if (inB) {
    // The background set changes by a few columns per iteration (the O2 tests move slowly), so
    // its mean is updated from the columns that entered or left it instead of re-gathered
    // (bmean_kernel).  Here: the changes of membership in list order -- +(spaxel+1) entered,
    // -(spaxel+1) left -- and the new membership flags.
    __syncthreads();  // bmap complete
    unsigned newmask = 0;
#pragma unroll
    for (int e = 0; e < SEL_EPT; ++e)
      if (sp[e] >= 0) newmask |= ((bmap[(i0 + e) >> 5] >> ((i0 + e) & 31)) & 1u) << e;
    const unsigned diff = newmask ^ oldmask;
    unsigned long long tot3;
    int r = (int)block_scan_u64((unsigned long long)__popc(diff), wsum, tot3);
#pragma unroll
    for (int e = 0; e < SEL_EPT; ++e)
      if ((diff >> e) & 1u) {
        const bool in = (newmask >> e) & 1u;
        dlist[o0 + r++] = in ? sp[e] + 1 : -(sp[e] + 1);
        inB[o0 + i0 + e] = (uint8_t)in;
      }
    if (tid == 0) ndiff[a] = (int)tot3;
  }
  if (tid == 0) {
    nb_out[a] = nb;
    if (n == 1) {  // if x_red.shape[1] == 1: break                              (lib :927-928)
      active[a] = 0;
      n_out[a] = 0;
    } else {
      n_out[a] = n;
    }
  }
}

// The host sizes every launch of an iteration from n / nb of each area.  Instead of a D2H copy
// and a stream synchronisation (~30 us of wake-up latency per iteration), each block stores its
// two numbers straight into mapped, coherent host memory; the last block to finish (device
// counter) raises the generation flag the host spins on.
// patchD / kidx / nw: the work list of the iteration this selection opens, already on the device
// with counts from the previous selection; the real n / nb of the area go into its slot.
__device__ __forceinline__ void select_publish(int a, int na, const int *n_out, const int *nb_out,
                                               int *host_out, unsigned *counter, int gen,
                                               long *patchD, const int *kidx, int nw) {
  if (threadIdx.x != 0) return;
  if (patchD) {
    const int k = kidx[a];
    if (k >= 0) {
      patchD[(long)DF_N * nw + k] = n_out[a];
      patchD[(long)DF_NB * nw + k] = nb_out[a];
    }
  }
  if (!host_out) return;
  host_out[a] = n_out[a];
  host_out[na + a] = nb_out[a];
  __threadfence_system();
  const unsigned old = atomicAdd(counter, 1u);
  if (old + 1u == (unsigned)na * (unsigned)gen) {
    __threadfence_system();
    __hip_atomic_store(host_out + 2 * na, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(1024) void pca_select_kernel(
    const int *__restrict__ spx, const long *__restrict__ spx_off, const double *__restrict__ test,
    const double *__restrict__ thr_, double noise_pop, int itermax, int *__restrict__ active,
    int *__restrict__ nbiter, int *__restrict__ nstop, int *__restrict__ mapO2,
    int *__restrict__ nuis, int *__restrict__ bg, int *__restrict__ nuis_pos,
    int *__restrict__ bg_pos, int *__restrict__ n_out, int *__restrict__ nb_out, int lds_cap,
    int *host_out, unsigned *counter, int gen, long *patchD, const int *kidx, int nw) {
  pca_select_body(spx, spx_off, test, thr_, noise_pop, itermax, active, nbiter, nstop, mapO2, nuis,
                  bg, nuis_pos, bg_pos, n_out, nb_out, lds_cap);
  select_publish(blockIdx.x, gridDim.x, n_out, nb_out, host_out, counter, gen, patchD, kidx, nw);
}

__global__ __launch_bounds__(1024) void pca_select_fast_kernel(
    const int *__restrict__ spx, const long *__restrict__ spx_off, const double *__restrict__ test,
    const double *__restrict__ thr_, double noise_pop, int itermax, int *__restrict__ active,
    int *__restrict__ nbiter, int *__restrict__ nstop, int *__restrict__ mapO2,
    int *__restrict__ nuis, int *__restrict__ bg, int *__restrict__ nuis_pos,
    int *__restrict__ bg_pos, int *__restrict__ n_out, int *__restrict__ nb_out, int *host_out,
    unsigned *counter, int gen, uint8_t *__restrict__ inB, int *__restrict__ dlist,
    int *__restrict__ ndiff, long *patchD, const int *kidx, int nw) {
  pca_select_fast_body(spx, spx_off, test, thr_, noise_pop, itermax, active, nbiter, nstop, mapO2,
                       nuis, bg, nuis_pos, bg_pos, n_out, nb_out, inB, dlist, ndiff);
  select_publish(blockIdx.x, gridDim.x, n_out, nb_out, host_out, counter, gen, patchD, kidx, nw);
}

// ------------------------------------------------------------------------------------
// cbar_k[q] = mean_{i in bg_k} C[q][bg_pos_i], q < T_k           grid (nw), block 1024
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void cbar_kernel(const double *__restrict__ C, long ntot,
                                                    const int *__restrict__ bg_pos,
                                                    const long *__restrict__ D, int nw,
                                                    double *__restrict__ cbar) {
  const int k = blockIdx.x;
  if (PCA_SLOT_DONE(k)) return;
  const int T = (int)DSC(DF_T, k);
  const int nb = (int)DSC(DF_NB, k);
  const long o0 = DSC(DF_LIST0, k);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // every read below is an L2 round trip (~0.7 us): fetch the positions once, then keep the
  // gathers of a row independent (first 8 positions per lane from registers)
  int pos[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) pos[e] = lane + 64 * e < nb ? bg_pos[o0 + lane + 64 * e] : -1;
  for (int q = wave; q < T; q += 16) {
    const double *row = C + (long)q * ntot;
    double v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = pos[e] >= 0 ? row[pos[e]] : 0.0;
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += v[e];  // same order as the plain loop
    for (int i = lane + 512; i < nb; i += 64) acc += row[bg_pos[o0 + i]];
    acc = wave_sum_d(acc);
    if (lane == 0) cbar[(long)k * PCA_CAP + q] = acc / (double)nb;
  }
}

// ------------------------------------------------------------------------------------
// b_k[z] = mean_{i in bg_k} F_t[z, bg_i] = mean_i X[z, bg_i] - sum_q U[z][q] cbar[q]
// grid (ceil(Nz/4), nw), block (64,4).  U layout: [area][z][PCA_CAP].
// A background column is a scattered gather (a 64-byte sector per sample), and the set moves by
// a few columns per iteration: with Ssum (float64 [area][Nz], the sum over the set as the
// previous iteration left it) and DF_SVALID the sum is updated from the columns that entered /
// left (dlist, signed, in list order: a fixed summation order) instead of re-gathered.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bmean_kernel(const float *__restrict__ X, int Nz, long S,
                                                    const int *__restrict__ bg,
                                                    const long *__restrict__ D, int nw,
                                                    const double *__restrict__ U,
                                                    const double *__restrict__ cbar,
                                                    double *__restrict__ b,
                                                    const int *__restrict__ dlist,
                                                    const int *__restrict__ ndiff,
                                                    double *__restrict__ Ssum) {
  const int k = blockIdx.y;
  const int z = blockIdx.x * 4 + threadIdx.y;
  if (z >= Nz || PCA_SLOT_DONE(k)) return;
  const long o0 = DSC(DF_LIST0, k);
  const int nb = (int)DSC(DF_NB, k), T = (int)DSC(DF_T, k);
  const int a = (int)DSC(DF_AREA, k);
  const float *row = X + (long)z * S;
  double part = 0.0;
  const bool delta = Ssum && DSC(DF_SVALID, k);
  if (delta) {
    const int nd = ndiff[a];
    for (int i = threadIdx.x; i < nd; i += 64) {
      const int e = dlist[o0 + i];
      const double v = (double)row[(e > 0 ? e : -e) - 1];
      part += e > 0 ? v : -v;
    }
  } else {
    for (int i = threadIdx.x; i < nb; i += 64) part += (double)row[bg[o0 + i]];
  }
  double tot = wave_sum_d(part);
  if (delta) tot += Ssum[(long)a * Nz + z];
  if (Ssum && threadIdx.x == 0) Ssum[(long)a * Nz + z] = tot;
  double acc = 0.0;
  if ((int)threadIdx.x < T)  // T <= PCA_CAP == 64 lanes
    acc = U[((long)a * Nz + z) * PCA_CAP + threadIdx.x] * cbar[(long)k * PCA_CAP + threadIdx.x];
  acc = wave_sum_d(acc);
  if (threadIdx.x == 0) b[(long)k * Nz + z] = tot / (double)nb - acc;
}

// ------------------------------------------------------------------------------------
// Nuisance block of F_t (float64, [Nz][ld]):  Fb[z][j] = F_t[z, col_j]
// and the partial c_j = b^T Fb_j of the block's z slice.
// The nuisance set only shrinks (the O2 tests only decrease), so the columns of iteration t are
// a subsequence of those of iteration t-1: when the previous block of the area is at hand
// (DF_FBOLD >= 0) column j is its old column (found by the list position, binary search) minus
// the one deflation since,  F_t = F_{t-1} - u_{t-1} c_{t-1}^T  -- no read of the cube (a
// scattered column costs a 64-byte sector per 4-byte sample).  Otherwise (first iteration of an
// area, or after a flush)  Fb[z][j] = X[z, col_j] - sum_q U[z][q] C[q][pos_j].
// grid (ceil(ldmax/64), nw, z slices), block (64 columns, 16 waves over z)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gather_xp_kernel(
    const float *__restrict__ X, int Nz, long S, const int *__restrict__ nuis,
    const int *__restrict__ nuis_pos, const int *__restrict__ nuis_pos_old,
    const long *__restrict__ D, int nw, const double *__restrict__ b,
    const double *__restrict__ U, const double *__restrict__ C, long ntot,
    const double *__restrict__ Fold, double *__restrict__ Fb, double *__restrict__ cpart,
    long ctot, int zper) {
  __shared__ double red[16][64];
  __shared__ double Cn[PCA_CAP][64];  // coefficients of this block's 64 columns
  const int k = blockIdx.y;
  const int ld = (int)DSC(DF_LD, k);
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (blockIdx.x * 64 >= ld || PCA_SLOT_DONE(k)) return;  // whole block out of range (uniform)
  const int n = (int)DSC(DF_N, k), T = (int)DSC(DF_T, k);
  const bool live = j < n;   // real nuisance column
  const bool inld = j < ld;  // padded column (stored as zeros)
  const long list0 = DSC(DF_LIST0, k);
  const long col = live ? (long)nuis[list0 + j] : 0;
  const long pos = live ? (long)nuis_pos[list0 + j] : 0;
  const double *bk = b + (long)k * Nz;
  const double *Ua = U + (long)DSC(DF_AREA, k) * Nz * PCA_CAP;
  double *Fk = Fb + DSC(DF_XP, k);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int z0 = blockIdx.z * zper, z1 = min(Nz, z0 + zper);
  const long fbold = DSC(DF_FBOLD, k);
  double acc = 0.0;
  if (fbold >= 0) {
    // ---- from the previous block: column with the same list position
    const int nold = (int)DSC(DF_NOLD, k), ldold = (int)DSC(DF_LDOLD, k);
    int io = 0;
    if (live) {
      int lo = 0, hi = nold - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((long)nuis_pos_old[list0 + mid] < pos) lo = mid + 1;
        else hi = mid;
      }
      io = lo;
    }
    const double cprev = live ? C[(long)(T - 1) * ntot + pos] : 0.0;
    const double *Fo = Fold + fbold;
    // four channels per trip with their loads issued together (each is an L2 round trip)
    for (int z = z0 + w; z < z1; z += 64) {
      double fo[4], uu[4], bb[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int zz = z + 16 * s;
        const long zc = zz < z1 ? zz : z0;
        fo[s] = live ? Fo[zc * ldold + io] : 0.0;
        uu[s] = Ua[zc * PCA_CAP + (T - 1)];
        bb[s] = bk[zc];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int zz = z + 16 * s;
        if (zz < z1) {
          const double v = live ? fma(-uu[s], cprev, fo[s]) : 0.0;
          if (inld) Fk[(long)zz * ld + j] = v;
          acc = fma(bb[s], v, acc);
        }
      }
    }
  } else {
    for (int q = w; q < T; q += 16) Cn[q][threadIdx.x] = live ? C[(long)q * ntot + pos] : 0.0;
    __syncthreads();
    for (int z = z0 + w; z < z1; z += 16) {
      double v = live ? (double)X[(long)z * S + col] : 0.0;
      const double *uz = Ua + (long)z * PCA_CAP;  // wave-uniform row of U -> scalar loads
      for (int q = 0; q < T; ++q) v = fma(-uz[q], Cn[q][threadIdx.x], v);
      if (inld) Fk[(long)z * ld + j] = v;
      acc = fma(bk[z], v, acc);
    }
  }
  red[threadIdx.y][threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.y == 0 && inld) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][threadIdx.x];
    cpart[(long)blockIdx.z * ctot + DSC(DF_C, k) + j] = t;
  }
}

// Xp[z][j] = Fb[z][j] - b[z] c[j]       grid (ceil(Nz/16), nw), block 256 (lanes over columns)
// (out of place: Fb stays as the next iteration's source)
__global__ __launch_bounds__(256) void project_xp_kernel(const double *__restrict__ b, int Nz,
                                                         const long *__restrict__ D, int nw,
                                                         const double *__restrict__ Fb,
                                                         double *__restrict__ Xp,
                                                         const double *__restrict__ cpart,
                                                         long ctot, int nzb) {
  const int k = blockIdx.y;
  if (PCA_SLOT_DONE(k)) return;
  const int ld = (int)DSC(DF_LD, k);
  const double *F = Fb + DSC(DF_XP, k);
  double *X = Xp + DSC(DF_XP, k);
  const double *c = cpart + DSC(DF_C, k);
  const double *bk = b + (long)k * Nz;
  const int z0 = blockIdx.x * 16, z1 = min(Nz, z0 + 16);
  for (int j = threadIdx.x; j < ld; j += 256) {
    // all loads of the column first (independent), then the arithmetic
    double cq[16], f[16], bz[16];  // nzb <= 16, 16 channels per block
#pragma unroll
    for (int q = 0; q < 16; ++q) cq[q] = q < nzb ? c[(long)q * ctot + j] : 0.0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int z = z0 + e < z1 ? z0 + e : z0;
      f[e] = F[(long)z * ld + j];
      bz[e] = bk[z];
    }
    double cj = 0.0;  // c_j = b^T F_j, summed over the z slices of the gather in fixed order
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (q < nzb) cj += cq[q];
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (z0 + e < z1) X[(long)(z0 + e) * ld + j] = fma(-bz[e], cj, f[e]);
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
                                                  const long *__restrict__ ld_,
                                                  const int *__restrict__ tile_i,
                                                  const int *__restrict__ tile_j,
                                                  const int *__restrict__ tile_a, int Nz,
                                                  int ksplit, double *__restrict__ slab,
                                                  const long *__restrict__ g_off,
                                                  long slab_stride,
                                                  const long *__restrict__ n_) {
  const int t = blockIdx.x;
  const int a = tile_a[t];
  if (n_ && n_[a] < 2) return;  // the area finished with this iteration's selection
  const int ld = (int)ld_[a];
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
  // eight k-steps (32 channels) per trip: the 32 loads are issued together -- a trip with one
  // k-step waits a whole L2 round trip (~0.7 us) for four loads
  for (int z = z0; z < z1; z += 32) {
    double a0[8], a1[8], b0[8], b1[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int zz = z + 4 * s + kq;
      const bool zin = zz < z1;
      const double *row = X + (long)(zin ? zz : z0) * ld;
      a0[s] = (zin && ia0) ? row[i0 + r16] : 0.0;
      a1[s] = (zin && ia1) ? row[i0 + 16 + r16] : 0.0;
      b0[s] = (zin && jb0) ? row[j0 + r16] : 0.0;
      b1[s] = (zin && jb1) ? row[j0 + 16 + r16] : 0.0;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b0[s], acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b1[s], acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b0[s], acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b1[s], acc11, 0, 0, 0);
    }
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
                                                          const long *__restrict__ ld_,
                                                          const int *__restrict__ tile_i,
                                                          const int *__restrict__ tile_j,
                                                          const int *__restrict__ tile_a,
                                                          double *__restrict__ G,
                                                          const long *__restrict__ g_off,
                                                          const long *__restrict__ n_) {
  const int t = blockIdx.x;
  const int a = tile_a[t];
  if (n_ && n_[a] < 2) return;
  const int ld = (int)ld_[a];
  const int i0 = tile_i[t] * 32, j0 = tile_j[t] * 32;
  double *Ga = G + g_off[a];
  for (int e = threadIdx.x; e < 1024; e += 256) {
    const int i = i0 + (e >> 5), j = j0 + (e & 31);
    if (i >= ld || j >= ld) continue;
    double acc = 0.0;
    for (int ks = 0; ks < ksplit; ++ks)
      acc += slab[(long)ks * slab_stride + g_off[a] + (long)i * ld + j];
    Ga[(long)i * ld + j] = acc;
    Ga[(long)j * ld + i] = acc;
  }
}

// ------------------------------------------------------------------------------------
// Leading eigenvector of the symmetric PSD matrix G (n x n, row stride ld): restarted
// Lanczos with full (twice-applied classical Gram-Schmidt) re-orthogonalisation.  One block
// of 1024 threads per matrix; the Krylov basis Q lives in global scratch (L2 resident).
// The small tridiagonal problem is solved by 64-way multisection on Sturm counts (wave 0)
// and inverse iteration with partial pivoting (thread 0).
// ------------------------------------------------------------------------------------
constexpr int LANCZOS_M = 48;

__device__ __forceinline__ double block_sum(double v, double *red) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < 16; ++w) t += red[w];
  return t;
}

// Number of eigenvalues < x of the symmetric tridiagonal with diagonal a[0..m) and squared
// off-diagonals bb[0..m-1), both already divided by the norm of T (so |a - x| <= 3, bb <= 1):
// sign changes of the leading principal minors p_i = (a_i - x) p_{i-1} - bb_{i-1} p_{i-2}.
// One multiply-add on the dependency chain per row (the pivot form q_i = a_i - x - bb/q_{i-1}
// carries a float64 division, ~15 dependent instructions).  A minor that is exactly zero takes
// the sign opposite to its predecessor -- the same convention as replacing a zero pivot by a
// tiny negative one.  With the scaling the minors cannot overflow for m <= 48; they are
// rescaled every 8 rows against underflow.
// The rows are taken eight at a time with their coefficients loaded up front (the loads do not
// depend on the chain); the arrays are padded by 8 and rows >= m are not counted.
constexpr int TRI_PAD = LANCZOS_M + 8;
__device__ __forceinline__ int sturm_count(const double *a, const double *bb, int m, double x) {
  double p0 = 1.0, p1 = a[0] - x;
  bool s1 = p1 < 0.0 || p1 == 0.0;  // sign of p_{i-1} (true: negative), p_{-1} = 1
  int cnt = s1;
  for (int i0 = 0; i0 < m - 1; i0 += 8) {
    double av[8], bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) av[e] = a[i0 + 1 + e] - x, bv[e] = bb[i0 + e];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double p2 = fma(av[e], p1, -bv[e] * p0);
      const bool s2 = p2 < 0.0 || (p2 == 0.0 && !s1);
      cnt += (i0 + 1 + e < m) && (s2 != s1);
      p0 = p1, p1 = p2, s1 = s2;
    }
    if (fabs(p1) < 1e-100 && fabs(p0) < 1e-100) p0 *= 1e100, p1 *= 1e100;
  }
  return cnt;
}

// Largest eigenpair of the symmetric tridiagonal T (alpha[0..m), beta[0..m-1)), executed by
// ONE wave: 64-way multisection on Sturm counts for the eigenvalue, then two steps of inverse
// iteration (lane 0) for the unit eigenvector, left in ws.x.  Returns the eigenvalue.
// The shift of the inverse iteration lies just above the largest eigenvalue, so T - sigma I is
// negative definite and its LDL^T factorisation needs no pivoting; it is formed once, with the
// pivot reciprocals kept for both solves.
template <int PAD_>
struct TriWorkT {
  static constexpr int PAD = PAD_;
  double x[PAD_], a[PAD_], bb[PAD_], l[PAD_], rd[PAD_];
};
typedef TriWorkT<TRI_PAD> TriWork;

__device__ __forceinline__ double tridiag_top(const double *alpha, const double *beta, int m,
                                              int lane, TriWork &ws) {
  double lo = 1e300, hi = -1e300, tn = 0.0;
  for (int i = 0; i < m; ++i) {
    const double r = (i > 0 ? fabs(beta[i - 1]) : 0.0) + (i < m - 1 ? fabs(beta[i]) : 0.0);
    lo = fmin(lo, alpha[i] - r);
    hi = fmax(hi, alpha[i] + r);
    tn = fmax(tn, fabs(alpha[i]) + r);
  }
  if (!(tn > 0.0)) {  // T == 0
    if (lane == 0)
      for (int i = 0; i < m; ++i) ws.x[i] = i == 0 ? 1.0 : 0.0;
    return 0.0;
  }
  // scaled copy: a = alpha / tn, bb = (beta / tn)^2
  const double itn = 1.0 / tn;
  for (int i = lane; i < m; i += 64) {
    ws.a[i] = alpha[i] * itn;
    const double b = i < m - 1 ? beta[i] * itn : 0.0;
    ws.bb[i] = b * b;
  }
  lo *= itn;
  hi = hi * itn + 1e-14;
  for (int it = 0; it < 10; ++it) {  // 65^10 > 2^53 * (hi - lo)
    // lane l tests x_l = lo + (l+1) (hi-lo)/65 ; count(x) == m  <=>  x > theta_max
    const double x = lo + (hi - lo) * (double)(lane + 1) / 65.0;
    const bool above = sturm_count(ws.a, ws.bb, m, x) >= m;
    const unsigned long long bal = __ballot(above);
    const int first = bal ? __ffsll((long long)bal) - 1 : 64;  // first lane above
    const double nlo = first == 0 ? lo : lo + (hi - lo) * (double)first / 65.0;
    const double nhi = first == 64 ? hi : lo + (hi - lo) * (double)(first + 1) / 65.0;
    lo = nlo;
    hi = nhi;
  }
  const double theta_s = 0.5 * (lo + hi);  // in units of tn
  if (lane == 0) {
    double *x = ws.x, *l = ws.l, *rd = ws.rd;
    const double sigma = theta_s + 4e-16;
    // LDL^T of (T - sigma I) / tn:  d_0 = a_0 - sigma, l_i = b_i / d_i,
    // d_{i+1} = a_{i+1} - sigma - b_i^2 / d_i   (all d < 0; a pivot that rounding pushed to
    // zero or above is replaced by a tiny negative one).  Every loop below is a serial
    // recurrence: eight rows at a time, operands loaded before and results stored after the
    // chain, so that an LDS round trip is paid per eight rows instead of per row.
    double d = ws.a[0] - sigma;
    for (int i0 = 0; i0 < m; i0 += 8) {
      double an[8], bq[8], bl[8], rr[8], ll[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        an[e] = ws.a[i0 + e + 1] - sigma;  // row i0 + e + 1 (padding beyond m: unused)
        bq[e] = ws.bb[i0 + e];
        bl[e] = i0 + e < m - 1 ? beta[i0 + e] * itn : 0.0;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (!(d < -1e-30)) d = -1e-30;
        const double r = 1.0 / d;
        rr[e] = r;
        ll[e] = bl[e] * r;
        d = an[e] - bq[e] * r;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) rd[i0 + e] = rr[e], l[i0 + e] = ll[e];
    }
    // rows >= m of l / rd hold padding values: the solves below never use them
    const double x0 = 1.0 / sqrt((double)m);
    for (int i = 0; i < TRI_PAD; ++i) x[i] = i < m ? x0 : 0.0;
    for (int iter = 0; iter < 2; ++iter) {
      // L y = x   (y_0 = x_0, y_i = x_i - l_{i-1} y_{i-1})
      double y = x[0];
      for (int i0 = 1; i0 < m; i0 += 8) {
        double lv[8], xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) lv[e] = l[i0 + e - 1], xv[e] = x[i0 + e];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          y = fma(-lv[e], y, xv[e]);
          xv[e] = y;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (i0 + e < m) x[i0 + e] = xv[e];
      }
      // D z = y ; L^T w = z   (w_{m-1} = y_{m-1} / d_{m-1}, w_i = y_i / d_i - l_i w_{i+1})
      double w = 0.0, nx = 0.0;
      for (int i0 = (m - 1) & ~7; i0 >= 0; i0 -= 8) {
        double lv[8], zv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool in = i0 + e < m;
          lv[e] = (i0 + e < m - 1) ? l[i0 + e] : 0.0;
          zv[e] = in ? x[i0 + e] * rd[i0 + e] : 0.0;
        }
#pragma unroll
        for (int e = 7; e >= 0; --e) {
          w = fma(-lv[e], w, zv[e]);  // rows >= m: lv = zv = 0 keep w = 0
          zv[e] = w;
          nx = fma(w, w, nx);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (i0 + e < m) x[i0 + e] = zv[e];
      }
      nx = 1.0 / sqrt(nx);
      for (int i0 = 0; i0 < m; i0 += 8) {
        double xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = x[i0 + e] * nx;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (i0 + e < m) x[i0 + e] = xv[e];
      }
    }
  }
  return theta_s * tn;
}

// Work area of the small-matrix solver (n <= LANCZOS_M), in dynamic LDS, zeroed by the kernel.
struct SmallWork {
  double G[LANCZOS_M][LANCZOS_M + 1];      // the matrix, zero beyond n
  double Q[LANCZOS_M + 1][LANCZOS_M + 1];  // B_k, ping
  double P[LANCZOS_M][LANCZOS_M + 1];      // B_k, pong
  double w[LANCZOS_M + 1];
};

// Small matrices (n <= LANCZOS_M) by repeated squaring: B_0 = G / tr G, B_{k+1} = B_k^2 / tr B_k^2 tends to
// v v^T for the leading eigenvector v, the weight of the second eigenvalue being squared at
// every step (tr B_k^2 -> 1).  One squaring of a <= 48 x 48 matrix is at most nine 16 x 16
// tiles of v_mfma_f64_16x16x4_f64, one wave each, operands straight from LDS -- about a dozen
// block barriers in all (a whole-space Lanczos pass in one wave, n dependent steps and a
// tridiagonal eigen-solve, took 50-70 us for n = 20 where this takes about 15).  All 1024 threads of the block take part; sw.G holds the matrix
// (zero beyond n) and is kept for the residual.  B_k is symmetric by construction (both
// operands are read as rows k of B_{k-1}: B^T B).
// NW: waves of the calling block (16 in lanczos_kernel, 8 in lanczos_plain_kernel).
template <int NW>
__device__ void eig_small_power(int n, int ld, SmallWork &sw, double *__restrict__ v, double *info3) {
  __shared__ double s_tr, s_part[4];
  constexpr int NT = 64 * NW, NS = (9 + NW - 1) / NW;  // at most nine tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) {
    const double d = lane < n ? sw.G[lane][lane] : 0.0;
    const double tr = wave_sum_d(d);
    if (lane == 0) s_tr = tr;
  }
  __syncthreads();
  const double tr0 = s_tr;
  if (!(tr0 > 0.0)) {  // G == 0: any unit vector
    for (int e = tid; e < ld; e += NT) v[e] = e == 0 ? 1.0 : 0.0;
    if (tid == 0 && info3) info3[0] = 0.0, info3[1] = 0.0, info3[2] = 0.0;
    return;
  }
  double (*cur)[LANCZOS_M + 1] = sw.Q, (*nxt)[LANCZOS_M + 1] = sw.P;
  for (int i = tid; i < LANCZOS_M * (LANCZOS_M + 1); i += NT) {
    const int r = i / (LANCZOS_M + 1), c = i - r * (LANCZOS_M + 1);
    cur[r][c] = sw.G[r][c] / tr0;
  }
  __syncthreads();
  const int T = (n + 15) >> 4, ntile = T * T;
  const int l16 = lane & 15, l4 = lane >> 4;
  double t_prev = 0.0;
  bool last = false;
  int it = 0;
  for (; it < 64; ++it) {
    double4_t acc[NS];
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      acc[sidx] = double4_t{0, 0, 0, 0};
      const int tile = wave + NW * sidx;
      if (tile < ntile) {
        const int ti = tile / T, tj = tile - ti * T;
        for (int k0 = 0; k0 < 16 * T; k0 += 4) {
          const double a = cur[k0 + l4][ti * 16 + l16];
          const double b = cur[k0 + l4][tj * 16 + l16];
          acc[sidx] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[sidx], 0, 0, 0);
        }
        if (ti == tj) {  // trace of the new matrix: D[row = l4 + 4r][col = l16]
          double d = 0.0;
#pragma unroll
          for (int r = 0; r < 4; ++r) d += (l16 == l4 + 4 * r) ? acc[sidx][r] : 0.0;
          d = wave_sum_d(d);
          if (lane == 0) s_part[ti] = d;
        }
      }
    }
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < T; ++i) t += s_part[i];
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      const int tile = wave + NW * sidx;
      if (tile < ntile) {
        const int ti = tile / T, tj = tile - ti * T;
#pragma unroll
        for (int r = 0; r < 4; ++r) nxt[ti * 16 + l4 + 4 * r][tj * 16 + l16] = acc[sidx][r] / t;
      }
    }
    __syncthreads();
    double (*sw_)[LANCZOS_M + 1] = cur;
    cur = nxt;
    nxt = sw_;
    // 1 - t ~ twice the relative weight of the rest of the spectrum: once it is below 1e-8 one
    // more squaring takes it below the rounding level; a stalled t means a repeated leading
    // eigenvalue (any vector of its eigenspace will do)
    if (last || fabs(t - t_prev) <= 2e-16 * t) {
      ++it;
      break;
    }
    last = 1.0 - t <= 1e-8;
    t_prev = t;
  }
  if (wave == 0) {
    // B ~ v v^T: the column through the largest diagonal entry, normalised
    const double d = lane < n ? cur[lane][lane] : -1.0;
    double best = d;
    int bj = lane;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double od = __shfl_xor(best, off, 64);
      const int oj = __shfl_xor(bj, off, 64);
      if (od > best || (od == best && oj < bj)) best = od, bj = oj;
    }
    const double x = lane < n ? cur[lane][bj] : 0.0;
    const double nrm = sqrt(wave_sum_d(x * x));
    const double vi = nrm > 0.0 ? x / nrm : (lane == 0 ? 1.0 : 0.0);
    if (lane < n) v[lane] = vi;
    for (int e = n + lane; e < ld; e += 64) v[e] = 0.0;
    if (info3) {  // Rayleigh quotient and residual against the original matrix
      if (lane < LANCZOS_M) sw.w[lane] = lane < n ? vi : 0.0;
      double y = 0.0;
      if (lane < n)
        for (int c = 0; c < n; ++c) y = fma(sw.G[lane][c], sw.w[c], y);
      const double theta = wave_sum_d(lane < n ? y * vi : 0.0);
      const double r = lane < n ? y - theta * vi : 0.0;
      const double res = sqrt(wave_sum_d(r * r));
      if (lane == 0) info3[0] = theta, info3[1] = res, info3[2] = 0.0;
    }
  }
}

// The same iteration for LANCZOS_M < n <= PW_N with both buffers in the block's dynamic LDS
// (2 x 96 x 97 float64 = 146 KB): up to 36 tiles, at most three per wave.  G is read from
// global memory (these sizes never take the slab path).  Lanczos needs 24-48 dependent steps of
// ~10 us for such a matrix; a squaring is ~6 us and a dozen of them suffice.
constexpr int PW_N = 96, PW_LD = 97;
constexpr size_t PW_BYTES = (size_t)2 * PW_N * PW_LD * sizeof(double);
template <int NW>
__device__ void eig_mid_power(const double *__restrict__ Gk, int n, int ld, double *lds,
                              double *__restrict__ v, double *info3) {
  __shared__ double s_tr, s_part[16], s_vec[PW_N];
  constexpr int NT = 64 * NW, NS = (36 + NW - 1) / NW;  // at most 36 tiles
  double (*cur)[PW_LD] = reinterpret_cast<double (*)[PW_LD]>(lds);
  double (*nxt)[PW_LD] = cur + PW_N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) {
    double d = lane < n ? Gk[(long)lane * ld + lane] : 0.0;
    if (lane + 64 < n) d += Gk[(long)(lane + 64) * ld + lane + 64];
    const double tr = wave_sum_d(d);
    if (lane == 0) s_tr = tr;
  }
  __syncthreads();
  const double tr0 = s_tr;
  if (!(tr0 > 0.0)) {
    for (int e = tid; e < ld; e += NT) v[e] = e == 0 ? 1.0 : 0.0;
    if (tid == 0 && info3) info3[0] = 0.0, info3[1] = 0.0, info3[2] = 0.0;
    return;
  }
  for (int i = tid; i < PW_N * PW_LD; i += NT) {
    const int r = i / PW_LD, c = i - r * PW_LD;
    cur[r][c] = (r < n && c < n) ? Gk[(long)r * ld + c] / tr0 : 0.0;
  }
  __syncthreads();
  const int T = (n + 15) >> 4, ntile = T * T;
  const int l16 = lane & 15, l4 = lane >> 4;
  double t_prev = 0.0;
  bool last = false;
  int it = 0;
  for (; it < 64; ++it) {
    double4_t acc[NS];
    double dsum = 0.0;
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      acc[sidx] = double4_t{0, 0, 0, 0};
      const int tile = wave + NW * sidx;
      if (tile < ntile) {
        const int ti = tile / T, tj = tile - ti * T;
        for (int k0 = 0; k0 < 16 * T; k0 += 4) {
          const double a = cur[k0 + l4][ti * 16 + l16];
          const double b = cur[k0 + l4][tj * 16 + l16];
          acc[sidx] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[sidx], 0, 0, 0);
        }
        if (ti == tj) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dsum += (l16 == l4 + 4 * r) ? acc[sidx][r] : 0.0;
        }
      }
    }
    dsum = wave_sum_d(dsum);
    if (lane == 0) s_part[wave] = dsum;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += s_part[w];
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      const int tile = wave + NW * sidx;
      if (tile < ntile) {
        const int ti = tile / T, tj = tile - ti * T;
#pragma unroll
        for (int r = 0; r < 4; ++r) nxt[ti * 16 + l4 + 4 * r][tj * 16 + l16] = acc[sidx][r] / t;
      }
    }
    __syncthreads();
    double (*sw_)[PW_LD] = cur;
    cur = nxt;
    nxt = sw_;
    if (last || fabs(t - t_prev) <= 2e-16 * t) {
      ++it;
      break;
    }
    last = 1.0 - t <= 1e-8;
    t_prev = t;
  }
  if (wave == 0) {
    // column through the largest diagonal entry (two rows per lane), normalised
    const double d0 = lane < n ? cur[lane][lane] : -1.0;
    const double d1 = lane + 64 < n ? cur[lane + 64][lane + 64] : -1.0;
    double best = d1 > d0 ? d1 : d0;
    int bj = d1 > d0 ? lane + 64 : lane;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double od = __shfl_xor(best, off, 64);
      const int oj = __shfl_xor(bj, off, 64);
      if (od > best || (od == best && oj < bj)) best = od, bj = oj;
    }
    const double x0 = lane < n ? cur[lane][bj] : 0.0;
    const double x1 = lane + 64 < n ? cur[lane + 64][bj] : 0.0;
    const double nrm = sqrt(wave_sum_d(x0 * x0 + x1 * x1));
    const double v0 = nrm > 0.0 ? x0 / nrm : (lane == 0 ? 1.0 : 0.0), v1 = nrm > 0.0 ? x1 / nrm : 0.0;
    if (lane < n) v[lane] = v0;
    if (lane + 64 < n) v[lane + 64] = v1;
    for (int e = n + lane; e < ld; e += 64) v[e] = 0.0;
    if (info3) {
      s_vec[lane] = v0;
      if (lane < PW_N - 64) s_vec[lane + 64] = v1;
      double y0 = 0.0, y1 = 0.0;
      for (int c = 0; c < n; ++c) {
        if (lane < n) y0 = fma(Gk[(long)lane * ld + c], s_vec[c], y0);
        if (lane + 64 < n) y1 = fma(Gk[(long)(lane + 64) * ld + c], s_vec[c], y1);
      }
      const double theta = wave_sum_d(y0 * v0 + y1 * v1);
      const double r0 = lane < n ? y0 - theta * v0 : 0.0, r1 = lane + 64 < n ? y1 - theta * v1 : 0.0;
      const double res = sqrt(wave_sum_d(r0 * r0 + r1 * r1));
      if (lane == 0) info3[0] = theta, info3[1] = res, info3[2] = 0.0;
    }
  }
}

// QLDS: the Krylov basis (LANCZOS_M + 2 rows of length ld) lives in dynamic LDS instead of global
// memory -- the Gram-Schmidt passes of this one-block kernel are chains of dependent reads, and
// an L2 round trip costs ~0.7 us against ~0.05 us for LDS.  The host picks it when the basis of
// the largest matrix of the launch fits (ld <= LANCZOS_QLDS_LD).
constexpr int LANCZOS_QLDS_LD = 368;  // (48 + 2) * 368 * 8 B = 147 KB
template <bool QLDS>
__global__ __launch_bounds__(1024) void lanczos_kernel(const double *__restrict__ G,
                                                       const long *__restrict__ g_off,
                                                       const long *__restrict__ ld_,
                                                       const long *__restrict__ n_,
                                                       double *__restrict__ Q,
                                                       const long *__restrict__ q_off,
                                                       double *__restrict__ vout,
                                                       const long *__restrict__ v_off,
                                                       int max_restart, double tol,
                                                       double *__restrict__ info,
                                                       const double *__restrict__ slab,
                                                       long slab_stride, int ksplit,
                                                       int power_nmax,
                                                       double *__restrict__ dbg = nullptr) {
  __shared__ double alpha[LANCZOS_M], beta[LANCZOS_M], h[LANCZOS_M + 1];
  __shared__ TriWork ws;
  __shared__ double red[16];
  __shared__ double upd[4][256];
  __shared__ double s_theta;
  extern __shared__ __align__(16) double lz_dyn[];  // SmallWork, or the basis when QLDS
  double *svec = ws.x;
  const int k = blockIdx.x;
  const int n = (int)n_[k], ld = (int)ld_[k];
  if (n < 1) return;  // (n == 1 is a valid 1 x 1 problem for origin_pca_eig; the PCA loop never
                      // sends fewer than two columns except for areas that have just finished)
  const unsigned long long t_dbg = dbg ? wall_clock64() : 0ull;  // 100 MHz
  int steps_dbg = 0;
  const double *Gk = G + g_off[k];
  double *Qk;  // (LANCZOS_M + 2) rows of length ld; last row = Ritz vector
  if constexpr (QLDS) Qk = lz_dyn;
  else Qk = Q + q_off[k];
  double *y = Qk + (long)(LANCZOS_M + 1) * ld;
  double *v = vout + v_off[k];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (n <= LANCZOS_M) {  // small matrix: repeated squaring on the matrix cores (eig_small_power)
    SmallWork &sw = *reinterpret_cast<SmallWork *>(lz_dyn);
    {  // the solver relies on zeros beyond n
      double *z8 = reinterpret_cast<double *>(&sw);
      for (int i = tid; i < (int)(sizeof(SmallWork) / sizeof(double)); i += 1024) z8[i] = 0.0;
      __syncthreads();
    }
    if (slab) {
      // G straight from the K-split slabs of gram_kernel (the reduction kernel is skipped when
      // every matrix of the launch is small): same sums, in the same order, as
      // gram_reduce_kernel.  Only tiles (ti <= tj) exist: the lower-left one is read mirrored.
      // All 16 waves share the rows; the <= 32 slab values of an element are independent loads
      // (one L2 round trip), summed in slab order.
      const double *sk = slab + g_off[k];
      for (int c = wave; c < n; c += 16) {
        if (lane < n) {
          const bool up = (c >> 5) <= (lane >> 5);
          const long off = up ? (long)c * ld + lane : (long)lane * ld + c;
          double part[32];
#pragma unroll
          for (int ks = 0; ks < 32; ++ks) part[ks] = ks < ksplit ? sk[(long)ks * slab_stride + off] : 0.0;
          double acc = 0.0;
#pragma unroll
          for (int ks = 0; ks < 32; ++ks) acc += part[ks];  // trailing zeros do not change the sum
          sw.G[c][lane] = acc;
        }
      }
      __syncthreads();
    }
    else {
      for (int c = wave; c < n; c += 16)
        if (lane < n) sw.G[c][lane] = Gk[(long)c * ld + lane];
      __syncthreads();
    }
    eig_small_power<16>(n, ld, sw, v, info ? info + 3 * k : nullptr);
    if (dbg && threadIdx.x == 0) dbg[2 * k] = -1.0, dbg[2 * k + 1] = (double)(wall_clock64() - t_dbg) * 0.01;
    return;
  }
  if (n <= power_nmax) {  // mid-size matrix: repeated squaring with both buffers in LDS
    eig_mid_power<16>(Gk, n, ld, lz_dyn, v, info ? info + 3 * k : nullptr);
    if (dbg && threadIdx.x == 0) dbg[2 * k] = -2.0, dbg[2 * k + 1] = (double)(wall_clock64() - t_dbg) * 0.01;
    return;
  }
  const int mfull = min(LANCZOS_M, n);

  // start vector: G * ones (a few power-like steps come for free in the Krylov space)
  for (int r = wave; r < n; r += 16) {
    double acc = 0.0;
    for (int c = lane; c < n; c += 64) acc += Gk[(long)r * ld + c];
    acc = wave_sum_d(acc);
    if (lane == 0) y[r] = acc;
  }
  __syncthreads();
  double theta = 0.0, resid = 0.0;
  int restarts = 0;
  for (; restarts < max_restart; ++restarts) {
    // q_0 = y / |y|
    double p = 0.0;
    for (int e = tid; e < n; e += 1024) p = fma(y[e], y[e], p);
    const double nrm = sqrt(block_sum(p, red));
    const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
    for (int e = tid; e < n; e += 1024) Qk[e] = (nrm > 0.0) ? y[e] * inv : (e == 0 ? 1.0 : 0.0);
    __syncthreads();
    int m = 0;
    double beta_last = 0.0;
    // a dominant nuisance converges in a few steps: the Ritz pair is tested at LANCZOS_M/2
    // and 3/4 LANCZOS_M, and the recurrence goes on from there (no restart) when it fails
    const int mmax = mfull;
    bool solved = false;
    for (int j = 0; j < mmax; ++j) {
      const double *qj = Qk + (long)j * ld;
      double *w = Qk + (long)(j + 1) * ld;
      // w = G q_j.  G is symmetric: thread r accumulates sum_c G[c][r] q_j[c], so that lanes
      // read consecutive addresses and no cross-lane reduction is needed; the columns are
      // split over 4 groups of 256 threads and combined through LDS.
      {
        const int part = tid >> 8, r0 = tid & 255;
        for (int rb = 0; rb < n; rb += 256) {
          const int r = rb + r0;
          double acc = 0.0;
          if (r < n) {
            // G comes from L2 (~0.7 us per round trip): 16 loads per lane in flight
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int c = part;
            for (; c + 60 < n; c += 64) {
              double g[16];
#pragma unroll
              for (int q = 0; q < 16; ++q) g[q] = Gk[(long)(c + 4 * q) * ld + r];
#pragma unroll
              for (int q = 0; q < 16; q += 4) {
                a0 = fma(g[q], qj[c + 4 * q], a0);
                a1 = fma(g[q + 1], qj[c + 4 * q + 4], a1);
                a2 = fma(g[q + 2], qj[c + 4 * q + 8], a2);
                a3 = fma(g[q + 3], qj[c + 4 * q + 12], a3);
              }
            }
            for (; c < n; c += 4) a0 = fma(Gk[(long)c * ld + r], qj[c], a0);
            acc = (a0 + a1) + (a2 + a3);
          }
          upd[part][r0] = acc;
          __syncthreads();
          if (part == 0 && r < n) w[r] = (upd[0][r0] + upd[1][r0]) + (upd[2][r0] + upd[3][r0]);
          __syncthreads();
        }
      }
      // classical Gram-Schmidt against q_0..q_j, twice; alpha_j = first-pass h_j (+ fix)
      double aj = 0.0;
      for (int pass = 0; pass < 2; ++pass) {
        for (int i = wave; i <= j; i += 16) {
          const double *qi = Qk + (long)i * ld;
          double acc = 0.0;
#pragma unroll 4
          for (int c = lane; c < n; c += 64) acc = fma(qi[c], w[c], acc);
          acc = wave_sum_d(acc);
          if (lane == 0) h[i] = acc;
        }
        __syncthreads();
        aj += h[j];
        // w -= Q h with the j+1 rows split over 4 thread groups (independent loads in
        // flight instead of one dependent chain per element), combined through LDS
        {
          const int part = tid >> 8, e0 = tid & 255;
          for (int eb = 0; eb < n; eb += 256) {
            const int e = eb + e0;
            double acc = 0.0;
            if (e < n) {
#pragma unroll 4
              for (int i = part; i <= j; i += 4) acc = fma(h[i], Qk[(long)i * ld + e], acc);
            }
            upd[part][e0] = acc;
            __syncthreads();
            if (part == 0 && e < n)
              w[e] -= (upd[0][e0] + upd[1][e0]) + (upd[2][e0] + upd[3][e0]);
            __syncthreads();
          }
        }
      }
      double pw = 0.0;
      for (int e = tid; e < n; e += 1024) pw = fma(w[e], w[e], pw);
      const double bj = sqrt(block_sum(pw, red));
      if (tid == 0) alpha[j] = aj, beta[j] = bj;
      m = j + 1;
      beta_last = bj;
      // invariant subspace reached (also the exact case m == n)
      if (bj <= 1e-300 || bj <= 1e-15 * fabs(aj)) break;
      if (j + 1 < mmax) {
        const double ib = 1.0 / bj;
        for (int e = tid; e < n; e += 1024) w[e] *= ib;
      }
      __syncthreads();
      if (j + 1 < mmax && (j + 1 == LANCZOS_M / 2 || j + 1 == 3 * LANCZOS_M / 4)) {
        if (wave == 0) {
          const double th = tridiag_top(alpha, beta, m, lane, ws);
          if (lane == 0) s_theta = th;
        }
        __syncthreads();
        solved = fabs(bj * svec[m - 1]) <= tol * fabs(s_theta);  // block-uniform
        if (solved) break;
      }
    }
    __syncthreads();
    // ---- largest eigenpair of T_m (wave 0)
    if (!solved) {
      if (wave == 0) {
        const double th = tridiag_top(alpha, beta, m, lane, ws);
        if (lane == 0) s_theta = th;
      }
      __syncthreads();
    }
    theta = s_theta;
    // y = Q s
    for (int e = tid; e < n; e += 1024) {
      double acc = 0.0;
#pragma unroll 8
      for (int i = 0; i < m; ++i) acc = fma(svec[i], Qk[(long)i * ld + e], acc);
      y[e] = acc;
    }
    resid = fabs(beta_last * svec[m - 1]);
    steps_dbg += m;
    __syncthreads();
    if (m >= n || resid <= tol * fabs(theta)) break;
  }
  // final normalisation of the eigenvector
  double p = 0.0;
  for (int e = tid; e < n; e += 1024) p = fma(y[e], y[e], p);
  const double nrm = sqrt(block_sum(p, red));
  const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
  for (int e = tid; e < n; e += 1024) v[e] = y[e] * inv;
  for (int e = n + tid; e < ld; e += 1024) v[e] = 0.0;
  if (tid == 0 && info) {
    info[3 * k] = theta;
    info[3 * k + 1] = resid;
    info[3 * k + 2] = (double)restarts;
  }
  if (tid == 0 && dbg) dbg[2 * k] = (double)steps_dbg, dbg[2 * k + 1] = (double)(wall_clock64() - t_dbg) * 0.01;
}

// ------------------------------------------------------------------------------------
// Plain Lanczos with the matrix resident on the CU  (round 3).
//
// What the per-matrix statistics of the round-2 kernel showed (ORIGIN_PCA_DEBUG_EIG, 3681 x 600 x 600
// and 900 x 900): a Lanczos step cost 7-19 us (59 us once the basis of the launch's largest matrix
// no longer fitted in LDS) of which the float64 arithmetic is a few hundred cycles -- the rest were
// L2 round trips for G (re-read every step), two Gram-Schmidt passes over the whole basis and a
// dozen block barriers; the Ritz pair was tested at 24 / 36 / 48 vectors only, so every matrix paid
// at least 24 steps and 48 + 24 when 48 were not enough.  Measured on the Gram matrices of the
// bench field (NumPy prototype): the leading pair converges to 1e-14 in 10-52 steps, and the
// three-term recurrence WITHOUT re-orthogonalisation takes exactly as many steps and gives the
// same vector -- orthogonality is only lost once a Ritz pair has converged (Paige), and the first
// to converge is the one wanted here; the iteration stops there.
//
// So: one block of 512 threads (8 waves, 256 VGPRs each) per matrix, G loaded ONCE.  Lane l of wave
// g keeps G[g + 8 i][l + 64 k] for its RS = 4 RC rows k and its column slots i: the first
// LP_NREG / RS slots in registers, the next LP_NLDS / RS in LDS (a column of 512 doubles per
// element: conflict free), slots beyond (n > 208) are streamed from L2 per step.  A mat-vec reads
// q[g + 8 i] once per slot (wave-uniform: one LDS broadcast serves RS FMAs per lane), the eight
// column groups are summed through LDS in a fixed order.  Per step: mat-vec, two block reductions
// (alpha, beta), four barriers; the basis goes to global memory (written once, read once for
// y = V s).  The Ritz pair is tested on a schedule that thins out (8, 12, .. 24, 32, .. 64, 80, ..);
// the accepted vector is verified against G itself (true residual <= LP_VERIFY_TOL * theta),
// otherwise -- and when LP_MAXS steps did not converge -- the recurrence restarts from it.
// n <= 96 takes the repeated-squaring solvers; a launch with n > LP_NMAX is left to lanczos_kernel.
// ------------------------------------------------------------------------------------
constexpr int LP_NT = 512, LP_NW = LP_NT / 64;  // threads / waves (= column groups) per block
constexpr int LP_NREG = 80, LP_NLDS = 24;       // doubles per thread in registers / in LDS
constexpr int LP_MAXS = 192;                    // Lanczos vectors before a restart
constexpr int LP_NMAX = 512;                    // eight rows per lane at most
constexpr int LP_PAD = LP_MAXS + 8;
constexpr int EIG_QROWS = LP_MAXS + 2;          // basis rows per matrix in the scratch (>= LANCZOS_M + 2)
constexpr double LP_VERIFY_TOL = 1e-12;
typedef TriWorkT<LP_PAD> TriWorkL;

struct PlainLds {  // carved from the dynamic LDS of the block
  double alpha[LP_MAXS], beta[LP_MAXS];
  TriWorkL ws;
  double qv[LP_NMAX + 16], qp[LP_NMAX];  // qv: zero beyond n, always
  double psum[LP_NW][LP_NMAX];
  double red[8];
  double glds[LP_NLDS][LP_NT];
};
constexpr size_t LP_BYTES = sizeof(PlainLds);
static_assert(LP_BYTES <= 156 * 1024, "plain Lanczos work area must fit the CU's LDS");

// The tridiagonal solver of lanczos_kernel with a small register footprint (the matrix sits in
// registers next to it): bounds and every row-independent quantity are computed lane-parallel,
// the serial recurrences (Sturm chains, pivots of the LDL^T factorisation, the two triangular
// solves) walk four rows per LDS round trip.  Same algorithm, same results to rounding; any m
// up to the padded size of the work area.
__device__ __forceinline__ int sturm_count4(const double *a, const double *bb, int m, double x) {
  double p0 = 1.0, p1 = a[0] - x;
  bool s1 = p1 < 0.0 || p1 == 0.0;
  int cnt = s1;
  for (int i0 = 0; i0 < m - 1; i0 += 4) {
    double av[4], bv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) av[e] = a[i0 + 1 + e] - x, bv[e] = bb[i0 + e];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double p2 = fma(av[e], p1, -bv[e] * p0);
      const bool s2 = p2 < 0.0 || (p2 == 0.0 && !s1);
      cnt += (i0 + 1 + e < m) && (s2 != s1);
      p0 = p1, p1 = p2, s1 = s2;
    }
    if (fabs(p1) < 1e-100 && fabs(p0) < 1e-100) p0 *= 1e100, p1 *= 1e100;
    if (fabs(p1) > 1e100 || fabs(p0) > 1e100) p0 *= 1e-100, p1 *= 1e-100;
  }
  return cnt;
}

__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}

template <class TW>
__device__ __forceinline__ double tridiag_top_lean(const double *alpha, const double *beta, int m,
                                                   int lane, TW &ws) {
  double lo = 1e300, hi = -1e300, tn = 0.0;
  for (int i = lane; i < m; i += 64) {
    const double r = (i > 0 ? fabs(beta[i - 1]) : 0.0) + (i < m - 1 ? fabs(beta[i]) : 0.0);
    lo = fmin(lo, alpha[i] - r);
    hi = fmax(hi, alpha[i] + r);
    tn = fmax(tn, fabs(alpha[i]) + r);
  }
  lo = -wave_max_d(-lo);
  hi = wave_max_d(hi);
  tn = wave_max_d(tn);
  if (!(tn > 0.0)) {  // T == 0
    for (int i = lane; i < m; i += 64) ws.x[i] = i == 0 ? 1.0 : 0.0;
    return 0.0;
  }
  const double itn = 1.0 / tn;
  for (int i = lane; i < TW::PAD; i += 64) {  // scaled copy, zero padding behind row m - 1
    ws.a[i] = i < m ? alpha[i] * itn : 0.0;
    const double b = i < m - 1 ? beta[i] * itn : 0.0;
    ws.bb[i] = b * b;
  }
  lo *= itn;
  hi = hi * itn + 1e-14;
  for (int it = 0; it < 10; ++it) {  // 65^10 > 2^53 * (hi - lo)
    const double x = lo + (hi - lo) * (double)(lane + 1) / 65.0;
    const bool above = sturm_count4(ws.a, ws.bb, m, x) >= m;
    const unsigned long long bal = __ballot(above);
    const int first = bal ? __ffsll((long long)bal) - 1 : 64;
    const double nlo = first == 0 ? lo : lo + (hi - lo) * (double)first / 65.0;
    const double nhi = first == 64 ? hi : lo + (hi - lo) * (double)(first + 1) / 65.0;
    lo = nlo;
    hi = nhi;
  }
  const double theta_s = 0.5 * (lo + hi);
  const double sigma = theta_s + 4e-16;
  // pivots of the LDL^T of (T - sigma I) / tn: d_0 = a_0 - sigma, d_{i+1} = a_{i+1} - sigma -
  // bb_i / d_i (all negative; one that rounding pushed to zero or above becomes a tiny negative
  // one); rd = 1 / d
  if (lane == 0) {
    double d = ws.a[0] - sigma;
    for (int i0 = 0; i0 < m; i0 += 4) {
      double an[4], bq[4], rr[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) an[e] = ws.a[i0 + e + 1] - sigma, bq[e] = ws.bb[i0 + e];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (!(d < -1e-30)) d = -1e-30;
        rr[e] = 1.0 / d;
        d = an[e] - bq[e] * rr[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) ws.rd[i0 + e] = rr[e];
    }
  }
  const double x0 = 1.0 / sqrt((double)m);
  for (int i = lane; i < TW::PAD; i += 64) {
    ws.l[i] = i < m - 1 ? beta[i] * itn * ws.rd[i] : 0.0;  // l_i = b_i / d_i
    ws.x[i] = i < m ? x0 : 0.0;
  }
  for (int iter = 0; iter < 2; ++iter) {
    double nx = 0.0;
    if (lane == 0) {
      // L y = x   (y_0 = x_0, y_i = x_i - l_{i-1} y_{i-1})
      double y = ws.x[0];
      for (int i0 = 1; i0 < m; i0 += 4) {
        double lv[4], xv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) lv[e] = ws.l[i0 + e - 1], xv[e] = ws.x[i0 + e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y = fma(-lv[e], y, xv[e]);
          xv[e] = y;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (i0 + e < m) ws.x[i0 + e] = xv[e];
      }
      // D z = y ; L^T w = z   (w_{m-1} = y_{m-1} / d_{m-1}, w_i = y_i / d_i - l_i w_{i+1})
      double w = 0.0;
      for (int i0 = (m - 1) & ~3; i0 >= 0; i0 -= 4) {
        double lv[4], zv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lv[e] = (i0 + e < m - 1) ? ws.l[i0 + e] : 0.0;
          zv[e] = (i0 + e < m) ? ws.x[i0 + e] * ws.rd[i0 + e] : 0.0;
        }
#pragma unroll
        for (int e = 3; e >= 0; --e) {
          w = fma(-lv[e], w, zv[e]);  // rows >= m: lv = zv = 0 keep w = 0
          zv[e] = w;
          nx = fma(w, w, nx);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (i0 + e < m) ws.x[i0 + e] = zv[e];
      }
      nx = 1.0 / sqrt(nx);
    }
    nx = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(nx)),
                          __builtin_amdgcn_readfirstlane(__double2loint(nx)));
    for (int i = lane; i < m; i += 64) ws.x[i] *= nx;
  }
  return theta_s * tn;
}

// Top eigenpair of T for the convergence checks of lanczos_plain, by ONE wave with T in registers.
//
// Measured (ORIGIN_PCA_DEBUG_EIG phase timers, n = 204, 64 steps): mat-vecs 82 us, vector part
// 58 us, and 278 us in twelve checks with the LDS-resident solver -- 0.74 us per row of T and
// check: ten multisection rounds of a Sturm chain that waits for an LDS round trip every four
// rows, then the serial LDL^T sweeps.  Here instead:
//  * lane i of three register pairs holds row i (64 j + i) of the scaled T; a chain step takes its
//    row through v_readlane (scalar operands, no memory at all);
//  * the eigenvalue bracket starts at the previous check's Ritz value (Ritz values only grow with
//    m) and its first round is GEOMETRIC towards that end -- theta_m - theta_{m'} is of the order of
//    the previous residual squared --, so a check needs ~4-6 rounds instead of 10;
//  * the eigenvector comes from the three-term recurrence run BOTTOM-UP (x_{m-1} = 1,
//    x_{i-1} = ((theta - a_i) x_i - b_i x_{i+1}) / b_{i-1}): the twisted factorisation with the
//    twist at row 0, x = (T - theta)^{-1} e_0 up to scale.  The top Ritz vector of a Lanczos
//    tridiagonal is largest at the top and decays downwards as the pair converges, so the
//    recurrence runs in its direction of growth (stable; NumPy prototype on the bench field's Gram
//    matrices: estimate and vector equal LAPACK's to 1e-12 at every check).  Every lane runs the
//    same chain on scalar operands and keeps "its" entries.
// The vector a check accepts is verified against G itself afterwards; when that fails the LDL^T
// solver (tridiag_top_lean) takes over.  m <= 192.
__device__ __forceinline__ double readlane_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                          __builtin_amdgcn_readlane(__double2loint(v), l));
}

struct TriRegs {
  double a[3];    // a_i / tn
  double b[3];    // b_i / tn      (couples rows i, i + 1; 0 for i >= m - 1)
  double bbp[3];  // (b_{i-1} / tn)^2  at position i  (0 at i = 0)
  double rb[3];   // tn / b_i      (0 for i >= m - 1)
};

// eigenvalues of T (scaled) below x: sign changes of the leading principal minors
// p_i = (a_i - x) p_{i-1} - bb_{i-1} p_{i-2}.  A step is three float64 operations and one
// v_alignbit that shifts the sign bit of p_i into a 32-bit history; the changes are counted per
// 31 rows (popcount of history ^ history >> 1).  A minor that is exactly zero counts as positive:
// the next one, -bb p_{i-2}, then has the sign opposite to p_{i-2}, which gives the same number
// of changes as the "zero takes the sign opposite to its predecessor" rule of sturm_count for
// every interior row, and makes the count that of the eigenvalues strictly below x.
__device__ __forceinline__ int sturm_count_rl(const TriRegs &t, int m, double x) {
  // hist: the signs of the last `held` minors, newest at bit 0; starts with p_{-1} = 1
  unsigned hist = 0u;
  int held = 1, cnt = 0;
  double p0 = 1.0, p1 = readlane_d(t.a[0], 0) - x;
  hist = __builtin_amdgcn_alignbit(hist, (unsigned)__double2hiint(p1), 31);  // hist << 1 | sign
  held = 2;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int l0 = j == 0 ? 1 : 0, l1 = min(64, m - 64 * j);
    for (int l = l0; l < l1; ++l) {
      const double ai = readlane_d(t.a[j], l), bi = readlane_d(t.bbp[j], l);
      const double p2 = fma(ai - x, p1, -bi * p0);
      hist = __builtin_amdgcn_alignbit(hist, (unsigned)__double2hiint(p2), 31);
      p0 = p1, p1 = p2;
      if (++held == 32) {  // (uniform) 31 adjacent pairs; the newest sign seeds the next batch
        cnt += __popc((hist ^ (hist >> 1)) & 0x7fffffffu);
        hist &= 1u;
        held = 1;
      }
      if ((l & 7) == 7) {
        if (fabs(p1) < 1e-100 && fabs(p0) < 1e-100) p0 *= 1e100, p1 *= 1e100;
        if (fabs(p1) > 1e100 || fabs(p0) > 1e100) p0 *= 1e-100, p1 *= 1e-100;
      }
    }
  }
  // the held - 1 pairs still in the history
  cnt += __popc((hist ^ (hist >> 1)) & ((1u << (held - 1)) - 1u));
  return cnt;
}

// theta_hint: a lower bound of the eigenvalue (the previous check's Ritz value) or -inf
template <class TW>
__device__ __forceinline__ double tridiag_top_fast(const double *alpha, const double *beta, int m,
                                                   int lane, TW &ws, double theta_hint,
                                                   double *tdbg = nullptr) {
  unsigned long long tq = tdbg ? wall_clock64() : 0ull;
  auto lap = [&](int slot) {
    if (tdbg) {
      const unsigned long long t = wall_clock64();
      if (lane == 0) tdbg[slot] += (double)(t - tq);
      tq = t;
    }
  };
  double lo = 1e300, hi = -1e300, tn = 0.0;
  for (int i = lane; i < m; i += 64) {
    const double r = (i > 0 ? fabs(beta[i - 1]) : 0.0) + (i < m - 1 ? fabs(beta[i]) : 0.0);
    lo = fmin(lo, alpha[i] - r);
    hi = fmax(hi, alpha[i] + r);
    tn = fmax(tn, fabs(alpha[i]) + r);
  }
  lo = -wave_max_d(-lo);
  hi = wave_max_d(hi);
  tn = wave_max_d(tn);
  if (!(tn > 0.0) || m == 1) {
    for (int i = lane; i < m; i += 64) ws.x[i] = i == 0 ? 1.0 : 0.0;
    return m == 1 ? alpha[0] : 0.0;
  }
  const double itn = 1.0 / tn;
  TriRegs t;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int i = 64 * j + lane;
    t.a[j] = i < m ? alpha[i] * itn : 0.0;
    const double b = i < m - 1 ? beta[i] * itn : 0.0;
    t.b[j] = b;
    t.rb[j] = i < m - 1 ? 1.0 / b : 0.0;  // (b_i > 0: the recurrence stops at breakdown)
    const double bp = (i >= 1 && i < m) ? beta[i - 1] * itn : 0.0;
    t.bbp[j] = bp * bp;
  }
  lo *= itn;
  hi = hi * itn + 1e-14;
  if (theta_hint * itn > lo) lo = theta_hint * itn - 1e-14;  // Ritz values grow with m
  lap(0);
  // first round geometric towards lo: x_l = lo + W rho^(63 - l), rho = 0.7 (x_0 = lo + 1.7e-10 W)
  {
    const double W = hi - lo;
    const double x = lo + W * exp2(-0.5145731728297583 * (double)(63 - lane));
    const bool above = sturm_count_rl(t, m, x) >= m;
    const unsigned long long bal = __ballot(above);
    const int first = bal ? __ffsll((long long)bal) - 1 : 64;
    const double nhi = first == 64 ? hi : readlane_d(x, first & 63);
    const double nlo = first == 0 ? lo : readlane_d(x, (first - 1) & 63);
    lo = nlo, hi = nhi;
  }
  // uniform 65-way rounds down to three units in the last place of the scaled T (|theta| <= 1)
  for (int it = 0; it < 11 && hi - lo > 6.7e-16; ++it) {
    const double x = lo + (hi - lo) * (double)(lane + 1) / 65.0;
    const bool above = sturm_count_rl(t, m, x) >= m;
    const unsigned long long bal = __ballot(above);
    const int first = bal ? __ffsll((long long)bal) - 1 : 64;
    const double nlo = first == 0 ? lo : lo + (hi - lo) * (double)first / 65.0;
    const double nhi = first == 64 ? hi : lo + (hi - lo) * (double)(first + 1) / 65.0;
    lo = nlo;
    hi = nhi;
  }
  const double th = 0.5 * (lo + hi);
  lap(1);
  // bottom-up recurrence: every lane runs the chain, lane (i & 63) keeps x_i in xr[i >> 6]
  double xr[3] = {0.0, 0.0, 0.0};
  double x1 = 1.0;  // x_{m-1}
  double x0 = (th - readlane_d(t.a[(m - 1) >> 6], (m - 1) & 63)) * readlane_d(t.rb[(m - 2) >> 6], (m - 2) & 63);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (((m - 1) >> 6) == j && ((m - 1) & 63) == lane) xr[j] = x1;
    if (((m - 2) >> 6) == j && ((m - 2) & 63) == lane) xr[j] = x0;
  }
#pragma unroll
  for (int j = 2; j >= 0; --j) {  // rows i = 64 j + l give x_{i-1}
    const int lhi = min(63, m - 2 - 64 * j);
    for (int l = lhi; l >= (j == 0 ? 1 : 0); --l) {
      const int im1 = 64 * j + l - 1;
      const double ai = readlane_d(t.a[j], l), bi = readlane_d(t.b[j], l);
      const double rbi = im1 >= 64 * j ? readlane_d(t.rb[j], l - 1)
                                       : readlane_d(t.rb[j > 0 ? j - 1 : 0], 63);
      const double xn = ((th - ai) * x0 - bi * x1) * rbi;
      x1 = x0, x0 = xn;
      const int jj = im1 >> 6;
      if ((im1 & 63) == lane) {
        if (jj == 0) xr[0] = xn;
        else if (jj == 1) xr[1] = xn;
        else xr[2] = xn;
      }
      if ((l & 7) == 0 && fabs(x0) > 1e100) {  // (uniform; growth <= 1e6 per row) scale the live
                                                 // pair and everything stored so far
        x0 *= 1e-100, x1 *= 1e-100;
        xr[0] *= 1e-100, xr[1] *= 1e-100, xr[2] *= 1e-100;
      }
    }
  }
  double p = 0.0;
#pragma unroll
  for (int j = 0; j < 3; ++j) p = fma(xr[j], xr[j], p);
  p = wave_sum_d(p);
  const double inv = p > 0.0 ? 1.0 / sqrt(p) : 0.0;
#pragma unroll
  for (int j = 0; j < 3; ++j)
    if (64 * j + lane < m) ws.x[64 * j + lane] = xr[j] * inv;
  lap(2);
  return th * tn;
}

// When to test the Ritz pair next.  A test costs ~0.3 us per row of T, a step ~2 us, and the
// residual estimate falls geometrically: after the tests at 8 and 12 vectors the next one goes
// where the last two estimates (e_prev at m_prev, e at m; relative to theta) predict
// e = tol, a little early (85 % of the way; 2..32 steps ahead, at most m / 2).  On the bench field's Gram
// matrices this takes 2-6 tests per solve instead of 2-18 on a fixed schedule and stops within
// two steps of the first converged m (NumPy prototype).
__device__ __forceinline__ int lp_next_check(int m, double e, int m_prev, double e_prev, double tol) {
  if (m_prev == 0 || !(e > 0.0) || !(e_prev > 0.0)) return m + 4;
  const double rate = fmax(log10(e_prev / e) / (double)(m - m_prev), 0.05);  // decades per step
  const double togo = log10(e / tol) / rate;
  // (never more than half of what has been done: an estimate that has not started to fall yet
  // -- a plateau before the Krylov space reaches the leading vector -- predicts nothing)
  const double ahead = fmin(fmax(rint(0.85 * togo), 2.0), fmin(32.0, fmax(4.0, 0.5 * (double)m)));
  return m + (int)ahead;
}

// sum over the four waves that hold the vector entries (threads 0..255); every thread of the
// block gets the total.  Ends with a barrier; `slot` alternates so that a sum can be written
// while stragglers still read the previous one.
__device__ __forceinline__ double lp_vec_sum(double v, double *red, int slot) {
  const int tid = threadIdx.x;
  if (tid < 256) {
    v = wave_sum_d(v);
    if ((tid & 63) == 0) red[slot * 4 + (tid >> 6)] = v;
  }
  __syncthreads();
  return (red[slot * 4] + red[slot * 4 + 1]) + (red[slot * 4 + 2] + red[slot * 4 + 3]);
}

template <int RC>
__device__ void lanczos_plain(const double *__restrict__ Gk, int n, int ld, double *__restrict__ Vk,
                              double *__restrict__ v, PlainLds &L, int max_restart, double tol,
                              double *info3, int *steps_out, double *tph = nullptr) {
  constexpr int RS = 4 * RC;                        // rows per lane
  // (tph: optional per-phase times of thread 0 in 10 ns ticks -- ORIGIN_PCA_DEBUG_EIG)
  unsigned long long tq = tph ? wall_clock64() : 0ull;
  auto lap = [&](int slot) {
    if (tph) {
      const unsigned long long t = wall_clock64();
      if (threadIdx.x == 0) tph[slot] += (double)(t - tq);
      tq = t;
    }
  };
  // column slots in registers / in LDS.  (Eight rows per lane, n > 256, stream most of the matrix
  // from L2 every step: ~70 GB/s per CU, 13 us per step at n = 315.  Trading register slots for
  // 32 instead of 16 loads in flight was measured and is slower -- 1255 against 1134 us for that
  // matrix's 86 steps: the CU's L2 bandwidth bounds it, not the latency.)
  constexpr int NRC = LP_NREG / RS, NLC = LP_NLDS / RS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);  // column group = wave (uniform)
  const int ncol8 = (n + 7) >> 3;
  // ---- G: registers, LDS, the rest stays in global memory.  No bounds tests: rows and columns
  // beyond n are clamped to n - 1 (valid memory); a clamped column meets q[c] = 0 (qv is zero
  // beyond n, always), a clamped row produces a sum nobody reads.
  int rows[RS];
#pragma unroll
  for (int k = 0; k < RS; ++k) rows[k] = min(lane + 64 * k, n - 1);
  double greg[NRC][RS];
#pragma unroll
  for (int i = 0; i < NRC; ++i) {
    const double *row = Gk + (long)min(g + 8 * i, n - 1) * ld;  // wave-uniform
#pragma unroll
    for (int k = 0; k < RS; ++k) greg[i][k] = row[rows[k]];
  }
#pragma unroll
  for (int i = 0; i < NLC; ++i) {
    const double *row = Gk + (long)min(g + 8 * (NRC + i), n - 1) * ld;
#pragma unroll
    for (int k = 0; k < RS; ++k) L.glds[i * RS + k][tid] = row[rows[k]];
  }
  for (int e = tid; e < LP_NMAX + 16; e += LP_NT) L.qv[e] = e < n ? 1.0 : 0.0;
  for (int e = tid; e < LP_NMAX; e += LP_NT) L.qp[e] = 0.0;
  __syncthreads();
  // partial sums of G q over this wave's columns, for the lane's rows -> psum[g][row]; q = L.qv
  auto matvec = [&]() {
    double acc[RS];
#pragma unroll
    for (int k = 0; k < RS; ++k) acc[k] = 0.0;
#pragma unroll
    for (int i = 0; i < NRC; ++i) {
      const double q = L.qv[g + 8 * i];
#pragma unroll
      for (int k = 0; k < RS; ++k) acc[k] = fma(greg[i][k], q, acc[k]);
    }
#pragma unroll
    for (int i = 0; i < NLC; ++i) {
      const double q = L.qv[g + 8 * (NRC + i)];
#pragma unroll
      for (int k = 0; k < RS; ++k) acc[k] = fma(L.glds[i * RS + k][tid], q, acc[k]);
    }
    // column slots beyond the resident ones: from L2, SU slots (16 loads per lane) in flight --
    // one slot at a time the loop is a chain of L2 round trips (measured: 18 us per step at
    // n = 298).  A slot past the last one is clamped to column n - 1 and meets q = 0.
    constexpr int SU = 16 / RS;
    for (int i = NRC + NLC; i < ncol8; i += SU) {
      double gl[SU][RS], qs[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int c = g + 8 * (i + u);  // (< LP_NMAX + 8 + 8 SU... clamped for q below)
        qs[u] = L.qv[min(c, LP_NMAX + 15)];
        const double *row = Gk + (long)min(c, n - 1) * ld;
#pragma unroll
        for (int k = 0; k < RS; ++k) gl[u][k] = row[rows[k]];
      }
#pragma unroll
      for (int u = 0; u < SU; ++u)
#pragma unroll
        for (int k = 0; k < RS; ++k) acc[k] = fma(gl[u][k], qs[u], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < RS; ++k)
      if (lane + 64 * k < LP_NMAX) L.psum[g][lane + 64 * k] = acc[k];
    __syncthreads();
  };
  auto row_w = [&](int r) {  // fixed order over the eight column groups
    return ((L.psum[0][r] + L.psum[1][r]) + (L.psum[2][r] + L.psum[3][r])) +
           ((L.psum[4][r] + L.psum[5][r]) + (L.psum[6][r] + L.psum[7][r]));
  };

  // start vector: G * ones (a few power-like steps come for free in the Krylov space)
  matvec();
  double wr[RC];
  {
    double p = 0.0;
    if (tid < 256) {
#pragma unroll
      for (int rc = 0; rc < RC; ++rc) {
        const int r = tid + 256 * rc;
        wr[rc] = r < n ? row_w(r) : 0.0;
        p = fma(wr[rc], wr[rc], p);
      }
    }
    const double nrm = sqrt(lp_vec_sum(p, L.red, 0));
    if (tid < 256) {
#pragma unroll
      for (int rc = 0; rc < RC; ++rc) {
        const int r = tid + 256 * rc;
        if (r < n) L.qv[r] = nrm > 0.0 ? wr[rc] / nrm : (r == 0 ? 1.0 : 0.0);
      }
    }
    __syncthreads();
  }
  double theta = 0.0, resid = 0.0;
  int restarts = 0, steps = 0;
  const int mmax = min(LP_MAXS, n);
  lap(0);
  for (;; ++restarts) {
    // q_0 = L.qv (unit norm), q_{-1} = 0
    if (tid < 256) {
#pragma unroll
      for (int rc = 0; rc < RC; ++rc) {
        const int r = tid + 256 * rc;
        if (r < n) {
          L.qp[r] = 0.0;
          Vk[r] = L.qv[r];
        }
      }
    }
    __syncthreads();
    double beta_prev = 0.0;
    int m = 0, m_checked = 0;  // m_checked: rows of the last check of THIS recurrence (0: none)
    int next_check = 8;
    double e_checked = 0.0;
    bool solved = false;
    for (int j = 0; j < mmax; ++j) {
      matvec();
      lap(1);
      // w = G q_j - beta_{j-1} q_{j-1};  alpha_j = q_j . w
      double a_loc = 0.0;
      if (tid < 256) {
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
          const int r = tid + 256 * rc;
          wr[rc] = r < n ? fma(-beta_prev, L.qp[r], row_w(r)) : 0.0;
          a_loc = fma(r < n ? L.qv[r] : 0.0, wr[rc], a_loc);
        }
      }
      const double aj = lp_vec_sum(a_loc, L.red, 0);
      // w -= alpha_j q_j;  beta_j = |w|
      double b_loc = 0.0;
      if (tid < 256) {
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
          const int r = tid + 256 * rc;
          if (r < n) wr[rc] = fma(-aj, L.qv[r], wr[rc]);
          b_loc = fma(wr[rc], wr[rc], b_loc);
        }
      }
      const double bj = sqrt(lp_vec_sum(b_loc, L.red, 1));
      if (tid == 0) L.alpha[j] = aj, L.beta[j] = bj;
      m = j + 1;
      const bool invariant = bj <= 1e-300 || bj <= 1e-15 * fabs(aj);
      if (!invariant && j + 1 < mmax && tid < 256) {
        const double ib = 1.0 / bj;
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
          const int r = tid + 256 * rc;
          if (r < n) {
            const double qn = wr[rc] * ib;
            L.qp[r] = L.qv[r];
            L.qv[r] = qn;
            Vk[(long)(j + 1) * ld + r] = qn;
          }
        }
      }
      beta_prev = bj;
      __syncthreads();
      lap(2);
      if (invariant || m == mmax || m == next_check) {
        if (tid < 64) {
          // (the recurrence form needs b_i > 0 and a decaying vector: breakdown and the forced
          // stop at mmax take the LDL^T solver)
          const double th = (invariant || m == mmax)
                                ? tridiag_top_lean(L.alpha, L.beta, m, lane, L.ws)
                                : tridiag_top_fast(L.alpha, L.beta, m, lane, L.ws,
                                                   theta > 0.0 && m_checked > 0 ? theta : -1e300,
                                                   tph ? tph + 5 : nullptr);
          if (lane == 0) L.red[0] = th;
        }
        __syncthreads();
        theta = L.red[0];
        resid = fabs(bj * L.ws.x[m - 1]);
        solved = invariant || m >= n || resid <= tol * fabs(theta);
        {
          const double e = fabs(theta) > 0.0 ? resid / fabs(theta) : 0.0;
          next_check = lp_next_check(m, e, m_checked, e_checked, tol);
          m_checked = m, e_checked = e;
        }
        __syncthreads();  // (red[0] is reused by the next reduction)
        lap(3);
        if (solved || m == mmax) break;
      }
    }
    steps += m;
    // y = V s, normalised, into qv; then the true residual |G y - theta y| against G itself.  A
    // vector from the recurrence form that fails the test is replaced by the LDL^T solver's.
    bool robust = m == mmax || L.beta[m - 1] <= 1e-15 * fabs(L.alpha[m - 1]);
    bool good = false;
    double rtrue = 0.0, th2 = 0.0;
    for (;;) {
      double yr[RC], p = 0.0;
      if (tid < 256) {
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
          const int r = tid + 256 * rc;
          double acc = 0.0;
          if (r < n) {
#pragma unroll 8
            for (int i = 0; i < m; ++i) acc = fma(L.ws.x[i], Vk[(long)i * ld + r], acc);
          }
          yr[rc] = acc;
          p = fma(acc, acc, p);
        }
      }
      const double ny = sqrt(lp_vec_sum(p, L.red, 0));
      if (tid < 256) {
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
          const int r = tid + 256 * rc;
          if (r < n) L.qv[r] = ny > 0.0 ? yr[rc] / ny : (r == 0 ? 1.0 : 0.0);
        }
      }
      __syncthreads();
      matvec();
      double t_loc = 0.0;
      if (tid < 256) {
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
          const int r = tid + 256 * rc;
          wr[rc] = r < n ? row_w(r) : 0.0;
          t_loc = fma(r < n ? L.qv[r] : 0.0, wr[rc], t_loc);
        }
      }
      th2 = lp_vec_sum(t_loc, L.red, 1);
      double r_loc = 0.0;
      if (tid < 256) {
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
          const int r = tid + 256 * rc;
          const double d = r < n ? fma(-th2, L.qv[r], wr[rc]) : 0.0;
          r_loc = fma(d, d, r_loc);
        }
      }
      rtrue = sqrt(lp_vec_sum(r_loc, L.red, 0));
      good = rtrue <= LP_VERIFY_TOL * fabs(th2) || !(th2 > 0.0);
      __syncthreads();
      if (good || robust) break;
      if (tid < 64) (void)tridiag_top_lean(L.alpha, L.beta, m, lane, L.ws);
      robust = true;
      __syncthreads();
    }
    theta = th2;
    if (!solved || !good) resid = rtrue;
    lap(4);
    if ((solved && good) || restarts + 1 >= max_restart) break;
  }
  // the eigenvector is in qv (unit norm)
  for (int e = tid; e < ld; e += LP_NT) v[e] = e < n ? L.qv[e] : 0.0;
  if (tid == 0 && info3) info3[0] = theta, info3[1] = resid, info3[2] = (double)restarts;
  *steps_out = steps;
}

__global__ __launch_bounds__(LP_NT) void lanczos_plain_kernel(
    const double *__restrict__ G, const long *__restrict__ g_off, const long *__restrict__ ld_,
    const long *__restrict__ n_, double *__restrict__ Q, const long *__restrict__ q_off,
    double *__restrict__ vout, const long *__restrict__ v_off, int max_restart, double tol,
    double *__restrict__ info, double *__restrict__ dbg) {
  extern __shared__ __align__(16) double lz_dyn[];  // SmallWork, the squaring buffers, or PlainLds
  const int k = blockIdx.x;
  const int n = (int)n_[k], ld = (int)ld_[k];
  if (n < 1) return;
  const unsigned long long t_dbg = dbg ? wall_clock64() : 0ull;
  const double *Gk = G + g_off[k];
  double *v = vout + v_off[k];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int steps = 0;
  if (n <= LANCZOS_M) {
    SmallWork &sw = *reinterpret_cast<SmallWork *>(lz_dyn);
    double *z8 = reinterpret_cast<double *>(&sw);
    for (int i = tid; i < (int)(sizeof(SmallWork) / sizeof(double)); i += LP_NT) z8[i] = 0.0;
    __syncthreads();
    for (int c = wave; c < n; c += LP_NW)
      if (lane < n) sw.G[c][lane] = Gk[(long)c * ld + lane];
    __syncthreads();
    eig_small_power<LP_NW>(n, ld, sw, v, info ? info + 3 * k : nullptr);
    steps = -1;
  } else if (n <= PW_N) {
    eig_mid_power<LP_NW>(Gk, n, ld, lz_dyn, v, info ? info + 3 * k : nullptr);
    steps = -2;
  } else {
    PlainLds &L = *reinterpret_cast<PlainLds *>(lz_dyn);
    double *Vk = Q + q_off[k];
    double *tph = dbg ? dbg + 2 * (long)gridDim.x + 8 * (long)k : nullptr;
    if (tph && tid == 0)
      for (int e = 0; e < 8; ++e) tph[e] = 0.0;
    if (n <= 256)
      lanczos_plain<1>(Gk, n, ld, Vk, v, L, max_restart, tol, info ? info + 3 * k : nullptr, &steps,
                       tph);
    else
      lanczos_plain<2>(Gk, n, ld, Vk, v, L, max_restart, tol, info ? info + 3 * k : nullptr, &steps,
                       tph);
  }
  if (dbg && tid == 0)
    dbg[2 * k] = (double)steps, dbg[2 * k + 1] = (double)(wall_clock64() - t_dbg) * 0.01;
}

// ------------------------------------------------------------------------------------
// u = Xp v, then normalised and appended to U.     grid (ceil(Nz/4), nw), block (64,4)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xv_kernel(const double *__restrict__ Xp,
                                                 const long *__restrict__ D, int nw, int Nz,
                                                 const double *__restrict__ v,
                                                 double *__restrict__ u) {
  const int k = blockIdx.y;
  const int z = blockIdx.x * 4 + threadIdx.y;
  if (z >= Nz || PCA_SLOT_DONE(k)) return;
  const int ld = (int)DSC(DF_LD, k), n = (int)DSC(DF_N, k);
  const double *row = Xp + DSC(DF_XP, k) + (long)z * ld;
  const double *vk = v + DSC(DF_C, k);
  double acc = 0.0;
  for (int j = threadIdx.x; j < n; j += 64) acc = fma(row[j], vk[j], acc);
  acc = wave_sum_d(acc);
  if (threadIdx.x == 0) u[(long)k * Nz + z] = acc;
}

// w_k[q] = u_k . U[:, q] (q < T_k) and |u_k|^2 of the not yet normalised u_k, as partial sums over
// UW_SLICES slices of z: grid (UW_SLICES, nw), block 256; lanes over q, waves over the rows of a
// slice.  part[(k * UW_SLICES + b) * (PCA_CAP + 1) + q], entry PCA_CAP = the slice's sum of squares.
constexpr int UW_SLICES = 32;
__global__ __launch_bounds__(256) void uw_partial_kernel(const double *__restrict__ u, int Nz,
                                                         const long *__restrict__ D, int nw,
                                                         const double *__restrict__ U,
                                                         double *__restrict__ part) {
  __shared__ double wred[4][PCA_CAP + 1];
  const int k = blockIdx.y, b = blockIdx.x;
  if (PCA_SLOT_DONE(k)) return;
  const int T = (int)DSC(DF_T, k);
  const double *uk = u + (long)k * Nz;
  const double *Ua = U + (long)DSC(DF_AREA, k) * Nz * PCA_CAP;
  const int zs = (Nz + UW_SLICES - 1) / UW_SLICES;
  const int z0 = b * zs, z1 = min(Nz, z0 + zs);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double a = 0.0, s2 = 0.0;
  for (int z = z0 + wave; z < z1; z += 4) {
    const double uz = uk[z];
    s2 = fma(uz, uz, s2);
    if (lane < T) a = fma(uz, Ua[(long)z * PCA_CAP + lane], a);
  }
  wred[wave][lane] = a;
  if (lane == 0) wred[wave][PCA_CAP] = s2;
  __syncthreads();
  if (threadIdx.x <= PCA_CAP) {
    const int q = threadIdx.x;
    part[((long)k * UW_SLICES + b) * (PCA_CAP + 1) + q] =
        ((wred[0][q] + wred[1][q]) + wred[2][q]) + wred[3][q];
  }
}

#ifdef PCA_EXP_DUMMY
// experiment: what one more dependent launch costs a tail iteration
__global__ void pca_dummy_kernel(const long *__restrict__ D, int nw, double *__restrict__ out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (double)D[blockIdx.x % nw];
}
#endif

// normalise u_k, store it as column T_k of U, and w_k[q] = u_k . U[:, q] for q < T_k
__global__ __launch_bounds__(1024) void normalize_kernel(double *__restrict__ u, int Nz,
                                                         const long *__restrict__ D, int nw,
                                                         double *__restrict__ U,
                                                         const double *__restrict__ part,
                                                         double *__restrict__ wq) {
  __shared__ double s_inv;
  const int k = blockIdx.x;
  if (PCA_SLOT_DONE(k)) return;
  const int T = (int)DSC(DF_T, k);
  double *uk = u + (long)k * Nz;
  double *Ua = U + (long)DSC(DF_AREA, k) * Nz * PCA_CAP;
  const double *pk = part + (long)k * UW_SLICES * (PCA_CAP + 1);
  double acc = 0.0;
  if (threadIdx.x <= PCA_CAP) {  // fixed-order sums over the slices
#pragma unroll 8
    for (int b = 0; b < UW_SLICES; ++b) acc += pk[(long)b * (PCA_CAP + 1) + threadIdx.x];
    if (threadIdx.x == PCA_CAP) s_inv = acc > 0.0 ? 1.0 / sqrt(acc) : 0.0;
  }
  __syncthreads();
  const double inv = s_inv;
  if ((int)threadIdx.x < T) wq[(long)k * PCA_CAP + threadIdx.x] = acc * inv;
  for (int z = threadIdx.x; z < Nz; z += 1024) {
    const double x = uk[z] * inv;
    uk[z] = x;
    Ua[(long)z * PCA_CAP + T] = x;
  }
}

// ------------------------------------------------------------------------------------
// deflation of whole areas in coefficient form.
//   dot    : cpart[zs][i] = sum_{z in slice zs} u_k[z] X[z, spx[i]]
//   finish : c_i = sum_zs cpart - sum_q w[q] C[q][i] (= u^T F_{t-1,i});  C[T][i] = c_i;
//            test[spx[i]] -= c_i^2 / Nz          (|F_t|^2 = |F_{t-1}|^2 - c^2, |u| = 1)
// grid (ceil(nsmax/256), ZS, nw) / (ceil(nsmax/256), nw); 256 lanes over the area's list
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void deflate_dot_kernel(const float *__restrict__ F, int Nz,
                                                          long S, const int *__restrict__ spx,
                                                          const long *__restrict__ D, int nw,
                                                          const double *__restrict__ u, int zper,
                                                          double *__restrict__ cpart, long ntot) {
  const int k = blockIdx.z;
  const int ns = (int)DSC(DF_NS, k);
  const int li = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= ns || PCA_SLOT_DONE(k)) return;
  const bool live = li < ns;
  const long col = spx[DSC(DF_LIST0, k) + (live ? li : ns - 1)];
  const double *uk = u + (long)k * Nz;
  const int z0 = blockIdx.y * zper, z1 = min(Nz, z0 + zper);
  double acc = 0.0;
#pragma unroll 4
  for (int z = z0; z < z1; ++z) acc = fma(uk[z], (double)F[(long)z * S + col], acc);
  if (live) cpart[(long)blockIdx.y * ntot + DSC(DF_CBASE, k) + li] = acc;
}

// The same dot products with the block -> memory mapping of the cube instead of the areas':
// 256 consecutive spaxels of the flattened (Ny, Nx) plane per block, each lane looking up its
// area (area_of), the area's slot in this iteration's work list (kidx, -1: not iterating) and
// its list position (pos_of).  An area row of 100 float32 cuts the 128-byte lines at both ends,
// and with one block set per area every cut line is fetched twice (measured: 4.7 TB/s of
// algorithmic bytes with 100-wide areas against 6.3 TB/s with 128-wide ones); here every line
// is fetched once, whole.  Same z slices and summation order as deflate_dot_kernel: identical partial sums.
// Used while the iterating areas cover most of the field.  grid (ceil(S/256), ZS)
__global__ __launch_bounds__(256) void deflate_dot_rows_kernel(
    const float *__restrict__ F, int Nz, long S, const int *__restrict__ area_of,
    const int *__restrict__ pos_of, const int *__restrict__ kidx, const long *__restrict__ D,
    int nw, const double *__restrict__ u, int zper, double *__restrict__ cpart, long ntot) {
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  int k = -1;
  if (s < S) {
    const int a = area_of[s];
    if (a >= 0) k = kidx[a];
    if (k >= 0 && PCA_SLOT_DONE(k)) k = -1;
  }
  if (!__any(k >= 0)) return;  // nothing of this wave's 64 spaxels iterates (no block barrier used)
  const long sc = k >= 0 ? s : (long)blockIdx.x * 256;  // idle lanes re-read the block's first spaxel
  const double *uk = u + (long)(k >= 0 ? k : 0) * Nz;  // per-lane load; lanes of one area share it
  const int z0 = blockIdx.y * zper, z1 = min(Nz, z0 + zper);
  double acc = 0.0;
#pragma unroll 4
  for (int z = z0; z < z1; ++z) acc = fma(uk[z], (double)F[(long)z * S + sc], acc);
  if (k >= 0)
    cpart[(long)blockIdx.y * ntot + DSC(DF_CBASE, k) + (pos_of[s] - DSC(DF_LIST0, k))] = acc;
}

// area_of[s] / pos_of[s]: the area (index) holding spaxel s and its position in the
// concatenated lists; area_of is preset to -1.   grid (ceil(nsmax/256), na)
__global__ __launch_bounds__(256) void invert_lists_kernel(const int *__restrict__ spx,
                                                           const long *__restrict__ spx_off,
                                                           int *__restrict__ area_of,
                                                           int *__restrict__ pos_of) {
  const int a = blockIdx.y;
  const long o0 = spx_off[a];
  const long i = o0 + (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= spx_off[a + 1]) return;
  const int s = spx[i];
  area_of[s] = a;
  pos_of[s] = (int)i;
}

__global__ __launch_bounds__(256) void deflate_finish_kernel(const int *__restrict__ spx,
                                                             const long *__restrict__ D, int nw,
                                                             int nzs, int Nz, long cb_tot,
                                                             const double *__restrict__ cpart,
                                                             const double *__restrict__ wq,
                                                             double *__restrict__ C, long ntot,
                                                             double *__restrict__ test) {
  const int k = blockIdx.y;
  const int ns = (int)DSC(DF_NS, k);
  const int li = blockIdx.x * 256 + threadIdx.x;
  if (li >= ns || PCA_SLOT_DONE(k)) return;
  const int T = (int)DSC(DF_T, k);
  const long pos = DSC(DF_LIST0, k) + li;
  const long ci = DSC(DF_CBASE, k) + li;
  double c = 0.0;
  {  // loads of a batch are independent (L2 latency), the sums keep their order
    int q = 0;
    for (; q + 8 <= nzs; q += 8) {
      double v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = cpart[(long)(q + e) * cb_tot + ci];
#pragma unroll
      for (int e = 0; e < 8; ++e) c += v[e];
    }
    for (; q < nzs; ++q) c += cpart[(long)q * cb_tot + ci];
  }
  const double *w = wq + (long)k * PCA_CAP;
  {
    int q = 0;
    for (; q + 8 <= T; q += 8) {
      double v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = C[(long)(q + e) * ntot + pos];
#pragma unroll
      for (int e = 0; e < 8; ++e) c = fma(-w[q + e], v[e], c);
    }
    for (; q < T; ++q) c = fma(-w[q], C[(long)q * ntot + pos], c);
  }
  C[(long)T * ntot + pos] = c;
  const long sp = spx[pos];
  test[sp] = test[sp] - c * c / (double)Nz;
}

// ------------------------------------------------------------------------------------
// flush: F[z, s] = X[z, s] - sum_{q < T_a} U[z][q] C[q][i]  (X may alias F; one float32 rounding)
// 1-D grid over (spaxel chunk, channel block, area), see the decoding below; thread = one spaxel x
// 32 channels, U rows in LDS
// ------------------------------------------------------------------------------------
constexpr int FLUSH_ZB = 32;  // channels per block
__global__ __launch_bounds__(256, 4) void flush_kernel(const float *X, float *F, int Nz, long S,
                                                    const int *__restrict__ spx,
                                                    const long *__restrict__ FD, int nf,
                                                    const double *__restrict__ U,
                                                    const double *__restrict__ C, long ntot,
                                                    int nxb, int nzb, int out_nx, long out_py,
                                                    long out_pz) {
  // out_nx > 0: F is a box inside a larger cube (the halo-extended tile of the tiled path):
  // spaxel col = y * out_nx + x of the (Nz, S) input goes to F[z * out_pz + y * out_py + x]
  // FD: [4][nf] = area, list0, ns, T
  __shared__ double Us[FLUSH_ZB][PCA_CAP];  // this block's rows of U (zero beyond T / Nz)
  // 1-D grid decoded so that the blocks of neighbouring areas for the same rows and channels
  // are 8 workgroup ids apart: same XCD (id mod 8), a few dispatches apart in time.  An area
  // row of 100 float32 cuts the 128-byte lines at both ends; when the two halves of a cut line
  // are written while the line is still in that XCD's L2 it goes to memory once, whole,
  // instead of as two partial writes (a read-modify-write each).
  const long id = blockIdx.x;
  const int low = (int)(id & 7);
  const long rest = id >> 3;
  const int k = (int)(rest % nf);
  const long cz = (rest / nf) * 8 + low;  // index over (spaxel chunk, channel block)
  if (cz >= (long)nxb * nzb) return;
  const int bx = (int)(cz % nxb), by = (int)(cz / nxb);
  const int ns = (int)FD[(long)2 * nf + k], T = (int)FD[(long)3 * nf + k];
  const int li = bx * 256 + threadIdx.x;
  if (bx * 256 >= ns) return;
  const bool live = li < ns;
  const long pos = FD[(long)1 * nf + k] + (live ? li : ns - 1);
  const long col = spx[pos];
  const long fcol = out_nx > 0 ? (col / out_nx) * out_py + (col % out_nx) : col;
  const long fS = out_nx > 0 ? out_pz : S;
  const double *Ua = U + FD[k] * (long)Nz * PCA_CAP;
  const int z0 = by * FLUSH_ZB;
  for (int i = threadIdx.x; i < FLUSH_ZB * PCA_CAP; i += 256) {
    const int r = i / PCA_CAP, q = i - r * PCA_CAP;
    Us[r][q] = (z0 + r < Nz && q < T) ? Ua[(long)(z0 + r) * PCA_CAP + q] : 0.0;
  }
  __syncthreads();
  // Register budget: 4 waves per SIMD (128 VGPRs) keep enough loads in flight for this pass to
  // stream; left alone the compiler hoists all 128 LDS reads of a coefficient batch (250 VGPRs,
  // 2 waves per SIMD, 2.9 TB/s).  Hence two rows of U per scheduling group, and the X column
  // loaded after the coefficient loop.
  double acc[FLUSH_ZB];
#pragma unroll
  for (int r = 0; r < FLUSH_ZB; ++r) acc[r] = 0.0;
  for (int q0 = 0; q0 < T; q0 += 8) {
    double c[8];  // independent loads; entries beyond T multiply zero rows of Us
#pragma unroll
    for (int e = 0; e < 8; ++e) c[e] = q0 + e < T ? C[(long)(q0 + e) * ntot + pos] : 0.0;
#pragma unroll
    for (int r = 0; r < FLUSH_ZB; r += 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        acc[r] = fma(Us[r][q0 + e], c[e], acc[r]);
        acc[r + 1] = fma(Us[r + 1][q0 + e], c[e], acc[r + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float xv[FLUSH_ZB];
#pragma unroll
  for (int r = 0; r < FLUSH_ZB; ++r) xv[r] = X[(long)min(z0 + r, Nz - 1) * S + col];
  if (live) {
#pragma unroll
    for (int r = 0; r < FLUSH_ZB; ++r) {
      const int z = z0 + r;
      if (z < Nz) F[(long)z * fS + fcol] = (float)((double)xv[r] - acc[r]);
    }
  }
}

// The same pass with the block -> memory mapping of the cube instead of the areas' (round 4): a
// block is 256 consecutive spaxels of the flattened (Ny, Nx) plane x 32 channels, every lane looks
// up its area (area_of), whether that area is written now and with how many vectors (aT[a] >= 0),
// and its list position (pos_of).  An area row of 100 float32 cuts the 128-byte lines at both
// ends: with one block set per area (flush_kernel) cut lines are fetched twice and written as two
// partial lines unless the neighbour's block happens to run while the line is still in L2 --
// 15.3 GB moved for 10.6 algorithmic at 3681 x 600 x 600 (profiles/r03_pmc_fetch_write.json).  Here
// every line of X is read once and every line of F written once, whole.
// The rows of U of every area present in the block sit in LDS, one table per RUN of equal areas
// along the block's 256 spaxels (a row of the field crosses an area border every ~100 spaxels:
// three or four runs), laid out [vector][channel] with the block's largest vector count as the
// common depth, zero beyond an area's own count; lanes of different areas read different tables
// (two or three distinct addresses per LDS instruction).  Blocks with more runs, or deeper tables,
// than fit FLUSH_ROWS_LDS take several rounds.  Same sums in the same order as flush_kernel:
// identical bits.
constexpr int FLUSH_ROWS_LDS = 48 * 1024;
__global__ __launch_bounds__(256, 3) void flush_rows_kernel(
    const float *X, float *F, int Nz, long S, const int *__restrict__ area_of,
    const int *__restrict__ pos_of, const int *__restrict__ aT, const double *__restrict__ U,
    const double *__restrict__ C, long ntot, int out_nx, long out_py, long out_pz) {
  extern __shared__ __align__(16) double fr_us[];  // [slot][q][FLUSH_ZB]
  __shared__ int fr_wave_heads[4], fr_slot_area[256], fr_tmax;
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  const int z0 = blockIdx.y * FLUSH_ZB;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int a = -1, T = -1;
  if (s < S) {
    a = area_of[s];
    if (a >= 0) T = aT[a];
    if (T < 0) a = -1;
  }
  if (!__syncthreads_or(a >= 0)) return;  // nothing of this block is written now
  // runs of equal areas: a lane opens one when its area differs from its left neighbour's
  int left = __shfl_up(a, 1);
  if (lane == 0) left = -2;  // (a wave's first lane always opens a run: no look across waves)
  const bool head = a >= 0 && a != left;
  const unsigned long long hb = __ballot(head);
  if (lane == 0) fr_wave_heads[wv] = __popcll(hb);
  if (threadIdx.x == 0) fr_tmax = 0;
  __syncthreads();
  int slot = __popcll(hb & ((2ull << lane) - 1ull)) - 1;  // run index inside the wave
  for (int w = 0; w < wv; ++w) slot += fr_wave_heads[w];
  const int nslots = fr_wave_heads[0] + fr_wave_heads[1] + fr_wave_heads[2] + fr_wave_heads[3];
  if (head) fr_slot_area[slot] = a;
  if (a >= 0) atomicMax(&fr_tmax, T);
  __syncthreads();
  // common table depth: a power of two >= 8 (index arithmetic of the fill by shifts)
  int lt = 3;
  while ((1 << lt) < fr_tmax) ++lt;
  const int Tp = fr_tmax > 0 ? 1 << lt : 0;
  const long pos = a >= 0 ? pos_of[s] : 0;
  const long fcol = out_nx > 0 ? (s / out_nx) * out_py + (s % out_nx) : s;
  const long fS = out_nx > 0 ? out_pz : S;
  double acc[FLUSH_ZB];
#pragma unroll
  for (int r = 0; r < FLUSH_ZB; ++r) acc[r] = 0.0;
  if (Tp > 0) {
    const int per_round = max(1, FLUSH_ROWS_LDS / (Tp * FLUSH_ZB * (int)sizeof(double)));
    for (int base = 0; base < nslots; base += per_round) {
      const int nhere = min(per_round, nslots - base);
      // tables of the runs base .. base + nhere - 1: Us[slot][r][q] = U[area][z0 + r][q], q fastest
      // on both sides (rows of U are PCA_CAP doubles: 8-double pieces, coalesced; no LDS conflicts)
      for (int i = threadIdx.x; i < (nhere * FLUSH_ZB) << lt; i += 256) {
        const int q = i & (Tp - 1), r = (i >> lt) & (FLUSH_ZB - 1), sl = i >> (lt + 5);
        const int ar = fr_slot_area[base + sl];
        fr_us[i] = (z0 + r < Nz && q < aT[ar])
                       ? U[((long)ar * Nz + z0 + r) * PCA_CAP + q] : 0.0;
      }
      __syncthreads();
      if (a >= 0 && slot >= base && slot < base + nhere) {
        const double *us = fr_us + ((long)(slot - base) * FLUSH_ZB << lt);
        for (int q0 = 0; q0 < T; q0 += 8) {
          double c[8];  // independent loads; entries beyond T multiply zero entries of the table
#pragma unroll
          for (int e = 0; e < 8; ++e) c[e] = q0 + e < T ? C[(long)(q0 + e) * ntot + pos] : 0.0;
          const double *uq = us + q0;
#pragma unroll
          for (int r = 0; r < FLUSH_ZB; r += 2) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              const double2 ua = *reinterpret_cast<const double2 *>(uq + ((long)r << lt) + e);
              const double2 ub = *reinterpret_cast<const double2 *>(uq + ((long)(r + 1) << lt) + e);
              acc[r] = fma(ua.x, c[e], acc[r]);
              acc[r] = fma(ua.y, c[e + 1], acc[r]);
              acc[r + 1] = fma(ub.x, c[e], acc[r + 1]);
              acc[r + 1] = fma(ub.y, c[e + 1], acc[r + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      __syncthreads();
    }
  }
  if (a < 0) return;
  float xv[FLUSH_ZB];
#pragma unroll
  for (int r = 0; r < FLUSH_ZB; ++r) xv[r] = X[(long)min(z0 + r, Nz - 1) * S + s];
#pragma unroll
  for (int r = 0; r < FLUSH_ZB; ++r) {
    const int z = z0 + r;
    if (z < Nz) F[(long)z * fS + fcol] = (float)((double)xv[r] - acc[r]);
  }
}

// ------------------------------------------------------------------------------------
// host helpers
// ------------------------------------------------------------------------------------
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  int reserve(origin_ctx *ctx, size_t want) {
    if (want <= bytes) return ORIGIN_OK;
    if (p) {
      ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
      ORIGIN_HIP(hipFree(p));
      p = nullptr;
      bytes = 0;
    }
    const size_t n = want + want / 4 + 4096;
    ORIGIN_HIP(hipMalloc(&p, n));
    bytes = n;
    return ORIGIN_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  ~DevBuf() { release(); }
};

// pinned host memory that lives as long as the workspace (hipHostMalloc / hipHostFree cost
// hundreds of microseconds and the free waits for the device)
struct HostBuf {
  void *p = nullptr;
  size_t bytes = 0;
  unsigned flags = 0;
  int reserve(size_t want, unsigned fl) {
    if (want <= bytes && fl == flags) return ORIGIN_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    bytes = 0;
    const size_t n = want + want / 2 + 256;
    ORIGIN_HIP(hipHostMalloc(&p, n, fl));
    bytes = n;
    flags = fl;
    return ORIGIN_OK;
  }
  ~HostBuf() {
    if (p) (void)hipHostFree(p);
  }
};

struct PcaWorkspace {
  DevBuf b[21];
  HostBuf h_nnb, h_stage;
};

// launches lanczos_kernel with the basis in LDS when the largest matrix allows it
int eig_launch(origin_ctx *ctx, int nmat, long ldmax, const double *d_G, const long *d_g_off,
               const long *d_ld, const long *d_n, double *d_q, const long *d_q_off, double *d_v,
               const long *d_v_off, double *d_info, const double *d_slab = nullptr,
               long slab_stride = 0, int ksplit = 0, double *d_dbg = nullptr) {
  static OriginPerDeviceOnce attr_once;
  // dynamic + static LDS must stay within the 160 KB of a CU: ask for exactly what is used
  const int dyn_max = (int)std::max(PW_BYTES, (size_t)(LANCZOS_M + 2) * LANCZOS_QLDS_LD * sizeof(double));
  ORIGIN_ONCE_PER_DEVICE(ctx, attr_once,
                         ORIGIN_HIP(hipFuncSetAttribute((const void *)lanczos_kernel<true>,
                                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        dyn_max));
                         ORIGIN_HIP(hipFuncSetAttribute((const void *)lanczos_kernel<false>,
                                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        dyn_max)));
  // Above PW_N columns: plain Lanczos with the matrix resident on the CU (lanczos_plain_kernel;
  // ORIGIN_PCA_EIG=cgs2 keeps the round-2 kernel for A/B runs).  A launch whose largest matrix
  // exceeds LP_NMAX columns goes to the round-2 kernel as a whole.
  static const bool cgs2 = getenv("ORIGIN_PCA_EIG") && !strcmp(getenv("ORIGIN_PCA_EIG"), "cgs2");
  if (!d_slab && ldmax > PW_N && ldmax <= LP_NMAX && !cgs2) {
    static OriginPerDeviceOnce attr_plain;
    const size_t lds = std::max(std::max(PW_BYTES, sizeof(SmallWork)), LP_BYTES);
    ORIGIN_ONCE_PER_DEVICE(ctx, attr_plain,
                           ORIGIN_HIP(hipFuncSetAttribute((const void *)lanczos_plain_kernel,
                                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                                          (int)lds)));
    hipLaunchKernelGGL(lanczos_plain_kernel, dim3(nmat), dim3(LP_NT), lds, ctx->stream, d_G, d_g_off,
                       d_ld, d_n, d_q, d_q_off, d_v, d_v_off, 8, 1e-14, d_info, d_dbg);
    ORIGIN_LAUNCH_CHECK();
    return ORIGIN_OK;
  }
  // matrices of up to PW_N columns are solved by repeated squaring in LDS (PW_BYTES of it)
  const bool mid = ldmax > LANCZOS_M;
  const bool qlds = ldmax <= LANCZOS_QLDS_LD;
  if (qlds) {
    size_t lds = std::max(sizeof(SmallWork),
                          (size_t)(LANCZOS_M + 2) * (size_t)ldmax * sizeof(double));
    if (mid) lds = std::max(lds, PW_BYTES);
    hipLaunchKernelGGL(lanczos_kernel<true>, dim3(nmat), dim3(1024), lds, ctx->stream, d_G, d_g_off,
                       d_ld, d_n, d_q, d_q_off, d_v, d_v_off, 60, 1e-14, d_info, d_slab, slab_stride,
                       ksplit, mid ? PW_N : 0, d_dbg);
  } else {
    hipLaunchKernelGGL(lanczos_kernel<false>, dim3(nmat), dim3(1024),
                       std::max(sizeof(SmallWork), PW_BYTES), ctx->stream, d_G, d_g_off, d_ld, d_n,
                       d_q, d_q_off, d_v, d_v_off, 60, 1e-14, d_info, d_slab, slab_stride, ksplit,
                       PW_N, d_dbg);
  }
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

int gram_launch(origin_ctx *ctx, const double *d_Xp, const long *d_xp_off, const long *d_ld, int Nz,
                int ntiles, const int *d_ti, const int *d_tj, const int *d_ta, long g_total,
                double *d_G, const long *d_g_off, bool skip_reduce = false,
                const double **slab_out = nullptr, int *ksplit_out = nullptr,
                const long *d_n = nullptr) {
  // K-split so that small problems still put >= ~8 waves on every CU
  int ksplit = (int)(((long)ctx->num_cu * 8 + ntiles - 1) / ntiles);
  if (ksplit < 1) ksplit = 1;
  if (ksplit > 32) ksplit = 32;
  if (ksplit > Nz / 64) ksplit = Nz / 64 > 0 ? Nz / 64 : 1;
  void *scr = nullptr;
  int rc = origin_scratch(ctx, (size_t)ksplit * g_total * sizeof(double), &scr);
  if (rc) return rc;
  ProfScope ps(ctx, K_PCA_GRAM, 2);
  hipLaunchKernelGGL(gram_kernel, dim3(ntiles, ksplit), dim3(64), 0, ctx->stream, d_Xp, d_xp_off,
                     d_ld, d_ti, d_tj, d_ta, Nz, ksplit, (double *)scr, d_g_off, g_total, d_n);
  if (slab_out) *slab_out = (const double *)scr;
  if (ksplit_out) *ksplit_out = ksplit;
  // the small-matrix eigen-solver sums the slabs itself when every matrix of the launch is small
  if (!skip_reduce)
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(ntiles), dim3(256), 0, ctx->stream,
                       (const double *)scr, g_total, ksplit, d_ld, d_ti, d_tj, d_ta, d_G, d_g_off,
                       d_n);
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

}  // namespace

extern "C" {

// Stand-alone Gram product (used by tests): nmat matrices, offsets/ld as int64 arrays.
int origin_pca_gram(origin_ctx *ctx, const double *d_Xp, const long *d_xp_off, const long *d_ld,
                    int Nz, int ntiles, const int *d_tile_i, const int *d_tile_j,
                    const int *d_tile_a, long g_total, double *d_G, const long *d_g_off) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_Xp && d_xp_off && d_ld && d_tile_i && d_tile_j && d_tile_a && d_G &&
                       d_g_off && Nz > 0 && ntiles > 0 && g_total > 0,
                   "bad arguments");
  return gram_launch(ctx, d_Xp, d_xp_off, d_ld, Nz, ntiles, d_tile_i, d_tile_j, d_tile_a, g_total,
                     d_G, d_g_off);
}

// Stand-alone leading eigenvector of nmat symmetric PSD matrices (used by tests).
// d_info (may be NULL): per matrix (theta, residual, restarts).
int origin_pca_eig(origin_ctx *ctx, const double *d_G, const long *d_g_off, const long *d_ld,
                   const long *d_n, int nmat, long q_total, const long *d_q_off, double *d_v,
                   const long *d_v_off, double *d_info) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_G && d_g_off && d_ld && d_n && d_q_off && d_v && d_v_off && nmat > 0 &&
                       q_total > 0,
                   "bad arguments");
  void *scr = nullptr;
  int rc = origin_scratch(ctx, (size_t)q_total * sizeof(double), &scr);
  if (rc) return rc;
  // same kernel choice as origin_pca_run
  std::vector<long> h_ld(nmat);
  ORIGIN_HIP(hipMemcpyAsync(h_ld.data(), d_ld, (size_t)nmat * sizeof(long), hipMemcpyDeviceToHost,
                            ctx->stream));
  ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
  long ldmax = 0;
  for (long x : h_ld) ldmax = std::max(ldmax, x);
  ProfScope ps(ctx, K_PCA_EIG, 2);
  return eig_launch(ctx, nmat, ldmax, d_G, d_g_off, d_ld, d_n, (double *)scr, d_q_off, d_v, d_v_off,
                    d_info);
}

int origin_pca_eig_qrows(void) { return EIG_QROWS; }

// The whole greedy PCA of `na` areas; d_F receives cube_faint.  out_nx > 0: d_F is the first
// element of an (Nz, S / out_nx, out_nx) box inside a larger cube with row pitch out_py and plane
// pitch out_pz (elements) -- the interior of a halo-extended tile, so that the tiled path needs no
// copy between the PCA and the halo exchange; d_X must then be a different, contiguous cube.
int origin_pca_set_tail_hook(origin_ctx *ctx, void (*hook)(void *, int, const int *), void *user,
                             int max_active) {
  ORIGIN_CHECK_ARG(ctx && max_active >= 0, "bad argument");
  ctx->pca_tail_hook = hook;
  ctx->pca_tail_user = user;
  ctx->pca_tail_max = max_active;
  return ORIGIN_OK;
}

int origin_pca_run_into(origin_ctx *ctx, const float *d_X, float *d_F, int Nz, long S, int na,
                        const int *d_spx, const long *h_spx_off, const double *d_test0,
                        const double *h_thr, double noise_pop, int itermax, int *d_mapO2,
                        int *h_nstop, int *h_iters, long *h_trace, int trace_cap, int out_nx,
                        long out_py, long out_pz) {
  ORIGIN_USE(ctx);
  const bool strided = out_nx > 0;
  ORIGIN_CHECK_ARG(!strided || (d_X && d_X != d_F && S % out_nx == 0 && out_py >= out_nx &&
                                out_pz >= (S / out_nx) * out_py),
                   "strided output needs a separate contiguous input and consistent pitches");
  ORIGIN_CHECK_ARG(d_F && d_spx && h_spx_off && d_test0 && h_thr && d_mapO2 && h_nstop &&
                       Nz > 0 && S > 0 && na > 0,
                   "bad arguments");
  ORIGIN_CHECK_ARG(noise_pop > 0 && itermax >= 0, "bad Noise_population / itermax");
  ProfScope ps_total(ctx, K_PCA_TOTAL, 1);
  if (!d_X) d_X = d_F;
  const float *src = d_X;  // where the not-yet-deflated cube is read from
  const long ntot = h_spx_off[na];
  ORIGIN_CHECK_ARG(ntot >= 0 && ntot <= S, "area lists longer than the field");
  hipStream_t st = ctx->stream;
  if (h_iters) *h_iters = 0;
  *h_nstop = 0;
  ORIGIN_HIP(hipMemsetAsync(d_mapO2, 0, (size_t)S * sizeof(int), st));
  if (src != d_F && ntot < S) {  // spaxels outside every area are simply copied
    if (strided) {
      int rcb = origin_copy_box(ctx, 2, d_F, out_py, out_pz, d_X, out_nx, S, Nz, (int)(S / out_nx),
                                out_nx, (int)sizeof(float));
      if (rcb) return rcb;
    } else {
      ORIGIN_HIP(hipMemcpyAsync(d_F, d_X, (size_t)Nz * S * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
  }
  if (ntot == 0) return ORIGIN_OK;

  // ---- persistent state on the device
  // device buffers are kept in the context between calls (hipMalloc/hipFree of a few hundred
  // MB per call cost milliseconds)
  if (!ctx->pca_ws) {
    ctx->pca_ws = new PcaWorkspace();
    ctx->pca_ws_free = [](void *p) { delete (PcaWorkspace *)p; };
  }
  PcaWorkspace &W = *(PcaWorkspace *)ctx->pca_ws;
  DevBuf &b_state = W.b[0], &b_lists = W.b[1], &b_test = W.b[2], &b_desc = W.b[3],
         &b_tiles = W.b[4], &b_xp = W.b[5], &b_g = W.b[6], &b_cv = W.b[7], &b_bu = W.b[8],
         &b_part = W.b[9], &b_cpart = W.b[10], &b_info = W.b[11], &b_U = W.b[12], &b_C = W.b[13],
         &b_small = W.b[14], &b_fd = W.b[15];
  DevBuf *b_fb[2] = {&W.b[16], &W.b[17]};  // nuisance blocks of this / the previous iteration
  const bool debug = getenv("ORIGIN_PCA_DEBUG") != nullptr;
  const size_t st_bytes = (size_t)na * (sizeof(double) + 4 * sizeof(int)) + sizeof(int) * 2 +
                          (size_t)(na + 1) * sizeof(long) + 64;
  int rc;
  if ((rc = b_state.reserve(ctx, st_bytes))) return rc;
  char *sp = (char *)b_state.p;
  double *d_thr = (double *)sp;
  sp += (size_t)na * sizeof(double);
  long *d_spx_off = (long *)sp;
  sp += (size_t)(na + 1) * sizeof(long);
  int *d_active = (int *)sp;
  sp += (size_t)na * sizeof(int);
  int *d_nbiter = (int *)sp;
  sp += (size_t)na * sizeof(int);
  int *d_n = (int *)sp;  // n and nb are contiguous: one read-back
  sp += (size_t)na * sizeof(int);
  int *d_nb = (int *)sp;
  sp += (size_t)na * sizeof(int);
  int *d_nstop = (int *)sp;
  unsigned *d_selcnt = (unsigned *)(d_nstop + 1);  // blocks of all selections that have finished
  std::vector<int> ones(na, 1);
  ORIGIN_HIP(hipMemcpyAsync(d_thr, h_thr, (size_t)na * sizeof(double), hipMemcpyHostToDevice, st));
  ORIGIN_HIP(hipMemcpyAsync(d_spx_off, h_spx_off, (size_t)(na + 1) * sizeof(long),
                            hipMemcpyHostToDevice, st));
  ORIGIN_HIP(hipMemcpyAsync(d_active, ones.data(), (size_t)na * sizeof(int), hipMemcpyHostToDevice,
                            st));
  ORIGIN_HIP(hipMemsetAsync(d_nbiter, 0, (size_t)na * sizeof(int), st));
  ORIGIN_HIP(hipMemsetAsync(d_nstop, 0, 2 * sizeof(int), st));
  ORIGIN_HIP(hipStreamSynchronize(st));  // `ones` goes out of use
  if ((rc = b_lists.reserve(ctx, (size_t)5 * ntot * sizeof(int)))) return rc;
  int *d_nuis = (int *)b_lists.p, *d_bg = d_nuis + ntot;
  int *d_bg_pos = d_bg + ntot;
  int *d_npos[2] = {d_bg_pos + ntot, d_bg_pos + 2 * ntot};  // nuisance positions, ping-pong
  // nuisance block of each area as the previous iteration left it (host bookkeeping)
  std::vector<long> fb_off(na, -1), fb_ld(na, 0), fb_n(na, 0);
  // running sum of the background columns of each area (bmean_kernel): float64 [na][Nz], then
  // the signed list of columns that entered / left the set [ntot], their number per area [na]
  // and the membership flag of every list position [ntot].  Only the register-resident
  // selection kernel keeps them up to date.
  const bool bg_delta = getenv("ORIGIN_PCA_FULL_BMEAN") == nullptr;
  if ((rc = W.b[18].reserve(ctx, (size_t)na * Nz * sizeof(double) +
                                     (size_t)(ntot + na) * sizeof(int) + (size_t)ntot)))
    return rc;
  double *d_ssum = (double *)W.b[18].p;
  int *d_dlist = (int *)(d_ssum + (size_t)na * Nz);
  int *d_ndiff = d_dlist + ntot;
  uint8_t *d_inb = (uint8_t *)(d_ndiff + na);
  ORIGIN_HIP(hipMemsetAsync(d_inb, 0, (size_t)ntot, st));
  std::vector<char> s_valid(na, 0);
  // spaxel -> (area, list position) for the kernels that walk the cube in memory order
  // (deflate_dot_rows_kernel); the per-iteration area -> work-list slot table travels with the
  // descriptors
  const bool row_dot = getenv("ORIGIN_PCA_AREA_DOT") == nullptr;
  if ((rc = W.b[19].reserve(ctx, (size_t)2 * S * sizeof(int)))) return rc;
  int *d_area_of = (int *)W.b[19].p, *d_pos_of = d_area_of + S;
  // (flush_rows_kernel, round 4: 4.15 ms against 2.96 for the area-order flush_kernel at
  // 3681 x 600 x 600 -- per block of 256 spaxels x 32 channels the run detection, the tables of
  // three or four areas and four block barriers cost more than the cut lines they save; kept
  // behind ORIGIN_PCA_ROW_FLUSH=1)
  const bool row_flush = getenv("ORIGIN_PCA_ROW_FLUSH") != nullptr;
  if (row_dot || row_flush) {
    int nsm = 0;
    for (int a = 0; a < na; ++a) nsm = std::max(nsm, (int)(h_spx_off[a + 1] - h_spx_off[a]));
    ORIGIN_HIP(hipMemsetAsync(d_area_of, 0xFF, (size_t)S * sizeof(int), st));
    if (nsm > 0)
      hipLaunchKernelGGL(invert_lists_kernel, dim3(cdiv(nsm, 256), na), dim3(256), 0, st, d_spx,
                         d_spx_off, d_area_of, d_pos_of);
  }
  std::vector<int> kidx(na, -1);
  if ((rc = b_test.reserve(ctx, (size_t)S * sizeof(double)))) return rc;
  double *d_test = (double *)b_test.p;
  ORIGIN_HIP(hipMemcpyAsync(d_test, d_test0, (size_t)S * sizeof(double), hipMemcpyDeviceToDevice,
                            st));
  // removed vectors U[area][z][PCA_CAP] and coefficient rows C[PCA_CAP][list position]
  if ((rc = b_U.reserve(ctx, (size_t)na * Nz * PCA_CAP * sizeof(double)))) return rc;
  if ((rc = b_C.reserve(ctx, (size_t)PCA_CAP * ntot * sizeof(double)))) return rc;
  double *d_U = (double *)b_U.p, *d_C = (double *)b_C.p;
  std::vector<int> T(na, 0);  // vectors held per area since the last flush

  // read-back buffer [2*na] + [1] generation flag: mapped, coherent host memory the selection
  // kernel writes into directly
  if ((rc = W.h_nnb.reserve((size_t)(2 * na + 1) * sizeof(int),
                            hipHostMallocMapped | hipHostMallocCoherent)))
    return rc;
  int *h_nnb = (int *)W.h_nnb.p;
  memset(h_nnb, 0, (size_t)(2 * na + 1) * sizeof(int));
  int *d_hostout = nullptr;
  ORIGIN_HIP(hipHostGetDevicePointer((void **)&d_hostout, h_nnb, 0));

  // F = X - U C for every area that holds vectors (every area at all when the output is a
  // different buffer and has not been written yet); afterwards T = 0 and the cube is read
  // from d_F
  // (Flushing the areas that have finished at once, on a second low-priority or CU-masked stream
  // in the shadow of the iterations that go on, was built and measured: no gain -- 25.4-25.5 ms
  // against 25.0-25.1; the one-block kernels of the chain slow down by what the flush overlaps.)
  // (strided output: a flush in the middle of the run -- an area used up its PCA_CAP slots -- goes
  // to a contiguous work cube the later passes read; only the final one writes d_F)
  float *d_work = nullptr;
  // Areas written to d_F ahead of the final flush (tail hook, below): they have stopped iterating
  // for good -- the final flush leaves them alone.  only_done: write just the areas that are not
  // in the work list any more (counts n_done[a] < 2); nothing else changes (src, T of the others).
  std::vector<char> flushed(na, 0);
  const int *n_done = nullptr;
  auto flush = [&](bool final, bool only_done = false) -> int {
    float *dst = d_F;
    if (strided && !final) {
      int rw = W.b[20].reserve(ctx, (size_t)Nz * S * sizeof(float));
      if (rw) return rw;
      if (!d_work && ntot < S)
        ORIGIN_HIP(hipMemcpyAsync(W.b[20].p, d_X, (size_t)Nz * S * sizeof(float),
                                  hipMemcpyDeviceToDevice, st));
      d_work = (float *)W.b[20].p;
      dst = d_work;
    }
    const bool to_strided = strided && final;
    const bool all = src != dst;
    std::vector<long> fd;
    int nf = 0, nsmax = 0;
    auto wanted = [&](int a) {
      if (flushed[a] || (only_done && n_done[a] >= 2)) return false;
      return (all || T[a] > 0) && h_spx_off[a + 1] > h_spx_off[a];
    };
    for (int a = 0; a < na; ++a) nf += wanted(a);
    if (nf > 0) {
      fd.assign((size_t)4 * nf, 0);
      int k = 0;
      for (int a = 0; a < na; ++a) {
        const int ns = (int)(h_spx_off[a + 1] - h_spx_off[a]);
        if (!wanted(a)) continue;
        fd[k] = a;
        fd[(size_t)nf + k] = h_spx_off[a];
        fd[(size_t)2 * nf + k] = ns;
        fd[(size_t)3 * nf + k] = T[a];
        nsmax = std::max(nsmax, ns);
        ++k;
      }
      // behind the descriptors: aT[a] = vectors of area a, -1 = not written now (flush_rows_kernel)
      const size_t fd_desc = fd.size();
      fd.resize(fd_desc + ((size_t)na + 1) / 2, 0);
      {
        int *at = reinterpret_cast<int *>(fd.data() + fd_desc);
        for (int a = 0; a < na; ++a) at[a] = -1;
        for (int k2 = 0; k2 < nf; ++k2) at[fd[k2]] = (int)fd[(size_t)3 * nf + k2];
      }
      int r;
      if ((r = b_fd.reserve(ctx, fd.size() * sizeof(long)))) return r;
      ORIGIN_HIP(hipMemcpyAsync(b_fd.p, fd.data(), fd.size() * sizeof(long), hipMemcpyHostToDevice,
                                st));
      ORIGIN_HIP(hipStreamSynchronize(st));
      ProfScope ps(ctx, K_PCA_FLUSH, 2);
      // (Round 3 measured two more forms of this pass at 3681 x 600 x 600 and dropped them: the
      // cube's memory order with one pass per area present in a wave, 3.39 ms; two / four spaxels
      // per lane so that one LDS read of U serves several products, 3.0 / 3.8 ms; this form 2.83.
      // Round 4: 16 / 24 / 8 channels per block instead of 32 (80 / 96 / 64 VGPRs, six / five / eight
      // waves per SIMD): 3.01 / 2.93 / 3.61 ms against 3.04 in the same session -- not occupancy.)
      const int nxb = cdiv(nsmax, 256), nzb = cdiv(Nz, FLUSH_ZB);
      long nsum_f = 0;
      for (int k2 = 0; k2 < nf; ++k2) nsum_f += fd[(size_t)2 * nf + k2];
      if (row_flush && 2 * nsum_f >= S && Nz / FLUSH_ZB < 65535) {
        // the areas written now cover at least half of the field: walk the cube in memory order
        // (flush_rows_kernel); aT[a] = vectors of area a, -1 = not written now
        const int *d_aT = (const int *)((const long *)b_fd.p + fd_desc);
        static OriginPerDeviceOnce attr_rows;
        ORIGIN_ONCE_PER_DEVICE(ctx, attr_rows,
                               ORIGIN_HIP(hipFuncSetAttribute(
                                   (const void *)flush_rows_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, FLUSH_ROWS_LDS)));
        hipLaunchKernelGGL(flush_rows_kernel, dim3((unsigned)cdiv(S, 256), nzb), dim3(256),
                           FLUSH_ROWS_LDS, st, src, dst, Nz, S, d_area_of, d_pos_of, d_aT, d_U, d_C,
                           ntot, to_strided ? out_nx : 0, out_py, out_pz);
      } else {
        const long ngroups = ((long)nxb * nzb + 7) / 8;  // groups of 8 (spaxel chunk, channel block)
        hipLaunchKernelGGL(flush_kernel, dim3((unsigned)(ngroups * nf * 8)), dim3(256), 0, st, src,
                           dst, Nz, S, d_spx, (const long *)b_fd.p, nf, d_U, d_C, ntot, nxb, nzb,
                           to_strided ? out_nx : 0, out_py, out_pz);
      }
      ORIGIN_LAUNCH_CHECK();
    }
    if (only_done) {  // the areas that still iterate keep their vectors and go on reading src
      for (int a = 0; a < na; ++a)
        if (n_done[a] < 2) flushed[a] = 1, T[a] = 0;
      return ORIGIN_OK;
    }
    for (int a = 0; a < na; ++a) T[a] = 0;
    std::fill(fb_off.begin(), fb_off.end(), -1L);  // (blocks refer to columns of U)
    std::fill(s_valid.begin(), s_valid.end(), (char)0);  // (sums refer to the cube read so far)
    src = dst;
    return ORIGIN_OK;
  };
  // LDS cache of the select kernel: the largest area, if it fits in 120 KiB
  int nsmax_all = 0;
  for (int a = 0; a < na; ++a) nsmax_all = std::max(nsmax_all, (int)(h_spx_off[a + 1] - h_spx_off[a]));
  int sel_cap = nsmax_all <= 12288 ? nsmax_all : 0;  // 12 B per spaxel: O2 value + spaxel index
  size_t sel_lds = (size_t)sel_cap * (sizeof(double) + sizeof(int));
  if (sel_lds > 48 * 1024 &&
      hipFuncSetAttribute((const void *)pca_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)sel_lds) != hipSuccess) {
    (void)hipGetLastError();
    sel_cap = 0;
    sel_lds = 0;
  }

  char *h_stage = (char *)W.h_stage.p;  // pinned staging of descriptors + Gram tile lists

  std::vector<long> D;
  int iters = 0;
  int gen = 1;  // selections launched so far + 1
  const bool use_delta = bg_delta && nsmax_all <= 1024 * SEL_EPT;
  // The selection of iteration t; with patchD the work list of that iteration is on the device
  // already and receives the real counts (see PCA_SLOT_DONE).
  auto launch_select = [&](long *patchD, const int *d_kidx, int nw_) {
    ProfScope ps(ctx, K_PCA_SELECT, 2);
    if (nsmax_all <= 1024 * SEL_EPT)
      hipLaunchKernelGGL(pca_select_fast_kernel, dim3(na), dim3(1024),
                         (size_t)nsmax_all * sizeof(int), st, d_spx, d_spx_off, d_test, d_thr,
                         noise_pop, itermax, d_active, d_nbiter, d_nstop, d_mapO2, d_nuis, d_bg,
                         d_npos[iters & 1], d_bg_pos, d_n, d_nb, d_hostout, d_selcnt, gen,
                         use_delta ? d_inb : nullptr, d_dlist, d_ndiff, patchD, d_kidx, nw_);
    else
      hipLaunchKernelGGL(pca_select_kernel, dim3(na), dim3(1024), sel_lds, st, d_spx, d_spx_off,
                         d_test, d_thr, noise_pop, itermax, d_active, d_nbiter, d_nstop, d_mapO2,
                         d_nuis, d_bg, d_npos[iters & 1], d_bg_pos, d_n, d_nb, sel_cap, d_hostout,
                         d_selcnt, gen, patchD, d_kidx, nw_);
  };
  // n / nb arrive in mapped host memory; wait for the generation flag of the newest selection
  // (everything enqueued before it on the stream is complete by then).  Fall back to a stream
  // synchronisation after 5 s (a faulted kernel never raises the flag).
  auto wait_select = [&]() -> int {
    int *flag = h_nnb + 2 * na;
    const auto t_start = std::chrono::steady_clock::now();
    long polls = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != gen) {
      __builtin_ia32_pause();
      if ((++polls & 0xfffff) == 0 &&
          std::chrono::steady_clock::now() - t_start > std::chrono::seconds(5)) {
        ORIGIN_HIP(hipStreamSynchronize(st));
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != gen) {
          origin_set_error("greedy PCA: the selection kernel did not report back");
          return ORIGIN_E_HIP;
        }
      }
    }
    ++gen;
    return ORIGIN_OK;
  };
  // Iteration 0: the selection first, then the chain with exact counts.  From iteration 1 on the
  // host runs one selection AHEAD of the device: the work list of iteration t is built from the
  // counts of selection t-1 (upper bounds), uploaded, and selection t + the chain of iteration t
  // are enqueued before the host waits for selection t -- the device never waits for the host
  // (the hand-shake + descriptor upload cost ~60 us of idle device per iteration before).
  std::vector<int> n_lay(2 * (size_t)na, 0);  // counts the next work list is laid out for
  launch_select(nullptr, nullptr, 0);
  ORIGIN_LAUNCH_CHECK();
  if ((rc = wait_select())) return rc;
  memcpy(n_lay.data(), h_nnb, 2 * (size_t)na * sizeof(int));
  bool exact = true;
  bool tail_fired = false;
  int tail_run = 0, tail_n0 = 0;  // iterations in a row with few areas; their nuisance count at the first
  // Default: selection, hand-shake, exact work list, chain.  ORIGIN_PCA_PIPELINED=1 lets the host
  // run one selection ahead (below).  Measured A/B at 3681 x 600 x 600 (57 iterations): 25.7-25.9
  // against 26.0-26.3 ms -- the loop is bound by the device (host: 2 ms of enqueueing, 20 ms of
  // waiting), the hand-shake hides behind the chain either way, and lists laid out for the
  // previous counts make the Gram / dot kernels of iterations 1-7 a little larger.  (The 120 us
  // gaps per tail iteration in rocprofv3 timelines are per-dispatch profiler overhead: the
  // unprofiled tail runs at its kernels' busy time, ~125 us per iteration.)
  const bool pipelined = getenv("ORIGIN_PCA_PIPELINED") != nullptr;
  // host-side phase times of the loop (ORIGIN_PCA_DEBUG): building the list, enqueueing, waiting
  double t_build = 0, t_enq = 0, t_wait = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  };
  for (;;) {
    const auto tp0 = now();
    if (!pipelined && iters > 0) {
      launch_select(nullptr, nullptr, 0);
      ORIGIN_LAUNCH_CHECK();
      if ((rc = wait_select())) return rc;
      memcpy(n_lay.data(), h_nnb, 2 * (size_t)na * sizeof(int));
          exact = true;
    }
    // ---- work list of this iteration
    int nw = 0;
    bool full = false;
    for (int a = 0; a < na; ++a)
      if (n_lay[a] >= 2) {
        ++nw;
        full = full || T[a] >= PCA_CAP;
      }
    if (nw == 0) break;
    // Tail hook: once, when few areas are left.  The areas that are done are written to d_F now
    // (the pass they would have had at the end), then the caller learns which areas go on -- it may
    // start the next stage on everything that does not depend on them (origin_glr_run_rows on the
    // side stream).  Needs exact counts (the default, non-pipelined loop) and no work cube between
    // the input and d_F.
    // It fires only when the few areas left look like stragglers: three iterations in a row with
    // at most pca_tail_max areas, whose nuisance count has not fallen below 0.6 of what it was at
    // the first of them.  (Bench fields, (areas, nuisance spaxels) per iteration: 600^2 ... (3,50)
    // (2,31) (2,28) (1,24) (1,21) (1,20) and 44 more down to (1,12); 200^2 ... (2,74) (2,58) (1,52)
    // (1,51) and 34 more; 300^2 ... (7,93) (1,23) (1,13) (1,3), end -- there the run is over
    // before anything started beside it could pay: 17.5 ms per step with the hook at the first
    // iteration with one area, 16.3 without it.)
    int n_active_sum = 0;
    for (int a = 0; a < na; ++a)
      if (n_lay[a] >= 2) n_active_sum += n_lay[a];
    if (nw > 0 && nw <= ctx->pca_tail_max && iters >= 1) {
      if (tail_run++ == 0) tail_n0 = n_active_sum;
    } else {
      tail_run = 0;
    }
    if (ctx->pca_tail_hook && !tail_fired && exact && tail_run >= 3 &&
        10 * (long)n_active_sum >= 6 * (long)tail_n0 && !d_work && !full) {
      tail_fired = true;
      n_done = n_lay.data();
      if ((rc = flush(true, true))) return rc;
      n_done = nullptr;
      std::vector<int> act;
      for (int a = 0; a < na; ++a)
        if (n_lay[a] >= 2) act.push_back(a);
      ctx->pca_tail_hook(ctx->pca_tail_user, (int)act.size(), act.data());
    }
    if (full && (rc = flush(false))) return rc;  // an area used up its PCA_CAP slots
    D.assign((size_t)DF_COUNT * nw, 0);
    long xp = 0, c = 0, g = 0, q = 0, cb = 0;
    int ldmax = 0, nsmax = 0, k = 0;
    std::vector<int> ti, tj, ta;
    long nsum = 0;
    for (int a = 0; a < na; ++a) {
      const int n = n_lay[a];
      if (n < 2) continue;
      const int ld = (n + 15) / 16 * 16;
      const int ns = (int)(h_spx_off[a + 1] - h_spx_off[a]);
      D[(size_t)DF_AREA * nw + k] = a;
      D[(size_t)DF_LIST0 * nw + k] = h_spx_off[a];
      D[(size_t)DF_N * nw + k] = n;
      D[(size_t)DF_NB * nw + k] = n_lay[na + a];
      D[(size_t)DF_LD * nw + k] = ld;
      D[(size_t)DF_XP * nw + k] = xp;
      D[(size_t)DF_C * nw + k] = c;
      D[(size_t)DF_G * nw + k] = g;
      D[(size_t)DF_Q * nw + k] = q;
      D[(size_t)DF_NS * nw + k] = ns;
      D[(size_t)DF_CBASE * nw + k] = cb;
      D[(size_t)DF_T * nw + k] = T[a];
      // previous nuisance block of the area, usable while at least one vector is held
      const bool have = fb_off[a] >= 0 && T[a] >= 1;
      D[(size_t)DF_FBOLD * nw + k] = have ? fb_off[a] : -1;
      D[(size_t)DF_LDOLD * nw + k] = fb_ld[a];
      D[(size_t)DF_NOLD * nw + k] = fb_n[a];
      D[(size_t)DF_SVALID * nw + k] = use_delta && s_valid[a];
      s_valid[a] = use_delta;  // this iteration's bmean_kernel leaves the sum behind
      xp += (long)Nz * ld;
      c += ld;
      g += (long)ld * ld;
      q += (long)EIG_QROWS * ld;
      cb += ns;
      ldmax = std::max(ldmax, ld);
      nsmax = std::max(nsmax, ns);
      nsum += n;
      const int Tt = (ld + 31) / 32;
      for (int i = 0; i < Tt; ++i)
        for (int j = i; j < Tt; ++j) ti.push_back(i), tj.push_back(j), ta.push_back(k);
      ++k;
    }
    if (h_trace && iters < trace_cap) {
      h_trace[2 * iters] = nw;
      h_trace[2 * iters + 1] = nsum;
    }
    const int ntiles = (int)ti.size();
    // descriptors and tile lists go up in ONE asynchronous copy from pinned staging (free
    // again: the selection that has just reported back is behind every earlier upload)
    const size_t dbytes = D.size() * sizeof(long), tbytes = (size_t)3 * ntiles * sizeof(int);
    const size_t kbytes = (size_t)na * sizeof(int);  // area -> slot of this iteration
    if ((rc = b_desc.reserve(ctx, dbytes + tbytes + kbytes))) return rc;
    long *dD = (long *)b_desc.p;
    int *d_ti = (int *)((char *)b_desc.p + dbytes), *d_tj = d_ti + ntiles, *d_ta = d_tj + ntiles;
    if (dbytes + tbytes + kbytes > W.h_stage.bytes) {
      // (free again: the selection that has just reported back is behind every earlier upload)
      if ((rc = W.h_stage.reserve((dbytes + tbytes + kbytes) * 2, hipHostMallocDefault))) return rc;
      h_stage = (char *)W.h_stage.p;
    }
    memcpy(h_stage, D.data(), dbytes);
    {
      int *ht = (int *)(h_stage + dbytes);
      memcpy(ht, ti.data(), (size_t)ntiles * sizeof(int));
      memcpy(ht + ntiles, tj.data(), (size_t)ntiles * sizeof(int));
      memcpy(ht + 2 * (size_t)ntiles, ta.data(), (size_t)ntiles * sizeof(int));
      std::fill(kidx.begin(), kidx.end(), -1);
      for (int w = 0; w < nw; ++w) kidx[(size_t)D[(size_t)DF_AREA * nw + w]] = w;
      memcpy(ht + 3 * (size_t)ntiles, kidx.data(), kbytes);
    }
    const int *d_kidx_it = (const int *)((char *)b_desc.p + dbytes + tbytes);
    const auto tp1 = now();
    t_build += secs(tp0, tp1);
    ORIGIN_HIP(hipMemcpyAsync(b_desc.p, h_stage, dbytes + tbytes + kbytes, hipMemcpyHostToDevice,
                              st));
    if (!exact) {  // this iteration's selection: fills DF_N / DF_NB of the list just uploaded
      launch_select(dD, d_kidx_it, nw);
      ORIGIN_LAUNCH_CHECK();
    }
    if ((rc = b_xp.reserve(ctx, (size_t)xp * sizeof(double)))) return rc;
    if ((rc = b_fb[iters & 1]->reserve(ctx, (size_t)xp * sizeof(double)))) return rc;
    double *d_Fb = (double *)b_fb[iters & 1]->p;
    if ((rc = b_g.reserve(ctx, (size_t)g * sizeof(double)))) return rc;
    if ((rc = b_cv.reserve(ctx, (size_t)c * sizeof(double)))) return rc;
    if ((rc = b_bu.reserve(ctx, (size_t)2 * nw * Nz * sizeof(double)))) return rc;
    if ((rc = b_small.reserve(ctx, ((size_t)2 * nw * PCA_CAP +
                                    (size_t)nw * UW_SLICES * (PCA_CAP + 1)) * sizeof(double))))
      return rc;
    double *d_Xp = (double *)b_xp.p, *d_G = (double *)b_g.p;
    double *d_v = (double *)b_cv.p;
    double *d_b = (double *)b_bu.p, *d_u = d_b + (size_t)nw * Nz;
    double *d_cbar = (double *)b_small.p, *d_wq = d_cbar + (size_t)nw * PCA_CAP;
    double *d_uwpart = d_wq + (size_t)nw * PCA_CAP;
    const long *dLD = dD + (size_t)DF_LD * nw, *dXP = dD + (size_t)DF_XP * nw;
    const long *dG = dD + (size_t)DF_G * nw, *dQ = dD + (size_t)DF_Q * nw;
    const long *dN = dD + (size_t)DF_N * nw, *dC = dD + (size_t)DF_C * nw;

    {
      ProfScope ps(ctx, K_PCA_BMEAN, 2);
      hipLaunchKernelGGL(cbar_kernel, dim3(nw), dim3(1024), 0, st, d_C, ntot, d_bg_pos, dD, nw,
                         d_cbar);
      hipLaunchKernelGGL(bmean_kernel, dim3(cdiv(Nz, 4), nw), dim3(64, 4), 0, st, src, Nz, S, d_bg,
                         dD, nw, d_U, d_cbar, d_b, d_dlist, d_ndiff, use_delta ? d_ssum : nullptr);
    }
    // z slices of the gather: enough blocks to fill the chip even when few areas iterate
    int nzb = (int)(((long)ctx->num_cu * 2 + (long)cdiv(ldmax, 64) * nw - 1) /
                    ((long)cdiv(ldmax, 64) * nw));
    nzb = std::max(1, std::min(nzb, 16));
    const int gzper = (cdiv(Nz, nzb) + 15) / 16 * 16;
    nzb = cdiv(Nz, gzper);
    if ((rc = b_cpart.reserve(ctx, (size_t)nzb * c * sizeof(double)))) return rc;
    double *d_cpart = (double *)b_cpart.p;
    {
      ProfScope ps(ctx, K_PCA_GATHER, 2);
      hipLaunchKernelGGL(gather_xp_kernel, dim3(cdiv(ldmax, 64), nw, nzb), dim3(64, 16), 0, st,
                         src, Nz, S, d_nuis, d_npos[iters & 1], d_npos[(iters & 1) ^ 1], dD, nw, d_b,
                         d_U, d_C, ntot, (const double *)b_fb[(iters & 1) ^ 1]->p, d_Fb, d_cpart, c,
                         gzper);
    }
    {
      ProfScope ps(ctx, K_PCA_PROJECT, 2);
      hipLaunchKernelGGL(project_xp_kernel, dim3(cdiv(Nz, 16), nw), dim3(256), 0, st, d_b, Nz, dD,
                         nw, d_Fb, d_Xp, d_cpart, c, nzb);
    }
    ORIGIN_LAUNCH_CHECK();
    const bool all_small = ldmax <= LANCZOS_M;  // (ld is n rounded up to 16: n <= 48)
    const double *d_slab = nullptr;
    int gram_ksplit = 0;
    if ((rc = gram_launch(ctx, d_Xp, dXP, dLD, Nz, ntiles, d_ti, d_tj, d_ta, g, d_G, dG, all_small,
                          &d_slab, &gram_ksplit, dN)))
      return rc;
    {
      void *scr = nullptr;
      if ((rc = b_part.reserve(ctx, (size_t)q * sizeof(double)))) return rc;
      scr = b_part.p;
      double *d_info = nullptr;
      if (debug) {
        if ((rc = b_info.reserve(ctx, (size_t)13 * nw * sizeof(double)))) return rc;
        d_info = (double *)b_info.p;
      }
      {
        ProfScope ps(ctx, K_PCA_EIG, 2);
        if ((rc = eig_launch(ctx, nw, ldmax, d_G, dG, dLD, dN, (double *)scr, dQ, d_v, dC, d_info,
                             all_small ? d_slab : nullptr, g, gram_ksplit,
                             debug ? d_info + (size_t)3 * nw : nullptr)))
          return rc;
      }
      if (debug) {
        std::vector<double> info((size_t)13 * nw, 0.0);
        ORIGIN_HIP(hipMemcpyAsync(info.data(), d_info, info.size() * sizeof(double),
                                  hipMemcpyDeviceToHost, st));
        ORIGIN_HIP(hipStreamSynchronize(st));
        if (getenv("ORIGIN_PCA_DEBUG_EIG")) {  // per matrix: n:steps:restarts:us
          fprintf(stderr, "[pca-eig] iter %d:", iters);
          for (int w = 0; w < nw; ++w)
            fprintf(stderr, " %d:%.0f:%.0f:%.0f", (int)D[(size_t)DF_N * nw + w], info[3 * nw + 2 * w],
                    info[3 * w + 2], info[3 * nw + 2 * w + 1]);
          fprintf(stderr, "\n");
          // phases (us) of the slowest plain-Lanczos block: setup, mat-vec, vector part, checks, final
          int ws_ = -1;
          for (int w = 0; w < nw; ++w)
            if (info[3 * nw + 2 * w] > 0 && (ws_ < 0 || info[3 * nw + 2 * w + 1] > info[3 * nw + 2 * ws_ + 1]))
              ws_ = w;
          if (ws_ >= 0) {
            const double *t = &info[5 * (size_t)nw + 8 * (size_t)ws_];
            fprintf(stderr, "[pca-eig]   slowest n %d steps %.0f: setup %.1f matvec %.1f vector %.1f checks %.1f "
                    "(prep %.1f rounds %.1f vector %.1f) final %.1f us\n",
                    (int)D[(size_t)DF_N * nw + ws_], info[3 * nw + 2 * ws_], t[0] * 0.01, t[1] * 0.01,
                    t[2] * 0.01, t[3] * 0.01, t[5] * 0.01, t[6] * 0.01, t[7] * 0.01, t[4] * 0.01);
          }
        }
        double rmax = 0, rsum = 0, resmax = 0;
        int nmax = 0;
        for (int w = 0; w < nw; ++w) {
          rmax = std::max(rmax, info[3 * w + 2]);
          rsum += info[3 * w + 2];
          resmax = std::max(resmax, info[3 * w + 1] / std::max(info[3 * w], 1e-300));
          nmax = std::max(nmax, (int)D[(size_t)DF_N * nw + w]);
        }
        fprintf(stderr, "[pca] iter %3d areas %3d nmax %4d restarts max %2.0f mean %.2f relres max %.1e\n",
                iters, nw, nmax, rmax, rsum / nw, resmax);
      }
    }
    {
      ProfScope ps(ctx, K_PCA_UVEC, 2);
      hipLaunchKernelGGL(xv_kernel, dim3(cdiv(Nz, 4), nw), dim3(64, 4), 0, st, d_Xp, dD, nw, Nz,
                         d_v, d_u);
      hipLaunchKernelGGL(uw_partial_kernel, dim3(UW_SLICES, nw), dim3(256), 0, st, d_u, Nz, dD, nw,
                         d_U, d_uwpart);
      hipLaunchKernelGGL(normalize_kernel, dim3(nw), dim3(1024), 0, st, d_u, Nz, dD, nw, d_U,
                         d_uwpart, d_wq);
    }
    ORIGIN_LAUNCH_CHECK();
    // ---- deflation (coefficient form): one read pass over the iterating areas
    const long blocks = (long)cdiv(nsmax, 256) * nw;
    int nzs = (int)(((long)ctx->num_cu * 8 + blocks - 1) / blocks);
    nzs = std::max(1, std::min(nzs, 32));
    nzs = std::min(nzs, Nz);
    const int zper = cdiv(Nz, nzs);
    nzs = cdiv(Nz, zper);
    void *scr = nullptr;
    if ((rc = origin_scratch(ctx, (size_t)nzs * cb * sizeof(double), &scr))) return rc;
    double *cpart = (double *)scr;
    {
      ProfScope ps(ctx, K_PCA_DEFLATE_DOT, 2);
      if (row_dot && 2 * cb >= S)  // the iterating areas cover at least half of the field
        hipLaunchKernelGGL(deflate_dot_rows_kernel, dim3(cdiv(S, 256), nzs), dim3(256), 0, st, src,
                           Nz, S, d_area_of, d_pos_of, d_kidx_it, dD, nw, d_u, zper, cpart, cb);
      else
        hipLaunchKernelGGL(deflate_dot_kernel, dim3(cdiv(nsmax, 256), nzs, nw), dim3(256), 0, st,
                           src, Nz, S, d_spx, dD, nw, d_u, zper, cpart, cb);
    }
    {
      ProfScope ps(ctx, K_PCA_DEFLATE_UPDATE, 2);
      hipLaunchKernelGGL(deflate_finish_kernel, dim3(cdiv(nsmax, 256), nw), dim3(256), 0, st, d_spx,
                         dD, nw, nzs, Nz, cb, cpart, d_wq, d_C, ntot, d_test);
    }
#ifdef PCA_EXP_DUMMY
    {
      static const int nd = getenv("ORIGIN_PCA_EXP_DUMMY") ? atoi(getenv("ORIGIN_PCA_EXP_DUMMY")) : 0;
      for (int i = 0; i < nd; ++i)
        hipLaunchKernelGGL(pca_dummy_kernel, dim3(nw), dim3(64), 0, st, dD, nw, d_wq);
    }
#endif
    ORIGIN_LAUNCH_CHECK();
    // the real counts of this iteration (the device is busy with its chain meanwhile)
    const auto tp2 = now();
    t_enq += secs(tp1, tp2);
    if (!exact) {
      if ((rc = wait_select())) return rc;
      t_wait += secs(tp2, now());
      memcpy(n_lay.data(), h_nnb, 2 * (size_t)na * sizeof(int));
    }
    exact = false;
    for (int w = 0; w < nw; ++w) {
      const int a = (int)D[(size_t)DF_AREA * nw + w];
      if (n_lay[a] < 2) continue;  // finished with this selection: its slot was left alone
      T[a] += 1;
      fb_off[a] = D[(size_t)DF_XP * nw + w];  // this iteration's block becomes the source
      fb_ld[a] = D[(size_t)DF_LD * nw + w];
      fb_n[a] = n_lay[a];
    }
    ++iters;
  }
  if (getenv("ORIGIN_PCA_TIMING"))
    fprintf(stderr, "[pca] host loop: %d iterations, build %.2f ms, enqueue %.2f ms, wait %.2f ms\n",
            iters, 1e3 * t_build, 1e3 * t_enq, 1e3 * t_wait);
  if ((rc = flush(true))) return rc;
  ORIGIN_HIP(hipMemcpyAsync(h_nnb, d_nstop, sizeof(int), hipMemcpyDeviceToHost, st));
  ORIGIN_HIP(hipStreamSynchronize(st));
  *h_nstop = h_nnb[0];
  if (h_iters) *h_iters = iters;
  return ORIGIN_OK;
}

int origin_pca_run(origin_ctx *ctx, const float *d_X, float *d_F, int Nz, long S, int na,
                   const int *d_spx, const long *h_spx_off, const double *d_test0,
                   const double *h_thr, double noise_pop, int itermax, int *d_mapO2, int *h_nstop,
                   int *h_iters, long *h_trace, int trace_cap) {
  return origin_pca_run_into(ctx, d_X, d_F, Nz, S, na, d_spx, h_spx_off, d_test0, h_thr, noise_pop,
                             itermax, d_mapO2, h_nstop, h_iters, h_trace, trace_cap, 0, 0, 0);
}

}  // extern "C"
