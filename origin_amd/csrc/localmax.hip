// compute_local_max  (reference muse_origin/lib_origin.py:1220-1256): size^3 maximum
// filter, keep voxels equal to their window maximum and not masked, zero elsewhere; the
// same on -correl_min.  scipy's default border mode 'reflect' duplicates edge samples,
// which for a maximum is the same as clamping the window to the cube.
#include <algorithm>
#include <cstdlib>
#include <vector>
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

// SPARSE (round 4): the two output cubes are > 98 % zeros (a voxel in ~70 of a smoothed cube is a
// 3x3x3 maximum), and writing them dense is half of the pass's traffic (17 B per voxel: 9 in, 8
// out).  The sparse form writes (linear index, value) pairs of the non-zero outputs instead: every
// wave owns a segment of `seg_cap` entries per cube in idx0 / val0 (maxima of a0) and idx1 / val1
// (maxima of -a1) and appends to it -- a wave-uniform counter, ballot + mbcnt for the lanes' slots,
// no atomics, no second pass; the number of entries of wave w ends up in counts[w] (cube 0) and
// counts[nwaves + w] (cube 1), entries beyond the capacity are counted but not stored (the host
// sees counts > seg_cap and falls back to the dense form).  9 B per voxel read, ~0.4 B written.
struct LmSparse {
  long long *idx0, *idx1;
  float *val0, *val1;
  int *counts;
  int seg_cap;
};

template <int NC, int R, bool SPARSE = false>
__global__ __launch_bounds__(256) void local_max3v_kernel(const float *__restrict__ a0,
                                                          const float *__restrict__ a1,
                                                          const uint8_t *__restrict__ mask, int Nz,
                                                          int Ny, int Nx, int zper, float sign0,
                                                          float *__restrict__ out0,
                                                          float *__restrict__ out1,
                                                          LmSparse sp = LmSparse()) {
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
  // sparse form: this wave's segments and how many entries they hold so far (wave-uniform)
  const long wave_id = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);
  const long seg = SPARSE ? wave_id * sp.seg_cap : 0;
  int cnt[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q) cnt[q] = 0;
  for (int z = z0; z < z1; ++z) {
    plane(a0, sign0, z + 1, pc[0], cc[0]);
    if constexpr (NC == 2) plane(a1, -1.0f, z + 1, pc[1], cc[1]);
    unsigned km[NC];   // SPARSE: bit 4 r + e = output (r, e) of this lane is a non-zero maximum
#pragma unroll
    for (int q = 0; q < NC; ++q) km[q] = 0u;
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
            const bool keep = cb[q][r][e] == m && unmasked;     // local_max *= local_mask (lib :1247)
            o[q][e] = keep ? m : 0.0f;
            if constexpr (SPARSE) km[q] |= (keep && m != 0.0f) ? (1u << (4 * r + e)) : 0u;
          }
        if constexpr (!SPARSE) {
          *reinterpret_cast<float4 *>(out0 + idx) = make_float4(o[0][0], o[0][1], o[0][2], o[0][3]);
          if constexpr (NC == 2)
            *reinterpret_cast<float4 *>(out1 + idx) = make_float4(o[1][0], o[1][1], o[1][2], o[1][3]);
        }
      }
#pragma unroll
      for (int q = 0; q < NC; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          pa[q][r][e] = pb[q][r][e], pb[q][r][e] = pc[q][r][e], cb[q][r][e] = cc[q][r][e];
    }
    if constexpr (SPARSE) {
      // append the lanes' maxima of this channel to the wave's segments: one round per entry of
      // the busiest lane (usually one or two); the value is read back from the cube (the line is
      // in cache: this lane loaded it a channel ago)
      const long zbase = (long)z * S + (long)yb * Nx + 4 * x4;
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        unsigned m = km[q];
        const float *src = q == 0 ? a0 : a1;
        const float sg = q == 0 ? sign0 : -1.0f;
        long long *ix = q == 0 ? sp.idx0 : sp.idx1;
        float *vl = q == 0 ? sp.val0 : sp.val1;
        for (;;) {
          const unsigned long long bal = __ballot(m != 0u);
          if (bal == 0ull) break;
          if (m != 0u) {
            const int b = __builtin_ctz(m);
            m &= m - 1u;
            const long at = zbase + (long)(b >> 2) * Nx + (b & 3);
            const int slot = cnt[q] + (int)__builtin_amdgcn_mbcnt_hi(
                                          (unsigned)(bal >> 32),
                                          __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            if (slot < sp.seg_cap) {
              ix[seg + slot] = at;
              vl[seg + slot] = sg * src[at];
            }
          }
          cnt[q] += __popcll(bal);
        }
      }
    }
  }
  if constexpr (SPARSE) {
    if (lane == 0) {
      const long nwaves = (long)gridDim.x * gridDim.y * 4;
#pragma unroll
      for (int q = 0; q < NC; ++q) sp.counts[q * nwaves + wave_id] = cnt[q];
    }
  }
}

// ---- the sparse pass, second form (round 4) ------------------------------------------------
// The kernel above, compiled with SPARSE, takes 4.05 ms at 3681 x 600 x 600 against 4.6 ms for the
// dense cubes: not the bytes (9 B per voxel: 2.9 TB/s) but ~770 instructions per wave and channel
// -- sign multiplies, canonicalising maxima, 115 register moves that rotate the three planes,
// per-output compare / select / bit-merge chains -- on two waves per SIMD.  This form does the same
// arithmetic in ~100 VALU instructions per wave and channel:
//   * one cube per wave (grid z = cube): half the registers, four or more waves per SIMD; the
//     minima of correl_min are found as minima (v_min3), not as maxima of a negated copy;
//   * the three planes rotate by NAME (the channel loop is unrolled by three), no moves;
//   * every maximum is one v_max3 / v_min3 (inline asm: no canonicalisation of its operands);
//   * "is this voxel its window's maximum" is a v_cmp into a scalar register pair per output; the
//     wave tests each pair (SALU) and only for the few that are non-zero -- 5 of 16 per channel --
//     looks at the lanes: value non-zero, mask byte (read there, not per row), slot by
//     ballot + mbcnt, two stores.
template <int SIGN>
__device__ __forceinline__ float lm_ext3(float a, float b, float c) {
  float d;
  if constexpr (SIGN > 0) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  else asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

template <int R, int NC>
struct LmPlane {
  float p[R][NC];  // 3 x 3 extrema of the lane's R x NC outputs in this plane
  float c[R][NC];  // the plane's own samples there
};

// NC consecutive samples of a row as one load (16 / 8 / 4 bytes per lane)
template <int NC>
__device__ __forceinline__ void lm_load_row(const char *p, float (&v)[NC]) {
  if constexpr (NC == 4) {
    const float4 q = *reinterpret_cast<const float4 *>(p);
    v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
  } else if constexpr (NC == 2) {
    const float2 q = *reinterpret_cast<const float2 *>(p);
    v[0] = q.x, v[1] = q.y;
  } else {
    v[0] = *reinterpret_cast<const float *>(p);
  }
}

// NC: samples per lane and row (a wave's row segment is 64 NC samples, 62 NC of them outputs); R rows
// per lane: R + 2 rows are read for R rows of outputs, so the same register budget (R NC outputs per
// lane) spent on fewer columns and more rows reads less twice.
template <int SIGN, int R, int NC, bool HAS_MASK, bool PREFETCH, bool STRIPS>
__device__ __forceinline__ void lm_sparse_march(const float *__restrict__ a,
                                                const uint8_t *__restrict__ mask, int Nz, int Ny,
                                                int Nx, int zper, long long *__restrict__ idx_out,
                                                float *__restrict__ val_out,
                                                int *__restrict__ counts, int seg_cap, long bxi,
                                                int bzi, long nbx) {
  // (bxi, bzi: this block's position among the nbx spaxel blocks and the z chunks)
  const int nx4 = Nx / NC, ngrp = (Ny + R - 1) / R;   // (nx4: lane positions per row)
  const int lane = threadIdx.x & 63;
  bool live;
  int grp, x4;
  if constexpr (STRIPS) {
    // The four waves of a block take four row groups ONE ABOVE THE OTHER (a strip of 4 R rows) at
    // the same columns: rows yb - 1 and yb + R of a wave are own rows of its siblings, loaded on the
    // same CU within the same channel step -- they meet in the CU's L1 / the XCD's L2 instead of
    // coming from memory twice (the counter passes of the flattened mapping below: 13.9 B per voxel
    // fetched for 8 read, profiles/r04_pmc_fetch_write.json).  Only a strip's outer two rows are
    // shared with other blocks: (4 R + 2) / 4 R = 1.125 instead of (R + 2) / R = 1.5 row fetches.
    // Price: a row is cut into chunks of 62 producing lanes, the last one partly empty.
    const int nxc = (nx4 + 61) / 62;
    const int sb = (int)(bxi / nxc), xc = (int)(bxi - (long)sb * nxc);
    grp = min(4 * sb + (int)(threadIdx.x >> 6), ngrp - 1);
    const int x_raw = 62 * xc - 1 + lane;
    live = lane >= 1 && lane <= 62 && x_raw < nx4 && 4 * sb + (int)(threadIdx.x >> 6) < ngrp;
    x4 = min(max(x_raw, 0), nx4 - 1);
  } else {
    const long total = (long)ngrp * nx4;
    const long wv = bxi * 4 + (threadIdx.x >> 6);
    const long t_raw = 62 * wv - 1 + lane;
    live = lane >= 1 && lane <= 62 && t_raw < total;
    const long t = min(max(t_raw, 0L), total - 1);
    grp = (int)(t / nx4);
    x4 = (int)(t - (long)grp * nx4);
  }
  const int yb = grp * R;
  const int z0 = bzi * zper, z1 = min(Nz, z0 + zper);
  const long S = (long)Ny * Nx;
  const bool first = x4 == 0, last = x4 == nx4 - 1;
  unsigned roff[R + 2];  // byte offsets of rows yb - 1 .. yb + R (clamped) at this lane's float4
#pragma unroll
  for (int r = 0; r < R + 2; ++r)
    roff[r] = 4u * (unsigned)((long)min(max(yb - 1 + r, 0), Ny - 1) * Nx + NC * x4);
  // rows of the lane's group that exist, and the lane produces at all
  unsigned long long okrow[R];   // lanes whose output row r exists (scalar register pairs)
#pragma unroll
  for (int r = 0; r < R; ++r) okrow[r] = __ballot(live && yb + r < Ny);
  const unsigned lane_lo = lane < 32 ? 1u << lane : 0u, lane_hi = lane < 32 ? 0u : 1u << (lane - 32);

  // the R + 2 rows of plane z around the lane's outputs
  auto fetch = [&](int z, float (&v)[R + 2][NC]) {
    const char *pz = reinterpret_cast<const char *>(a + (long)min(max(z, 0), Nz - 1) * S);
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
      // (the empty asm keeps the 32-bit offset's zero-extension next to the load: "SGPR base +
      // 32-bit lane offset" is one addressing mode; hoisted, each row costs a 64-bit VGPR pair)
      unsigned o = roff[r];
      asm volatile("" : "+v"(o));
      lm_load_row<NC>(pz + o, v[r]);
    }
  };
  float vnext[R + 2][NC];   // PREFETCH: the rows of the plane after next, requested a channel ahead
  auto plane = [&](int z, LmPlane<R, NC> &o) {
    float v[R + 2][NC];
    if constexpr (PREFETCH) {
#pragma unroll
      for (int r = 0; r < R + 2; ++r)
#pragma unroll
        for (int e = 0; e < NC; ++e) v[r][e] = vnext[r][e];
      fetch(z + 1, vnext);
    } else {
      fetch(z, v);
    }
    float xm[R + 2][NC];
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
      // (mov_dpp without an `old` operand: lanes 0 / 63 get zeros, and they produce no output)
      float l = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v[r][NC - 1]), 0x138, 0xF, 0xF, true));
      float rr = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v[r][0]), 0x130, 0xF, 0xF, true));
      l = first ? v[r][0] : l;   // at a row's ends the window is clamped to the row
      rr = last ? v[r][NC - 1] : rr;
#pragma unroll
      for (int e = 0; e < NC; ++e)
        xm[r][e] = lm_ext3<SIGN>(e == 0 ? l : v[r][e - 1], v[r][e], e == NC - 1 ? rr : v[r][e + 1]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int e = 0; e < NC; ++e) {
        o.p[r][e] = lm_ext3<SIGN>(xm[r][e], xm[r + 1][e], xm[r + 2][e]);
        o.c[r][e] = v[r + 1][e];
      }
    }
  };

  const long wave_id = ((long)bzi * nbx + bxi) * 4 + (threadIdx.x >> 6);
  // this wave's segment: uniform base pointers, 32-bit lane offsets
  char *seg_idx = reinterpret_cast<char *>(idx_out + wave_id * seg_cap);
  char *seg_val = reinterpret_cast<char *>(val_out + wave_id * seg_cap);
  int cnt = 0;  // entries of this wave's segment so far (wave-uniform)
  // channel z: extrema of planes z - 1 (pa), z (pb), z + 1 (pc, made here); centre = plane z
  auto step = [&](int z, const LmPlane<R, NC> &pa, const LmPlane<R, NC> &pb, LmPlane<R, NC> &pc) {
    plane(z + 1, pc);
    const long zbase = (long)z * S + (long)yb * Nx + NC * x4;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int e = 0; e < NC; ++e) {
        const float m = lm_ext3<SIGN>(pa.p[r][e], pb.p[r][e], pc.p[r][e]);
        // one v_cmp into a scalar register pair, the rest of the test on the scalar unit (a bool
        // that goes through __ballot() costs two more VALU instructions per output: the compiler
        // widens it to 0 / 1 per lane and compares again)
        const unsigned long long eq = __builtin_amdgcn_fcmpf(pb.c[r][e], m, 1) & okrow[r];  // OEQ
        if (eq != 0ull) {  // (uniform; ~1 output in 3 has a candidate in some lane)
          const long at = zbase + (long)r * Nx + e;
          const bool cand = (((unsigned)eq & lane_lo) | ((unsigned)(eq >> 32) & lane_hi)) != 0u;
          bool hit = cand && m != 0.0f;
          if constexpr (HAS_MASK) {
            if (hit) hit = mask[at] == 0;               // local_max *= local_mask   (lib :1247)
          }
          const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
          if (hit) {
            const unsigned slot = (unsigned)cnt + __builtin_amdgcn_mbcnt_hi(
                                                      (unsigned)(bal >> 32),
                                                      __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            if (slot < (unsigned)seg_cap) {
              *reinterpret_cast<long long *>(seg_idx + 8u * slot) = at;
              *reinterpret_cast<float *>(seg_val + 4u * slot) = SIGN > 0 ? m : -m;
            }
          }
          cnt += __popcll(bal);
        }
      }
  };
  LmPlane<R, NC> A, B, C;
  if constexpr (PREFETCH) fetch(z0 - 1, vnext);
  plane(z0 - 1, A);
  plane(z0, B);
  int z = z0;
  for (; z + 3 <= z1; z += 3) {
    step(z, A, B, C);
    step(z + 1, B, C, A);
    step(z + 2, C, A, B);
  }
  if (z < z1) {
    step(z, A, B, C);
    if (z + 1 < z1) step(z + 1, B, C, A);
  }
  if (lane == 0) counts[wave_id] = cnt;
}

#ifndef LMS_R_N
#define LMS_R_N 4
#endif
#ifndef LMS_NC_N
#define LMS_NC_N 2
#endif
// Measured at 3681 x 600 x 600 (tools/localmax_sparse_time.py, R x NC: ms): 4 x 4 (119 VGPRs, four waves
// per SIMD) 3.76; 8 x 2 (the same registers, 1.25 instead of 1.5 row fetches per row of outputs) 3.76;
// 6 x 2 3.42; 5 x 2 3.51; 4 x 2 (72 VGPRs, seven waves per SIMD) 3.36; 2 x 4 3.65; 3 x 2 3.76; 4 x 1 3.77;
// 12 x 2 and 16 x 1 (spills) 5.1 / 4.5.  Rows fetched twice do not show; waves in flight do.
constexpr int LMS_R = LMS_R_N;    // rows per lane of the sparse pass
constexpr int LMS_NC = LMS_NC_N;  // samples per lane and row (4, 2 or 1)
static_assert(LMS_NC == 4 || LMS_NC == 2 || LMS_NC == 1, "samples per lane and row");
#ifdef LMS_BLOCKS_N
constexpr int LMS_BLOCKS = LMS_BLOCKS_N;
#else
constexpr int LMS_BLOCKS =   // blocks per CU the registers allow
    LMS_R * LMS_NC <= 8 ? 6 : (LMS_R * LMS_NC <= 16 ? 4 : (LMS_R * LMS_NC <= 24 ? 3 : 2));
#endif

// 1-D grid of nbx * nzc * 2 blocks.  Workgroups go to the 8 XCDs round robin by their id, and a
// lane's rows yb - 1 and yb + R are the own rows of lanes nx4 = Nx / 4 flattened positions away --
// two or three waves on, mostly in the NEXT block.  With the natural numbering that block runs on
// another XCD, behind another L2, and the shared rows come from HBM twice (the dense form's PMC
// passes: 1.4 x the read bytes).  So the ids are decoded such that an XCD gets a contiguous range
// of the (cube, z chunk, spaxel block) order: neighbours in that order run on the same XCD at about
// the same time and find each other's rows in its L2.
template <bool HAS_MASK, bool PREFETCH, bool STRIPS>
__global__ __launch_bounds__(256, PREFETCH ? (LMS_BLOCKS > 3 ? 3 : LMS_BLOCKS) : LMS_BLOCKS) void local_max3s_kernel(const float *__restrict__ a0,
                                                          const float *__restrict__ a1,
                                                          const uint8_t *__restrict__ mask, int Nz,
                                                          int Ny, int Nx, int zper, long nbx, int nzc,
                                                          int xcd_order, LmSparse sp) {
  const long nb = (long)gridDim.x;
  const long xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
  const long base = nb >> 3, rem = nb & 7;
  long logical = xcd * base + (xcd < rem ? xcd : rem) + within;  // (a bijection of [0, nb))
  if (!xcd_order) logical = blockIdx.x;
  const long bxi = logical % nbx;
  const long t = logical / nbx;
  const int bzi = (int)(t % nzc), cube = (int)(t / nzc);
  const long nwaves = nbx * nzc * 4;
  if (cube == 0)
    lm_sparse_march<1, LMS_R, LMS_NC, HAS_MASK, PREFETCH, STRIPS>(a0, mask, Nz, Ny, Nx, zper, sp.idx0, sp.val0, sp.counts,
                                        sp.seg_cap, bxi, bzi, nbx);
  else
    lm_sparse_march<-1, LMS_R, LMS_NC, HAS_MASK, PREFETCH, STRIPS>(a1, mask, Nz, Ny, Nx, zper, sp.idx1, sp.val1,
                                         sp.counts + nwaves, sp.seg_cap, bxi, bzi, nbx);
}

// ---- consumers of the sparse form ------------------------------------------------------------
// one block per segment; entries beyond the capacity were never stored
__global__ __launch_bounds__(256) void sparse_to_dense_kernel(const long long *__restrict__ idx,
                                                              const float *__restrict__ val,
                                                              const int *__restrict__ counts,
                                                              int seg_cap, float *__restrict__ dense) {
  const long seg = (long)blockIdx.x * seg_cap;
  const int n = min(counts[blockIdx.x], seg_cap);
  for (int i = threadIdx.x; i < n; i += 256) dense[idx[seg + i]] = val[seg + i];
}

// hist[b] += 1 for every kept entry whose value exceeds exactly the b + 1 smallest thresholds
__global__ __launch_bounds__(256) void sparse_count_above_kernel(
    const long long *__restrict__ idx, const float *__restrict__ val, const int *__restrict__ counts,
    long nseg, int seg_cap, const uint8_t *__restrict__ keep, long S, int nthr,
    const double *__restrict__ thr, unsigned long long *__restrict__ hist) {
  __shared__ double sthr[1024];
  __shared__ unsigned shist[1024];
  for (int i = threadIdx.x; i < nthr; i += 256) sthr[i] = thr[i], shist[i] = 0u;
  __syncthreads();
  const double t0 = sthr[0];
  for (long sgi = blockIdx.x; sgi < nseg; sgi += gridDim.x) {
    const long seg = sgi * seg_cap;
    const int n = min(counts[sgi], seg_cap);
    for (int i = threadIdx.x; i < n; i += 256) {
      const double v = (double)val[seg + i];
      if (!(v > t0)) continue;
      if (keep && !keep[idx[seg + i] % S]) continue;
      int lo = 0, hi = nthr;                   // number of thresholds < v
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sthr[mid] < v) lo = mid + 1;
        else hi = mid;
      }
      atomicAdd(&shist[lo - 1], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nthr; i += 256)
    if (shist[i]) atomicAdd(&hist[i], (unsigned long long)shist[i]);
}

// entries above the threshold, appended in any order (the host sorts the few that come back)
__global__ __launch_bounds__(256) void sparse_where_above_kernel(
    const long long *__restrict__ idx, const float *__restrict__ val, const int *__restrict__ counts,
    int seg_cap, double thr, const uint8_t *__restrict__ aux, long cap,
    long long *__restrict__ out_idx, float *__restrict__ out_val, uint8_t *__restrict__ out_aux,
    unsigned long long *__restrict__ total) {
  const long seg = (long)blockIdx.x * seg_cap;
  const int n = min(counts[blockIdx.x], seg_cap);
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = val[seg + i];
    if (!((double)v > thr)) continue;
    const unsigned long long at = atomicAdd(total, 1ull);
    if ((long)at < cap) {
      const long long ii = idx[seg + i];
      out_idx[at] = ii;
      out_val[at] = v;
      if (aux) out_aux[at] = aux[ii];
    }
  }
}

// map[s] = max(map[s], v) for the POSITIVE entries (the map starts at 0: every column of a
// local-maximum cube holds zeros); positive floats order like their bit patterns
__global__ __launch_bounds__(256) void sparse_zmax_kernel(const long long *__restrict__ idx,
                                                          const float *__restrict__ val,
                                                          const int *__restrict__ counts,
                                                          int seg_cap, const uint8_t *__restrict__ keep,
                                                          long S, int *__restrict__ map_bits) {
  const long seg = (long)blockIdx.x * seg_cap;
  const int n = min(counts[blockIdx.x], seg_cap);
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = val[seg + i];
    if (!(v > 0.0f)) continue;
    const long s_ = idx[seg + i] % S;
    if (keep && !keep[s_]) continue;
    atomicMax(&map_bits[s_], __float_as_int(v));
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
                           d_mask, Nz, Ny, Nx, zp, sgn, oa, ob, LmSparse());
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

// ---- sparse form: C ABI ---------------------------------------------------------------------
namespace {

struct LmGeom {
  long bx;
  int zp, nzc;
};

// ORIGIN_LOCALMAX_STRIPS=1: strips of four row groups per block instead of the flattened (row
// group, column) mapping.  Measured (tools/localmax_sparse_time.py): 4.16 against 3.77 ms at
// 3681 x 600 x 600, 9.57 against 8.31 ms at 900 x 900 -- the partly empty last chunk of every row
// (150 float4 columns = 62 + 62 + 26 lanes) costs more than the shared rows save; off by default.
bool lm_strips() {
  static const bool on = getenv("ORIGIN_LOCALMAX_STRIPS") && atoi(getenv("ORIGIN_LOCALMAX_STRIPS")) != 0;
  return on;
}

// the launch geometry of the sparse pass (local_max3s_kernel)
LmGeom lm_geometry(const origin_ctx *ctx, int Nz, int Ny, int Nx) {
  const int R = LMS_R;
  const long threads = (long)cdiv(Ny, R) * (Nx / LMS_NC);
  LmGeom g;
  g.bx = (threads + 4 * 62 - 1) / (4 * 62);  // 62 producing lanes per wave
  if (lm_strips())  // strips of four row groups x chunks of 62 columns
    g.bx = (long)cdiv(cdiv(Ny, R), 4) * cdiv(Nx / LMS_NC, 62);
  int nzc = (int)(((long)ctx->num_cu * 16 + g.bx - 1) / g.bx);  // ~16 blocks per CU
  nzc = nzc < 1 ? 1 : (nzc > cdiv(Nz, 32) ? cdiv(Nz, 32) : nzc);
  g.zp = cdiv(Nz, nzc);
  g.nzc = cdiv(Nz, g.zp);
  return g;
}

}  // namespace

extern "C" {

int origin_local_max_sparse_plan(origin_ctx *ctx, int Nz, int Ny, int Nx, long *nseg, int *seg_cap) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(Nz > 0 && Ny > 0 && Nx > 0 && nseg && seg_cap, "bad arguments");
  *nseg = 0, *seg_cap = 0;
  if ((Nx & 3) != 0 || Nz > 65535 * 32) return ORIGIN_OK;  // no sparse form for this shape
  const LmGeom g = lm_geometry(ctx, Nz, Ny, Nx);
  *nseg = g.bx * g.nzc * 4;
  // a wave sees zp channels of 62 lanes x 16 outputs; white noise has one 3x3x3 maximum in 27
  // voxels, a smoothed cube one in ~70: room for one in 8
  const long per_wave = (long)g.zp * 62 * LMS_NC * LMS_R;
  long cap = (per_wave / 8 + 63) / 64 * 64;
  *seg_cap = (int)(cap < 256 ? 256 : cap);
  return ORIGIN_OK;
}

int origin_local_max_sparse(origin_ctx *ctx, const float *d_correl, const float *d_correl_min,
                            const uint8_t *d_mask, int Nz, int Ny, int Nx, long nseg, int seg_cap,
                            long long *d_idx_max, float *d_val_max, long long *d_idx_min,
                            float *d_val_min, int *d_counts) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_correl && d_correl_min && d_idx_max && d_val_max && d_idx_min && d_val_min &&
                       d_counts, "null pointer");
  ORIGIN_CHECK_ARG(Nz > 0 && Ny > 0 && (Nx & 3) == 0 && Nx > 0, "the sparse form needs Nx % 4 == 0");
  ORIGIN_CHECK_ARG(((uintptr_t)d_correl & 15) == 0 && ((uintptr_t)d_correl_min & 15) == 0 &&
                       ((uintptr_t)d_mask & 3) == 0, "cubes must be 16-byte aligned");
  const LmGeom g = lm_geometry(ctx, Nz, Ny, Nx);
  ORIGIN_CHECK_ARG(nseg == g.bx * g.nzc * 4 && seg_cap > 0,
                   "segments do not match origin_local_max_sparse_plan for this shape");
  LmSparse sp;
  sp.idx0 = d_idx_max, sp.val0 = d_val_max, sp.idx1 = d_idx_min, sp.val1 = d_val_min;
  sp.counts = d_counts, sp.seg_cap = seg_cap;
  ProfScope ps(ctx, K_LOCAL_MAX);
  if (getenv("ORIGIN_LOCALMAX_SPARSE_V1") && LMS_R == 4 && LMS_NC == 4)   // the first form (same segments)
    hipLaunchKernelGGL((local_max3v_kernel<2, 4, true>), dim3((unsigned)g.bx, g.nzc), dim3(256), 0,
                       ctx->stream, d_correl, d_correl_min, d_mask, Nz, Ny, Nx, g.zp, 1.0f,
                       (float *)nullptr, (float *)nullptr, sp);
  else
  {
    // (XCD-contiguous block order: measured 3.39 ms against 3.26 with the natural order at
    // 3681 x 600 x 600 -- the rows two neighbouring blocks share are not what the pass waits for;
    // off unless ORIGIN_LOCALMAX_XCD=1)
    static const int xcd_order = getenv("ORIGIN_LOCALMAX_XCD") ? atoi(getenv("ORIGIN_LOCALMAX_XCD")) : 0;
    // ORIGIN_LOCALMAX_PREFETCH=1: the rows of the plane after next are requested a channel ahead
    // (24 more registers: three waves per SIMD instead of four, twice the loads in flight per wave)
    static const int prefetch = getenv("ORIGIN_LOCALMAX_PREFETCH") ? atoi(getenv("ORIGIN_LOCALMAX_PREFETCH")) : 0;
    const dim3 grid((unsigned)(g.bx * g.nzc * 2));
#define LM_GO(M, P, T)                                                                           \
  hipLaunchKernelGGL((local_max3s_kernel<M, P, T>), grid, dim3(256), 0, ctx->stream, d_correl,   \
                     d_correl_min, d_mask, Nz, Ny, Nx, g.zp, g.bx, g.nzc, xcd_order, sp)
    if (lm_strips()) {
      if (d_mask) LM_GO(true, false, true);
      else LM_GO(false, false, true);
    } else if (d_mask && prefetch) LM_GO(true, true, false);
    else if (d_mask) LM_GO(true, false, false);
    else if (prefetch) LM_GO(false, true, false);
    else LM_GO(false, false, false);
#undef LM_GO
  }
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

int origin_sparse_to_dense(origin_ctx *ctx, const long long *d_idx, const float *d_val,
                           const int *d_counts, long nseg, int seg_cap, float *d_dense, long n) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_idx && d_val && d_counts && d_dense && nseg > 0 && seg_cap > 0 && n > 0,
                   "bad arguments");
  ORIGIN_HIP(hipMemsetAsync(d_dense, 0, (size_t)n * sizeof(float), ctx->stream));
  ProfScope ps(ctx, K_SMALL);
  hipLaunchKernelGGL(sparse_to_dense_kernel, dim3((unsigned)nseg), dim3(256), 0, ctx->stream, d_idx,
                     d_val, d_counts, seg_cap, d_dense);
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

int origin_sparse_count_above(origin_ctx *ctx, const long long *d_idx, const float *d_val,
                              const int *d_counts, long nseg, int seg_cap, const uint8_t *d_keep,
                              long S, int nthr, const double *h_thr, long *h_counts) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_idx && d_val && d_counts && h_thr && h_counts && nseg > 0 && S > 0,
                   "bad arguments");
  ORIGIN_CHECK_ARG(nthr >= 1 && nthr <= 1024, "1..1024 thresholds");
  for (int i = 0; i < nthr; ++i) ORIGIN_CHECK_ARG(h_thr[i] == h_thr[i], "NaN threshold");
  std::vector<int> order(nthr);
  for (int i = 0; i < nthr; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return h_thr[a] < h_thr[b]; });
  std::vector<double> sorted(nthr);
  for (int i = 0; i < nthr; ++i) sorted[i] = h_thr[order[i]];
  void *scr = nullptr;
  const size_t tb = (size_t)nthr * sizeof(double), hb = (size_t)nthr * sizeof(unsigned long long);
  int rc = origin_scratch(ctx, tb + hb, &scr);
  if (rc) return rc;
  double *d_thr = (double *)scr;
  unsigned long long *d_hist = (unsigned long long *)((char *)scr + tb);
  ORIGIN_HIP(hipMemcpyAsync(d_thr, sorted.data(), tb, hipMemcpyHostToDevice, ctx->stream));
  ORIGIN_HIP(hipMemsetAsync(d_hist, 0, hb, ctx->stream));
  {
    ProfScope ps(ctx, K_SMALL);
    const long blocks = nseg < (long)ctx->num_cu * 16 ? nseg : (long)ctx->num_cu * 16;
    hipLaunchKernelGGL(sparse_count_above_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       d_idx, d_val, d_counts, nseg, seg_cap, d_keep, S, nthr, d_thr, d_hist);
  }
  ORIGIN_LAUNCH_CHECK();
  std::vector<unsigned long long> hist(nthr);
  ORIGIN_HIP(hipMemcpyAsync(hist.data(), d_hist, hb, hipMemcpyDeviceToHost, ctx->stream));
  ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
  unsigned long long run = 0;
  for (int j = nthr - 1; j >= 0; --j) {
    run += hist[j];
    h_counts[order[j]] = (long)run;
  }
  return ORIGIN_OK;
}

int origin_sparse_where_above(origin_ctx *ctx, const long long *d_idx, const float *d_val,
                              const int *d_counts, long nseg, int seg_cap, double threshold,
                              const uint8_t *d_aux, long cap, long long *d_out_idx,
                              float *d_out_val, uint8_t *d_out_aux, long *h_count) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_idx && d_val && d_counts && d_out_idx && d_out_val && h_count && nseg > 0 &&
                       cap > 0, "bad arguments");
  ORIGIN_CHECK_ARG(!d_aux || d_out_aux, "aux cube without an output for it");
  void *scr = nullptr;
  int rc = origin_scratch(ctx, sizeof(unsigned long long), &scr);
  if (rc) return rc;
  ORIGIN_HIP(hipMemsetAsync(scr, 0, sizeof(unsigned long long), ctx->stream));
  {
    ProfScope ps(ctx, K_SMALL);
    hipLaunchKernelGGL(sparse_where_above_kernel, dim3((unsigned)nseg), dim3(256), 0, ctx->stream,
                       d_idx, d_val, d_counts, seg_cap, threshold, d_aux, cap, d_out_idx, d_out_val,
                       d_out_aux, (unsigned long long *)scr);
  }
  ORIGIN_LAUNCH_CHECK();
  unsigned long long total = 0;
  ORIGIN_HIP(hipMemcpyAsync(&total, scr, sizeof(total), hipMemcpyDeviceToHost, ctx->stream));
  ORIGIN_HIP(hipStreamSynchronize(ctx->stream));
  *h_count = (long)total;
  return ORIGIN_OK;
}

int origin_sparse_zmax_map(origin_ctx *ctx, const long long *d_idx, const float *d_val,
                           const int *d_counts, long nseg, int seg_cap, const uint8_t *d_keep,
                           long S, float *d_map) {
  ORIGIN_USE(ctx);
  ORIGIN_CHECK_ARG(d_idx && d_val && d_counts && d_map && nseg > 0 && S > 0, "bad arguments");
  ORIGIN_HIP(hipMemsetAsync(d_map, 0, (size_t)S * sizeof(float), ctx->stream));
  ProfScope ps(ctx, K_SMALL);
  hipLaunchKernelGGL(sparse_zmax_kernel, dim3((unsigned)nseg), dim3(256), 0, ctx->stream, d_idx,
                     d_val, d_counts, seg_cap, d_keep, S, (int *)d_map);
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

}  // extern "C"
