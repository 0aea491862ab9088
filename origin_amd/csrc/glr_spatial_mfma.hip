// GLR spatial stage on the matrix cores  (reference _convolve_fsf, lib_origin.py:1027-1043,
// for one field without weight map:  cube_fsf[z] = corr2(cube[z], psf_z - mean(psf_z)), 'same',
// zeros outside the field).
//
// Round 1 put only the x axis of the PSF on the K dimension of the MFMA (banded Toeplitz, 32
// outputs against a 64-column window: 25 of 64 products useful, 300 MFMAs per 1024 outputs).
// Here BOTH axes are: an M tile is a PATCH of 8 (x) by 4 (y) outputs, its K dimension the
// (8 + P - 1) x (4 + P - 1) input window around it (32 x 28 for P = 25), so that for output
// m = (my, mx) and window pixel k = (wy, wx)
//     A[m][k] = psf[wy - my][wx - mx]        (zero outside the P x P support)
//     B[k][n] = in[oy_n + wy][ox_n + wx]     (n = one of 32 patch origins, 8 apart in x, 4 in y)
// -- 625 of 896 products useful, 56 k-steps of 16 = 168 MFMAs (three per k-step, two-term f16
// split as everywhere in the GLR) per 1024 outputs instead of 300.  A wave owns a 32 x 32 output
// area = 4 x 8 patches; a k-step is half a window row (16 consecutive wx: lane half h takes 8).
//   * B fragments are 16-byte reads of an f16 image of the input region in LDS (row pitch =
//     16 mod 64 bytes: the 16 lanes a ds_read_b128 serves at once fall on distinct banks);
//   * A fragments are 16-byte reads of a per-channel table: for every kernel row 8 copies of
//     the zero-padded row shifted by 0..7 elements (entry pitch 80 bytes: conflict free), rows
//     -3..-1 and P..P+2 all zero for the outputs of a patch whose my puts wy - my outside.
//
// The PSF differs per channel, so image AND table change with every channel.  A block (8 waves
// = two groups of 4) owns a 64 x 64 region and marches z; the groups take alternate channels and
// alternate ROLES, separated by one block barrier per phase: while one group issues the 168
// MFMAs of its channel (one wave per SIMD: the matrix pipe's full rate), the other converts its
// next channel's input tile to the f16 hi / lo images and builds its table (VALU + LDS writes:
// ~2200 cycles against ~5400), then they swap.  Global loads of the tile after next and its
// taps are issued at the start of a group's MFMA phase and land in registers; the tile's
// power-of-two scale (max |x|) is reduced at the end of that phase, so the conversion phase
// starts with everything at hand.  LDS: 2 x (36.6 KB images + 39.7 KB table + 2.5 KB taps).
//
// TERMS = 1: bf16 operands, one MFMA per k-step (BASELINE config 4's bf16 GLR), no scaling.
//
// Eligible shapes: odd P from 5 to 25 (float4 tile loads when P/2 is a multiple of four; any
// field size: 16-byte accesses when Nx % 4 == 0, element
// accesses otherwise), one field or a mosaic of weighted fields (WEIGHTED); other PSF sizes stay
// on spatial4x4_kernel / spatial_kernel.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace {

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int S2_R = 64;           // region side (outputs) of a block
constexpr int S2_TAP_LOG2 = 12;    // f16 taps are stored times 2^12
constexpr int S2_ENTRY = 80;       // bytes of one (row, copy) table entry: 40 taps, 5 groups of 8

// -DS2_TIMING: clock64 stamps of block (1, 1, 0), phases 10..25, every wave
// (tools/s2_phase_times.py; each stamp costs an s_memtime round trip and drains the LDS queue:
// read the numbers as shares, not as cycle counts)
#ifdef S2_TIMING
__device__ long long s2_tim[16 * 8 * 8];
#define S2_STAMP(k)                                                                             \
  if (blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 0 && lane == 0 && p >= 10 && p < 26) \
  s2_tim[((p - 10) * 8 + wave) * 8 + (k)] = clock64()
#else
#define S2_STAMP(k)
#endif

template <int P>
struct S2Geom {
  static constexpr int H = P - 1;                       // halo
  // input tile; its width rounded up to whole float4s (P - 1 not a multiple of four: up to two
  // extra columns of real data that only ever meet zero taps)
  static constexpr int IW = (S2_R + H + 3) / 4 * 4, IH = S2_R + H;
  static constexpr int WROWS = 4 + H;                   // window rows of a patch
  static constexpr int WCOLS = 8 + H;                   // window columns (<= 32)
  static constexpr int KROW = (WCOLS + 15) / 16;        // k-steps per window row
  static constexpr int NKS = WROWS * KROW;              // k-steps per channel
  static constexpr int PITCH = ((IW * 2 + 63) / 64) * 64 + 16;  // bytes, = 16 mod 64, > 2 IW
  static constexpr int IMG = IH * PITCH;                // one f16 / bf16 image
  static constexpr int TROWS = P + 6;                   // table rows: dy = -3 .. P + 2
  static constexpr int TAB = TROWS * 8 * S2_ENTRY;      // one table (hi or lo)
  static constexpr int NQ = (IW / 4 * IH + 255) / 256;  // staged float4 per thread of a group
  static constexpr int NT = (P * P + 255) / 256;        // staged taps per thread
  // (staged taps, row pitch P: the table build's reads of four rows per 32-lane group meet on banks
  // for P = 25; a pitch of 40 floats -- banks 8 dy + copy, all distinct -- was measured in round 3:
  // 8.15-8.2 ms either way, the conversion group's conflicts are not on the critical path)
  static constexpr int TAPS = (P * P * 4 + 15) / 16 * 16;
  static_assert(IW % 4 == 0 && WCOLS <= 32 && PITCH >= 2 * IW + 16, "geometry");
};

template <int P, int TERMS>
constexpr size_t s2_group_bytes() {
  using G = S2Geom<P>;
  return (size_t)(TERMS == 3 ? 2 : 1) * (G::IMG + G::TAB) + G::TAPS + 16;
}

// two-term split of y into f16 hi + lo
__device__ __forceinline__ void s2_split(float y, _Float16 &hi, _Float16 &lo) {
  hi = (_Float16)y;
  lo = (_Float16)(y - (float)hi);
}

// VEC: Nx % 4 == 0 -- rows of the cube are 16-byte aligned, the tile is loaded and the outputs are
// stored four at a time; otherwise element by element.
// WEIGHTED: one field of a mosaic (lib_origin.py:1029-1031, :1134-1147): the input is cube * W
// (W [Ny][Nx], the field's weight map, multiplied in while the tile is staged) and, with accf,
// the result is added to what the fields before left in `out`.
// SOLO (round 4): a block is ONE group of four waves (256 threads, half the LDS) that converts and
// multiplies its channels one after the other, and a CU holds TWO such blocks.  The two-group
// block keeps its groups in lock step (one block barrier per phase): while one group runs its MFMA
// phase (~9.6 k cycles: 1.2 k of loads in front, 6.3 k of k-steps, 1.5 k of scale reduction and
// stores behind) the other converts (2.2 k) and then WAITS at the barrier -- one channel per 9.6 k
// cycles and CU, the matrix pipe ~55 % busy.  Two independent blocks drift apart and fill each
// other's gaps: everything around a block's k-step loop runs beside the other block's MFMAs.
template <int P, int TERMS, bool VEC, bool WEIGHTED, bool SOLO = false>
__global__ __launch_bounds__(SOLO ? 256 : 512, SOLO ? 2 : 1) void spatial2_kernel(const float *__restrict__ A,
                                                          const float *__restrict__ W,
                                                          const float *__restrict__ taps, int Nz,
                                                          int Ny, int Nx, int zper, int accf,
                                                          float *__restrict__ out, int ry0,
                                                          int rx0) {
  // (ry0, rx0: first row / column of 64 x 64 regions of this launch -- a run may be split into
  // rectangles of regions)
  using G = S2Geom<P>;
  extern __shared__ __align__(16) char s2_lds[];
  constexpr int c = P / 2;
  constexpr size_t GB = s2_group_bytes<P, TERMS>();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = SOLO ? 0 : wave >> 2, gw = wave & 3, gt = tid & 255;  // group, wave / thread in group
  char *base = s2_lds + grp * GB;
  char *img_h = base, *img_l = base + (TERMS == 3 ? G::IMG : 0);
  char *tab_h = base + (TERMS == 3 ? 2 : 1) * G::IMG, *tab_l = tab_h + (TERMS == 3 ? G::TAB : 0);
  float *tapst = reinterpret_cast<float *>(tab_h + (TERMS == 3 ? 2 : 1) * G::TAB);
  unsigned *maxw = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(tapst) + G::TAPS);

  const int x0 = (blockIdx.x + rx0) * S2_R, y0 = (blockIdx.y + ry0) * S2_R;
  const long S = (long)Ny * Nx;
  const int z0 = blockIdx.z * zper, z1 = min(Nz, z0 + zper);
  const int n = lane & 31, h = lane >> 5;

  // ---- zero everything once: table rows outside the PSF stay zero for good, image pads finite
  for (int i = tid; i < (int)((SOLO ? 1 : 2) * GB / 16); i += (SOLO ? 256 : 512))
    reinterpret_cast<uint4 *>(s2_lds)[i] = make_uint4(0u, 0u, 0u, 0u);

  // ---- register staging of this group's NEXT channel: input tile, taps
  float4 stage[G::NQ];
  float tapreg[G::NT];
  auto prefetch = [&](int z) {
    const float *Az = A + (long)z * S;
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
      const int e = gt + 256 * q;
      const int ry = e / (G::IW / 4), cx = e - ry * (G::IW / 4);
      const int y = y0 - c + ry, x = x0 - c + 4 * cx;  // x % 4 == 0 (c % 4 == 0), Nx % 4 == 0
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (VEC) {
        if (ry < G::IH && y >= 0 && y < Ny && x >= 0 && x < Nx) {
          v = *reinterpret_cast<const float4 *>(Az + (long)y * Nx + x);
          if constexpr (WEIGHTED) {
            const float4 w = *reinterpret_cast<const float4 *>(W + (long)y * Nx + x);
            v.x *= w.x, v.y *= w.y, v.z *= w.z, v.w *= w.w;
          }
        }
      } else if (ry < G::IH && y >= 0 && y < Ny) {
        const float *row = Az + (long)y * Nx;
        if (x >= 0 && x < Nx) v.x = row[x];
        if (x + 1 >= 0 && x + 1 < Nx) v.y = row[x + 1];
        if (x + 2 >= 0 && x + 2 < Nx) v.z = row[x + 2];
        if (x + 3 >= 0 && x + 3 < Nx) v.w = row[x + 3];
        if constexpr (WEIGHTED) {
          const float *wr = W + (long)y * Nx;
          if (x >= 0 && x < Nx) v.x *= wr[x];
          if (x + 1 >= 0 && x + 1 < Nx) v.y *= wr[x + 1];
          if (x + 2 >= 0 && x + 2 < Nx) v.z *= wr[x + 2];
          if (x + 3 >= 0 && x + 3 < Nx) v.w *= wr[x + 3];
        }
      }
      stage[q] = v;
    }
    const float *kz = taps + (long)z * P * P;
#pragma unroll
    for (int q = 0; q < G::NT; ++q) tapreg[q] = gt + 256 * q < P * P ? kz[gt + 256 * q] : 0.f;
  };
  // max |x| of the staged tile -> maxw[slot]; taps -> LDS staging (both read after a barrier)
  auto publish = [&](int slot) {
    if constexpr (TERMS == 3) {
      float m = 0.f;
#pragma unroll
      for (int q = 0; q < G::NQ; ++q)
        m = fmaxf(fmaxf(m, fmaxf(fabsf(stage[q].x), fabsf(stage[q].y))),
                  fmaxf(fabsf(stage[q].z), fabsf(stage[q].w)));
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      if (lane == 0) atomicMax(&maxw[slot], __float_as_uint(m));
    }
#pragma unroll
    for (int q = 0; q < G::NT; ++q)
      if (gt + 256 * q < P * P) tapst[gt + 256 * q] = tapreg[q];
  };

  // ---- conversion phase: staged tile -> f16 hi / lo (or bf16) image; taps -> fragment table
  int p = 0;  // phase counter (kernel scope: the timing stamps inside the lambdas name it)
  auto convert = [&](int slot, float &inv_out) {
    float scale = 1.f, inv = 1.f;
    if constexpr (TERMS == 3) {
      const int ex = (int)((maxw[slot] >> 23) & 0xffu);
      const bool tiny = ex < 40 || ex == 255;  // zero / denormal-small / non-finite: no scaling
      scale = __uint_as_float((unsigned)(tiny ? 127 : 268 - ex) << 23);  // max |y| in [2^14, 2^15)
      inv = __uint_as_float((unsigned)(tiny ? 127 - S2_TAP_LOG2 : ex - 14 - S2_TAP_LOG2) << 23);
      if (gt == 0) maxw[slot ^ 1] = 0u;  // the word the NEXT publish of this group adds to
    }
    inv_out = inv;
    S2_STAMP(4);
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
      const int e = gt + 256 * q;
      const int ry = e / (G::IW / 4), cx = e - ry * (G::IW / 4);
      if (ry < G::IH) {
        const float v[4] = {stage[q].x * scale, stage[q].y * scale, stage[q].z * scale,
                            stage[q].w * scale};
        if constexpr (TERMS == 3) {
          f16x4 vh, vl;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            _Float16 a, b;
            s2_split(v[j], a, b);
            vh[j] = a, vl[j] = b;
          }
          *reinterpret_cast<f16x4 *>(img_h + ry * G::PITCH + cx * 8) = vh;
          *reinterpret_cast<f16x4 *>(img_l + ry * G::PITCH + cx * 8) = vl;
        } else {
          bf16x2 a, b;
          a[0] = (__bf16)v[0], a[1] = (__bf16)v[1], b[0] = (__bf16)v[2], b[1] = (__bf16)v[3];
          u32x2v w;
          w[0] = __builtin_bit_cast(unsigned, a), w[1] = __builtin_bit_cast(unsigned, b);
          *reinterpret_cast<u32x2v *>(img_h + ry * G::PITCH + cx * 8) = w;
        }
      }
    }
    S2_STAMP(5);
    // table entry (row t = dy + 3, copy cp): G_dy[e + cp], e = 0..39, G_dy[e] = k[dy][e - 7]
    for (int ent = gt; ent < P * 8; ent += 256) {
      const int dy = ent >> 3, cp = ent & 7;
      char *dst_h = tab_h + ((dy + 3) * 8 + cp) * S2_ENTRY;
      char *dst_l = tab_l + ((dy + 3) * 8 + cp) * S2_ENTRY;
#pragma unroll
      for (int g8 = 0; g8 < 5; ++g8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int d = 8 * g8 + j + cp - 7;
          // branch-free: all eight LDS reads of a group are in flight together
          const float t0 = tapst[dy * P + min(max(d, 0), P - 1)];
          t[j] = (d >= 0 && d < P) ? t0 : 0.f;
        }
        if constexpr (TERMS == 3) {
          f16x8 gh, gl;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            _Float16 a, b;
            s2_split(t[j] * (float)(1 << S2_TAP_LOG2), a, b);
            gh[j] = a, gl[j] = b;
          }
          *reinterpret_cast<f16x8 *>(dst_h + 16 * g8) = gh;
          *reinterpret_cast<f16x8 *>(dst_l + 16 * g8) = gl;
        } else {
          bf16x8 gb;
#pragma unroll
          for (int j = 0; j < 8; ++j) gb[j] = (__bf16)t[j];
          *reinterpret_cast<bf16x8 *>(dst_h + 16 * g8) = gb;
        }
      }
    }
  };

  // ---- MFMA phase: this wave's 32 x 32 area (ax, ay) of the region
  const int ax = gw & 1, ay = gw >> 1;
  const int ox = 8 * (n & 3), oy = 4 * (n >> 2);  // patch origin of column n
  const int mx = n & 7, my = n >> 3;              // output of row m = n (A operand: lane = row)
  // A: entry (row wy - my + 3, copy 7 - mx), group 2 half + h;  B: image row 32 ay + oy + wy,
  // columns 32 ax + ox + 16 half + 8 h + j
  const int a_off = ((3 - my) * 8 + (7 - mx)) * S2_ENTRY + 16 * h;
  const int b_off = (32 * ay + oy) * G::PITCH + (32 * ax + ox + 8 * h) * 2;
  auto mfma_phase = [&](int z, float inv) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const char *ap = tab_h + a_off, *bp = img_h + b_off;
    constexpr int LOA = G::TAB, LOB = G::IMG;
    constexpr int DEPTH = 2;  // k-steps of fragments in flight ahead of the MFMAs
    u32x4v fah[DEPTH + 1], fal[DEPTH + 1], fbh[DEPTH + 1], fbl[DEPTH + 1];
    auto fetch = [&](int ks, int slot) {
      const int wy = ks / G::KROW, half = ks - wy * G::KROW;
      const int ao = wy * 8 * S2_ENTRY + half * 32, bo = wy * G::PITCH + half * 32;
      fah[slot] = *reinterpret_cast<const u32x4v *>(ap + ao);
      fbh[slot] = *reinterpret_cast<const u32x4v *>(bp + bo);
      if constexpr (TERMS == 3) {
        fal[slot] = *reinterpret_cast<const u32x4v *>(ap + ao + LOA);
        fbl[slot] = *reinterpret_cast<const u32x4v *>(bp + bo + LOB);
      }
    };
#pragma unroll
    for (int ks = 0; ks < DEPTH; ++ks) fetch(ks, ks);
#pragma unroll
    for (int ks = 0; ks < G::NKS; ++ks) {
      const int cur = ks % (DEPTH + 1);
      if (ks + DEPTH < G::NKS) fetch(ks + DEPTH, (ks + DEPTH) % (DEPTH + 1));
      // asm volatile statements keep their order, and the fragment reads above stay where they
      // are written: DEPTH k-steps ahead of their use (with the builtins hipcc sinks every
      // ds_read to just in front of its MFMA and waits for it there)
      if constexpr (TERMS == 3) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(fah[cur]), "v"(fbh[cur]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(fah[cur]), "v"(fbl[cur]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(fal[cur]), "v"(fbh[cur]));
      } else {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(fah[cur]), "v"(fbh[cur]));
      }
    }
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc));  // last MFMA result -> VALU readers
    // store: accumulator register i of lane (n, h) is output m = (i&3) + 8 (i>>2) + 4 h of patch
    // n, i.e. patch row i >> 2, patch columns 4 h + (i & 3): one float4 per patch row
    const int xo = x0 + 32 * ax + ox + 4 * h;
    if (xo < Nx) {
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const int y = y0 + 32 * ay + oy + pr;
        if (y >= Ny) continue;
        float *o = out + (long)z * S + (long)y * Nx + xo;
        if constexpr (VEC) {  // Nx % 4 == 0: a float4 is inside or outside as a whole
          float4 r = make_float4(acc[4 * pr] * inv, acc[4 * pr + 1] * inv, acc[4 * pr + 2] * inv,
                                 acc[4 * pr + 3] * inv);
          if constexpr (WEIGHTED) {
            if (accf) {
              const float4 b = *reinterpret_cast<const float4 *>(o);
              r.x += b.x, r.y += b.y, r.z += b.z, r.w += b.w;
            }
          }
          *reinterpret_cast<float4 *>(o) = r;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (xo + e < Nx) {
              float r = acc[4 * pr + e] * inv;
              if constexpr (WEIGHTED) {
                if (accf) r += o[e];
              }
              o[e] = r;
            }
        }
      }
    }
  };

  // ---- schedule.  Group g takes channels z0 + g + 2 i; in phase p it converts channel i when
  // p = 2 i + g and runs the MFMAs of channel i when p = 2 i + g + 1.  One barrier per phase.
  const int nch = z1 - z0;
  if constexpr (SOLO) {
    // one group, every channel of the chunk: convert (tile and taps are in registers / staging),
    // barrier (image and table complete), request the next channel's tile and taps, MFMAs + stores,
    // publish the next tile's maximum and taps, barrier (everybody is done reading image and table)
    __syncthreads();  // zero fill done
    if (nch > 0) {
      prefetch(z0);
      publish(0);
    }
    __syncthreads();
    float inv_s = 1.f;
    for (int i = 0; i < nch; ++i) {
      p = i;
      convert(i & 1, inv_s);
      __syncthreads();
      if (i + 1 < nch) prefetch(z0 + i + 1);
      mfma_phase(z0 + i, inv_s);
      if (i + 1 < nch) publish((i + 1) & 1);
      __syncthreads();
    }
    return;
  }
  const int ng = (nch - grp + 1) / 2;  // channels of this group
  __syncthreads();                     // zero fill done
  if (ng > 0) {
    prefetch(z0 + grp);
    publish(0);
  }
  __syncthreads();
  float inv_cur = 1.f;
  const int nphase = 2 * ((nch + 1) / 2) + 2;
  for (p = 0; p < nphase; ++p) {
    const int q = p - grp;
    S2_STAMP(0);
    if (q >= 0) {
      const int i = q >> 1;
      if ((q & 1) == 0) {
        if (i < ng) convert(i & 1, inv_cur);
      } else if (i < ng) {
        if (i + 1 < ng) prefetch(z0 + grp + 2 * (i + 1));
        S2_STAMP(1);
        mfma_phase(z0 + grp + 2 * i, inv_cur);
        S2_STAMP(2);
        if (i + 1 < ng) publish((i + 1) & 1);
      }
    }
    S2_STAMP(3);
    __syncthreads();
  }
}

}  // namespace

// 1 if this shape can run on spatial2_kernel
int origin_spatial_mfma_ok(int Ny, int Nx, int P) {
  return P >= 5 && P <= 25 && (P & 1) && Nx >= 1 && Ny >= 1;
}

template <int P, int TERMS, bool VEC, bool WEIGHTED>
static int s2_launch(origin_ctx *ctx, const float *A, const float *W, const float *taps, int Nz,
                     int Ny, int Nx, int accf, float *out, int ry0, int nry, int rx0, int nrx) {
  // ORIGIN_GLR_SPATIAL_SOLO=0: the two-group block of rounds 2-3
  static const bool solo = !(getenv("ORIGIN_GLR_SPATIAL_SOLO") && atoi(getenv("ORIGIN_GLR_SPATIAL_SOLO")) == 0);
  if (solo) {
    const size_t lds1 = s2_group_bytes<P, TERMS>();
    static OriginPerDeviceOnce attr1;
    ORIGIN_ONCE_PER_DEVICE(ctx, attr1,
                           ORIGIN_HIP(hipFuncSetAttribute(
                               (const void *)spatial2_kernel<P, TERMS, VEC, WEIGHTED, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1)));
    if (nry <= 0) ry0 = 0, nry = cdiv(Ny, S2_R);
    if (nrx <= 0) rx0 = 0, nrx = cdiv(Nx, S2_R);
    const long regions = (long)nrx * nry;
    const int slots = 2 * std::max(1, ctx->num_cu);  // two blocks per CU at a time
    int best_nzb = 1;
    double best_eff = 0.0;
    for (int nzb = 1; nzb <= std::max(1, Nz / 16); ++nzb) {
      const int zp = cdiv(Nz, nzb);
      const long blocks = regions * cdiv(Nz, zp);
      const long rounds = (blocks + slots - 1) / slots;
      // useful channel slots / (rounds x chunk length x block slots), ~2 channels of start-up per block
      const double eff = (double)regions * Nz / ((double)rounds * slots * (zp + 2));
      if (eff > best_eff) best_eff = eff, best_nzb = nzb;
    }
    const int zper = cdiv(Nz, best_nzb);
    dim3 grid(nrx, nry, cdiv(Nz, zper));
    hipLaunchKernelGGL((spatial2_kernel<P, TERMS, VEC, WEIGHTED, true>), grid, dim3(256), lds1,
                       ctx->stream, A, W, taps, Nz, Ny, Nx, zper, accf, out, ry0, rx0);
    ORIGIN_LAUNCH_CHECK();
    return ORIGIN_OK;
  }
  const size_t lds = 2 * s2_group_bytes<P, TERMS>();
  static OriginPerDeviceOnce attr_once;
  ORIGIN_ONCE_PER_DEVICE(ctx, attr_once,
                         ORIGIN_HIP(hipFuncSetAttribute(
                             (const void *)spatial2_kernel<P, TERMS, VEC, WEIGHTED>,
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)));
  // one block per CU at a time (LDS): choose the number of z chunks so that the blocks fill
  // whole rounds of the chip (an even number of channels per chunk keeps both groups busy)
  if (nry <= 0) ry0 = 0, nry = cdiv(Ny, S2_R);
  if (nrx <= 0) rx0 = 0, nrx = cdiv(Nx, S2_R);
  const long regions = (long)nrx * nry;
  const int ncu = std::max(1, ctx->num_cu);
  int best_nzb = 1;
  double best_eff = 0.0;
  for (int nzb = 1; nzb <= std::max(1, Nz / 32); ++nzb) {
    int zp = cdiv(Nz, nzb);
    zp += zp & 1;
    const long blocks = regions * cdiv(Nz, zp);
    const long rounds = (blocks + ncu - 1) / ncu;
    // useful channel slots / (rounds x chunk length x CUs), with a per-block cost of ~3 channels
    const double eff = (double)regions * Nz / ((double)rounds * ncu * (zp + 3));
    if (eff > best_eff) best_eff = eff, best_nzb = nzb;
  }
  int zper = cdiv(Nz, best_nzb);
  zper += zper & 1;
  dim3 grid(nrx, nry, cdiv(Nz, zper));
  hipLaunchKernelGGL((spatial2_kernel<P, TERMS, VEC, WEIGHTED>), grid, dim3(512), lds, ctx->stream,
                     A, W, taps, Nz, Ny, Nx, zper, accf, out, ry0, rx0);
  ORIGIN_LAUNCH_CHECK();
  return ORIGIN_OK;
}

// W: weight map of the field or NULL; accf: add to `out` (fields after the first)
int origin_spatial_mfma_launch(origin_ctx *ctx, int terms, const float *A, const float *W,
                               const float *taps, int Nz, int Ny, int Nx, int P, int accf,
                               float *out, int ry0, int nry, int rx0, int nrx) {
  // (tile loads start at x0 - P/2: float4-aligned only when P/2 is a multiple of four)
  const bool vec = (Nx & 3) == 0 && ((P / 2) & 3) == 0;
  if (!W && accf) {
    origin_set_error("spatial MFMA kernel: accumulation needs a weight map");
    return ORIGIN_E_ARG;
  }
#define S2_GO(PP, TT, VV)                                                                      \
  return W ? s2_launch<PP, TT, VV, true>(ctx, A, W, taps, Nz, Ny, Nx, accf, out, ry0, nry, rx0,  \
                                         nrx)                                                      \
           : s2_launch<PP, TT, VV, false>(ctx, A, nullptr, taps, Nz, Ny, Nx, 0, out, ry0, nry, rx0, \
                                          nrx)
#define S2_CASE(PP)                        \
  case PP:                                 \
    if (vec) {                             \
      if (terms == 3) S2_GO(PP, 3, true);  \
      S2_GO(PP, 1, true);                  \
    }                                      \
    if (terms == 3) S2_GO(PP, 3, false);   \
    S2_GO(PP, 1, false)
#define S2_CASE_E(PP) /* P/2 not a multiple of four: element-wise tile loads only */ \
  case PP:                                                                          \
    if (terms == 3) S2_GO(PP, 3, false);                                            \
    S2_GO(PP, 1, false)
  switch (P) {
    S2_CASE_E(5);
    S2_CASE_E(7);
    S2_CASE(9);
    S2_CASE_E(11);
    S2_CASE_E(13);
    S2_CASE_E(15);
    S2_CASE(17);
    S2_CASE_E(19);
    S2_CASE_E(21);
    S2_CASE_E(23);
    S2_CASE(25);
  }
#undef S2_CASE_E
#undef S2_CASE
#undef S2_GO
  origin_set_error("spatial MFMA kernel: PSF size %d not supported", P);
  return ORIGIN_E_ARG;
}

// MFMA instructions one launch issues (what SQ_INSTS_MFMA counts): every 64 x 64 region of the
// field (partial ones at the edges included) runs, per channel, four waves of
// (4 + P - 1) x ceil((8 + P - 1) / 16) k-steps with `terms` MFMAs each.
long origin_spatial_mfma_count(int terms, int Nz, int Ny, int Nx, int P) {
  const long regions = (long)cdiv(Nx, S2_R) * cdiv(Ny, S2_R);
  const long nks = (long)(4 + P - 1) * ((8 + P - 1 + 15) / 16);
  return regions * Nz * 4 * nks * terms;
}

#ifdef S2_TIMING
extern "C" int origin_debug_s2_timing(long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(s2_tim), sizeof(long long) * 16 * 8 * 8);
}
#endif
