// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_f16 on gfx950 for the operand placements
// the GLR spectral kernel can choose between, alone and with VALU work between the MFMAs.
//   MODE 0: builtin (compiler-chosen registers), 2 accumulator chains
//   MODE 1: asm, D/C in VGPRs, B in AGPRs (literal), A in VGPRs, 2 chains
//   MODE 2: as 1 with 5 independent VALU ops after every MFMA
//   MODE 3: as 2 with 8 VALU ops
//   MODE 4: builtin with 5 VALU ops (compiler interleave via sched_group_barrier)
//   MODE 5: as 1 but 4 chains
//   MODE 6: as 1 but ONE chain (every MFMA waits for the previous one)
// Build: hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) a[j] = (_Float16)(threadIdx.x * 0.001f + j), b[j] = (_Float16)(j * 0.5f);
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c)
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = out[(threadIdx.x + i) & 63];
  if (MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6) {
    asm volatile("" ::: "a200", "a255");
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 u = __builtin_bit_cast(u4, b);
    asm volatile("v_accvgpr_write_b32 a200, %0\n\tv_accvgpr_write_b32 a201, %1\n\t"
                 "v_accvgpr_write_b32 a202, %2\n\tv_accvgpr_write_b32 a203, %3\n\ts_nop 4"
                 :: "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]));
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      constexpr int NV = MODE == 2 ? 5 : MODE == 3 ? 8 : MODE == 4 ? 5 : 0;
      const int c = MODE == 5 ? (r & 3) : MODE == 6 ? 0 : (r & 1);
      if (MODE == 0 || MODE == 4) {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
      } else {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[200:203], %0" : "+v"(acc[c]) : "v"(a));
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q] = __builtin_fmaf(v[q], 1.0001f, 0.5f);
      if (MODE == 4) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      } else if (NV) {
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0;
  for (int c = 0; c < 4; ++c)
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) out[0] = s;
}

template <int MODE>
void run(const char *name) {
  float *d;
  hipMalloc(&d, 4096);
  hipMemset(d, 0, 4096);
  const int iters = 4000, blocks = 256;  // one 4-wave block per CU: 1 wave per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 10);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 32;
  printf("%-44s %.3f ms  %.1f ns per MFMA (= %.1f cycles at 2.4 GHz)  %.0f TFLOP/s\n", name, ms,
         ms * 1e6 / n, ms * 1e6 / n * 2.4, 1024.0 * n * 32768 / (ms * 1e-3) / 1e12);
  hipFree(d);
}

// two waves per SIMD (512-thread blocks): ROLE 0 = every wave runs the mixed loop (NV VALU per
// MFMA), ROLE 1 = waves 0-3 MFMA only, waves 4-7 VALU only (NV per "slot"), same totals
template <int ROLE, int NV>
__global__ __launch_bounds__(512, 1) void k2(float *out, int iters) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) a[j] = (_Float16)(threadIdx.x * 0.001f + j), b[j] = (_Float16)(j * 0.5f);
  f32x16 acc[2];
  for (int c = 0; c < 2; ++c)
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = out[(threadIdx.x + i) & 63];
  asm volatile("" ::: "a100", "a103");
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 u = __builtin_bit_cast(u4, b);
  asm volatile("v_accvgpr_write_b32 a100, %0\n\tv_accvgpr_write_b32 a101, %1\n\t"
               "v_accvgpr_write_b32 a102, %2\n\tv_accvgpr_write_b32 a103, %3\n\ts_nop 4"
               :: "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]));
  const bool second = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;
  const bool do_m = ROLE == 0 || !second, do_v = ROLE == 0 || second;
  const int n = ROLE == 0 ? iters / 2 : iters;  // same total work per SIMD in both roles
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      if (do_m)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[100:103], %0" : "+v"(acc[r & 1]) : "v"(a));
      if (do_v) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = __builtin_fmaf(v[q], 1.0001f, 0.5f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int c = 0; c < 2; ++c)
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) out[0] = s;
}

template <int ROLE, int NV>
void run2(const char *name) {
  float *d;
  hipMalloc(&d, 4096);
  hipMemset(d, 0, 4096);
  const int iters = 4000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k2<ROLE, NV><<<blocks, 512>>>(d, 10);
  hipEventRecord(e0);
  k2<ROLE, NV><<<blocks, 512>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 32;
  printf("%-52s %.3f ms  %.1f ns per MFMA of the SIMD\n", name, ms, ms * 1e6 / n);
  hipFree(d);
}

int main() {
  run2<0, 5>("2 waves/SIMD, both mixed 5 VALU per MFMA");
  run2<1, 5>("2 waves/SIMD, one MFMA-only + one VALU-only (5)");
  run2<0, 8>("2 waves/SIMD, both mixed 8 VALU per MFMA");
  run2<1, 8>("2 waves/SIMD, one MFMA-only + one VALU-only (8)");
  run2<1, 0>("2 waves/SIMD, one MFMA-only + one idle");

  run<0>("builtin, 2 chains");
  run<1>("asm D=VGPR B=AGPR, 2 chains");
  run<5>("asm D=VGPR B=AGPR, 4 chains");
  run<6>("asm D=VGPR B=AGPR, 1 chain");
  run<2>("asm + 5 VALU per MFMA");
  run<3>("asm + 8 VALU per MFMA");
  run<4>("builtin + 5 VALU per MFMA (sched_group)");
  return 0;
}
