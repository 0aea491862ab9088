// Micro-benchmark: fp32 FMA issue rate on gfx950 for (a) VGPR operands, (b) one SGPR
// operand, (c) v_pk_fma_f32.  Build: hipcc -O3 --offload-arch=gfx950 fma_rate.hip -o fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *sc, int iters) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  float b = out[threadIdx.x & 7];
  const float s0 = sc[0], s1 = sc[1];  // uniform -> SGPR
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8];
  for (int i = 0; i < 8; ++i) p[i] = (f2){a[2 * i], a[2 * i + 1]};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], b, b);
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], s0, b);
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], (f2){b, b}, (f2){s1, s1});
    }
  }
  float acc = 0;
  for (int i = 0; i < 16; ++i) acc += a[i];
  for (int i = 0; i < 8; ++i) acc += p[i].x + p[i].y;
  if (acc == 123.456f) out[0] = acc;
}

template <int MODE>
void run(const char *name, int wavesPerSimd) {
  float *d, *sc;
  hipMalloc(&d, 4096);
  hipMalloc(&sc, 64);
  hipMemset(d, 0, 4096);
  hipMemset(sc, 0, 64);
  const int iters = 20000;
  const int blocks = 256 * wavesPerSimd;  // 256 threads = 4 waves = 1 wave per SIMD per block
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, sc, 10);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, sc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double fma = (double)blocks * 256 * iters * 128.0;
  printf("%-28s waves/SIMD %d : %.2f ms  %.1f TFLOP/s  (%.1f FMA lanes/clk/CU at 2.4 GHz)\n", name,
         wavesPerSimd, ms, 2 * fma / ms / 1e9, fma / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32 vgpr", w);
    run<1>("v_fma_f32 sgpr operand", w);
    run<2>("v_pk_fma_f32", w);
  }
  return 0;
}
