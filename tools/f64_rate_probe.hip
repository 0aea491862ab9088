// Probe: issue rate of v_fma_f64 / v_fma_f32 / v_pk_fma_f32 per SIMD on gfx950.
//   hipcc --offload-arch=gfx950 -O2 -o tools/f64_rate_probe tools/f64_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, int iters, double s0, double s1) {
  double a[16];
  float f[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i, f[i] = threadIdx.x + i;
  double m = s0;
  float mf = (float)s0;
  long t0 = __builtin_readcyclecounter();
  long c0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (MODE == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(s1));
        if (MODE == 1) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "s"(s1), "v"(m));
        if (MODE == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(mf), "v"(mf));
        if (MODE == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (MODE == 4) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (MODE == 5) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
      }
    }
  }
  long t1 = __builtin_readcyclecounter();
  long c1 = wall_clock64();
  double acc = 0;
  for (int i = 0; i < 16; ++i) acc += a[i] + f[i];
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = (double)(t1 - t0);
    out[1] = (double)(c1 - c0);
  }
  if (acc == 12345.678) out[2] = acc;
}

template <int MODE>
void run(const char *name, int waves_per_simd) {
  double *d;
  hipMalloc(&d, 64);
  const int iters = 20000;
  const int blocks = 256 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 10, 1.0000001, 0.5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0000001, 0.5);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double h[2];
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  const double n = 64.0 * iters;  // instructions per wave
  printf("%-14s waves/SIMD %d: %.2f shader-cycles/inst/wave, %.2f per SIMD-inst; wall %.3f ms -> %.2f ns/inst/SIMD; "
         "clk %.0f MHz (cycle counter / 100 MHz wall clock)\n",
         name, waves_per_simd, h[0] / n, h[0] / n / waves_per_simd, ms,
         ms * 1e6 / (n * waves_per_simd), h[0] / h[1] * 100.0);
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f64 vvv", w);
    run<1>("v_fmac_f64 s", w);
    run<2>("v_fma_f32", w);
    run<3>("v_mul_f64", w);
    run<4>("v_add_f64", w);
    run<5>("v_cvt_f64_f32", w);
  }
  return 0;
}
