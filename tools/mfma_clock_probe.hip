// What core clock does an MI355X hold under a dense MFMA load?  Every wave runs a chain of
// v_mfma_f32_32x32x16_f16 for a few milliseconds; clock64() (s_memtime: core clock) against
// wall_clock64() (constant 100 MHz) gives the frequency each block saw, the MFMA count the rate.
//   hipcc --offload-arch=gfx950 -O3 mfma_clock_probe.hip -o mfma_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <bool MFMA>
__global__ __launch_bounds__(256) void burn(int iters, float *out, long long *stat, float seed) {
  f32x16 a0, a1;
  for (int i = 0; i < 16; ++i) a0[i] = 0.f, a1[i] = 0.f;
  f16x8 x, y;
  for (int i = 0; i < 8; ++i) x[i] = (_Float16)(seed * (threadIdx.x + i)), y[i] = (_Float16)(seed * (i + 1));
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (MFMA) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 64; ++k) a0[k & 15] = fmaf(a0[k & 15], 1.0001f, seed);
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) stat[2 * blockIdx.x] = c1 - c0, stat[2 * blockIdx.x + 1] = w1 - w0;
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[3];
}

template <bool MFMA>
void run(const char *name, int iters, float seed) {
  const int nb = 256 * 2;  // 2 blocks of 4 waves per CU: 2 waves per SIMD
  float *out; long long *stat;
  hipMalloc(&out, nb * 256 * 4); hipMalloc(&stat, nb * 16);
  burn<MFMA><<<nb, 256>>>(iters / 10, out, stat, seed);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  burn<MFMA><<<nb, 256>>>(iters, out, stat, seed);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(2 * nb); hipMemcpy(h.data(), stat, nb * 16, hipMemcpyDeviceToHost);
  std::vector<double> mhz(nb);
  for (int i = 0; i < nb; ++i) mhz[i] = 100.0 * h[2 * i] / (double)h[2 * i + 1];
  std::sort(mhz.begin(), mhz.end());
  const double mf = MFMA ? (double)iters * 16 * nb * 4 : 0;  // wave-level MFMAs
  printf("%-34s %7.2f ms  core clock min %.0f median %.0f max %.0f MHz", name, ms, mhz[0], mhz[nb / 2], mhz[nb - 1]);
  if (MFMA) printf("  %.0f TFLOP/s dense f16 (%.1f cycles per MFMA and SIMD at the median clock)",
                   mf * 32768 * 2 / 2 / (ms * 1e-3) / 1e12, ms * 1e-3 * mhz[nb / 2] * 1e6 / (mf / 1024));
  printf("\n");
  hipFree(out); hipFree(stat);
}

int main() {
  run<false>("light VALU loop", 200000, 0.5f);
  run<true>("MFMA chain, zero operands", 60000, 0.f);
  run<true>("MFMA chain, varied operands", 60000, 0.37f);
  run<true>("MFMA chain, varied operands (long)", 400000, 0.37f);
  return 0;
}
