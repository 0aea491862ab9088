// Micro-benchmark (gfx950): how much VALU epilogue work hides behind v_mfma_f32_32x32x16_f16 when
// both are issued by ONE wave in a hand-placed order, in real shader cycles (s_memtime), for the
// instruction mixes the GLR spectral epilogue can choose between.
//   KIND 0: MFMA only
//   KIND 1: N x v_mul_f32 reading an older accumulator
//   KIND 2: N x v_pk_mul_f32
//   KIND 3: N x "exact" epilogue of one output: mul, cmp_gt, cndmask, max, min      (5 ops)
//   KIND 4: N x "keyed" epilogue of one output for TWO profiles: 2 mul, 2 and_or, max3 key,
//           max3 best, min3 worst                                                  (7 ops)
//   KIND 5: N x v_max3_f32 (independent)
// Three accumulators in rotation: the VALU reads the one written two MFMAs earlier.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int N, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(float *out, unsigned long long *cyc, int iters) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j)
    a[j] = (_Float16)((threadIdx.x & 63) * 0.01f + j), b[j] = (_Float16)(j * 0.25f - 1.f);
  f32x16 acc[3];
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float best[16], worst[16], f[16], key[16];
  int bk[16];
  for (int i = 0; i < 16; ++i) {
    best[i] = -1e30f, worst[i] = 1e30f, bk[i] = 0, key[i] = -1e30f;
    f[i] = out[(threadIdx.x + i) & 63] + 1.0f;
  }
  unsigned maskv = 0xffffffe0u;
  asm volatile("" : "+v"(maskv));
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      constexpr int dummy = 0;
      f32x16 &d = acc[r % 3];
      const f32x16 &s = acc[(r + 1) % 3];
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
#pragma unroll
      for (int q = 0; q < N; ++q) {
        const int i = (r * N + q) & 15;
        if (KIND == 1) {
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(best[i]) : "v"(s[i]), "v"(f[i]));
        } else if (KIND == 2) {
          f32x2 t;
          asm volatile("v_pk_mul_f32 %0, %1, %2"
                       : "=v"(t)
                       : "v"((f32x2){s[i & 14], s[(i & 14) + 1]}), "v"((f32x2){f[i & 14], f[(i & 14) + 1]}));
          best[i & 14] = t.x, best[(i & 14) + 1] = t.y;
        } else if (KIND == 3) {
          float T;
          asm volatile(
              "v_mul_f32 %0, %4, %5\n\t"
              "v_cmp_gt_f32 vcc, %0, %1\n\t"
              "v_cndmask_b32 %2, %2, %6, vcc\n\t"
              "v_max_f32 %1, %1, %0\n\t"
              "v_min_f32 %3, %3, %0"
              : "=&v"(T), "+v"(best[i]), "+v"(bk[i]), "+v"(worst[i])
              : "v"(s[i]), "v"(f[i]), "v"(r)
              : "vcc");
        } else if (KIND == 4) {
          float T0, T1, K0, K1;
          asm volatile(
              "v_mul_f32 %0, %7, %8\n\t"
              "v_mul_f32 %1, %9, %8\n\t"
              "v_and_or_b32 %2, %0, %10, %11\n\t"
              "v_and_or_b32 %3, %1, %10, %12\n\t"
              "v_max3_f32 %4, %4, %2, %3\n\t"
              "v_max3_f32 %5, %5, %0, %1\n\t"
              "v_min3_f32 %6, %6, %0, %1"
              : "=&v"(T0), "=&v"(T1), "=&v"(K0), "=&v"(K1), "+v"(key[i]), "+v"(best[i]), "+v"(worst[i])
              : "v"(s[i]), "v"(f[i]), "v"(s[(i + 1) & 15]), "v"(maskv), "s"(r), "s"(r + 1));
        } else if (KIND == 5) {
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(best[i]) : "v"(s[i]), "v"(f[i]));
        }
      }
      (void)dummy;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0;
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < 16; ++i) sum += acc[c][i];
  for (int i = 0; i < 16; ++i) sum += best[i] + worst[i] + key[i] + bk[i];
  if (sum == 123.456f) out[0] = sum;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int N, int WAVES>
void run(const char *name, int valu_per_mfma) {
  float *d;
  unsigned long long *c;
  const int blocks = 256, iters = 3000;
  hipMalloc(&d, 4096);
  hipMemset(d, 0, 4096);
  hipMalloc(&c, sizeof(unsigned long long) * blocks * WAVES);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<KIND, N, WAVES><<<blocks, 64 * WAVES>>>(d, c, 20);
  hipEventRecord(e0);
  k<KIND, N, WAVES><<<blocks, 64 * WAVES>>>(d, c, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * WAVES);
  hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  s /= h.size();
  const double nm = (double)iters * 12;  // MFMAs per wave
  // s_memtime counts at 100 MHz on some parts; report both the raw ratio and ns
  printf("%-34s waves/SIMD %d  VALU/MFMA %2d : %7.2f memtime ticks per MFMA(wave), %6.2f ns per MFMA "
         "of the SIMD  (%.3f ms)\n",
         name, WAVES / 4, valu_per_mfma, s / nm, ms * 1e6 / (nm * (WAVES / 4)), ms);
  hipFree(d);
  hipFree(c);
}

int main() {
  run<0, 0, 4>("mfma only", 0);
  run<0, 0, 8>("mfma only", 0);
  run<1, 2, 4>("v_mul", 2);
  run<1, 4, 4>("v_mul", 4);
  run<1, 5, 4>("v_mul", 5);
  run<1, 6, 4>("v_mul", 6);
  run<1, 8, 4>("v_mul", 8);
  run<1, 5, 8>("v_mul", 5);
  run<1, 8, 8>("v_mul", 8);
  run<2, 2, 4>("v_pk_mul", 2);
  run<2, 4, 4>("v_pk_mul", 4);
  run<5, 4, 4>("v_max3", 4);
  run<5, 6, 4>("v_max3", 6);
  run<3, 1, 4>("exact epilogue (5 ops/out)", 5);
  run<3, 2, 4>("exact epilogue (5 ops/out)", 10);
  run<3, 1, 8>("exact epilogue (5 ops/out)", 5);
  run<4, 1, 4>("keyed epilogue (7 ops/2 prof)", 7);
  run<4, 1, 8>("keyed epilogue (7 ops/2 prof)", 7);
  return 0;
}
