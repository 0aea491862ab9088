// ds_read_b128 throughput per CU on gfx950 for the address patterns of the GLR spatial kernel
// (csrc/glr_spatial_mfma.hip): how many cycles does one wave-wide 16-byte read cost when 4 or 8
// waves of a block read at once?   hipcc --offload-arch=gfx950 -O3 lds_read_probe.hip -o lds_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512, 1) void probe(int iters, unsigned *out, long long *cyc) {
  extern __shared__ __align__(16) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 150 * 1024 / 4; i += blockDim.x) reinterpret_cast<unsigned *>(lds)[i] = i;
  __syncthreads();
  const int n = lane & 31, h = lane >> 5;
  int off;
  if (PAT == 0) off = lane * 16;                                            // linear
  else if (PAT == 1) off = 832 * (n >> 2) + 16 * (n & 3) + 16 * h;         // B fragment (pitch 208)
  else if (PAT == 2) off = ((3 - (n >> 3)) * 8 + (7 - (n & 7))) * 80 + 16 * h;  // A fragment (entry 80)
  else if (PAT == 3) off = 832 * (n >> 2) + 32 * (n & 3) + 16 * h;         // B without the duplicates
  else off = (lane & 15) * 16 + (lane >> 4) * 272;                          // 4 rows, pitch 272
  off += (wave & 3) * 32 * 208;  // the waves of a group read different areas
  const char *p = lds + off;
  u32x4v acc = {0, 0, 0, 0};
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      u32x4v v;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)p), "n"(k * 208 * 2 % 4096 / 16 * 16));
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      acc ^= v;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  const long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + tid] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

template <int PAT>
void run(const char *name, int waves) {
  unsigned *out; long long *cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  hipFuncSetAttribute((const void *)probe<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  probe<PAT><<<256, waves * 64, 150 * 1024>>>(10, out, cyc);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  probe<PAT><<<256, waves * 64, 150 * 1024>>>(iters, out, cyc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  const double reads = (double)iters * 16 * waves;  // wave-wide reads per CU
  printf("%-28s %d waves: %6.2f clock64-cycles per wave-read per CU, %6.1f ns/read -> %6.1f B/ns per CU (%.2f ms)\n",
         name, waves, h[0] / reads, ms * 1e6 / reads, 1024.0 / (ms * 1e6 / reads), ms);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {4, 8}) {
    run<0>("linear (lane*16)", w);
    run<1>("B fragment, pitch 208", w);
    run<2>("A fragment, entry 80", w);
    run<3>("B pattern without duplicates", w);
    run<4>("16 lanes x 4 rows pitch 272", w);
  }
  return 0;
}
