// Probe: which way of marching a (Nz, S) float cube per spaxel reaches HBM bandwidth on MI355X?
// Every variant reads raw (f32), var (f32), mask (u8) of each voxel once and does a token amount
// of arithmetic.   hipcc --offload-arch=gfx950 -O3 -o tools/rowmarch_probe tools/rowmarch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// XCD-contiguous remap: hardware deals consecutive blocks round-robin over the 8 XCDs
__device__ __forceinline__ unsigned remap(unsigned b, unsigned n, int mode) {
  if (!mode) return b;
  const unsigned per = (n + 7) / 8;
  const unsigned g = (b & 7) * per + (b >> 3);
  return g;  // (callers check g < n)
}

// A: one wave per block, 64 spaxels, all z, plain loads, unroll U
template <int U, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void march_plain(const float *raw, const float *var, const uint8_t *mask,
                                                          int Nz, long S, float *out, int xcd) {
  const unsigned nb = gridDim.x;
  const unsigned g = remap(blockIdx.x, nb, xcd);
  if (g >= nb) return;
  const long s = (long)g * (64 * WAVES) + threadIdx.x;
  if (s >= S) return;
  float acc = 0.f;
  int z = 0;
  for (; z + U <= Nz; z += U) {
    float r[U], v[U]; int m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long i = (long)(z + u) * S + s; r[u] = raw[i]; v[u] = var[i]; m[u] = mask[i]; }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += m[u] ? 0.f : r[u] * v[u];
  }
  for (; z < Nz; ++z) { const long i = (long)z * S + s; acc += mask[i] ? 0.f : raw[i] * var[i]; }
  out[s] = acc;
}

// B: z-chunked grid (blockIdx.y = chunk), block of 64*WAVES spaxels
template <int U, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void march_chunk(const float *raw, const float *var, const uint8_t *mask,
                                                          int Nz, long S, int zchunk, float *out, int xcd) {
  const unsigned nb = gridDim.x;
  const unsigned g = remap(blockIdx.x, nb, xcd);
  if (g >= nb) return;
  const long s = (long)g * (64 * WAVES) + threadIdx.x;
  if (s >= S) return;
  const int z0 = blockIdx.y * zchunk, z1 = min(Nz, z0 + zchunk);
  float acc = 0.f;
  int z = z0;
  for (; z + U <= z1; z += U) {
    float r[U], v[U]; int m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long i = (long)(z + u) * S + s; r[u] = raw[i]; v[u] = var[i]; m[u] = mask[i]; }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += m[u] ? 0.f : r[u] * v[u];
  }
  for (; z < z1; ++z) { const long i = (long)z * S + s; acc += mask[i] ? 0.f : raw[i] * var[i]; }
  atomicAdd(&out[s], acc);
}

// C: 4 spaxels per lane (dwordx4 loads): block of 64*WAVES lanes covers 256*WAVES spaxels
template <int U, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void march_x4(const float *raw, const float *var, const uint8_t *mask,
                                                       int Nz, long S, int zchunk, float *out, int xcd) {
  const unsigned nb = gridDim.x;
  const unsigned g = remap(blockIdx.x, nb, xcd);
  if (g >= nb) return;
  const long s = ((long)g * (64 * WAVES) + threadIdx.x) * 4;
  if (s >= S) return;
  const int z0 = blockIdx.y * zchunk, z1 = min(Nz, z0 + zchunk);
  float acc = 0.f;
  int z = z0;
  for (; z + U <= z1; z += U) {
    float4 r[U], v[U]; uchar4 m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = (long)(z + u) * S + s;
      r[u] = *(const float4 *)(raw + i); v[u] = *(const float4 *)(var + i); m[u] = *(const uchar4 *)(mask + i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      acc += (m[u].x ? 0.f : r[u].x * v[u].x) + (m[u].y ? 0.f : r[u].y * v[u].y) + (m[u].z ? 0.f : r[u].z * v[u].z) +
             (m[u].w ? 0.f : r[u].w * v[u].w);
  }
  atomicAdd(&out[s / 4], acc);
}

int main(int argc, char **argv) {
  const int Nz = 3681;
  const int N = argc > 1 ? atoi(argv[1]) : 600;
  const long S = (long)N * N;
  const size_t n = (size_t)Nz * S;
  float *raw, *var, *out; uint8_t *mask;
  CK(hipMalloc(&raw, n * 4)); CK(hipMalloc(&var, n * 4)); CK(hipMalloc(&mask, n)); CK(hipMalloc(&out, S * 4));
  CK(hipMemset(raw, 0, n * 4)); CK(hipMemset(var, 0, n * 4)); CK(hipMemset(mask, 0, n));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double gb = 9.0 * n / 1e9;
  auto timeit = [&](const char *name, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    float best = 1e9f, tot = 0;
    for (int it = 0; it < 3; ++it) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; tot += ms;
    }
    printf("%-44s best %.3f ms  %.2f TB/s (mean %.3f ms)\n", name, best, gb / best, tot / 3); fflush(stdout);
  };
  for (int xcd = 0; xcd < 2; ++xcd) {
    printf("---- xcd remap %d\n", xcd);
    timeit("A plain  1 wave/block  all z  U4", [&] { hipLaunchKernelGGL((march_plain<4, 1>), dim3((S + 63) / 64), dim3(64), 0, 0, raw, var, mask, Nz, S, out, xcd); });
    timeit("A plain  1 wave/block  all z  U8", [&] { hipLaunchKernelGGL((march_plain<8, 1>), dim3((S + 63) / 64), dim3(64), 0, 0, raw, var, mask, Nz, S, out, xcd); });
    timeit("A plain  4 waves/block all z  U4", [&] { hipLaunchKernelGGL((march_plain<4, 4>), dim3((S + 255) / 256), dim3(256), 0, 0, raw, var, mask, Nz, S, out, xcd); });
    timeit("A plain  4 waves/block all z  U8", [&] { hipLaunchKernelGGL((march_plain<8, 4>), dim3((S + 255) / 256), dim3(256), 0, 0, raw, var, mask, Nz, S, out, xcd); });
    for (int nzc : {4, 16, 58}) {
      const int zchunk = (Nz + nzc - 1) / nzc;
      char nm[96];
      CK(hipMemset(out, 0, S * 4));
      snprintf(nm, 96, "B chunk  1 wave/block  nzc %d U8", nzc);
      timeit(nm, [&] { hipLaunchKernelGGL((march_chunk<8, 1>), dim3((S + 63) / 64, nzc), dim3(64), 0, 0, raw, var, mask, Nz, S, zchunk, out, xcd); });
      snprintf(nm, 96, "B chunk  4 waves/block nzc %d U8", nzc);
      timeit(nm, [&] { hipLaunchKernelGGL((march_chunk<8, 4>), dim3((S + 255) / 256, nzc), dim3(256), 0, 0, raw, var, mask, Nz, S, zchunk, out, xcd); });
      snprintf(nm, 96, "C x4     1 wave/block  nzc %d U4", nzc);
      timeit(nm, [&] { hipLaunchKernelGGL((march_x4<4, 1>), dim3((S / 4 + 63) / 64, nzc), dim3(64), 0, 0, raw, var, mask, Nz, S, zchunk, out, xcd); });
      snprintf(nm, 96, "C x4     4 waves/block nzc %d U4", nzc);
      timeit(nm, [&] { hipLaunchKernelGGL((march_x4<4, 4>), dim3((S / 4 + 255) / 256, nzc), dim3(256), 0, 0, raw, var, mask, Nz, S, zchunk, out, xcd); });
    }
  }
  return 0;
}
