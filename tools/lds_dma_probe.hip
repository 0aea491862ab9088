// Probe: what does global_load_lds_ubyte / _dword write into LDS on gfx950?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_dma_probe tools/lds_dma_probe.hip && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(const uint8_t *bytes, const float *words, unsigned *out) {
  __shared__ unsigned buf[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) buf[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)buf;  // LDS byte address (low 32 bits of the generic pointer)
  const unsigned b0 = __builtin_amdgcn_readfirstlane(base);
  const unsigned b1 = b0 + 512;
  unsigned keep;
  const unsigned off1 = lane, off4 = lane * 4;
  asm volatile("s_mov_b32 %0, m0\n\t"
               "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_ubyte %1, %5\n\t"
               "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, %6\n\t"
               "s_mov_b32 m0, %0\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&s"(keep)
               : "v"(off1), "v"(off4), "s"(b0), "s"(b1), "s"(bytes), "s"(words)
               : "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = buf[i];
}

int main() {
  std::vector<uint8_t> hb(256);
  std::vector<float> hw(256);
  for (int i = 0; i < 256; ++i) hb[i] = (uint8_t)(0x80 + i), hw[i] = 1000.0f + i;
  uint8_t *db; float *dw; unsigned *dout;
  hipMalloc(&db, 256); hipMalloc(&dw, 1024); hipMalloc(&dout, 1024);
  hipMemcpy(db, hb.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(dw, hw.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, db, dw, dout);
  std::vector<unsigned> ho(256);
  hipMemcpy(ho.data(), dout, 1024, hipMemcpyDeviceToHost);
  printf("ubyte region (dwords 0..71):\n");
  for (int i = 0; i < 72; ++i) printf("%08x%s", ho[i], (i % 8 == 7) ? "\n" : " ");
  printf("dword region (dwords 128..135 as float):\n");
  for (int i = 128; i < 136; ++i) printf("%g ", *(float *)&ho[i]);
  printf("... %g %g\n", *(float *)&ho[190], *(float *)&ho[191]);
  return 0;
}
