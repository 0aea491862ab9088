// Launch cost of a chain of small dependent kernels on one stream: plain launches against a
// captured hipGraph replayed per "iteration" (the shape of one greedy-PCA tail iteration:
// 13 kernels of a few microseconds).   hipcc -O3 --offload-arch=gfx950 graph_launch_rate.hip -o graph_launch_rate
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void tiny(double *p, int spin) {
  double v = p[threadIdx.x];
  for (int i = 0; i < spin; ++i) v = v * 1.0000001 + 1e-9;
  p[threadIdx.x] = v;
}

int main() {
  double *d;
  hipMalloc(&d, 1024 * sizeof(double));
  hipMemset(d, 0, 1024 * sizeof(double));
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const int NK = 13, ITERS = 2000;
  for (int spin : {100, 2000, 8000}) {
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, st, d, spin);
    hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < ITERS; ++it)
      for (int k = 0; k < NK; ++k) hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, st, d, spin);
    hipStreamSynchronize(st);
    double plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / ITERS;
    // one kernel alone, to know the pure execution time
    t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < 200; ++it) {
      hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, st, d, spin);
      hipStreamSynchronize(st);
    }
    double one = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int k = 0; k < NK; ++k) hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, st, d, spin);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < ITERS; ++it) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / ITERS;
    // with a host round trip per iteration (the PCA hand-shake): sync after every chain
    t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < 500; ++it) {
      for (int k = 0; k < NK; ++k) hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, st, d, spin);
      hipStreamSynchronize(st);
    }
    double plain_sync = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 500;
    t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < 500; ++it) {
      hipGraphLaunch(ge, st);
      hipStreamSynchronize(st);
    }
    double graph_sync = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 500;
    printf("spin %5d: one kernel+sync %6.1f us | chain of %d: plain %6.1f us, graph %6.1f us | with a sync per chain: plain %6.1f, graph %6.1f us\n",
           spin, one, NK, plain, graph, plain_sync, graph_sync);
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
  }
  return 0;
}
