// Micro-benchmark: stream time of a chain of dependent trivial kernels, plain launches vs a replayed hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
__global__ void wide(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
int main() {
  float* d; hipMalloc(&d, 64 << 20); hipMemset(d, 0, 64 << 20);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 2000;
  for (int variant = 0; variant < 2; ++variant) {
    auto launch = [&](int i) {
      if (variant == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d);
      else hipLaunchKernelGGL(wide, dim3(1024), dim3(256), 0, s, d, 1 << 18);
    };
    for (int i = 0; i < 100; ++i) launch(i);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < N; ++i) launch(i);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s plain launches : %.2f us per kernel\n", variant ? "wide (1024 WGs)" : "tiny (1 WG)    ", ms * 1e3 / N);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 100; ++i) launch(i);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < N / 100; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s hipGraph replay : %.2f us per kernel\n", variant ? "wide (1024 WGs)" : "tiny (1 WG)    ", ms * 1e3 / N);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
