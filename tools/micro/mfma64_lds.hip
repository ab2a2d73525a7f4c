// Micro-benchmark: f64 MFMA consumer loop fed from LDS the way k_schur_sym does it (9 tiles / wave, 8 waves).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 176, K = 96;
template <int MODE>   // 0: operands from LDS each k-step; 1: operands loaded once (registers only); 2: LDS loads but MFMAs use fixed regs
__global__ __launch_bounds__(512) void k(long long* out, double* sink, int iters) {
  extern __shared__ double panel[];
  for (int i = threadIdx.x; i < K * ROWS; i += 512) panel[i] = (i % 37) * 0.01;
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const double* pl = panel + (lane >> 4) * ROWS + (lane & 15);
  d4 acc[9];
  for (int i = 0; i < 9; ++i) acc[i] = d4{0, 0, 0, 0};
  double f[11];
  for (int b = 0; b < 11; ++b) f[b] = pl[16 * b];
  double keep = 0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < K / 4; ++ks) {
      double g[11];
      if (MODE != 1) {
#pragma unroll
        for (int b = 0; b < 11; ++b) g[b] = pl[ks * 4 * ROWS + 16 * b];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const double a = MODE == 0 ? g[t % 3] : f[t % 3], b = MODE == 0 ? g[2 + t] : f[2 + t];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
      if (MODE == 2) { for (int b = 0; b < 11; ++b) keep += g[b]; }
    }
  }
  long long t1 = clock64();
  double s = keep;
  for (int i = 0; i < 9; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0 && blockIdx.x == 0) out[wid] = t1 - t0;
}
int main() {
  long long* d; double* sk; hipMalloc(&d, 64); hipMalloc(&sk, 8 << 20);
  long long h[8];
  const int iters = 20;
  auto rep = [&](const char* n) { hipMemcpy(h, d, 64, hipMemcpyDeviceToHost); printf("%-28s %.1f cyc/mfma per wave (wave0), wave7 %.1f\n", n, (double)h[0] / (iters * 24 * 9), (double)h[7] / (iters * 24 * 9)); };
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  for (int blocks : {1, 256}) {
    printf("blocks %d\n", blocks);
    hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), K * ROWS * 8, 0, d, sk, iters); rep("LDS-fed operands");
    hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), K * ROWS * 8, 0, d, sk, iters); rep("register operands");
    hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), K * ROWS * 8, 0, d, sk, iters); rep("LDS loads + register operands");
  }
  return 0;
}
