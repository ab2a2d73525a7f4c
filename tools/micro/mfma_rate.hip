// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 on one SIMD (one wave), in shader cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k32(long long* out, float* sink, int iters) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int NACC>
__global__ void k64(long long* out, double* sink, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main() {
  long long* d; float* sf; double* sd;
  hipMalloc(&d, 8); hipMalloc(&sf, 4 << 20); hipMalloc(&sd, 8 << 20);
  long long h;
  const int iters = 1000;
  for (int threads : {64, 256, 512}) {
    for (int blocks : {1, 256}) {
      hipLaunchKernelGGL(k32<8>, dim3(blocks), dim3(threads), 0, 0, d, sf, iters); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      printf("f32 16x16x4  threads %3d blocks %3d  8 acc: %.1f cyc/mfma (per wave)\n", threads, blocks, (double)h / (iters * 8));
      hipLaunchKernelGGL(k64<8>, dim3(blocks), dim3(threads), 0, 0, d, sd, iters); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      printf("f64 16x16x4  threads %3d blocks %3d  8 acc: %.1f cyc/mfma (per wave)\n", threads, blocks, (double)h / (iters * 8));
    }
  }
  hipLaunchKernelGGL(k64<1>, dim3(1), dim3(64), 0, 0, d, sd, iters); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("f64 dependent chain: %.1f cyc/mfma\n", (double)h / iters);
  hipLaunchKernelGGL(k32<1>, dim3(1), dim3(64), 0, 0, d, sf, iters); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("f32 dependent chain: %.1f cyc/mfma\n", (double)h / iters);
  return 0;
}
