// Micro-benchmark: does a bf16 MFMA wave overlap with an f32 VALU wave on the same SIMD (unlike f32 MFMA)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
// same question for the 32x32x16 shape (8 passes = 32 cycles, same flops per cycle): does one issue block the VALU for less of its duration?
__global__ __launch_bounds__(512) void mix32(long long* out, float* sink, int it_valu, int it_mfma) {
  const int wid = threadIdx.x >> 6;
  long long t0 = clock64(), t1;
  float s = 0;
  if (wid < 4) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    float a = 1.0001f, b = 0.5f;
    for (int it = 0; it < it_valu; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = x[i] * a + b;
    }
    for (int i = 0; i < 16; ++i) s += x[i];
    t1 = clock64();
  } else {
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0;
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i); }
    for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
    t1 = clock64();
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wid] = t1 - t0;
}
__global__ __launch_bounds__(512) void mix(long long* out, float* sink, int it_valu, int it_mfma) {
  const int wid = threadIdx.x >> 6;
  long long t0 = clock64(), t1;
  float s = 0;
  if (wid < 4) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    float a = 1.0001f, b = 0.5f;
    for (int it = 0; it < it_valu; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = x[i] * a + b;
    }
    for (int i = 0; i < 16; ++i) s += x[i];
    t1 = clock64();
  } else {
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i); }
    for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    t1 = clock64();
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wid] = t1 - t0;
}
int main() {
  long long* d; float* sk; hipMalloc(&d, 64); hipMalloc(&sk, 8 << 20);
  long long h[8];
  const int NV = 1000, NM = 500;   // 16000 VALU (64000 cyc alone), 4000 MFMA (64000 cyc at 16)
  for (int mode = 0; mode < 3; ++mode) {
    int iv = mode == 1 ? 0 : NV, im = mode == 0 ? 0 : NM;
    hipLaunchKernelGGL(mix, dim3(256), dim3(512), 0, 0, d, sk, iv, im);
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("%-9s: VALU wave %7lld cyc (%.2f/instr)   bf16 MFMA wave %7lld cyc (%.1f/mfma)\n",
           mode == 0 ? "valu only" : mode == 1 ? "mfma only" : "both", h[0], iv ? (double)h[0] / (16.0 * iv) : 0.0, h[4], im ? (double)h[4] / (8.0 * im) : 0.0);
  }
  for (int mode = 0; mode < 3; ++mode) {
    int iv = mode == 1 ? 0 : NV, im = mode == 0 ? 0 : NM;
    hipLaunchKernelGGL(mix32, dim3(256), dim3(512), 0, 0, d, sk, iv, im);
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("32x32x16 %-9s: VALU wave %7lld cyc (%.2f/instr)   bf16 MFMA wave %7lld cyc (%.1f/mfma)\n",
           mode == 0 ? "valu only" : mode == 1 ? "mfma only" : "both", h[0], iv ? (double)h[0] / (16.0 * iv) : 0.0, h[4], im ? (double)h[4] / (4.0 * im) : 0.0);
  }
  return 0;
}
