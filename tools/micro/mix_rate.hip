// Micro-benchmark: one VALU-only wave and one MFMA-only wave per SIMD (512-thread workgroup), alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
template <typename T> struct M;
template <> struct M<float> { using acc = f4; static __device__ acc mma(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); } };
template <> struct M<double> { using acc = d4; static __device__ acc mma(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); } };
template <typename T>
__global__ __launch_bounds__(512) void mix(long long* out, T* sink, int it_valu, int it_mfma, int prio) {
  const int wid = threadIdx.x >> 6;
  long long t0 = clock64(), t1;
  T s = 0;
  if (wid < 4) {
    T x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * (T)0.001 + i;
    T a = (T)1.0001, b = (T)0.5;
    for (int it = 0; it < it_valu; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = x[i] * a + b;
    }
    for (int i = 0; i < 16; ++i) s += x[i];
    t1 = clock64();
  } else {
    if (prio) __builtin_amdgcn_s_setprio(2);
    typename M<T>::acc acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = typename M<T>::acc{0, 0, 0, 0};
    T a = threadIdx.x * (T)0.001, b = (T)1.0 + threadIdx.x;
    for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = M<T>::mma(a, b, acc[i]);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    t1 = clock64();
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wid] = t1 - t0;
}
template <typename T> void run(const char* name) {
  long long* d; T* sk; hipMalloc(&d, 64); hipMalloc(&sk, 8 << 20);
  long long h[8];
  const int NV = 1000, NM = 250;   // 16000 VALU (64000 cyc alone at 4/instr), 2000 MFMA (64000 cyc at 32)
  for (int prio : {0, 2})
  for (int mode = 0; mode < 3; ++mode) {
    int iv = mode == 1 ? 0 : NV, im = mode == 0 ? 0 : NM;
    hipLaunchKernelGGL(mix<T>, dim3(256), dim3(512), 0, 0, d, sk, iv, im, prio);
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("%s prio %d %-9s: VALU wave %7lld cyc (%.2f/instr)   MFMA wave %7lld cyc (%.1f/mfma)\n", name, prio,
           mode == 0 ? "valu only" : mode == 1 ? "mfma only" : "both", h[0], iv ? (double)h[0] / (16.0 * iv) : 0.0, h[4], im ? (double)h[4] / (8.0 * im) : 0.0);
  }
}
int main() { run<float>("f32"); run<double>("f64"); return 0; }
