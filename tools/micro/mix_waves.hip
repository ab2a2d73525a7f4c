// Micro-benchmark: ONE f32-MFMA wave per SIMD beside 1, 2 or 3 VALU-only waves per SIMD (512 / 768 / 1024-thread workgroups).
// Question: how many VALU instructions per 32-cycle MFMA slot can the partners of a saturated matrix pipe issue, as a function
// of the number of VALU waves and of their instruction-level parallelism (ILP independent FMA chains per wave)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int ILP>
__global__ __launch_bounds__(1024) void mix(long long* out, float* sink, int it_valu, int it_mfma, int prio) {
  const int wid = threadIdx.x >> 6;
  long long t0 = clock64(), t1;
  float s = 0;
  if (wid >= 4) {
    float x[ILP];
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 0.001f + i;
    float a = 1.0001f, b = 0.5f;
    for (int it = 0; it < it_valu * (16 / ILP); ++it) {
#pragma unroll
      for (int i = 0; i < ILP; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    for (int i = 0; i < ILP; ++i) s += x[i];
    t1 = clock64();
  } else {
    if (prio) __builtin_amdgcn_s_setprio(2);
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x;
    for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    t1 = clock64();
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wid] = t1 - t0;
}
template <int ILP> void run() {
  long long* d; float* sk; (void)hipMalloc(&d, 128); (void)hipMalloc(&sk, 16 << 20);
  long long h[16];
  const int NV = 500, NM = 250;   // 8000 VALU per wave, 2000 MFMA per MFMA wave (64000 cycles at 32)
  for (int threads : {512, 768, 1024})
  for (int mode = 0; mode < 2; ++mode) {     // 0: VALU waves alone (MFMA waves idle), 1: both
    hipLaunchKernelGGL(mix<ILP>, dim3(256), dim3(threads), 0, 0, d, sk, NV, mode ? NM : 0, 2);
    (void)hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    long long vmax = 0; for (int w = 4; w < threads / 64; ++w) vmax = h[w] > vmax ? h[w] : vmax;
    const int nvw = threads / 64 / 4 - 1;
    printf("ILP %2d  %d VALU wave(s)/SIMD %-5s: VALU waves %7lld cyc = %.2f cyc/instr/wave = %.2f SIMD-cycles per VALU instr;  MFMA wave %7lld cyc (%.1f/mfma)\n",
           ILP, nvw, mode ? "+MFMA" : "alone", vmax, (double)vmax / (16.0 * NV), (double)vmax / (16.0 * NV * nvw), h[0], mode ? (double)h[0] / (8.0 * NM) : 0.0);
  }
}
int main() { run<16>(); run<4>(); run<2>(); run<1>(); return 0; }
