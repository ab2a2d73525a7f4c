// Micro-benchmark: a wave issuing bf16 MFMAs back to back starves a VALU wave on the same SIMD (mix_bf16).  Does spacing the
// MFMAs with s_nop (so that the MFMA wave does not sit at the head of the vector issue port while the matrix pipe is busy)
// or a higher priority for the VALU wave let the two overlap?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int NOP, int PRIO_VALU, int PRIO_MFMA>
__global__ __launch_bounds__(512) void mix(long long* out, float* sink, int it_valu, int it_mfma) {
  const int wid = threadIdx.x >> 6;
  long long t0 = clock64(), t1;
  float s = 0;
  if (wid < 4) {
    if (PRIO_VALU) __builtin_amdgcn_s_setprio(PRIO_VALU);
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
    if (PRIO_MFMA) __builtin_amdgcn_s_setprio(PRIO_MFMA);
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i); }
    for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        if (NOP > 0) __builtin_amdgcn_sched_barrier(0);
        if (NOP > 0) asm volatile("s_nop %0" :: "n"(NOP > 16 ? 15 : (NOP > 0 ? NOP - 1 : 0)));
        if (NOP > 16) asm volatile("s_nop %0" :: "n"(NOP > 16 ? NOP - 17 : 0));
        if (NOP > 0) __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    t1 = clock64();
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wid] = t1 - t0;
}
template <int NOP, int PV, int PM>
void run(long long* d, float* sk) {
  long long h[8];
  const int NV = 1000, NM = 500;
  for (int mode = 1; mode < 3; ++mode) {
    int iv = mode == 1 ? 0 : NV, im = NM;
    hipLaunchKernelGGL((mix<NOP, PV, PM>), dim3(256), dim3(512), 0, 0, d, sk, iv, im);
    (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("nop %2d prio valu %d mfma %d %-9s: VALU wave %7lld cyc (%.2f/instr)   MFMA wave %7lld cyc (%.1f/mfma)\n", NOP, PV, PM,
           mode == 1 ? "mfma only" : "both", h[0], iv ? (double)h[0] / (16.0 * iv) : 0.0, h[4], (double)h[4] / (8.0 * im));
  }
}
int main() {
  long long* d; float* sk; (void)hipMalloc(&d, 64); (void)hipMalloc(&sk, 8 << 20);
  run<0, 0, 0>(d, sk);
  run<0, 3, 0>(d, sk);
  run<0, 0, 3>(d, sk);
  run<2, 0, 0>(d, sk);
  run<4, 0, 0>(d, sk);
  run<8, 0, 0>(d, sk);
  run<10, 0, 0>(d, sk);
  run<12, 0, 0>(d, sk);
  run<14, 0, 0>(d, sk);
  run<16, 0, 0>(d, sk);
  run<12, 3, 0>(d, sk);
  run<12, 0, 3>(d, sk);
  return 0;
}
