// Micro-benchmark: which VALU instruction types make progress while another wave of the same SIMD keeps the bf16 (or f32) MFMA
// pipe busy?  One VALU wave + one MFMA wave per SIMD; the VALU wave runs 16 independent chains of one instruction type.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define OPLOOP(ASM)                                                                   \
  for (int it = 0; it < it_valu; ++it) {                                              \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b)); \
  }
template <int OP, int F32MFMA>
__global__ __launch_bounds__(512) void mix(long long* out, float* sink, int it_valu, int it_mfma) {
  const int wid = threadIdx.x >> 6;
  long long t0 = clock64(), t1;
  float s = 0;
  if (wid < 4) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    float a = 1.0001f, b = 0.5f;
    if (OP == 0) { OPLOOP("v_fma_f32 %0, %0, %1, %2") }
    if (OP == 1) { OPLOOP("v_mul_f32 %0, %0, %1") }
    if (OP == 2) { OPLOOP("v_add_f32 %0, %0, %2") }
    if (OP == 3) { OPLOOP("v_xor_b32 %0, %0, %1") }
    if (OP == 4) { OPLOOP("v_add_u32 %0, %0, %1") }
    if (OP == 5) { OPLOOP("v_mov_b32 %0, %1") }
    if (OP == 6) { OPLOOP("v_and_or_b32 %0, %0, %1, %2") }
    if (OP == 7) { OPLOOP("v_perm_b32 %0, %0, %1, %2") }
    if (OP == 8) { OPLOOP("v_lshrrev_b32 %0, 16, %0") }
    if (OP == 9) { OPLOOP("v_sub_f32 %0, %0, %2") }
    if (OP == 10) { OPLOOP("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") }
    if (OP == 11) { OPLOOP("v_cndmask_b32 %0, %0, %1, vcc") }
    for (int i = 0; i < 16; ++i) s += x[i];
    t1 = clock64();
  } else {
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    if (F32MFMA) {
      float a = threadIdx.x * 0.001f, b = 1.5f;
      for (int it = 0; it < it_mfma / 2; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      }
    } else {
      bf8 a, b;
      for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i); }
      for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      }
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    t1 = clock64();
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wid] = t1 - t0;
}
template <int OP, int F>
void run(long long* d, float* sk, const char* name) {
  long long h[8], alone;
  const int NV = 1000, NM = 500;
  hipLaunchKernelGGL((mix<OP, F>), dim3(256), dim3(512), 0, 0, d, sk, NV, 0);
  (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  alone = h[0];
  hipLaunchKernelGGL((mix<OP, F>), dim3(256), dim3(512), 0, 0, d, sk, NV, NM);
  (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  printf("%-14s %s MFMA: VALU alone %6lld cyc (%.2f/instr), beside MFMA %7lld cyc, MFMA wave %6lld cyc -> VALU cycles hidden under the MFMAs: %5.1f %%\n",
         name, F ? "f32 " : "bf16", alone, alone / 16000.0, h[0], h[4], 100.0 * (double)(alone + h[4] - h[0]) / (double)(alone < h[4] ? alone : h[4]));
}
int main() {
  long long* d; float* sk; (void)hipMalloc(&d, 64); (void)hipMalloc(&sk, 8 << 20);
  run<0, 0>(d, sk, "v_fma_f32"); run<1, 0>(d, sk, "v_mul_f32"); run<2, 0>(d, sk, "v_add_f32"); run<9, 0>(d, sk, "v_sub_f32");
  run<3, 0>(d, sk, "v_xor_b32"); run<4, 0>(d, sk, "v_add_u32"); run<5, 0>(d, sk, "v_mov_b32"); run<6, 0>(d, sk, "v_and_or_b32");
  run<7, 0>(d, sk, "v_perm_b32"); run<8, 0>(d, sk, "v_lshrrev_b32"); run<10, 0>(d, sk, "v_add_f32_dpp"); run<11, 0>(d, sk, "v_cndmask_b32");
  run<0, 1>(d, sk, "v_fma_f32"); run<3, 1>(d, sk, "v_xor_b32"); run<5, 1>(d, sk, "v_mov_b32");
  return 0;
}
