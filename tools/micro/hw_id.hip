// Which SIMD does wave `wid` of a workgroup run on?  Reads HW_REG_HW_ID (gfx9: wave_id [3:0], simd_id [5:4], pipe, cu_id [11:8], ...)
// for workgroups of 256, 512 and 1024 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = v;
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 64);
  for (int nt : {256, 512, 1024}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(nt), 0, 0, d);
    unsigned h[16]; (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("%4d threads: SIMD of wave 0..%d:", nt, nt / 64 - 1);
    for (int w = 0; w < nt / 64; ++w) printf(" %u", (h[w] >> 4) & 3);
    printf("   (CU %u)\n", (h[0] >> 8) & 15);
  }
  return 0;
}
