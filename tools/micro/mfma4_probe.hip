// Probe of the lane layout of v_mfma_f64_4x4x4_4b_f64: each lane's value is 2^(lane & 15); the result per lane, printed as a bit mask,
// shows which lanes of the 16-lane block were summed (A = value, B = 1 and A = 1, B = value).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* o) {
  const double v = (double)(1 << (threadIdx.x & 15));
  o[threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
  o[64 + threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, v, 0.0, 0, 0, 0);
}
int main() {
  double* d; (void)hipMalloc(&d, 128 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[128]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int t = 0; t < 2; ++t) {
    printf("%s\n", t == 0 ? "A = 2^lane, B = 1:" : "A = 1, B = 2^lane:");
    for (int l = 0; l < 16; ++l) printf("  lane %2d: %04x\n", l, (unsigned)h[64 * t + l]);
    printf("  lane 16: %04x  lane 32: %04x\n", (unsigned)h[64 * t + 16], (unsigned)h[64 * t + 32]);
  }
  return 0;
}
