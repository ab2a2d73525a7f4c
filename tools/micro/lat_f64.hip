// Micro-benchmark: dependent-chain latency (cycles per link, one wave alone on its SIMD) of the operations on the Cholesky pivot chain.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
template <int OP>
__global__ __launch_bounds__(64) void k(double* io, long long* cyc, int n) {
  double x = io[threadIdx.x];
  float xf = (float)x;
  const long long t0 = clock64();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) x = __builtin_amdgcn_rsq(x);
      if (OP == 1) x = x * 1.0000001;
      if (OP == 2) x = __builtin_fma(x, 0.9999999, 1e-9);
      if (OP == 3) x = readlane_f64(x, 5) + 0.0 * x;          // readlane pair + one f64 add that consumes the SGPRs
      if (OP == 4) { xf = __builtin_amdgcn_rsqf(xf); }
      if (OP == 5) x = (double)__builtin_amdgcn_rsqf((float)x);
      if (OP == 6) x = x + 1e-9;
      if (OP == 7) { asm volatile("v_readlane_b32 s20, %1, 5\n v_readlane_b32 s21, %2, 5\n v_mov_b32 %0, s20" : "=v"(xf) : "v"(xf), "v"(xf) : "s20", "s21"); }
      if (OP == 8) x = __builtin_amdgcn_rcp(x);
      if (OP == 9) x = __builtin_sqrt(x);
    }
  }
  const long long t1 = clock64();
  io[threadIdx.x] = x + xf;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* d; long long* c; (void)hipMalloc(&d, 512); (void)hipMalloc(&c, 8);
  const char* names[] = {"v_rsq_f64", "v_mul_f64", "v_fma_f64", "readlane x2 + v_fma_f64", "v_rsq_f32", "cvt + v_rsq_f32 + cvt", "v_add_f64", "readlane x2 + v_mov", "v_rcp_f64", "sqrt(f64) (library)"};
  double h[64]; for (int i = 0; i < 64; ++i) h[i] = 1.5 + i * 0.01;
  const int n = 200;
  for (int op = 0; op < 10; ++op) {
    (void)hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
    switch (op) {
      case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, c, n); break;
      case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, c, n); break;
      case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, c, n); break;
      case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, c, n); break;
      case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d, c, n); break;
      case 5: hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, d, c, n); break;
      case 6: hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, d, c, n); break;
      case 7: hipLaunchKernelGGL(k<7>, dim3(1), dim3(64), 0, 0, d, c, n); break;
      case 8: hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, 0, d, c, n); break;
      case 9: hipLaunchKernelGGL(k<9>, dim3(1), dim3(64), 0, 0, d, c, n); break;
    }
    long long cc; (void)hipMemcpy(&cc, c, 8, hipMemcpyDeviceToHost);
    printf("%-28s %6.1f cycles per link\n", names[op], (double)cc / (8.0 * n));
  }
  return 0;
}
