// accuracy of v_rsq_f64 and of one / two Newton steps on it
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* y0, double* y1, double* y2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double y = __builtin_amdgcn_rsq(v);
  y0[i] = y;
  y = y * (1.5 - 0.5 * v * y * y);
  y1[i] = y;
  y = y * (1.5 - 0.5 * v * y * y);
  y2[i] = y;
}
int main() {
  const int n = 1 << 20;
  double* h = new double[n];
  for (int i = 0; i < n; ++i) h[i] = std::exp((i / (double)n) * 80.0 - 40.0) * (1.0 + (i % 977) * 1e-3);
  double *dx, *d0, *d1, *d2; hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, h, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  double *r0 = new double[n], *r1 = new double[n], *r2 = new double[n];
  hipMemcpy(r0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1, d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r2, d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double ex = 1.0L / sqrtl((long double)h[i]);
    e0 = fmax(e0, (double)fabsl((r0[i] - ex) / ex)); e1 = fmax(e1, (double)fabsl((r1[i] - ex) / ex)); e2 = fmax(e2, (double)fabsl((r2[i] - ex) / ex));
  }
  printf("max rel err: rsq %.3e   +1 NR %.3e   +2 NR %.3e\n", e0, e1, e2);
  return 0;
}
