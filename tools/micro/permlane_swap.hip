// Micro-benchmark / check: sum over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) of a double with the gfx950
// v_permlane16_swap / v_permlane32_swap instructions against __shfl_xor (two ds_bpermute_b32 per step).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__device__ inline double xrow_sum_swap(double v) {
  unsigned lo = __double2loint(v), hi = __double2hiint(v);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double p = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  lo = __double2loint(p); hi = __double2hiint(p);
  auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}
__device__ inline double xrow_sum_shfl(double v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
template <int MODE>
__global__ void k(const double* x, double* o, long long* cyc, int n) {
  double v = x[threadIdx.x];
  const long long t0 = clock64();
  for (int i = 0; i < n; ++i) v = (MODE ? xrow_sum_swap(v) : xrow_sum_shfl(v)) * 0.25;
  const long long t1 = clock64();
  o[threadIdx.x] = v;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double h[64], r0[64], r1[64]; double *dx, *dout; long long* dc; long long c0, c1;
  for (int i = 0; i < 64; ++i) h[i] = 1.0 + 0.37 * i + 1e-3 * i * i;
  (void)hipMalloc(&dx, 512); (void)hipMalloc(&dout, 512); (void)hipMalloc(&dc, 8);
  (void)hipMemcpy(dx, h, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dx, dout, dc, 1);
  (void)hipMemcpy(r0, dout, 512, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dx, dout, dc, 1);
  (void)hipMemcpy(r1, dout, 512, hipMemcpyDeviceToHost);
  double err = 0, ref_err = 0;
  for (int i = 0; i < 64; ++i) {
    const double ref = 0.25 * (h[i & 15] + h[(i & 15) + 16] + h[(i & 15) + 32] + h[(i & 15) + 48]);
    err = fmax(err, fabs(r1[i] - ref)); ref_err = fmax(ref_err, fabs(r0[i] - ref));
  }
  hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dx, dout, dc, 200); (void)hipMemcpy(&c0, dc, 8, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dx, dout, dc, 200); (void)hipMemcpy(&c1, dc, 8, hipMemcpyDeviceToHost);
  printf("cross-row sum of a double: permlane swap error %.1e (shfl %.1e); dependent chain: shfl_xor %.0f cycles per sum, permlane swap %.0f\n",
         err, ref_err, c0 / 200.0, c1 / 200.0);
  return 0;
}
