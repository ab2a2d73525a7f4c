// Micro-test: the sum over the 16 lanes of a DPP row of an f64 value with two v_mfma_f64_4x4x4_4b_f64 (one 4x4x4 product per
// 16-lane block): first A = value, B = ones (partial sums over k), then the result fed back as A with B = ones.
// Prints the worst deviation from the exact row sums and the cycles per sum against the DPP tree (4 stages of 2 v_mov_dpp + v_add_f64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum_dpp(double v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}
__device__ __forceinline__ double row16_sum_mfma(double v) {
  const double p = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
  return __builtin_amdgcn_mfma_f64_4x4x4f64(p, 1.0, 0.0, 0, 0, 0);
}
__global__ void k(const double* in, double* out_m, double* out_d, long long* cyc, int n) {
  double v[9];
  for (int u = 0; u < 9; ++u) v[u] = in[threadIdx.x * 9 + u];
  double m[9], d[9];
  long long t0 = clock64();
  for (int it = 0; it < n; ++it)
#pragma unroll
    for (int u = 0; u < 9; ++u) m[u] = row16_sum_mfma(v[u] + it * 1e-300);
  long long t1 = clock64();
  for (int it = 0; it < n; ++it)
#pragma unroll
    for (int u = 0; u < 9; ++u) d[u] = row16_sum_dpp(v[u] + it * 1e-300);
  long long t2 = clock64();
  for (int u = 0; u < 9; ++u) { out_m[threadIdx.x * 9 + u] = m[u]; out_d[threadIdx.x * 9 + u] = d[u]; }
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
  double h[64 * 9], hm[64 * 9], hd[64 * 9];
  for (int i = 0; i < 64 * 9; ++i) h[i] = sin(0.37 * i) * exp(0.01 * (i % 50));
  double *d, *om, *od; long long* c;
  (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&om, sizeof(h)); (void)hipMalloc(&od, sizeof(h)); (void)hipMalloc(&c, 16);
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  const int n = 200;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, om, od, c, n);
  (void)hipMemcpy(hm, om, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipMemcpy(hd, od, sizeof(h), hipMemcpyDeviceToHost);
  long long cc[2]; (void)hipMemcpy(cc, c, 16, hipMemcpyDeviceToHost);
  double worst_m = 0, worst_d = 0;
  for (int row = 0; row < 4; ++row)
    for (int u = 0; u < 9; ++u) {
      long double s = 0, a = 0;
      for (int l = 0; l < 16; ++l) { s += h[(row * 16 + l) * 9 + u]; a += fabsl(h[(row * 16 + l) * 9 + u]); }
      for (int l = 0; l < 16; ++l) {
        worst_m = fmax(worst_m, fabs((double)(hm[(row * 16 + l) * 9 + u] - s)) / (double)a);
        worst_d = fmax(worst_d, fabs((double)(hd[(row * 16 + l) * 9 + u] - s)) / (double)a);
      }
    }
  printf("row sums of 16 lanes, worst |error| / sum|terms|: two 4x4x4 f64 MFMAs %.3g, DPP tree %.3g\n", worst_m, worst_d);
  printf("cycles per sum (9 independent sums per round, one wave): MFMA pair %.1f, DPP tree %.1f\n", (double)cc[0] / (9.0 * n), (double)cc[1] / (9.0 * n));
  return 0;
}
