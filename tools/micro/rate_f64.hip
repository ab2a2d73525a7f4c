// Micro-benchmark: issue rate of INDEPENDENT f64 / f32 VALU instructions (8 accumulators per lane, no dependence inside a group of 8)
// with 1, 2 and 3 waves per SIMD, and the same beside a wave that issues v_mfma_f64_16x16x4 back to back.
// Output: cycles per VALU instruction as one wave sees them.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int OP>
__global__ __launch_bounds__(1024) void k(double* io, long long* cyc, int n, int mfma_waves_per_simd) {
  const int wid = threadIdx.x >> 6;
  const int wps = blockDim.x / 256;                       // waves per SIMD (waves are dealt round robin to the 4 SIMDs)
  const bool mf = (wid / 4) >= wps - mfma_waves_per_simd; // the last wave(s) of each SIMD issue MFMAs instead
  double a[8];
  float f[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) { a[u] = io[threadIdx.x & 63] + u; f[u] = (float)a[u]; }
  d4 acc = {0, 0, 0, 0};
  __syncthreads();
  const long long t0 = clock64();
  if (mf) {
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], a[1], acc, 0, 0, 0);
    }
  } else {
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (OP == 0) a[u] = __builtin_fma(a[u], 0.9999999, 1e-9);
          if (OP == 1) a[u] = a[u] * 1.0000001;
          if (OP == 2) a[u] = a[u] + 1e-9;
          if (OP == 3) f[u] = __builtin_fmaf(f[u], 0.9999999f, 1e-9f);
          if (OP == 4) a[u] = __builtin_fma(a[u], a[(u + 1) & 7], a[(u + 2) & 7]);   // three register operands (64-bit each)
        }
      }
    }
  }
  const long long t1 = clock64();
  double s = acc[0] + acc[1];
#pragma unroll
  for (int u = 0; u < 8; ++u) s += a[u] + f[u];
  io[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[wid] = t1 - t0;
}
int main() {
  double* d; long long* c; (void)hipMalloc(&d, 8192); (void)hipMalloc(&c, 16 * 8);
  double h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1.5 + i * 0.01;
  const char* names[] = {"v_fma_f64 (const operands)", "v_mul_f64", "v_add_f64", "v_fma_f32", "v_fma_f64 (3 VGPR pairs)"};
  const int n = 200;
  for (int mfw = 0; mfw <= 1; ++mfw)
    for (int op = 0; op < 5; ++op)
      for (int wps = 1 + mfw; wps <= 3 + mfw && wps <= 4; ++wps) {
        (void)hipMemcpy(d, h, 8192, hipMemcpyHostToDevice);
        const dim3 blk(256 * wps);
        switch (op) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(1), blk, 0, 0, d, c, n, mfw); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(1), blk, 0, 0, d, c, n, mfw); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(1), blk, 0, 0, d, c, n, mfw); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(1), blk, 0, 0, d, c, n, mfw); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(1), blk, 0, 0, d, c, n, mfw); break;
        }
        long long cc[16]; (void)hipMemcpy(cc, c, sizeof(cc), hipMemcpyDeviceToHost);
        const int nw = 4 * wps;
        printf("%-28s %d VALU wave(s) per SIMD%s: %6.2f cycles per VALU instruction and wave", names[op], wps - mfw, mfw ? " + 1 MFMA wave" : "",
               (double)cc[0] / (32.0 * n));
        if (mfw) printf("   (MFMA wave: %6.1f cycles per MFMA)", (double)cc[nw - 1] / (8.0 * n));
        printf("\n");
      }
  return 0;
}
