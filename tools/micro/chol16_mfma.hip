// Micro-benchmark: the 16x16 diagonal-tile factorisation with the tile in f64-MFMA ACCUMULATOR layout, every column step one rank-1
// update on the matrix pipe (v_mfma_f64_16x16x4 with a single live k-slot) instead of 3 (15 - k) v_readlane / v_fma instructions:
//     C (tile) and T (starts as the identity, ends as Linv): register rg of lane (lq, l15) holds entry [lq + 4 rg][l15]
//     step k (row k lives in lane group m = k & 3, register rg = k >> 2):
//         piv = 1 / sqrt(C[k][k])                      v_readlane x2 of lane 16 m + k, v_rsq_f64
//         v   = row k of C scaled by piv, entries <= k zeroed      = column k of L (the tile is symmetric)
//         z   = row k of T scaled by piv                            = row k of Linv (final: kept in place)
//         C  -= v v^T,  T -= v z^T                     two MFMAs: A operand v, B operand v / z, both in k-slot m only
// Prints cycles per tile and |A Ainv - I| with Ainv = Linv^T Linv.  Compare tools/micro/chol16_chain.hip (variant 0 = shipped).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
constexpr int CB = 16, CLD = 17;
typedef double acc_t __attribute__((ext_vector_type(4)));
__device__ inline double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
template <int LO, int HI, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (LO < HI) { f(std::integral_constant<int, LO>{}); static_for<LO + 1, HI>(f); }
}
template <bool NEWTON>
__global__ __launch_bounds__(64) void k(const double* A, double* out, long long* cyc, int reps) {
  __shared__ double blk[CB * CLD];
  const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
  long long total = 0;
  for (int rep = 0; rep < reps; ++rep) {
    for (int t = lane; t < CB * CB; t += 64) blk[(t >> 4) * CLD + (t & 15)] = A[t];
    __syncthreads();
    const long long t0 = clock64();
    acc_t C, T;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) { C[rg] = blk[(lq + 4 * rg) * CLD + l15]; T[rg] = (lq + 4 * rg == l15) ? 1.0 : 0.0; }
    double chk = 0;
    static_for<0, CB>([&](auto kc) {
      constexpr int kk = decltype(kc)::value, m = kk & 3, rg = kk >> 2;
      const double d = readlane_f64(C[rg], 16 * m + kk);
      double piv = __builtin_amdgcn_rsq(d);
      if (NEWTON) piv = piv * (1.5 - 0.5 * d * piv * piv);
      chk += piv;
      const bool mine = lq == m;
      const double x = C[rg] * piv, zt = T[rg] * piv;
      const double v = (mine && l15 > kk) ? x : 0.0;
      const double z = mine ? zt : 0.0;
      T[rg] = mine ? zt : T[rg];
      if (kk + 1 < CB) {
        C = __builtin_amdgcn_mfma_f64_16x16x4f64(-v, v, C, 0, 0, 0);
        T = __builtin_amdgcn_mfma_f64_16x16x4f64(-v, z, T, 0, 0, 0);
      }
    });
    // T = Linv in accumulator layout: Linv[lq + 4 rg][l15]; stored as Linv^T row-major (what the panel solves read)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) blk[l15 * CLD + lq + 4 * rg] = T[rg];
    __syncthreads();
    total += clock64() - t0;
    if (!(chk > 0.0)) blk[0] = NAN;
  }
  for (int t = lane; t < CB * CB; t += 64) out[t] = blk[(t >> 4) * CLD + (t & 15)];
  if (lane == 0) cyc[0] = total / reps;
}
int main() {
  std::vector<double> A(256), L(256);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) A[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
  double *dA, *dO; long long* dC;
  (void)hipMalloc(&dA, 2048); (void)hipMalloc(&dO, 2048); (void)hipMalloc(&dC, 8);
  (void)hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
  for (int newton = 0; newton < 2; ++newton) {
    if (newton) hipLaunchKernelGGL((k<true>), dim3(1), dim3(64), 0, 0, dA, dO, dC, 20);
    else hipLaunchKernelGGL((k<false>), dim3(1), dim3(64), 0, 0, dA, dO, dC, 20);
    long long c; (void)hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(L.data(), dO, 2048, hipMemcpyDeviceToHost);
    double err = 0, low = 0;
    for (int r = 0; r < 16; ++r) for (int c2 = 0; c2 < 16; ++c2) {
      double s = 0;
      for (int m = 0; m < 16; ++m) { double ainv = 0; for (int q = 0; q < 16; ++q) ainv += L[m * 16 + q] * L[c2 * 16 + q]; s += A[r * 16 + m] * ainv; }
      err = fmax(err, fabs(s - (r == c2 ? 1.0 : 0.0)));
      if (c2 < r) low = fmax(low, fabs(L[r * 16 + c2]));       // Linv^T is upper triangular: entries left of the diagonal must be 0
    }
    printf("MFMA rank-1 tile factorisation (%s pivots): %lld cycles per tile (%.0f per pivot), |A Ainv - I| = %.2e, max |below diagonal of Linv^T| = %.1e\n",
           newton ? "Newton-refined" : "rsq-estimate", c, c / 16.0, err, low);
  }
  return 0;
}
