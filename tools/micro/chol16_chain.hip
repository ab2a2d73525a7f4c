// Micro-benchmark: where do the ~235 cycles per pivot of chol16_wave (sba_chol_blocked.hpp) go?  One wave factors a 16x16 tile
// (rows in lanes 0..15, identity rows in lanes 16..31) in several variants:
//   0  as shipped: every column update broadcasts l_jk with two v_readlane_b32
//   1  chain only (no updates of the other columns; wrong result, timing of the pivot chain alone)
//   2  the update of column k+1 by v_readlane, the other columns from an LDS broadcast of the l_ik vector
//   3  like 0 but the broadcasts through ds_bpermute-free DPP row_newbcast... (not available on gfx950: skipped)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
constexpr int CB = 16, CLD = 17;
__device__ inline double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
template <int VAR, int SB>
__global__ __launch_bounds__(64) void k(const double* A, double* out, long long* cyc, int reps) {
  __shared__ double blk[CB * CLD];
  __shared__ __align__(16) double s_l[2][64];
  const int lane = threadIdx.x & 63, i = lane & 15;
  const bool ident = lane >= 16;
  long long total = 0;
  for (int rep = 0; rep < reps; ++rep) {
    for (int t = lane; t < CB * CB; t += 64) blk[(t >> 4) * CLD + (t & 15)] = A[t];
    __syncthreads();
    const long long t0 = clock64();
    double a[CB];
#pragma unroll
    for (int j = 0; j < CB; ++j) { const double v = blk[i * CLD + j]; a[j] = ident ? ((j == i) ? 1.0 : 0.0) : v; }
    double dg = blk[i * CLD + i];
    __builtin_amdgcn_wave_barrier();
    double akk = readlane_f64(dg, 0);
    double piv = __builtin_amdgcn_rsq(akk);
    double lprev = 0;
    double lq[CB];
#pragma unroll
    for (int j = 0; j < CB; ++j) lq[j] = 0;
#pragma unroll
    for (int k = 0; k < CB; ++k) {
      const double lik = a[k] * piv;
      a[k] = lik;
      if (VAR == 5) {
        // the order the hazards want: the broadcast of l_(k+1)k fills the wait state between the diagonal downdate and its
        // v_readlane, the column-(k+1) update fills the two wait states between that v_readlane and the rsq
        if (k + 1 < CB) {
          __builtin_amdgcn_sched_barrier(0);
          dg = __builtin_fma(-lik, lik, dg);
          __builtin_amdgcn_sched_barrier(0);
          const double bl = readlane_f64(lik, k + 1);
          __builtin_amdgcn_sched_barrier(0);
          akk = readlane_f64(dg, k + 1);
          __builtin_amdgcn_sched_barrier(0);
          a[k + 1] = __builtin_fma(-lik, bl, a[k + 1]);
          __builtin_amdgcn_sched_barrier(0);
          piv = __builtin_amdgcn_rsq(akk);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = k + 2; j < CB; ++j) a[j] -= lik * readlane_f64(lik, j);
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      if (k + 1 < CB) {
        dg = __builtin_fma(-lik, lik, dg);
        akk = readlane_f64(dg, k + 1);
        piv = __builtin_amdgcn_rsq(akk);
      }
      if (VAR == 4) {
        // software-pipelined LDS broadcast: the column published in step k-1 was read back during that step (lq), its
        // updates run in the shadow of this step's rsq; column k+1 takes this step's contribution by v_readlane
        if (k + 1 < CB) a[k + 1] -= lik * readlane_f64(lik, k + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (k >= 1) {
#pragma unroll
          for (int j = k + 1; j < CB; ++j) a[j] -= lprev * lq[j];
        }
        __builtin_amdgcn_sched_barrier(0);
        s_l[k & 1][lane] = lik;                    // every lane stores (no branch: one basic block keeps LLVM from sinking the updates)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = k + 2; j < CB; ++j) lq[j] = s_l[k & 1][j];
        lprev = lik;
        __builtin_amdgcn_sched_barrier(0);
      } else if (VAR == 0) {
#pragma unroll
        for (int j = k + 1; j < CB; ++j) a[j] -= lik * readlane_f64(lik, j);
      } else if (VAR == 1) {
        if (k + 1 < CB) a[k + 1] -= lik * readlane_f64(lik, k + 1);
      } else if (VAR == 3) {
        // column k+1 by v_readlane now; the other columns take step k-1's contribution, read back from LDS one step late
        if (k + 1 < CB) a[k + 1] -= lik * readlane_f64(lik, k + 1);
        if (lane < 16) s_l[k & 1][lane] = lik;
        __builtin_amdgcn_wave_barrier();
        if (k >= 1) {
#pragma unroll
          for (int j = k + 1; j < CB; ++j) a[j] -= lprev * s_l[(k - 1) & 1][j];
        }
        lprev = lik;
      } else if (VAR == 2) {
        if (k + 1 < CB) a[k + 1] -= lik * readlane_f64(lik, k + 1);
        if (k + 2 < CB) {
          if (lane < 16) s_l[k & 1][lane] = lik;
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int j = k + 2; j < CB; ++j) a[j] -= lik * s_l[k & 1][j];
        }
      }
      if (SB) __builtin_amdgcn_sched_barrier(0);
    }
    if (lane >= 16 && lane < 32) {
#pragma unroll
      for (int j = 0; j < CB; ++j) blk[i * CLD + j] = a[j];
    }
    __syncthreads();
    total += clock64() - t0;
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
  auto run = [&](int var) {
    if (var == 0) hipLaunchKernelGGL((k<0, 0>), dim3(1), dim3(64), 0, 0, dA, dO, dC, 20);
    if (var == 1) hipLaunchKernelGGL((k<1, 0>), dim3(1), dim3(64), 0, 0, dA, dO, dC, 20);
    if (var == 2) hipLaunchKernelGGL((k<2, 0>), dim3(1), dim3(64), 0, 0, dA, dO, dC, 20);
    if (var == 3) hipLaunchKernelGGL((k<3, 0>), dim3(1), dim3(64), 0, 0, dA, dO, dC, 20);
    if (var == 4) hipLaunchKernelGGL((k<4, 0>), dim3(1), dim3(64), 0, 0, dA, dO, dC, 20);
    if (var == 5) hipLaunchKernelGGL((k<5, 0>), dim3(1), dim3(64), 0, 0, dA, dO, dC, 20);
    if (var == 10) hipLaunchKernelGGL((k<0, 1>), dim3(1), dim3(64), 0, 0, dA, dO, dC, 20);
    if (var == 11) hipLaunchKernelGGL((k<1, 1>), dim3(1), dim3(64), 0, 0, dA, dO, dC, 20);
    if (var == 13) hipLaunchKernelGGL((k<3, 1>), dim3(1), dim3(64), 0, 0, dA, dO, dC, 20);
    long long c; (void)hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(L.data(), dO, 2048, hipMemcpyDeviceToHost);
    // check: out = Linv^T; (Linv^T)(Linv) should equal A^-1 -> verify A * (LinvT * Linv) = I
    double err = 0;
    for (int r = 0; r < 16; ++r) for (int c2 = 0; c2 < 16; ++c2) {
      double s = 0;
      for (int m = 0; m < 16; ++m) { double ainv = 0; for (int q = 0; q < 16; ++q) ainv += L[m * 16 + q] * L[c2 * 16 + q]; s += A[r * 16 + m] * ainv; }
      err = fmax(err, fabs(s - (r == c2 ? 1.0 : 0.0)));
    }
    printf("variant %d: %lld cycles per tile (%.0f per pivot), |A Ainv - I| = %.2e\n", var, c, c / 16.0, err);
  };
  run(0); run(1); run(4); run(5);
  return 0;
}
