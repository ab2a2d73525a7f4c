// Micro-benchmark: what bounds the trailing update of the one-workgroup Cholesky (k_cholesky_blocked, phase C)?
// One workgroup of 512 threads; NW of its waves run the strip loop of the kernel on LDS-resident 16x16 f64 blocks (row stride 17
// doubles): per tile 4 + 4 operand reads, 4 target reads, 4 dependent v_mfma_f64_16x16x4, 4 subtractions, 4 target writes.
// Variants drop one ingredient at a time.  Output: cycles per tile as one wave sees them, and per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int CLD = 17, CBS = 16 * CLD, NBLK = 66;
template <int VAR>
__global__ __launch_bounds__(512) void k(double* out, long long* cyc, int ntile, unsigned wave_mask) {
  extern __shared__ double Lb[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < NBLK * CBS; i += 512) Lb[i] = 1e-3 * ((i * 7919) % 1000);
  __syncthreads();
  const int po = (lane & 15) * CLD + (lane >> 4), to = (lane >> 4) * CLD + (lane & 15);
  long long t0 = 0, t1 = 0;
  double sink = 0;
  if ((wave_mask >> wid) & 1u) {
    const double* Ap = Lb + (wid * 8) * CBS + po;
    double a[4];
    for (int ks = 0; ks < 4; ++ks) a[ks] = Ap[4 * ks];
    double na[4];
    for (int ks = 0; ks < 4; ++ks) na[ks] = -a[ks];
    double* D = Lb + (wid * 8 + 1) * CBS + to;
    const double* Bp = Lb + (wid * 8 + 2) * CBS + po;
    d4 keep = {0, 0, 0, 0};
    t0 = clock64();
    if (VAR == 6) {
      // software-pipelined lean form: tile t + 1's loads and MFMA chain are issued before tile t's result is written back
      const int nt = __builtin_amdgcn_readfirstlane(ntile);
      double bq[2][4];
      d4 acc[2];
      for (int ks = 0; ks < 4; ++ks) bq[0][ks] = Bp[4 * ks];
      for (int rg = 0; rg < 4; ++rg) acc[0][rg] = D[4 * rg * CLD];
      for (int ks = 0; ks < 4; ++ks) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(na[ks], bq[0][ks], acc[0], 0, 0, 0);
      double* Dprev = D;
      for (int t = 1; t < nt; t += 2) {
        // odd tile into buffers 1
        double* D1 = Lb + (wid * 8 + 1) * CBS + to + (t & 3) * CBS;
        const double* B1 = Lb + (wid * 8 + 2) * CBS + po + (t & 3) * CBS;
        for (int ks = 0; ks < 4; ++ks) bq[1][ks] = B1[4 * ks];
        for (int rg = 0; rg < 4; ++rg) acc[1][rg] = D1[4 * rg * CLD];
        for (int ks = 0; ks < 4; ++ks) acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(na[ks], bq[1][ks], acc[1], 0, 0, 0);
        for (int rg = 0; rg < 4; ++rg) Dprev[4 * rg * CLD] = acc[0][rg];
        // even tile into buffers 0
        double* D0 = Lb + (wid * 8 + 1) * CBS + to + ((t + 1) & 3) * CBS;
        const double* B0 = Lb + (wid * 8 + 2) * CBS + po + ((t + 1) & 3) * CBS;
        for (int ks = 0; ks < 4; ++ks) bq[0][ks] = B0[4 * ks];
        for (int rg = 0; rg < 4; ++rg) acc[0][rg] = D0[4 * rg * CLD];
        for (int ks = 0; ks < 4; ++ks) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(na[ks], bq[0][ks], acc[0], 0, 0, 0);
        for (int rg = 0; rg < 4; ++rg) D1[4 * rg * CLD] = acc[1][rg];
        Dprev = D0;
      }
      for (int rg = 0; rg < 4; ++rg) Dprev[4 * rg * CLD] = acc[0][rg];
    } else {
    const int nt_s = __builtin_amdgcn_readfirstlane(ntile);
    for (int t = 0; t < nt_s; ++t) {
      double bq[4];
      d4 acc = {0, 0, 0, 0}, prod = {0, 0, 0, 0};
      if (VAR == 5) {
        // the lean form: D <- D + (-A) B^T as one MFMA chain seeded with D, nothing else on the VALU but the two address updates
        for (int ks = 0; ks < 4; ++ks) bq[ks] = Bp[4 * ks];
        for (int rg = 0; rg < 4; ++rg) acc[rg] = D[4 * rg * CLD];
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(na[ks], bq[ks], acc, 0, 0, 0);
        for (int rg = 0; rg < 4; ++rg) D[4 * rg * CLD] = acc[rg];
        D += CBS; Bp += CBS;
        if (((t + 1) & 3) == 0) { D -= 4 * CBS; Bp -= 4 * CBS; }
        continue;
      }
      if (VAR != 3) { for (int ks = 0; ks < 4; ++ks) bq[ks] = Bp[4 * ks]; } else { for (int ks = 0; ks < 4; ++ks) bq[ks] = a[ks] + t; }
      if (VAR != 2 && VAR != 3) for (int rg = 0; rg < 4; ++rg) acc[rg] = D[4 * rg * CLD];
      if (VAR != 4) { for (int ks = 0; ks < 4; ++ks) prod = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], bq[ks], prod, 0, 0, 0); }
      else { for (int ks = 0; ks < 4; ++ks) prod[ks] = a[ks] * bq[ks]; }
      if (VAR != 1 && VAR != 3) { for (int rg = 0; rg < 4; ++rg) D[4 * rg * CLD] = acc[rg] - prod[rg]; }
      else { for (int rg = 0; rg < 4; ++rg) keep[rg] += acc[rg] - prod[rg]; }
      D += CBS; Bp += CBS;
      if (((t + 1) & 3) == 0) { D -= 4 * CBS; Bp -= 4 * CBS; }
    }
    }
    t1 = clock64();
    sink = keep[0] + keep[1] + keep[2] + keep[3];
  }
  __syncthreads();
  out[threadIdx.x] = sink + Lb[threadIdx.x];
  if (lane == 0) cyc[wid] = t1 - t0;
}
int main() {
  double* d; long long* c; (void)hipMalloc(&d, 512 * 8); (void)hipMalloc(&c, 64);
  const size_t lds = (size_t)NBLK * CBS * 8;
  const char* names[] = {"full tile (reads, 4 MFMA, subtract, writes)", "no LDS writes", "no target reads", "MFMAs only (no LDS at all)", "no MFMAs (VALU products)", "lean: chain seeded with the target, -A kept", "lean + pipelined (next chain before this write-back)"};
  void (*kern[])(double*, long long*, int, unsigned) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>};
  for (auto f : kern) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int ntile = 400;
  // waves go round the SIMDs in the order 1, 3, 0, 2 (hw_id.hip): waves w and w + 4 share a SIMD
  const unsigned masks[] = {0x02, 0x22, 0x0e, 0xee, 0xfe};
  const char* mname[] = {"1 wave", "2 waves on one SIMD", "3 waves on 3 SIMDs", "6 waves on 3 SIMDs", "7 waves (one SIMD holds one)"};
  for (int v = 0; v < 7; ++v)
    for (int m = 0; m < 5; ++m) {
      hipLaunchKernelGGL(kern[v], dim3(1), dim3(512), lds, 0, d, c, ntile, masks[m]);
      long long cc[8]; (void)hipMemcpy(cc, c, 64, hipMemcpyDeviceToHost);
      long long mx = 0; int nw = 0;
      for (int w = 0; w < 8; ++w) if ((masks[m] >> w) & 1u) { mx = cc[w] > mx ? cc[w] : mx; ++nw; }
      printf("%-46s %-30s %7.0f cycles per tile and wave, %6.0f per tile of the workgroup\n", names[v], mname[m], (double)mx / ntile, (double)mx / ntile / nw);
    }
  return 0;
}
