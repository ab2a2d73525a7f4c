// sba_chol_blocked.hpp -- blocked Cholesky + solve of the reduced camera system, one workgroup, all in LDS.
//
//   A = S + lam*diag(D2c)  (n = 11*C <= 176, padded to n16 = 16*nb with an identity tail)
//   A = L L^T ;  delta_c = A^-1 rhs
//
// Block size 16 (= one f64 MFMA 16x16x4 tile); the lower triangle is kept as 16x16 blocks with a 17-double
// row stride (conflict-free ds_read_b64 for the MFMA operand pattern lane -> [row lane&15][col lane>>4]).
// Per block column jb:
//   B'  every thread owns one row below the diagonal block and forward-substitutes it against L11
//       (L21 = A21 L11^-T); one more thread does the same with the rhs row, which is the forward solve.
//   C   wave 0 updates the next diagonal tile and factors it at once (look-ahead), wave 1 downdates the rhs
//       tail, waves 1..7 apply the rank-16 trailing update tile by tile with 4 MFMAs per tile.
// After the last column one wave runs the blocked back substitution (16-step shuffle solves per block).
#pragma once
#include "sba_lm_kernels.hpp"

namespace sba {

constexpr int CB = 16;                 // block edge
constexpr int CLD = 17;                // row stride inside a block (doubles)
constexpr int CBS = CB * CLD;          // doubles per block
constexpr int CHOLB_THREADS = 512;
constexpr int CHOLB_MAX_NB = 11;       // 176 rows

__device__ inline int cb_off(int r, int c) { return (r * (r + 1) / 2 + c) * CBS; }

// wave-uniform broadcast of lane `src` (compile-time constant) through SGPRs: v_readlane_b32 x2, no LDS round trip
__device__ inline double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// 1/sqrt(x): hardware estimate + two Newton steps (the pivot only has to be consistent, not correctly rounded)
__device__ inline double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}

// Cholesky of one 16x16 diagonal block by the calling wave: lane i (< 16) owns row i; broadcasts go through
// v_readlane (SGPRs), so a step costs ~150 cycles instead of two LDS round trips.  Writes L (lower, upper zeroed)
// back into the block, L^T into `blkT` and 1/L[k][k] into inv[0..15].  Returns false (wave-uniform) when the block
// is not positive definite.
__device__ __forceinline__ bool chol16_wave(double* __restrict__ blk, double* __restrict__ blkT, double* __restrict__ inv) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15;
  double a[CB];
#pragma unroll
  for (int j = 0; j < CB; ++j) a[j] = blk[i * CLD + j];
  bool ok = true;
#pragma unroll
  for (int k = 0; k < CB; ++k) {
    const double akk = readlane_f64(a[k], k);
    if (!(akk > 0.0) || !isfinite(akk)) ok = false;
    const double piv = rsqrt_nr(ok ? akk : 1.0);
    const double lik = a[k] * piv;
    a[k] = lik;
    if (lane == k) inv[k] = piv;
#pragma unroll
    for (int j = k + 1; j < CB; ++j) a[j] -= lik * readlane_f64(lik, j);
  }
  if (lane < CB) {
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const double v = (j <= i) ? a[j] : 0.0;
      blk[i * CLD + j] = v;
      blkT[j * CLD + i] = v;
    }
  }
  return ok;
}

template <typename T>
__global__ __launch_bounds__(CHOLB_THREADS) void k_cholesky_blocked(
    const double* __restrict__ E /* summed exchange buffer [S | rhs | diagU | gc | cost] */, int C,
    LMState* __restrict__ st, double* __restrict__ D2c, const ParamSets<T> ps,
    double* __restrict__ delta_c, int n_sys /* size of the system in E: 11*C, or fewer when cameras share parameters */,
    const int32_t* __restrict__ tie /* [11*C] camera parameter -> system row, or NULL = identity */,
    const int32_t* __restrict__ first /* [n_sys] system row -> one camera parameter mapped to it, or NULL */,
    long long* __restrict__ dbg /* optional cycle stamps (diagnostic runs only) */) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (st->status >= 0) return;
  const int cur_ = ps_cur(ps, st);
  const double* __restrict__ cams = ps.cams[cur_];
  double* __restrict__ cams_new = ps.cams[cur_ ^ 1];
  T* __restrict__ campre_new = ps.campre[cur_ ^ 1];
  int nstamp = 0;
#define CHOL_STAMP() do { if (dbg && threadIdx.x == 0) dbg[nstamp] = clock64(); ++nstamp; } while (0)
  CHOL_STAMP();
  const int ncam = C * NCP;      // camera parameters
  const int n = n_sys;           // rows of the system being solved
  const int nb = (n + CB - 1) / CB;
  const int n16 = nb * CB;
  const int nblk = nb * (nb + 1) / 2;
  double* Lb = reinterpret_cast<double*>(smem);                   // nblk blocks
  double* s_LT = Lb + nblk * CBS;                                 // [2][CBS] transposed diagonal block (ping-pong)
  double* s_y = s_LT + 2 * CBS;                                   // [n16]  rhs -> y -> x
  double* s_inv = s_y + n16;                                      // [n16]  1 / L[k][k]
  double* s_d = s_inv + n16;                                      // [n16]  lam * D2c (diagonal damping)
  __shared__ int s_fail;
  __shared__ short s_rc[CHOLB_MAX_NB * (CHOLB_MAX_NB + 1) / 2];   // block index -> (r << 8 | c)
  __shared__ double s_scr[4][CHOLB_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const double* rhs = E + (size_t)n * n;
  const double* dU = rhs + n;
  const double* gct = dU + n;
  const double lam = st->lam;
  const bool fresh = st->fresh != 0;
  // operands of the epilogue, requested now so that their latency is hidden behind the factorisation
  const double my_cam = (tid < ncam) ? cams[tid] : 0.0;                       // camera parameter tid
  const double my_xs = (tid < n) ? cams[first ? first[tid] : tid] : 0.0;      // system unknown tid (a shared one counts once)
  const double my_g = (tid < n) ? gct[tid] : 0.0;

  // camera scaling: monotone max of the squared column norms (x_scale='jac', scipy trf.py:424,545)
  for (int i = tid; i < n16; i += CHOLB_THREADS) {
    double v = 0, dd = 0;
    if (i < n) {
      double d = D2c[i];
      if (fresh) { d = fmax(d, dU[i]); D2c[i] = d; }
      dd = lam * fmax_pos(d);
      v = rhs[i];
    }
    s_y[i] = v;
    s_d[i] = dd;
  }
  if (tid < nblk) {
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= tid) ++r;
    s_rc[tid] = (short)((r << 8) | (tid - r * (r + 1) / 2));
  }
  if (tid == 0) { s_fail = 0; st->cost = E[(size_t)n * n + 3 * n]; }
  __syncthreads();
  // load the lower block triangle (+ damping): every thread issues all of its (<= 33) loads before using any of
  // them, so the fabric latency of reading E (written by other CUs) is paid once; the padded tail is the identity
  {
    constexpr int MAXU = (CHOLB_MAX_NB * (CHOLB_MAX_NB + 1) / 2 + 1) / 2;     // 33
    const int e = tid & 255, ii = e >> 4, jj = e & 15, half = tid >> 8;
    double v[MAXU];
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
      const int b = 2 * u + half;
      v[u] = 0;
      if (b < nblk) {
        const int rc = s_rc[b];
        const int I = (rc >> 8) * CB + ii, J = (rc & 255) * CB + jj;
        if (I < n && J < n) v[u] = E[(size_t)I * n + J];
        else v[u] = (I == J) ? 1.0 : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
      const int b = 2 * u + half;
      if (b < nblk) {
        const int rc = s_rc[b];
        const int I = (rc >> 8) * CB + ii, J = (rc & 255) * CB + jj;
        Lb[b * CBS + ii * CLD + jj] = v[u] + ((I == J && I < n) ? s_d[I] : 0.0);
      }
    }
  }
  __syncthreads();
  CHOL_STAMP();
  if (wid == 0) {
    if (!chol16_wave(Lb + cb_off(0, 0), s_LT, s_inv)) { if (lane == 0) s_fail = 1; }
  }
  __syncthreads();
  CHOL_STAMP();

  for (int jb = 0; jb < nb && !s_fail; ++jb) {
    const int m = (nb - jb - 1) * CB;              // rows below the diagonal block
    const double* LT = s_LT + (jb & 1) * CBS;      // LT[k][j] = L11[j][k]: row k = column k of L11, contiguous
    const double* inv = s_inv + jb * CB;
    // ---- B': L21 = A21 L11^-T row by row, right-looking: x_k = a_k / L_kk ; a_j -= x_k L11[j][k] (j > k).
    //      One more thread does the same with the rhs block, which is the forward solve.
    if (tid <= m + CB) {
      double a[CB];
      // tid < m: panel row ; tid == m: rhs block ; tid in (m, m+16]: row e_j of the identity, whose solution
      // e_j L11^-T is row j of Linv^T -- stored over the diagonal block itself (B' reads L11 only through its
      // transposed copy, and nothing but the back substitution needs the diagonal block afterwards)
      const int jrow = tid - m - 1;
      double* row = (tid < m) ? Lb + cb_off(jb + 1 + (tid >> 4), jb) + (tid & 15) * CLD
                  : (tid == m) ? s_y + jb * CB : Lb + cb_off(jb, jb) + jrow * CLD;
      if (tid <= m) {
#pragma unroll
        for (int k = 0; k < CB; ++k) a[k] = row[k];
      } else {
#pragma unroll
        for (int k = 0; k < CB; ++k) a[k] = (k == jrow) ? 1.0 : 0.0;
      }
#pragma unroll
      for (int k = 0; k < CB; ++k) {
        const double xk = a[k] * inv[k];
        a[k] = xk;
#pragma unroll
        for (int j = k + 1; j < CB; ++j) a[j] -= xk * LT[k * CLD + j];     // wave-uniform address: LDS broadcast
        if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);                 // bound how far loads are hoisted
      }
#pragma unroll
      for (int k = 0; k < CB; ++k) row[k] = a[k];
    }
    __syncthreads();
    CHOL_STAMP();
    // ---- C: trailing update with look-ahead
    if (wid == 0) {
      if (jb + 1 < nb) {
        double* D = Lb + cb_off(jb + 1, jb + 1);
        const double* P = Lb + cb_off(jb + 1, jb) + (lane & 15) * CLD + (lane >> 4);
        Mfma<double>::acc_t acc;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) acc[rg] = D[((lane >> 4) + 4 * rg) * CLD + (lane & 15)];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = Mfma<double>::mma(-P[4 * ks], P[4 * ks], acc);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) D[((lane >> 4) + 4 * rg) * CLD + (lane & 15)] = acc[rg];
        __builtin_amdgcn_wave_barrier();
        if (!chol16_wave(D, s_LT + ((jb + 1) & 1) * CBS, s_inv + (jb + 1) * CB)) { if (lane == 0) s_fail = 1; }
      }
    } else {
      if (wid == 1) {
        // rhs tail: y_i -= sum_k L21[i][k] y_blk[k]
        for (int t = lane; t < m; t += 64) {
          const double* row = Lb + cb_off(jb + 1 + (t >> 4), jb) + (t & 15) * CLD;
          double s0 = 0, s1 = 0;
#pragma unroll
          for (int k = 0; k < CB; k += 2) { s0 += row[k] * s_y[jb * CB + k]; s1 += row[k + 1] * s_y[jb * CB + k + 1]; }
          s_y[(jb + 1) * CB + t] -= s0 + s1;
        }
      }
      // tiles (r,c), jb < c <= r < nb, except (jb+1,jb+1): enumerate the trailing block triangle from index 1
      const int q = nb - jb - 1;
      const int ntile = q * (q + 1) / 2;
      for (int t = 1 + (wid - 1); t < ntile; t += (CHOLB_THREADS / 64 - 1)) {
        const int rc = s_rc[t];
        const int r = jb + 1 + (rc >> 8), c = jb + 1 + (rc & 255);
        double* Dt = Lb + cb_off(r, c);
        const double* Pa = Lb + cb_off(r, jb) + (lane & 15) * CLD + (lane >> 4);
        const double* Pb = Lb + cb_off(c, jb) + (lane & 15) * CLD + (lane >> 4);
        Mfma<double>::acc_t acc;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) acc[rg] = Dt[((lane >> 4) + 4 * rg) * CLD + (lane & 15)];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = Mfma<double>::mma(-Pa[4 * ks], Pb[4 * ks], acc);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) Dt[((lane >> 4) + 4 * rg) * CLD + (lane & 15)] = acc[rg];
      }
    }
    __syncthreads();
    CHOL_STAMP();
  }
  const bool fail = s_fail != 0;
  // ---- back substitution, right-looking:  x_b = Linv_b^T y_b  (wave 0), then every earlier row t < 16 b takes
  //      y_t -= sum_i L[16b+i][t] x_b[i]  (all waves).  The diagonal blocks hold Linv^T.
  if (!fail) {
    for (int b = nb - 1; b >= 0; --b) {
      if (wid == 0) {
        const int j = lane & 15, part = lane >> 4;
        const double* LiT = Lb + cb_off(b, b) + j * CLD;    // row j of Linv^T
        double x = 0;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) { const int i = part + 4 * ii; x += LiT[i] * s_y[b * CB + i]; }
        x += __shfl_xor(x, 16, 64);
        x += __shfl_xor(x, 32, 64);
        __builtin_amdgcn_wave_barrier();
        if (part == 0) s_y[b * CB + j] = x;
      }
      __syncthreads();
      if (tid < b * CB) {
        const double* col = Lb + cb_off(b, tid >> 4) + (tid & 15);
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < CB; i += 2) { s0 += col[i * CLD] * s_y[b * CB + i]; s1 += col[(i + 1) * CLD] * s_y[b * CB + i + 1]; }
        s_y[tid] -= s0 + s1;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  CHOL_STAMP();
  double pred = 0, dx2 = 0, x2 = 0, gm = 0;
  if (tid < ncam) {                                  // 11*C <= 176 < CHOLB_THREADS: one camera parameter per thread
    const double d = fail ? 0.0 : s_y[tie ? tie[tid] : tid];
    delta_c[tid] = d;
    cams_new[tid] = my_cam + d;
  }
  if (tid < n) {                                     // scalars of the step live in the system's own unknowns
    const double d = fail ? 0.0 : s_y[tid];
    pred = 0.5 * d * (s_d[tid] * d - my_g);
    dx2 = d * d;
    x2 = my_xs * my_xs;
    gm = fabs(my_g);
  }
  pred = wave_sum(pred); dx2 = wave_sum(dx2); x2 = wave_sum(x2); gm = wave_max(gm);
  if (lane == 0) { s_scr[0][wid] = pred; s_scr[1][wid] = dx2; s_scr[2][wid] = x2; s_scr[3][wid] = gm; }
  __syncthreads();      // also orders the cams_new stores before the CamPre rebuild below
  if (tid == 0) {
    double p = 0, d2 = 0, xx = 0, g = 0;
    for (int w = 0; w < CHOLB_THREADS / 64; ++w) { p += s_scr[0][w]; d2 += s_scr[1][w]; xx += s_scr[2][w]; g = fmax(g, s_scr[3][w]); }
    st->pred_c = p; st->dx2_c = d2; st->x2_c = xx; st->gmax_c = g;
    st->chol_fail = fail ? 1 : 0;
    st->fresh = 0;
  }
  if (tid < C) campre_build<T>(cams_new + (size_t)tid * NCP, campre_new + (size_t)tid * CAMPRE);
  CHOL_STAMP();
#undef CHOL_STAMP
}

}  // namespace sba
