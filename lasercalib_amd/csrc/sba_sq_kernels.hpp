// sba_sq_kernels.hpp -- the two "squared pixel error" variants of the reference:
//   PySBA.bundle_adjustment_camonly        (pySBA.py:151-173)  cameras free, points fixed
//   PySBA.bundleAdjust_transform_points_3d (pySBA.py:176-205)  one 3x4 affine applied to all points, cameras fixed
// Both minimise rho = w * (project - uv)^2 per pixel component (the reference squares the error before handing it to
// least_squares, pySBA.py:155,185), with scipy's defaults x_scale = 1 and a dense Jacobian.  Here the Jacobian rows
// are analytic, J~ = 2 w Delta dDelta/dtheta, and J~^T J~, J~^T rho and rho.rho come out of the same LDS-tile + MFMA
// 16x16x4 trick as k_linearize_cams: rows [J~ | rho] (<= 13 columns) multiplied by themselves.
#pragma once
#include "sba_lm_kernels.hpp"

namespace SBA_NS {

constexpr int SQ_CHUNK = 1024;     // observations per workgroup of the transform-variant linearize

struct ThetaSets { double* th[2]; int base; };      // 12 affine parameters, current / trial (parity = LMState::cur)

// KIND 1: camera-major chunks of one camera each (cm arrays).  KIND 2: plain chunks of SQ_CHUNK observations (pm arrays).
template <typename T, int KIND>
__global__ __launch_bounds__(256) void k_sq_linearize(
    const ParamSets<T> ps, const ThetaSets ts, const LMState* __restrict__ st, int C,
    const typename Vec2<T>::type* __restrict__ uv, const T* __restrict__ w, const int32_t* __restrict__ ci,
    const int32_t* __restrict__ pi, const int32_t* __restrict__ chunk_cam, const int32_t* __restrict__ chunk_begin,
    const int32_t* __restrict__ chunk_end, int64_t M, double* __restrict__ part /* [n_chunks][256] */) {
  constexpr int LD = 130;
  extern __shared__ __align__(16) unsigned char smem[];
  T* s_tile = reinterpret_cast<T*>(smem);                 // [4][16*LD]
  T* s_cam = s_tile + 4 * 16 * LD;                        // KIND 1: [CAMPRE] ; KIND 2: [C][CAMPRE]
  __shared__ T s_th[12];
  using M_ = Mfma<T>;
  if (st && (st->status >= 0 || !st->need_lin)) return;
  const int cur_ = ps_cur(ps, st);
  const T* __restrict__ campre = ps.campre[cur_];
  const T* __restrict__ ptsT = ps.ptsT[cur_];
  const int chunk = blockIdx.x;
  int beg, end, cam = 0;
  if (KIND == 1) {
    cam = chunk_cam[chunk]; beg = chunk_begin[chunk]; end = chunk_end[chunk];
    if (threadIdx.x < CAMPRE) s_cam[threadIdx.x] = campre[(size_t)cam * CAMPRE + threadIdx.x];
  } else {
    beg = chunk * SQ_CHUNK; end = (int)min((int64_t)beg + SQ_CHUNK, M);
    stage_campre(campre, s_cam, C);
    if (threadIdx.x < 12) s_th[threadIdx.x] = (T)ts.th[(ts.base ^ (st ? st->cur : 0)) & 1][threadIdx.x];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  T* tile = s_tile + wid * 16 * LD;
  typename M_::acc_t acc = {0, 0, 0, 0};
  for (int o0 = beg + wid * 64; o0 < end; o0 += 256) {
    const int o = o0 + lane;
    T row0[16], row1[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { row0[k] = 0; row1[k] = 0; }
    if (o < end) {
      const int p = pi[o];
      const auto m = uv[o];
      const T ww = w ? w[o] : (T)1;
      const T Xo0 = ptsT[3 * (size_t)p], Xo1 = ptsT[3 * (size_t)p + 1], Xo2 = ptsT[3 * (size_t)p + 2];
      T X0 = Xo0, X1 = Xo1, X2 = Xo2;
      const T* cp = s_cam;
      if (KIND == 2) {
        cp = s_cam + ci[o] * CAMPRE;
        X0 = s_th[0] * Xo0 + s_th[1] * Xo1 + s_th[2] * Xo2 + s_th[3];
        X1 = s_th[4] * Xo0 + s_th[5] * Xo1 + s_th[6] * Xo2 + s_th[7];
        X2 = s_th[8] * Xo0 + s_th[9] * Xo1 + s_th[10] * Xo2 + s_th[11];
      }
      T d[2], Jc[2][NCP], Jp[2][3];
      obs_resjac<T>(cp, X0, X1, X2, m.x, m.y, (T)1, d, Jc, Jp);     // unweighted pixel error and its derivatives
      const T f0 = (T)2 * ww * d[0], f1 = (T)2 * ww * d[1];           // d rho / d Delta
      if (KIND == 1) {
#pragma unroll
        for (int k = 0; k < NCP; ++k) { row0[k] = f0 * Jc[0][k]; row1[k] = f1 * Jc[1][k]; }
        row0[NCP] = ww * d[0] * d[0]; row1[NCP] = ww * d[1] * d[1];
      } else {
        const T Xh[4] = {Xo0, Xo1, Xo2, (T)1};                        // theta = params.reshape(3,4): entry [k][l] multiplies Xh[l]
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int l = 0; l < 4; ++l) { row0[4 * k + l] = f0 * Jp[0][k] * Xh[l]; row1[4 * k + l] = f1 * Jp[1][k] * Xh[l]; }
        row0[12] = ww * d[0] * d[0]; row1[12] = ww * d[1] * d[1];
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) { tile[k * LD + lane] = row0[k]; tile[k * LD + 64 + lane] = row1[k]; }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 8
    for (int s = 0; s < 32; ++s) {
      const T v = tile[(lane & 15) * LD + 4 * s + (lane >> 4)];
      acc = M_::mma(v, v, acc);
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  double* s_acc = reinterpret_cast<double*>(s_tile);      // reuse: [4][256] doubles
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) s_acc[wid * 256 + M_::row_of(lane, rg) * 16 + (lane & 15)] = (double)acc[rg];
  __syncthreads();
  part[(size_t)chunk * 256 + threadIdx.x] =
      (s_acc[threadIdx.x] + s_acc[256 + threadIdx.x]) + (s_acc[512 + threadIdx.x] + s_acc[768 + threadIdx.x]);
}

// out16[g][256] = sum over chunks [start[g], start[g+1]) of part[chunk][256].   grid = groups, block = 1024.
__global__ __launch_bounds__(1024) void k_reduce16(const double* __restrict__ part, const int32_t* __restrict__ start,
                                                   double* __restrict__ out16, const LMState* __restrict__ st) {
  __shared__ double s_p[4][256];
  if (st && (st->status >= 0 || !st->need_lin)) return;
  const int g = blockIdx.x, e = threadIdx.x & 255, q = threadIdx.x >> 8;
  const int a = start[g], b = start[g + 1];
  double s0 = 0, s1 = 0;
  int k = a + q;
  for (; k + 4 < b; k += 8) { s0 += part[(size_t)k * 256 + e]; s1 += part[(size_t)(k + 4) * 256 + e]; }
  if (k < b) s0 += part[(size_t)k * 256 + e];
  s_p[q][e] = s0 + s1;
  __syncthreads();
  if (q == 0) out16[(size_t)g * 256 + e] = (s_p[0][e] + s_p[1][e]) + (s_p[2][e] + s_p[3][e]);
}

// cameras-only variant: exchange-layout system  [S = blockdiag(H_c) | rhs = -g | diagU = 1 | g | cost]  (x_scale = 1)
__global__ void k_sq_pack_cams(const double* __restrict__ U16, int C, LMState* __restrict__ st, double* __restrict__ E) {
  if (st->status >= 0) return;
  const int n = C * NCP;
  double* rhs = E + (size_t)n * n;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n * n; idx += gridDim.x * blockDim.x) {
    const int i = idx / n, j = idx - i * n;
    const int ci_ = i / NCP, cj_ = j / NCP;
    E[idx] = (ci_ == cj_) ? U16[(size_t)ci_ * 256 + (i - ci_ * NCP) * 16 + (j - cj_ * NCP)] : 0.0;
  }
  if (blockIdx.x == 0) {
    __shared__ double s_max[256], s_cost[256];
    double mx = 0, cs = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const int c = i / NCP, e = i - c * NCP;
      const double g = U16[(size_t)c * 256 + e * 16 + NCP];
      rhs[i] = -g; rhs[n + i] = 1.0; rhs[2 * n + i] = g;
      mx = fmax(mx, U16[(size_t)c * 256 + e * 16 + e]);
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) cs += U16[(size_t)c * 256 + NCP * 16 + NCP];
    s_max[threadIdx.x] = mx; s_cost[threadIdx.x] = cs;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int t = 1; t < (int)blockDim.x; ++t) { mx = fmax(mx, s_max[t]); cs += s_cost[t]; }
      rhs[3 * n] = 0.5 * cs;
      if (st->lam < 0) st->lam = -st->lam * fmax(mx, 1e-300);     // first call: lambda = tau * max diag(J^T J)
    }
  }
}

// transform variant: solve the 12x12 system in one thread (it is 12x12), write the trial affine and the step's scalars
__global__ void k_sq_solve12(const double* __restrict__ H16, LMState* __restrict__ st, const ThetaSets ts) {
  if (st->status >= 0 || threadIdx.x != 0 || blockIdx.x != 0) return;
  const int cur_ = (ts.base ^ st->cur) & 1;
  const double* th = ts.th[cur_];
  double* th_new = ts.th[cur_ ^ 1];
  double A[12][12], g[12], x[12];
  double mx = 0;
  for (int i = 0; i < 12; ++i) { g[i] = H16[i * 16 + 12]; mx = fmax(mx, H16[i * 16 + i]); }
  if (st->lam < 0) st->lam = -st->lam * fmax(mx, 1e-300);
  const double lam = st->lam;
  for (int i = 0; i < 12; ++i)
    for (int j = 0; j < 12; ++j) A[i][j] = H16[i * 16 + j] + (i == j ? lam : 0.0);
  bool fail = false;
  for (int k = 0; k < 12 && !fail; ++k) {                 // in-place Cholesky (lower)
    double d = A[k][k];
    for (int j = 0; j < k; ++j) d -= A[k][j] * A[k][j];
    if (!(d > 0.0) || !isfinite(d)) { fail = true; break; }
    const double lkk = sqrt(d);
    A[k][k] = lkk;
    for (int i = k + 1; i < 12; ++i) {
      double s = A[i][k];
      for (int j = 0; j < k; ++j) s -= A[i][j] * A[k][j];
      A[i][k] = s / lkk;
    }
  }
  for (int i = 0; i < 12; ++i) {                          // L y = -g
    double s = -g[i];
    for (int j = 0; j < i; ++j) s -= A[i][j] * x[j];
    x[i] = fail ? 0.0 : s / A[i][i];
  }
  for (int i = 11; i >= 0; --i) {                         // L^T delta = y
    double s = x[i];
    for (int j = i + 1; j < 12; ++j) s -= A[j][i] * x[j];
    x[i] = fail ? 0.0 : s / A[i][i];
  }
  double pred = 0, dx2 = 0, x2 = 0, gm = 0;
  for (int i = 0; i < 12; ++i) {
    th_new[i] = th[i] + x[i];
    pred += 0.5 * x[i] * (lam * x[i] - g[i]);
    dx2 += x[i] * x[i]; x2 += th[i] * th[i]; gm = fmax(gm, fabs(g[i]));
  }
  st->cost = 0.5 * H16[12 * 16 + 12];
  st->pred_c = pred; st->dx2_c = dx2; st->x2_c = x2; st->gmax_c = gm; st->chol_fail = fail ? 1 : 0; st->fresh = 0;
}

// trial cost 0.5 * sum rho^2 at the trial parameters.  trial_part layout = k_decide's: [cost | 0 | 0 | 0] x nblk
template <typename T, int KIND>
__global__ __launch_bounds__(PM_BLOCK) void k_sq_trial(
    const ParamSets<T> ps, const ThetaSets ts, const LMState* __restrict__ st, int C,
    const typename Vec2<T>::type* __restrict__ uv, const T* __restrict__ w, const int32_t* __restrict__ ci,
    const int32_t* __restrict__ pi, int64_t M, double* __restrict__ trial_part, int nblk) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* s_cam = reinterpret_cast<T*>(smem);
  __shared__ double s_red[PM_BLOCK / 64];
  __shared__ T s_th[12];
  if (st->status >= 0) return;
  const int cur_ = ps_cur(ps, st);
  // cameras-only: trial cameras, fixed points ; transform: fixed cameras, trial affine
  stage_campre(KIND == 1 ? ps.campre[cur_ ^ 1] : ps.campre[cur_], s_cam, C);
  if (KIND == 2 && threadIdx.x < 12) s_th[threadIdx.x] = (T)ts.th[((ts.base ^ st->cur) & 1) ^ 1][threadIdx.x];
  const T* __restrict__ ptsT = ps.ptsT[cur_];
  __syncthreads();
  const int64_t o = (int64_t)blockIdx.x * PM_BLOCK + threadIdx.x;
  double sq = 0;
  if (o < M) {
    const int p = pi[o];
    const auto m = uv[o];
    const T ww = w ? w[o] : (T)1;
    T X0 = ptsT[3 * (size_t)p], X1 = ptsT[3 * (size_t)p + 1], X2 = ptsT[3 * (size_t)p + 2];
    if (KIND == 2) {
      const T a0 = X0, a1 = X1, a2 = X2;
      X0 = s_th[0] * a0 + s_th[1] * a1 + s_th[2] * a2 + s_th[3];
      X1 = s_th[4] * a0 + s_th[5] * a1 + s_th[6] * a2 + s_th[7];
      X2 = s_th[8] * a0 + s_th[9] * a1 + s_th[10] * a2 + s_th[11];
    }
    T u, v;
    obs_project<T>(s_cam + ci[o] * CAMPRE, X0, X1, X2, u, v);
    const double r0 = (double)ww * (double)(u - m.x) * (double)(u - m.x), r1 = (double)ww * (double)(v - m.y) * (double)(v - m.y);
    sq = r0 * r0 + r1 * r1;
  }
  const double s = block_sum(sq, s_red);
  if (threadIdx.x == 0) {
    trial_part[blockIdx.x] = 0.5 * s;
    trial_part[nblk + blockIdx.x] = 0; trial_part[2 * nblk + blockIdx.x] = 0; trial_part[3 * nblk + blockIdx.x] = 0;
  }
}

}  // namespace SBA_NS
