// sba_lm_kernels.hpp -- the Levenberg-Marquardt control path on the device: reduced-system assembly,
// dense Cholesky of the camera system, back-substitution fused with the trial residual, and the
// accept/reject + termination logic.  Replaces the scipy TRF loop the reference drives
// (scipy/optimize/_lsq/trf.py:401-560) with an exact damped solve; termination tests and status
// codes keep scipy's meaning (scipy/optimize/_lsq/common.py:705-717).
#pragma once
#include <cstddef>
#include "sba_kernels.hpp"

namespace SBA_NS {

constexpr int CHOL_THREADS = 1024;
constexpr int CHOL_LDS_MAX_N = 176;    // packed lower triangle of 176x176 doubles = 124.6 KB of the 160 KB LDS

// packed exchange layout (multi-rank): row i of the upper triangle starts at i*n - i(i-1)/2
__host__ __device__ inline size_t exch_packed_index(int n, int i, int j) { return (size_t)i * n - (size_t)i * (i - 1) / 2 + (size_t)(j - i); }
__host__ __device__ inline size_t exch_packed_size(int n) { return (size_t)n * (n + 1) / 2 + 3 * (size_t)n + 1; }

// ------------------------------------------------------------------ reduced system assembly
// One launch builds the whole exchange buffer E = [S | rhs | diagU | gc | cost]:
//   blocks [0, 4*NT*npairs):   S(i,j) = [same camera] U(i,j) - sum_ks slab      (NT = 121 tile slots per pair)
//       block = (pair, tile, quarter of the tile's 256 entries); 1024 threads = 16 k-split groups x 64 entries,
//       group g sums slabs g, g+16, ...; the groups are folded through LDS in a fixed order (deterministic).
//   blocks [.., +nrow_blocks): rhs = -gc + sum_ks bpart ; diagU ; gc           (16 rows x 64 k-split groups)
//   last block:                cost = sum cost_part
template <typename T>
__global__ __launch_bounds__(1024) void k_build_exchange(
    const T* __restrict__ slabs, const double* __restrict__ bpart, int ksplit, const int32_t* __restrict__ pair_ga,
    const int32_t* __restrict__ pair_gb, int npairs, const double* __restrict__ U, const double* __restrict__ gc,
    const double* __restrict__ cost_part, int n_cost_part, int C, int free_cams, double* __restrict__ E,
    const LMState* __restrict__ st,
    const double* __restrict__ gdpart /* fused linearisation: per-workgroup g_c / diag U partials [ksplit][2][176] (then the
                                          slabs hold Schur partials - U and bpart holds b - g_c); NULL = classic U / gc */,
    double* __restrict__ Pk = nullptr /* multi-rank: the same system once more, upper triangle packed row by row
                                         [S n(n+1)/2 | rhs | diagU | gc | cost] -- what travels through the all-reduce */,
    int emajor_mode = 0 /* parameter-major tiles (tile (R, Tc) holds parameters (R, Tc) of the camera pairs, row 16 e + c):
                           1 = every pair (k_schur_fused_bf3), 2 = the diagonal pairs (k_schur_diag_bf3),
                           3 = k_schur_fused_wide: one slab per workgroup, compact rows e*C + c in ceil(11 C / 16) tiles, row partials
                               (bpart / gdpart) in the exchange buffer's own order with a stride of WIDE_ROWS */,
    int nt_launch = GROUP_TILES * GROUP_TILES /* tile slots per pair the grid covers: 121 with several pairs (an off-diagonal pair has
                           that many), the pair's own count -- 66, or the wide kernel's ntw (ntw + 1) / 2 -- when there is one pair: the
                           220 workgroups of 16 waves that returned at once cost a third of the launch */) {
  using M_ = Mfma<T>;
  // tiles: 64 entries x 16 k-split groups per block (256-byte segments per group load; 16 entries x 64 groups was
  // measured slower: 64-byte segments, four times the blocks).  Rows: 16 rows x 64 groups, see below.
  constexpr int EPB = 64, NG = 1024 / EPB, BPT = 256 / EPB;       // entries per block, groups, blocks per tile
  __shared__ double s_p[NG][EPB];
  // the record is requested now and looked at right before the first write: the slab loads do not wait for its round trip
  const int st_status = st->status;
  __shared__ double scr[16];
  constexpr int NT = GROUP_TILES * GROUP_TILES;
  const int n = C * NCP;
  const int tile_blocks = free_cams ? BPT * nt_launch * npairs : 0;
  const int row_blocks = free_cams ? (n + 15) / 16 : 0;
  const int g = threadIdx.x / EPB, l16 = threadIdx.x & (EPB - 1);
  int bid = blockIdx.x;
  if (bid < tile_blocks) {
    const bool wide = emajor_mode == 3;
    const int pair = wide ? 0 : bid / (BPT * nt_launch);
    const int rem = bid - pair * BPT * nt_launch;
    const int t = rem / BPT, part = rem - t * BPT;
    const int ga = pair_ga[pair], gb = pair_gb[pair];
    const bool diag = wide || (ga == gb);
    const int ntw = wide_ntw(C);
    const int ntile = wide ? ntw * (ntw + 1) / 2 : diag ? (GROUP_TILES * (GROUP_TILES + 1)) / 2 : NT;
    if (t >= ntile) return;
    int R, Tc;
    if (wide) wide_tile_rc(ntw, t, R, Tc);
    else schur_tile_rc(diag, t, R, Tc);
    const int e = part * EPB + l16;                   // entry of the tile's register dump: lane = e>>2, reg = e&3
    const int rg = e & 3, lane = e >> 2;            // slab layout [tile][lane][reg]
    const bool emajor = wide || emajor_mode == 1 || (emajor_mode == 2 && diag);
    int i, j;
    if (wide) {                                       // compact rows: row = e C + c  ->  exchange index c * NCP + e
      const int rho = 16 * R + M_::row_of(lane, rg), kap = 16 * Tc + (lane & 15);
      const int er = rho / C, ek = kap / C;
      i = rho < n ? (rho - er * C) * NCP + er : n;
      j = kap < n ? (kap - ek * C) * NCP + ek : n;
    } else {
      i = emajor ? (ga * GROUP_CAMS + M_::row_of(lane, rg)) * NCP + R : ga * GROUP_ROWS + 16 * R + M_::row_of(lane, rg);
      j = emajor ? (gb * GROUP_CAMS + (lane & 15)) * NCP + Tc : gb * GROUP_ROWS + 16 * Tc + (lane & 15);
    }
    const size_t stride = wide ? (size_t)WIDE_SLOTS * 256 : (size_t)NT * 256;
    const T* src = slabs + (size_t)pair * ksplit * stride + (size_t)t * 256 + e;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (ksplit == 16 * NG) {
      // one workgroup per CU (256 slabs): every group adds exactly 16 of them -- all 16 loads are in flight at once instead of four
      // rounds of four (the kernel is a chain of memory round trips, not a bandwidth problem); same summation order as the loop below
      T v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(g + u * NG) * stride];
#pragma unroll
      for (int u = 0; u < 16; u += 4) { s0 += (double)v[u]; s1 += (double)v[u + 1]; s2 += (double)v[u + 2]; s3 += (double)v[u + 3]; }
    } else {
      int k = g;
      for (; k + 3 * NG < ksplit; k += 4 * NG) {
        s0 += (double)src[(size_t)k * stride];
        s1 += (double)src[(size_t)(k + NG) * stride];
        s2 += (double)src[(size_t)(k + 2 * NG) * stride];
        s3 += (double)src[(size_t)(k + 3 * NG) * stride];
      }
      for (; k < ksplit; k += NG) s0 += (double)src[(size_t)k * stride];
    }
    const int ci_ = i / NCP, cj_ = j / NCP;
    s_p[g][l16] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (st_status >= 0) return;
    if (g == 0 && i < n && j < n) {
      double s = 0;
#pragma unroll
      for (int q = 0; q < NG; ++q) s += s_p[q][l16];
      double v = -s;                 // fused linearisation: the slabs already hold (Schur partials - U)
      if (!gdpart && ci_ == cj_) v += U[(size_t)ci_ * NCP * NCP + (i - ci_ * NCP) * NCP + (j - cj_ * NCP)];
      const bool sym_tile = diag && R == Tc;          // the tile holds both (i, j) and (j, i)
      if (!emajor) {
        // camera-major tiles: a symmetric tile is panel^T panel with the SAME fragments on both sides, bitwise symmetric by itself
        E[(size_t)i * n + j] = v;
        if (!sym_tile) E[(size_t)j * n + i] = v;
        if (Pk && i <= j) Pk[exch_packed_index(n, i, j)] = v;
      } else if (!sym_tile || i <= j) {
        // parameter-major tiles (bf16 x 3): the two halves of a symmetric tile add the same six partial products in a different
        // order (h m' and m h' swap), so they differ in the last bits -- the upper half is the value of both, as in the packed copy
        E[(size_t)i * n + j] = v;
        E[(size_t)j * n + i] = v;
        if (Pk) Pk[exch_packed_index(n, i < j ? i : j, i < j ? j : i)] = v;
      }
    }
    return;
  }
  bid -= tile_blocks;
  double* rhs = E + (size_t)n * n;
  double* dU = rhs + n;
  double* gv = dU + n;
  if (bid < row_blocks) {
    // 16 rows x 64 k-split groups per block: a thread adds at most ksplit/64 partials per array, so the dependent
    // chain of fabric round trips is 4 long at 256 k-splits (it was 16 with 64 rows x 16 groups)
    constexpr int RB = 16, RG = 1024 / RB;
    __shared__ double s_r[3][RG][RB];
    const int lr = threadIdx.x & (RB - 1), gr = threadIdx.x / RB;
    const int i = bid * RB + lr;
    double b0 = 0, g0 = 0, d0 = 0;
    if (i < n) {
      const bool wide = emajor_mode == 3;
      const int rstride = wide ? WIDE_ROWS : GROUP_ROWS;
      const int grp = wide ? 0 : i / GROUP_ROWS, rho = i - grp * GROUP_ROWS;
      const double* src = bpart + (size_t)grp * ksplit * GROUP_ROWS + rho;
      if (gdpart) {        // fused linearisation: g_c and diag(U) are per-workgroup partial rows laid out like bpart ([group][k-split][2][rows])
        const double* gs_ = gdpart + (size_t)grp * ksplit * 2 * GROUP_ROWS + rho;
        for (int k = gr; k < ksplit; k += RG) {
          b0 += src[(size_t)k * rstride];
          g0 += gs_[(size_t)k * 2 * rstride];
          d0 += gs_[((size_t)k * 2 + 1) * rstride];
        }
      } else {
        for (int k = gr; k < ksplit; k += RG) b0 += src[(size_t)k * GROUP_ROWS];
      }
    }
    s_r[0][gr][lr] = b0; s_r[1][gr][lr] = g0; s_r[2][gr][lr] = d0;
    __syncthreads();
    if (st_status >= 0) return;
    if (gr == 0 && i < n) {
      double bs = 0, gs = 0, dsv = 0;
#pragma unroll 8
      for (int q = 0; q < RG; ++q) { bs += s_r[0][q][lr]; gs += s_r[1][q][lr]; dsv += s_r[2][q][lr]; }
      double v_r, v_d, v_g;
      if (gdpart) {
        v_r = bs;                      // k_schur_fused stored b - g_c
        v_d = dsv;
        v_g = gs;
      } else {
        const int c = i / NCP, e = i - c * NCP;
        v_r = -gc[i] + bs;
        v_d = U[(size_t)c * NCP * NCP + e * NCP + e];
        v_g = gc[i];
      }
      rhs[i] = v_r; dU[i] = v_d; gv[i] = v_g;
      if (Pk) { double* pr = Pk + exch_packed_index(n, n - 1, n - 1) + 1; pr[i] = v_r; pr[n + i] = v_d; pr[2 * n + i] = v_g; }
    }
    return;
  }
  double s = 0;
  for (int i = threadIdx.x; i < n_cost_part; i += blockDim.x) s += cost_part[i];
  s = block_sum(s, scr);
  if (st_status >= 0) return;
  if (threadIdx.x == 0) {
    E[(size_t)n * n + 3 * n] = s;
    if (Pk) Pk[exch_packed_size(n) - 1] = s;
  }
}

// all-reduced packed system -> the full symmetric layout the factorisation kernels read
__global__ void k_unpack_exchange(const double* __restrict__ Pk, int n, int free_cams, double* __restrict__ E,
                                  const LMState* __restrict__ st) {
  if (st->status >= 0) return;
  const size_t nn = (size_t)n * n;
  const size_t tail = exch_packed_index(n, n - 1, n - 1) + 1;
  const size_t total = free_cams ? nn + 3 * (size_t)n : 0;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    if (idx < nn) {
      const int i = (int)(idx / n), j = (int)(idx - (size_t)i * n);
      E[idx] = Pk[exch_packed_index(n, i < j ? i : j, i < j ? j : i)];
    } else {
      E[idx] = Pk[tail + (idx - nn)];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) E[nn + 3 * (size_t)n] = Pk[exch_packed_size(n) - 1];
}

// ------------------------------------------------------------------ shared camera parameters (PySBA.bundleAdjust_sharedcam)
// The reference's shared-intrinsics variant (pySBA.py:252-325) optimises f, k1, k2 once for all cameras.  With
// delta_c = T delta_s (T a 0/1 matrix, `tie` = its row->column map) the reduced system in the tied unknowns is
// S_s = T^T S T, rhs_s = T^T rhs, and the Jacobian column norms add up: diagU_s = T^T diagU.  grid = n_s rows.
__global__ void k_tie_system(const double* __restrict__ E, int n, int n_s, const int32_t* __restrict__ pre_start,
                             const int32_t* __restrict__ pre_idx, double* __restrict__ Es, const LMState* __restrict__ st) {
  if (st->status >= 0) return;
  const int a = blockIdx.x;
  const double* rhs = E + (size_t)n * n;
  double* rhs_s = Es + (size_t)n_s * n_s;
  for (int b = threadIdx.x; b < n_s; b += blockDim.x) {
    double s = 0;
    for (int ia = pre_start[a]; ia < pre_start[a + 1]; ++ia) {
      const double* row = E + (size_t)pre_idx[ia] * n;
      for (int ib = pre_start[b]; ib < pre_start[b + 1]; ++ib) s += row[pre_idx[ib]];
    }
    Es[(size_t)a * n_s + b] = s;
  }
  if (threadIdx.x < 3) {       // rhs, diagU, gc
    double s = 0;
    for (int ia = pre_start[a]; ia < pre_start[a + 1]; ++ia) s += rhs[(size_t)threadIdx.x * n + pre_idx[ia]];
    rhs_s[(size_t)threadIdx.x * n_s + a] = s;
  }
  if (a == 0 && threadIdx.x == 0) Es[(size_t)n_s * n_s + 3 * n_s] = E[(size_t)n * n + 3 * n];
}

// ------------------------------------------------------------------ dense Cholesky + solve of the reduced camera system
// Single workgroup.  A = S + lam*diag(D2c) ; A = L L^T ; delta_c = A^-1 rhs.  LDSMODE keeps the packed
// lower triangle in LDS (n <= 176); otherwise factors in place in the caller's S copy in global memory.
struct TriLds {
  double* a;
  __device__ inline double& at(int i, int j) const { return a[(size_t)i * (i + 1) / 2 + j]; }   // i >= j
};
struct TriGlobal {
  double* a; int n;
  __device__ inline double& at(int i, int j) const { return a[(size_t)i * n + j]; }
};

template <bool LDSMODE, typename T>
__global__ __launch_bounds__(CHOL_THREADS) void k_cholesky_solve(
    double* __restrict__ E /* summed exchange buffer; S is destroyed in global mode */, int C,
    LMState* __restrict__ st, double* __restrict__ D2c, const ParamSets<T> ps,
    double* __restrict__ delta_c, int n_sys, const int32_t* __restrict__ tie, const int32_t* __restrict__ first) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (st->status >= 0) return;
  const int cur_ = ps_cur(ps, st);
  const double* __restrict__ cams = ps.cams[cur_];
  double* __restrict__ cams_new = ps.cams[cur_ ^ 1];
  T* __restrict__ campre_new = ps.campre[cur_ ^ 1];
  __shared__ double s_y[GROUP_ROWS * 8 > 1408 ? GROUP_ROWS * 8 : 1408];   // rhs / solution, n <= 1408
  __shared__ double s_piv;
  __shared__ int s_fail;
  __shared__ double s_scr[CHOL_THREADS / 64];
  const int ncam = C * NCP;
  const int n = n_sys;
  const int tid = threadIdx.x;
  double* S = E;
  const double* rhs = E + (size_t)n * n;
  const double* dU = rhs + n;
  const double* gct = dU + n;
  const double lam = st->lam;
  // camera scaling: monotone max of the column norms (x_scale='jac', scipy trf.py:424,545)
  const bool fresh = st->fresh != 0;
  for (int i = tid; i < n; i += CHOL_THREADS) {
    double d = D2c[i];
    if (fresh) { d = fmax(d, dU[i]); D2c[i] = d; }
    s_y[i] = rhs[i];
  }
  if (tid == 0) { s_fail = 0; if (tid == 0 && E) st->cost = E[(size_t)n * n + 3 * n]; }
  __syncthreads();
  TriLds Ll{reinterpret_cast<double*>(smem)};
  TriGlobal Lg{S, n};
  auto AT = [&](int i, int j) -> double& { if constexpr (LDSMODE) return Ll.at(i, j); else return Lg.at(i, j); };
  // load (+ damping)
  for (int idx = tid; idx < n * n; idx += CHOL_THREADS) {
    const int i = idx / n, j = idx - i * n;
    if (j > i) continue;
    double v = S[idx];
    if (i == j) v += lam * fmax_pos(D2c[i]);
    AT(i, j) = v;
  }
  __syncthreads();
  const int tx = tid & 31, ty = tid >> 5;
  for (int k = 0; k < n; ++k) {
    if (tid == 0) {
      const double d = AT(k, k);
      if (!(d > 0.0) || !isfinite(d)) { s_fail = 1; s_piv = 1.0; }
      else s_piv = 1.0 / sqrt(d);
    }
    __syncthreads();
    if (s_fail) break;
    const double ip = s_piv;
    for (int i = k + tid; i < n; i += CHOL_THREADS) AT(i, k) *= ip;    // includes the diagonal: L_kk = sqrt(d)
    __syncthreads();
    for (int i = k + 1 + ty; i < n; i += 32) {
      const double lik = AT(i, k);
      for (int j = k + 1 + tx; j <= i; j += 32) AT(i, j) -= lik * AT(j, k);
    }
    __syncthreads();
  }
  const bool fail = s_fail != 0;
  // triangular solves inside wave 0 (no workgroup barriers): L y = rhs ; L^T x = y
  if (!fail && tid < 64) {
    for (int k = 0; k < n; ++k) {
      const double yk = s_y[k] / AT(k, k);
      __builtin_amdgcn_wave_barrier();
      if (tid == 0) s_y[k] = yk;
      for (int i = k + 1 + tid; i < n; i += 64) s_y[i] -= AT(i, k) * yk;
      __builtin_amdgcn_wave_barrier();
    }
    for (int k = n - 1; k >= 0; --k) {
      const double xk = s_y[k] / AT(k, k);
      __builtin_amdgcn_wave_barrier();
      if (tid == 0) s_y[k] = xk;
      for (int i = tid; i < k; i += 64) s_y[i] -= AT(k, i) * xk;
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  double pred = 0, dx2 = 0, x2 = 0, gm = 0;
  for (int i = tid; i < ncam; i += CHOL_THREADS) {
    const double d = fail ? 0.0 : s_y[tie ? tie[i] : i];
    delta_c[i] = d;
    cams_new[i] = cams[i] + d;
  }
  for (int i = tid; i < n; i += CHOL_THREADS) {
    const double d = fail ? 0.0 : s_y[i];
    const double x = cams[first ? first[i] : i];
    pred += 0.5 * d * (lam * fmax_pos(D2c[i]) * d - gct[i]);
    dx2 += d * d;
    x2 += x * x;
    gm = fmax(gm, fabs(gct[i]));
  }
  pred = block_sum(pred, s_scr);
  dx2 = block_sum(dx2, s_scr);
  x2 = block_sum(x2, s_scr);
  gm = block_max(gm, s_scr);
  if (tid == 0) {
    st->pred_c = pred; st->dx2_c = dx2; st->x2_c = x2; st->gmax_c = gm;
    st->chol_fail = fail ? 1 : 0;
    st->fresh = 0;
  }
  __syncthreads();
  if (tid < C) campre_build<T>(cams_new + (size_t)tid * NCP, campre_new + (size_t)tid * CAMPRE);
}

// points-only mode: no camera system.  Zero step for the cameras, cost from the partials.
__global__ void k_nocam_step(LMState* __restrict__ st, const double* __restrict__ E, int n) {
  if (st->status >= 0) return;
  if (threadIdx.x == 0) {
    st->cost = E[(size_t)n * n + 3 * n];
    st->pred_c = 0; st->dx2_c = 0; st->x2_c = 0; st->gmax_c = 0; st->chol_fail = 0; st->fresh = 0;
  }
}

// ------------------------------------------------------------------ large camera systems: prepare / epilogue around a separate factorisation
// Systems that do not go through one of the all-in-one kernels (k_cholesky_blocked up to 176 unknowns, k_cholesky_ll up to 256)
// are factored by k_cholesky_stream or the multi-workgroup kernels of sba_chol_big.hpp, bracketed by these two kernels, which
// keep the LM-specific parts: Marquardt scaling + damping before; step, trial cameras and the step's scalars after.
__global__ void k_chol_prepare(double* __restrict__ E, int n, LMState* __restrict__ st, double* __restrict__ D2c,
                               double* __restrict__ sol /* [n] rhs in, solution out */) {
  if (st->status >= 0) return;
  const double* rhs = E + (size_t)n * n;
  const double* dU = rhs + n;
  const double lam = st->lam;
  const bool fresh = st->fresh != 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double d = D2c[i];
    if (fresh) { d = fmax(d, dU[i]); D2c[i] = d; }
    E[(size_t)i * n + i] += lam * fmax_pos(d);
    sol[i] = rhs[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) st->cost = E[(size_t)n * n + 3 * n];
}

// (a device function: k_chol_big_back_all's last workgroup runs it on the solution it holds in LDS instead of a launch of its own)
template <typename T>
__device__ __forceinline__ void chol_epilogue_body(
    const double* __restrict__ E, int C, int n, LMState* __restrict__ st, const double* __restrict__ D2c,
    const ParamSets<T>& ps, double* __restrict__ delta_c, const double* sol, const int* __restrict__ info,
    const int32_t* __restrict__ tie, const int32_t* __restrict__ first, double* __restrict__ s_scr /* [16] */, int* __restrict__ s_flag) {
  const int cur_ = ps_cur(ps, st);
  const double* __restrict__ cams = ps.cams[cur_];
  double* __restrict__ cams_new = ps.cams[cur_ ^ 1];
  T* __restrict__ campre_new = ps.campre[cur_ ^ 1];
  const int ncam = C * NCP;
  const double* gct = E + (size_t)n * n + 2 * n;
  const double lam = st->lam;
  bool fail = (*info != 0);
  // a factorisation of an indefinite matrix that "succeeds" numerically still shows up as a non-finite step
  double bad = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) if (!isfinite(sol[i])) bad = 1;
  bad = block_max(bad, s_scr);
  if (threadIdx.x == 0) *s_flag = (fail || bad > 0) ? 1 : 0;
  __syncthreads();
  fail = *s_flag != 0;
  double pred = 0, dx2 = 0, x2 = 0, gm = 0;
  for (int i = threadIdx.x; i < ncam; i += blockDim.x) {
    const double d = fail ? 0.0 : sol[tie ? tie[i] : i];
    delta_c[i] = d;
    cams_new[i] = cams[i] + d;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double d = fail ? 0.0 : sol[i];
    const double x = cams[first ? first[i] : i];
    pred += 0.5 * d * (lam * fmax_pos(D2c[i]) * d - gct[i]);
    dx2 += d * d;
    x2 += x * x;
    gm = fmax(gm, fabs(gct[i]));
  }
  pred = block_sum(pred, s_scr);
  dx2 = block_sum(dx2, s_scr);
  x2 = block_sum(x2, s_scr);
  gm = block_max(gm, s_scr);
  if (threadIdx.x == 0) {
    st->pred_c = pred; st->dx2_c = dx2; st->x2_c = x2; st->gmax_c = gm;
    st->chol_fail = fail ? 1 : 0;
    st->fresh = 0;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) campre_build<T>(cams_new + (size_t)c * NCP, campre_new + (size_t)c * CAMPRE);
}
template <typename T>
__global__ __launch_bounds__(1024) void k_chol_epilogue(
    const double* __restrict__ E, int C, int n, LMState* __restrict__ st, const double* __restrict__ D2c,
    const ParamSets<T> ps, double* __restrict__ delta_c, const double* __restrict__ sol, const int* __restrict__ info,
    const int32_t* __restrict__ tie, const int32_t* __restrict__ first) {
  __shared__ double s_scr[16];
  __shared__ int s_fail;
  if (st->status >= 0) return;
  chol_epilogue_body<T>(E, C, n, st, D2c, ps, delta_c, sol, info, tie, first, s_scr, &s_fail);
}

// ------------------------------------------------------------------ K6: back-substitution + trial point + trial residual
// Same point-aligned workgroups as k_linearize_points.
//   phase 1 (lane = observation): t_i = Jp^T (Jc delta_c[cam])                      -> LDS
//   phase 2 (lane = point):       delta_p = -(V + lam D)^-1 (gp + sum_i t_i) ; X_new -> LDS + global
//   phase 3 (lane = observation): residual at (cams_new, X_new)                     -> cost partial
// partials per block: trial_part[4][nblk] = cost_new, pred_p, |delta_p|^2, |X|^2
template <typename T>
__global__ __launch_bounds__(PM_BLOCK) void k_backsub_trial(
    const ParamSets<T> ps, int C, const typename Vec2<T>::type* __restrict__ uv, const T* __restrict__ w,
    const int32_t* __restrict__ ci, const int32_t* __restrict__ pi, const int32_t* __restrict__ pt_start,
    const int4* __restrict__ blk_desc /* {p_lo, p_hi, o_lo, o_hi} per block */,
    const T* __restrict__ pf, const double* __restrict__ gp, const double* __restrict__ D2p,
    const double* __restrict__ delta_c, const LMState* __restrict__ st, double* __restrict__ trial_part, int nblk) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (st->status >= 0) return;
  const int cur_ = ps_cur(ps, st);
  const T* __restrict__ campre = ps.campre[cur_];
  const T* __restrict__ campre_new = ps.campre[cur_ ^ 1];
  const double* __restrict__ pts = ps.pts[cur_];
  const T* __restrict__ ptsT = ps.ptsT[cur_];
  double* __restrict__ pts_new = ps.pts[cur_ ^ 1];
  T* __restrict__ ptsT_new = ps.ptsT[cur_ ^ 1];
  double* s_t = reinterpret_cast<double*>(smem);          // [256][3]
  double* s_xn = s_t + PM_BLOCK * 3;                      // [256][3] new point coordinates (per local point)
  T* s_cam = reinterpret_cast<T*>(s_xn + PM_BLOCK * 3);   // [C][CAMPRE] current
  T* s_camn = s_cam + C * CAMPRE;                         // [C][CAMPRE] trial
  T* s_dc = s_camn + C * CAMPRE;                          // [C*11]
  __shared__ double s_scr[PM_BLOCK / 64];
  const bool free_cams = st->free_cams != 0;
  const double lam = st->lam;
  stage_campre(campre, s_cam, C);
  stage_campre(campre_new, s_camn, C);
  for (int i = threadIdx.x; i < C * NCP; i += PM_BLOCK) s_dc[i] = (T)delta_c[i];
  const int4 bd = blk_desc[blockIdx.x];
  const int p_lo = bd.x, p_hi = bd.y;
  const int o_lo = bd.z, o_hi = bd.w;
  const int nobs = o_hi - o_lo, npts = p_hi - p_lo;
  const bool own_pt = (int)threadIdx.x < npts;      // phase 2: thread q < npts owns point p_lo + q
  const size_t myq = (size_t)(p_lo + (own_pt ? (int)threadIdx.x : 0));
  __syncthreads();
  int my_p = -1, my_c = 0;
  typename Vec2<T>::type my_uv; my_uv.x = 0; my_uv.y = 0;
  T my_w = (T)1;
  if ((int)threadIdx.x < nobs) {
    const int o = o_lo + threadIdx.x;
    my_p = pi[o];
    my_c = ci[o];
    my_uv = uv[o];
    my_w = w ? w[o] : (T)1;
    double t0 = 0, t1 = 0, t2 = 0;
    if (free_cams) {
      T r[2], Jc[2][NCP], Jp[2][3];
      obs_resjac<T>(s_cam + my_c * CAMPRE, ptsT[3 * (size_t)my_p], ptsT[3 * (size_t)my_p + 1],
                    ptsT[3 * (size_t)my_p + 2], my_uv.x, my_uv.y, my_w, r, Jc, Jp);
      if (ps.loss_delta > 0.f) (void)robust_apply<T>(ps.loss(), r, Jc, Jp);
      T s0 = 0, s1 = 0;
#pragma unroll
      for (int e = 0; e < NCP; ++e) { s0 += Jc[0][e] * s_dc[my_c * NCP + e]; s1 += Jc[1][e] * s_dc[my_c * NCP + e]; }
      t0 = (double)(Jp[0][0] * s0 + Jp[1][0] * s1);
      t1 = (double)(Jp[0][1] * s0 + Jp[1][1] * s1);
      t2 = (double)(Jp[0][2] * s0 + Jp[1][2] * s1);
    }
    s_t[threadIdx.x * 3 + 0] = t0; s_t[threadIdx.x * 3 + 1] = t1; s_t[threadIdx.x * 3 + 2] = t2;
  }
  __syncthreads();
  double pred = 0, dx2 = 0, x2 = 0;
  if (own_pt) {      // a point-aligned block holds at most 256 points: one per thread
    const int q = threadIdx.x;
    const size_t p = myq;
    // operands are read here, not ahead of phase 1: keeping ~27 more values live across it costs a wave of occupancy
    T f[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) f[k] = pf[myq * PF + k];
    const double g0 = gp[myq * 3], g1 = gp[myq * 3 + 1], g2 = gp[myq * 3 + 2];
    const double dd0 = fmax_pos(D2p[myq * 3]), dd1 = fmax_pos(D2p[myq * 3 + 1]), dd2 = fmax_pos(D2p[myq * 3 + 2]);
    const double X0 = pts[myq * 3], X1 = pts[myq * 3 + 1], X2 = pts[myq * 3 + 2];
    const int a = pt_start[p] - o_lo, b = pt_start[p + 1] - o_lo;
    double t0 = 0, t1 = 0, t2 = 0;
    for (int k = a; k < b; ++k) { t0 += s_t[k * 3]; t1 += s_t[k * 3 + 1]; t2 += s_t[k * 3 + 2]; }
    double e0 = 0, e1 = 0, e2 = 0;
    if (f[9] != (T)0) {     // delta = -(V + lam D)^-1 (g + t) = -L^-T ( z + L^-1 t ),  z = L^-1 g from k_point_factor
      const double l0 = f[0], l1 = f[1], l2 = f[2], l3 = f[3], l4 = f[4], l5 = f[5];
      const double y0 = -((double)f[6] + l0 * t0);
      const double y1 = -((double)f[7] + l1 * t0 + l2 * t1);
      const double y2 = -((double)f[8] + l3 * t0 + l4 * t1 + l5 * t2);
      e0 = l0 * y0 + l1 * y1 + l3 * y2;     // L^-T y
      e1 = l2 * y1 + l4 * y2;
      e2 = l5 * y2;
    }
    const double n0 = X0 + e0, n1 = X1 + e1, n2 = X2 + e2;
    pts_new[p * 3] = n0; pts_new[p * 3 + 1] = n1; pts_new[p * 3 + 2] = n2;
    ptsT_new[p * 3] = (T)n0; ptsT_new[p * 3 + 1] = (T)n1; ptsT_new[p * 3 + 2] = (T)n2;
    s_xn[q * 3] = n0; s_xn[q * 3 + 1] = n1; s_xn[q * 3 + 2] = n2;
    const double lam_ = lam;
    pred = 0.5 * (e0 * (lam_ * dd0 * e0 - g0) + e1 * (lam_ * dd1 * e1 - g1) + e2 * (lam_ * dd2 * e2 - g2));
    dx2 = e0 * e0 + e1 * e1 + e2 * e2;
    x2 = pt_fixed(ps, p) ? 0.0 : X0 * X0 + X1 * X1 + X2 * X2;      // a fixed point is not part of x
  }
  __syncthreads();
  double sq = 0;
  if (my_p >= 0) {
    const int q = my_p - p_lo;
    T u, v;
    obs_project<T>(s_camn + my_c * CAMPRE, (T)s_xn[q * 3], (T)s_xn[q * 3 + 1], (T)s_xn[q * 3 + 2], u, v);
    const T r0 = my_w * (u - my_uv.x), r1 = my_w * (v - my_uv.y);
    sq = (ps.loss_delta > 0.f) ? (double)robust_cost<T>(ps.loss(), r0, r1) : (double)r0 * r0 + (double)r1 * r1;
  }
  const double c_new = block_sum(sq, s_scr);
  const double b_pred = block_sum(pred, s_scr);
  const double b_dx2 = block_sum(dx2, s_scr);
  const double b_x2 = block_sum(x2, s_scr);
  if (threadIdx.x == 0) {
    trial_part[blockIdx.x] = 0.5 * c_new;
    trial_part[nblk + blockIdx.x] = b_pred;
    trial_part[2 * nblk + blockIdx.x] = b_dx2;
    trial_part[3 * nblk + blockIdx.x] = b_x2;
  }
}

// one block of DECIDE_THREADS threads: fold the per-block partials into this rank's 8 scalars (same gather, same fold as the decision
// itself, so a one-rank communicator reproduces the plain loop bit for bit)
__global__ __launch_bounds__(DECIDE_THREADS) void k_trial_scalars(const double* __restrict__ trial_part, const double* __restrict__ gmax_part,
                                                                  int nblk, int n_gmax, const LMState* __restrict__ st, double* __restrict__ scal) {
  __shared__ double s_scr[5 * DECIDE_THREADS];
  if (st->status >= 0) return;
  DecidePartials dp;
  decide_gather(dp, nullptr, trial_part, gmax_part, nblk, n_gmax);
  double f5[5] = {0, 0, 0, 0, 0};
  decide_fold(dp, s_scr, f5);
  if (threadIdx.x == 0) {
    scal[0] = f5[0]; scal[1] = f5[1]; scal[2] = f5[2]; scal[3] = f5[3]; scal[4] = f5[4]; scal[5] = (double)st->chol_fail;
    scal[6] = 0; scal[7] = 0;
  }
}

// ------------------------------------------------------------------ K6 for dense visibility, one camera group
// Same arithmetic as k_backsub_trial, laid out like k_schur_fused: lane (q, c) = (point of a 16-point chunk, camera),
// so W_p^T delta_c is a DPP row sum instead of an LDS staging pass with two barriers, the workgroup is persistent
// (camera tables staged once, chunks strided over the grid) and each partial row holds a whole workgroup's share.
// LW = lanes per point: 16 (one DPP row: up to 16 cameras), 64 (a whole wave, two cameras per lane: 33 .. 128 cameras, round 4) or 32 (a wave half: 17 .. 32 cameras, the rigs of k_schur_fused_wide and, since round 4, 24 .. 32 --
// sums over the half, 8 points per chunk; 17 of 32 lanes work at 17 cameras, so it only draws level with k_backsub_trial there
// (17 x 50k: 166.8 us per iteration either way) and gains from 20 cameras on (20 x 50k 218 -> 213.8 us).  LW = 0 (17 .. 21 cameras):
// three points per wave, packed -- 17 x 50k 166.8 -> 162.5 us per iteration in fp32, 273.0 -> 268.8 in fp64, 20 x 50k 213.8 -> 209.6.
template <typename T, int LW = 16>
__global__ __launch_bounds__(PM_BLOCK) void k_backsub_dense(
    const ParamSets<T> ps, int C, const typename Vec2<T>::type* __restrict__ uv /* observation (p, c) at p*C + c */,
    const T* __restrict__ w, int N, const T* __restrict__ pf, const double* __restrict__ gp, const double* __restrict__ D2p,
    const double* __restrict__ delta_c, const LMState* __restrict__ st, double* __restrict__ trial_part, int nparts,
    const uint16_t* __restrict__ vis = nullptr /* sparse rigs: per (16-camera group, point) visibility mask [groups][N]; observation (p, c) then
                                                  sits at pt_start[group][p] + popcount(mask below bit c & 15), a lane without one idles through
                                                  the row sums (one group: the point's mask and first observation) */,
    const int32_t* __restrict__ pt_start = nullptr) {
  // LW = 0: PACKED -- three points per wave, lane l -> point l / C, camera l % C (17 .. 21 cameras: 51 .. 63 of the 64 lanes work
  // instead of 34 .. 42 with a point per half); the sums over a point's cameras = segment differences of a wave-wide prefix scan
  constexpr bool PACK = LW == 0;
  // LW = 64 (round 4): a point per WAVE and up to two cameras per lane (c, c + 64): 33 .. 128 cameras, i.e. BASELINE configs 4 and 5, which
  // used to take the point-aligned k_backsub_trial (LDS staging of W^T dc, two barriers per block)
  constexpr int CPL = LW == 64 ? 2 : 1;                                                                                 // cameras per lane
  constexpr int MAXC = LW == 16 ? GROUP_CAMS : LW == 64 ? 128 : 32, PPC = PACK ? 3 * (PM_BLOCK / 64) : PM_BLOCK / (PACK ? 1 : LW);       // cameras the tables hold, points per chunk
  __shared__ T s_cam01[2][MAXC * CAMPRE], s_dc[MAXC * NCP];
  __shared__ double s_scr[PM_BLOCK / 64];
  // both camera tables are needed whichever is current: they and the camera step are requested before the state record's round trip
  stage_campre(ps.campre[0], s_cam01[0], C);
  stage_campre(ps.campre[1], s_cam01[1], C);
  for (int i = threadIdx.x; i < C * NCP; i += PM_BLOCK) s_dc[i] = (T)delta_c[i];
  if (st->status >= 0) return;
  const int cur_ = ps_cur(ps, st);
  const T* s_cam = s_cam01[cur_];
  const T* s_camn = s_cam01[cur_ ^ 1];
  const double* __restrict__ pts = ps.pts[cur_];
  const T* __restrict__ ptsT = ps.ptsT[cur_];
  double* __restrict__ pts_new = ps.pts[cur_ ^ 1];
  T* __restrict__ ptsT_new = ps.ptsT[cur_ ^ 1];
  const bool free_cams = st->free_cams != 0;
  const double lam = st->lam;
  __syncthreads();
  const int lane_ = threadIdx.x & 63;
  const int hseg = PACK ? ((lane_ >= C) + (lane_ >= 2 * C) + (lane_ >= 3 * C)) : 0;
  const int q = PACK ? 3 * (int)(threadIdx.x >> 6) + min(hseg, 2) : (int)threadIdx.x / (PACK ? 1 : LW);
  const int c = PACK ? lane_ - hseg * C : (int)threadIdx.x % (PACK ? 1 : LW);
  const bool cam_ok = PACK ? hseg < 3 : c < C;
  const int cc = c & 15;
  auto lane_sum = [&](T v) -> T {
    if constexpr (LW == 16) return row16_sum(v);
    else if constexpr (LW == 32) return half32_sum(v);
    else if constexpr (LW == 64) return wave64_sum(v);
    else {
      const T sc = wave_scan(v);
      const T e0 = lane_bcast(sc, C - 1), e1 = lane_bcast(sc, 2 * C - 1), e2 = lane_bcast(sc, 3 * C - 1);
      return hseg == 0 ? e0 : hseg == 1 ? e1 - e0 : e2 - e1;
    }
  };
  T dc[NCP];
#pragma unroll
  for (int e = 0; e < NCP; ++e) dc[e] = cam_ok ? s_dc[c * NCP + e] : (T)0;
  const T* cp = s_cam + (cam_ok ? c : 0) * CAMPRE;
  const T* cpn = s_camn + (cam_ok ? c : 0) * CAMPRE;
  double sq = 0, pred = 0, dx2 = 0, x2 = 0;
  const int nch = (N + PPC - 1) / PPC;
  for (int ch = blockIdx.x; ch < nch; ch += gridDim.x) {
    const int p = ch * PPC + q;
    const bool pt_ok = p < N;
    const size_t pp = (size_t)(pt_ok ? p : 0);
    // (the lane's first camera c; with CPL = 2 its second one, c + 64, goes through the same steps with index 1)
    bool valid[CPL];
    typename Vec2<T>::type m[CPL];
    T ww[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int cj = c + 64 * j;
      const bool cam_j = j == 0 ? cam_ok : cj < C;
      const int gj = (cam_j ? cj : 0) >> 4;
      unsigned mask = 0xffffu;
      size_t o = pp * C + cj;
      if (vis) { mask = vis[(size_t)gj * N + pp]; o = (size_t)pt_start[(size_t)gj * N + pp] + __builtin_popcount(mask & ((1u << cc) - 1u)); }
      valid[j] = pt_ok && cam_j && ((mask >> cc) & 1u);
      m[j].x = 0; m[j].y = 0;
      ww[j] = (T)1;
      if (valid[j]) { m[j] = uv[o]; if (w) ww[j] = w[o]; }
    }
    // every lane of the row reads the point's data (same addresses: one transaction) and solves for its step
    // redundantly.  (Requesting the next chunk's operands one chunk ahead was measured: no gain, the kernel is
    // issue-bound at three workgroups per CU and the extra registers cost one of them.)
    T f[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) f[k] = pf[pp * PF + k];
    const double g0 = gp[pp * 3], g1 = gp[pp * 3 + 1], g2 = gp[pp * 3 + 2];
    const double dd0 = fmax_pos(D2p[pp * 3]), dd1 = fmax_pos(D2p[pp * 3 + 1]), dd2 = fmax_pos(D2p[pp * 3 + 2]);
    const double X0 = pts[pp * 3], X1 = pts[pp * 3 + 1], X2 = pts[pp * 3 + 2];
    T t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      if (free_cams && valid[j]) {
        // W^T dc of this observation = Jp^T (Jc dc): the point block and the directional derivative along the camera step
        T r[2], Jp[2][3], sj[2];
        if (j == 0) obs_jp_jvp<T>(cp, ptsT[pp * 3], ptsT[pp * 3 + 1], ptsT[pp * 3 + 2], m[j].x, m[j].y, ww[j], dc, r, Jp, sj);
        else {
          T dcj[NCP];
#pragma unroll
          for (int e = 0; e < NCP; ++e) dcj[e] = s_dc[(c + 64 * j) * NCP + e];
          obs_jp_jvp<T>(s_cam + (c + 64 * j) * CAMPRE, ptsT[pp * 3], ptsT[pp * 3 + 1], ptsT[pp * 3 + 2], m[j].x, m[j].y, ww[j], dcj, r, Jp, sj);
        }
        if (ps.loss_delta > 0.f) robust_apply_jvp<T>(ps.loss(), r, Jp, sj);
        t0 += Jp[0][0] * sj[0] + Jp[1][0] * sj[1];
        t1 += Jp[0][1] * sj[0] + Jp[1][1] * sj[1];
        t2 += Jp[0][2] * sj[0] + Jp[1][2] * sj[1];
      }
    }
    const double T0 = (double)lane_sum(t0), T1 = (double)lane_sum(t1), T2 = (double)lane_sum(t2);
    double e0 = 0, e1 = 0, e2 = 0;
    if (f[9] != (T)0) {     // delta = -(V + lam D)^-1 (g + t) = -L^-T ( z + L^-1 t )
      const double l0 = f[0], l1 = f[1], l2 = f[2], l3 = f[3], l4 = f[4], l5 = f[5];
      const double y0 = -((double)f[6] + l0 * T0);
      const double y1 = -((double)f[7] + l1 * T0 + l2 * T1);
      const double y2 = -((double)f[8] + l3 * T0 + l4 * T1 + l5 * T2);
      e0 = l0 * y0 + l1 * y1 + l3 * y2;
      e1 = l2 * y1 + l4 * y2;
      e2 = l5 * y2;
    }
    const double n0 = X0 + e0, n1 = X1 + e1, n2 = X2 + e2;
    if (pt_ok && cam_ok && c == 0) {
      pts_new[pp * 3] = n0; pts_new[pp * 3 + 1] = n1; pts_new[pp * 3 + 2] = n2;
      ptsT_new[pp * 3] = (T)n0; ptsT_new[pp * 3 + 1] = (T)n1; ptsT_new[pp * 3 + 2] = (T)n2;
      pred += 0.5 * (e0 * (lam * dd0 * e0 - g0) + e1 * (lam * dd1 * e1 - g1) + e2 * (lam * dd2 * e2 - g2));
      dx2 += e0 * e0 + e1 * e1 + e2 * e2;
      if (!pt_fixed(ps, pp)) x2 += X0 * X0 + X1 * X1 + X2 * X2;      // a fixed point is not part of x
    }
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      if (valid[j]) {
        T u, v;
        obs_project<T>(j == 0 ? cpn : s_camn + (c + 64 * j) * CAMPRE, (T)n0, (T)n1, (T)n2, u, v);
        const T r0 = ww[j] * (u - m[j].x), r1 = ww[j] * (v - m[j].y);
        sq += (ps.loss_delta > 0.f) ? (double)robust_cost<T>(ps.loss(), r0, r1) : (double)r0 * r0 + (double)r1 * r1;
      }
    }
  }
  const double c_new = block_sum(sq, s_scr);
  const double b_pred = block_sum(pred, s_scr);
  const double b_dx2 = block_sum(dx2, s_scr);
  const double b_x2 = block_sum(x2, s_scr);
  if (threadIdx.x == 0) {
    trial_part[blockIdx.x] = 0.5 * c_new;
    trial_part[nparts + blockIdx.x] = b_pred;
    trial_part[2 * nparts + blockIdx.x] = b_dx2;
    trial_part[3 * nparts + blockIdx.x] = b_x2;
  }
}

// ------------------------------------------------------------------ K3a with lane = (point, camera): 24 .. 128 cameras (round 4)
// k_linearize_points (lane = observation, the nine per-point sums through an LDS pass in doubles, one workgroup per point-aligned
// block) laid out like k_backsub_dense<T, 32 / 64>: a point per wave half (24 .. 32 cameras) or per wave with up to two cameras per
// lane (33 .. 128), the sums V_p (6) and g_p (3) as DPP / row-swap lane sums, persistent workgroups (camera table staged once) and
// one cost / max|g_p| partial per workgroup instead of one per 256 observations (k_decide folded 50 000 of them at 64 x 200k).
template <typename T, int LW>
__global__ __launch_bounds__(PM_BLOCK) void k_linearize_points_wave(
    const ParamSets<T> ps, const LMState* __restrict__ st, int C, const typename Vec2<T>::type* __restrict__ uv, const T* __restrict__ w, int N,
    const uint16_t* __restrict__ vis /* [groups][N] visibility masks, or NULL: dense (observation (p, c) at p C + c) */,
    const int32_t* __restrict__ pt_start /* [groups][N] first observation of the point in the group */,
    double* __restrict__ V, double* __restrict__ gp, double* __restrict__ D2p, double* __restrict__ cost_part, double* __restrict__ gmax_part,
    T* __restrict__ pf /* non-null (inside the LM loop, free cameras): the damped point factors of THIS trial are formed here as well, for the
                          points this workgroup linearises -- k_point_factor's work without its launch; after a rejected step (nothing to
                          re-linearise, but a new lambda) that is all the launch does */,
    const unsigned char* __restrict__ fixed) {
  static_assert(LW == 32 || LW == 64, "a point per wave half or per wave");
  constexpr int CPL = LW == 64 ? 2 : 1, MAXC = LW == 64 ? 128 : 32, PPC = PM_BLOCK / LW;
  __shared__ T s_cam[MAXC * CAMPRE];
  __shared__ double s_scr[PM_BLOCK / 64];
  if (st && st->status >= 0) return;
  const int nch = (N + PPC - 1) / PPC;
  if (st && !st->need_lin) {
    if (pf) {
      const double lam = st->lam;
      const int nown = (nch - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
      for (int i = threadIdx.x; i < nown * PPC; i += PM_BLOCK) {
        const int p = ((int)blockIdx.x + (i / PPC) * (int)gridDim.x) * PPC + i % PPC;
        if (p < N) point_factor_one<T>(V, gp, D2p, lam, (size_t)p, pf, fixed);
      }
    }
    return;
  }
  const int cur_ = ps_cur(ps, st);
  const T* __restrict__ ptsT = ps.ptsT[cur_];
  stage_campre(ps.campre[cur_], s_cam, C);
  __syncthreads();
  const int q = (int)threadIdx.x / LW, c = (int)threadIdx.x % LW, cc = c & 15;
  const bool cam_ok = c < C;
  auto lane_sum = [&](T v) -> T {
    if constexpr (LW == 32) return half32_sum(v);
    else return wave64_sum(v);
  };
  double sq = 0, gmax = 0;
  for (int ch = blockIdx.x; ch < nch; ch += gridDim.x) {
    const int p = ch * PPC + q;
    const bool pt_ok = p < N;
    const size_t pp = (size_t)(pt_ok ? p : 0);
    const T X0 = ptsT[pp * 3], X1 = ptsT[pp * 3 + 1], X2 = ptsT[pp * 3 + 2];
    T v[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) v[e] = (T)0;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int cj = c + 64 * j;
      const bool cam_j = j == 0 ? cam_ok : cj < C;
      const int gj = (cam_j ? cj : 0) >> 4;
      unsigned mask = 0xffffu;
      size_t o = pp * C + cj;
      if (vis) { mask = vis[(size_t)gj * N + pp]; o = (size_t)pt_start[(size_t)gj * N + pp] + __builtin_popcount(mask & ((1u << cc) - 1u)); }
      if (pt_ok && cam_j && ((mask >> cc) & 1u)) {
        const auto m = uv[o];
        const T ww = w ? w[o] : (T)1;
        T r[2], Jc[2][NCP], Jp[2][3];
        obs_resjac<T>(s_cam + cj * CAMPRE, X0, X1, X2, m.x, m.y, ww, r, Jc, Jp);
        if (ps.loss_delta > 0.f) sq += (double)robust_apply<T>(ps.loss(), r, Jc, Jp);
        else sq += (double)r[0] * r[0] + (double)r[1] * r[1];
        v[0] += Jp[0][0] * Jp[0][0] + Jp[1][0] * Jp[1][0];
        v[1] += Jp[0][0] * Jp[0][1] + Jp[1][0] * Jp[1][1];
        v[2] += Jp[0][0] * Jp[0][2] + Jp[1][0] * Jp[1][2];
        v[3] += Jp[0][1] * Jp[0][1] + Jp[1][1] * Jp[1][1];
        v[4] += Jp[0][1] * Jp[0][2] + Jp[1][1] * Jp[1][2];
        v[5] += Jp[0][2] * Jp[0][2] + Jp[1][2] * Jp[1][2];
        v[6] += Jp[0][0] * r[0] + Jp[1][0] * r[1];
        v[7] += Jp[0][1] * r[0] + Jp[1][1] * r[1];
        v[8] += Jp[0][2] * r[0] + Jp[1][2] * r[1];
      }
    }
    double S[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) S[e] = (double)lane_sum(v[e]);
    if (pt_ok && c == 0) {
#pragma unroll
      for (int e = 0; e < 6; ++e) V[pp * 6 + e] = S[e];
      D2p[pp * 3] = fmax(D2p[pp * 3], S[0]); D2p[pp * 3 + 1] = fmax(D2p[pp * 3 + 1], S[3]); D2p[pp * 3 + 2] = fmax(D2p[pp * 3 + 2], S[5]);
      gp[pp * 3] = S[6]; gp[pp * 3 + 1] = S[7]; gp[pp * 3 + 2] = S[8];
      if (!pt_fixed(ps, pp)) gmax = fmax(gmax, fmax(fabs(S[6]), fmax(fabs(S[7]), fabs(S[8]))));      // a fixed point is not an unknown
    }
  }
  const double cs = block_sum(sq, s_scr);
  const double gm = block_max(gmax, s_scr);
  if (threadIdx.x == 0) { cost_part[blockIdx.x] = 0.5 * cs; gmax_part[blockIdx.x] = gm; }
  if (pf && st) {                                 // (the barriers of the block sums stand between the V / g_p / D2p stores above and these reads)
    const double lam = st->lam;
    const int nown = (nch - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    for (int i = threadIdx.x; i < nown * PPC; i += PM_BLOCK) {
      const int p = ((int)blockIdx.x + (i / PPC) * (int)gridDim.x) * PPC + i % PPC;
      if (p < N) point_factor_one<T>(V, gp, D2p, lam, (size_t)p, pf, fixed);
    }
  }
}

// ------------------------------------------------------------------ start of a solve
// What lm_begin has to reset on the device, in one launch instead of three memsets and two copies (each a separate enqueue of a few
// microseconds on the host): the monotone column scalings and the camera step are cleared, the trial camera buffers start as copies of
// the current ones (points-only mode never writes them).
template <typename T>
__global__ void k_lm_reset(double* __restrict__ D2p, size_t n_d2p, double* __restrict__ D2c, double* __restrict__ delta_c, int n,
                           const double* __restrict__ cams_cur, double* __restrict__ cams_trial,
                           const T* __restrict__ campre_cur, T* __restrict__ campre_trial, int n_campre) {
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = i0; i < n_d2p; i += stride) D2p[i] = 0.0;
  for (size_t i = i0; i < (size_t)n; i += stride) { D2c[i] = 0.0; delta_c[i] = 0.0; cams_trial[i] = cams_cur[i]; }
  for (size_t i = i0; i < (size_t)n_campre; i += stride) campre_trial[i] = campre_cur[i];
}

// ------------------------------------------------------------------ accept / reject / terminate
// The decision itself is decide_core (sba_kernels.hpp); this kernel runs it on the record in place.  k_schur_fused_bf3 runs
// the same function in its prologue instead (every workgroup redundantly, one of them publishing the result), which takes
// this launch out of the iteration.
// scal_all: n_ranks x 8 scalars (already gathered), or -- single rank -- nullptr, in which case the block folds the
// per-block partials itself (what k_trial_scalars does for the multi-rank path) and no separate launch is needed.
template <typename T>
__global__ __launch_bounds__(DECIDE_THREADS) void k_decide(LMState* st,
                                                const double* __restrict__ scal_all, int n_ranks,
                                                const double* __restrict__ trial_part, const double* __restrict__ gmax_part,
                                                int nblk, int n_gmax /* entries of gmax_part (one per linearisation workgroup) */,
                                                LMLogRow* __restrict__ log, int log_cap) {
  // The decision logic is a chain of dependent reads and writes of the state record; done against global memory every
  // link costs a memory round trip.  The record is staged in LDS by the whole block (its load overlaps the partial
  // sums), thread 0 works on that copy, writes it back with fire-and-forget stores and publishes `status` last.
  __shared__ LMState s_st;
  __shared__ LMLogRow s_row;
  __shared__ double s_scr[5 * DECIDE_THREADS];
  constexpr int NWORD = sizeof(LMState) / 4;
  DecidePartials dp;
  decide_gather(dp, scal_all, trial_part, gmax_part, nblk, n_gmax);
  if ((int)threadIdx.x < NWORD) reinterpret_cast<int*>(&s_st)[threadIdx.x] = reinterpret_cast<const int*>(st)[threadIdx.x];
  __syncthreads();
  if (s_st.status >= 0) return;
  const bool have_row = decide_core(&s_st, dp, scal_all, n_ranks, &s_row, log_cap, s_scr);      // result valid in thread 0
  if (threadIdx.x != 0) return;
  if (have_row && log) log[s_st.iter - 1] = s_row;
  {
    constexpr int SW = offsetof(LMState, status) / 4;
    const int* src = reinterpret_cast<const int*>(&s_st);
    int* dst = reinterpret_cast<int*>(st);
#pragma unroll
    for (int wd = 0; wd < NWORD; ++wd)
      if (wd != SW) dst[wd] = src[wd];
  }
  // no fence: every reader of the record (the next kernels of the stream, the host's copy in lm_poll) is ordered after
  // this kernel by the stream itself
  st->status = s_st.status;
}

}  // namespace SBA_NS
