// sba_chol_big.hpp -- Cholesky + solve of a LARGE reduced camera system (n > 512: 47 .. 128 cameras) on many CUs.
//
//   A = S + lam*diag(D2c)   (n = 11*C up to 1408, f64),  A = L L^T,  delta_c = A^-1 rhs
//
// Right-looking, 64-wide block columns, ONE launch per block column, no library call and no spin-waits:
//   k_chol_big_prepare   copies the damped system into a padded workspace W (npad x npad, npad = 64*nbr) and appends the
//                        right-hand side as one more ROW (index R = 16*ceil(n/16)): the forward substitution y = L^-1 rhs
//                        then simply falls out as row R of the factor.  Padding rows are the identity.
//   k_chol_big_step(j)   one workgroup per trailing 64x64 tile (r, c), j < c <= r.  Every workgroup
//                          1. factors the diagonal block W(j,j) itself, in LDS (16x16 sub-blocks, the building blocks of
//                             sba_chol_blocked.hpp) and inverts the 64x64 factor -- redundant work, but it runs in parallel
//                             and spares a launch boundary plus a hand-off per block column;
//                          2. forms its two panel blocks L(r,j) = W(r,j) Linv^T, L(c,j) = W(c,j) Linv^T (f64 MFMA);
//                          3. downdates its tile W(r,c) -= L(r,j) L(c,j)^T.
//                        The workgroups of tile column c = j+1 also publish L(r,j), into the UPPER block triangle of W
//                        (block (j,r)): the lower block (r,j) is still being read by the other workgroups of the launch.
//                        Workgroup 0 publishes the inverse of the diagonal factor (Minv) and the factor's sub-blocks (Ld).
//   k_chol_big_back_init gathers y (row R of L), then
//   k_chol_big_back(b)   b = last .. 0:  x_b = Minv_b^T y_b (every workgroup, redundantly), y_t -= L(b,t)^T x_b (workgroup t < b).
// Launches per solve: 2 + nbr + ceil(n/64), each a few microseconds; the chain of n pivots (~275 cycles each) inside
// the diagonal factorisations is the floor.  The LM-specific parts stay in k_chol_epilogue (sba_lm_kernels.hpp).
#pragma once
#include "sba_chol_blocked.hpp"

namespace SBA_NS {

constexpr int BB = 64;                      // block edge of the big factorisation
constexpr int BSUB = BB / CB;               // 4 sub-blocks of 16 per edge
constexpr int CHOLBIG_THREADS = 512;
constexpr double CHOLBIG_RHS_DIAG = 1e300;  // diagonal entry of the appended rhs row: keeps the augmented matrix PD

__host__ __device__ inline int cholbig_rhs_row(int n) { return ((n + CB - 1) / CB) * CB; }
__host__ __device__ inline int cholbig_npad(int n) { return ((cholbig_rhs_row(n) + 1 + BB - 1) / BB) * BB; }

// one 16x16x16 product on LDS sub-blocks (17-double rows): acc += sign * op(A) * op(B)
//   AT: A is read transposed (A[k][row]),  BT: B is read transposed (B[col][k])
template <bool AT, bool BT>
__device__ __forceinline__ Mfma<double>::acc_t mm16(const double* __restrict__ A, const double* __restrict__ B,
                                                    Mfma<double>::acc_t acc, double sign) {
  const int lane = threadIdx.x & 63;
  const int rc = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int k = 4 * ks + kq;
    const double a = AT ? A[k * CLD + rc] : A[rc * CLD + k];
    const double b = BT ? B[rc * CLD + k] : B[k * CLD + rc];
    acc = Mfma<double>::mma(sign * a, b, acc);
  }
  return acc;
}
__device__ __forceinline__ void store16(double* __restrict__ blk, const Mfma<double>::acc_t& acc) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) blk[((lane >> 4) + 4 * rg) * CLD + (lane & 15)] = acc[rg];
}

// 64x64 block of W (rows r0.., columns c0..) -> 4x4 sub-blocks in LDS ((sr*4 + sc) * CBS); lower_only: sub-blocks with
// sc > sr are skipped and the destination uses the packed lower numbering cb_off(sr, sc)
template <bool LOWER_ONLY>
__device__ __forceinline__ void load_block64(const double* __restrict__ W, int npad, int r0, int c0, double* __restrict__ dst) {
  double2 v[BB * BB / 2 / CHOLBIG_THREADS];
#pragma unroll
  for (int u = 0; u < BB * BB / 2 / CHOLBIG_THREADS; ++u) {
    const int e = threadIdx.x + u * CHOLBIG_THREADS;
    const int row = e >> 5, col = (e & 31) * 2;
    v[u] = *reinterpret_cast<const double2*>(W + (size_t)(r0 + row) * npad + c0 + col);
  }
#pragma unroll
  for (int u = 0; u < BB * BB / 2 / CHOLBIG_THREADS; ++u) {
    const int e = threadIdx.x + u * CHOLBIG_THREADS;
    const int row = e >> 5, col = (e & 31) * 2;
    const int sr = row >> 4, sc = col >> 4;
    if (LOWER_ONLY && sc > sr) continue;
    double* d = dst + (LOWER_ONLY ? cb_off(sr, sc) : (sr * BSUB + sc) * CBS) + (row & 15) * CLD + (col & 15);
    d[0] = v[u].x; d[1] = v[u].y;
  }
}

// ------------------------------------------------------------------ prepare: W <- [A rhs^T; rhs BIG], padded with the identity
// grid = nbr*(nbr+1)/2 (lower block triangle, diagonal blocks whole), 256 threads
__global__ __launch_bounds__(256) void k_chol_big_prepare(const double* __restrict__ E, int n, LMState* __restrict__ st,
                                                          double* __restrict__ D2c, double* __restrict__ W, int npad,
                                                          int* __restrict__ info) {
  if (st->status >= 0) return;
  int br = 0, t = blockIdx.x;
  while ((br + 1) * (br + 2) / 2 <= t) ++br;
  const int bc = t - br * (br + 1) / 2;
  const int R = cholbig_rhs_row(n);
  const double* rhs = E + (size_t)n * n;
  const double* dU = rhs + n;
  const double lam = st->lam;
  const bool fresh = st->fresh != 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { st->cost = E[(size_t)n * n + 3 * n]; *info = 0; }
  for (int e = threadIdx.x; e < BB * BB; e += 256) {
    const int i = br * BB + (e >> 6), j = bc * BB + (e & 63);
    double v;
    if (i < n && j < n) {
      v = E[(size_t)i * n + j];
      if (i == j) {      // camera scaling: monotone max of the squared column norms (x_scale='jac', scipy trf.py:424,545)
        double d = D2c[i];
        if (fresh) { d = fmax(d, dU[i]); D2c[i] = d; }
        v += lam * fmax_pos(d);
      }
    } else if (i == R && j == R) v = CHOLBIG_RHS_DIAG;
    else if (i == R && j < n) v = rhs[j];
    else if (j == R && i < n) v = rhs[i];
    else v = (i == j) ? 1.0 : 0.0;
    W[(size_t)i * npad + j] = v;
  }
}

// ------------------------------------------------------------------ one block column.  grid = max(1, q(q+1)/2), q = nbr-1-j
__global__ __launch_bounds__(CHOLBIG_THREADS) void k_chol_big_step(double* __restrict__ W, int npad, int j,
                                                                   double* __restrict__ Minv_ws, double* __restrict__ Ld_ws,
                                                                   int* __restrict__ info, const LMState* __restrict__ st) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (st->status >= 0) return;
  double* Dg = reinterpret_cast<double*>(smem);          // 10 lower sub-blocks of the diagonal block -> L / Linv^T
  double* Mi = Dg + 10 * CBS;                            // 6 strictly-lower sub-blocks of the inverse
  double* Ar = Mi + 6 * CBS;                             // 16 sub-blocks: W(r,j) -> L(r,j)
  double* Ac = Ar + 16 * CBS;                            // 16 sub-blocks: W(c,j) -> L(c,j)
  __shared__ int s_fail;
  const int nbr = npad / BB, q = nbr - 1 - j;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int r = j, c = j;
  if (q > 0) {
    int rr = 0, t = blockIdx.x;
    while ((rr + 1) * (rr + 2) / 2 <= t) ++rr;
    r = j + 1 + rr; c = j + 1 + (t - rr * (rr + 1) / 2);
  }
  if (threadIdx.x == 0) s_fail = 0;
  load_block64<true>(W, npad, j * BB, j * BB, Dg);
  if (q > 0) {
    load_block64<false>(W, npad, r * BB, j * BB, Ar);
    if (c != r) load_block64<false>(W, npad, c * BB, j * BB, Ac);
  }
  __syncthreads();
  // ---- 1. factor the 64x64 diagonal block: 4 sub-block columns
  for (int jb = 0; jb < BSUB; ++jb) {
    if (wid == 0) { if (!chol16_wave(Dg + cb_off(jb, jb))) { if (lane == 0) s_fail = 1; } }
    __syncthreads();
    if (wid < BSUB - 1 - jb) chol_panel_block(Dg + cb_off(jb + 1 + wid, jb), Dg + cb_off(jb, jb));
    __syncthreads();
    {
      const int qq = BSUB - 1 - jb, nt = qq * (qq + 1) / 2;
      if (wid < nt) {
        int a = 0;
        while ((a + 1) * (a + 2) / 2 <= wid) ++a;
        const int b = wid - a * (a + 1) / 2;
        chol_update_tile(Dg + cb_off(jb + 1 + a, jb + 1 + b), Dg + cb_off(jb + 1 + a, jb), Dg + cb_off(jb + 1 + b, jb));
      }
    }
    __syncthreads();
  }
  // ---- inverse of the factor, sub-block diagonal by sub-block diagonal:  Minv(a,b) = -Linv_a * sum_{k=b}^{a-1} L(a,k) Minv(k,b)
  //      (the diagonal sub-blocks hold T = Linv^T; Minv(b,b) = T_b^T)
  auto mi_blk = [&](int a, int b) { return Mi + (a * (a - 1) / 2 + b) * CBS; };
  for (int d = 1; d < BSUB; ++d) {
    if (wid < BSUB - d) {
      const int b = wid, a = wid + d;
      Mfma<double>::acc_t p = {0, 0, 0, 0};
      p = mm16<false, true>(Dg + cb_off(a, b), Dg + cb_off(b, b), p, 1.0);          // L(a,b) * T_b^T
      for (int k = b + 1; k < a; ++k) p = mm16<false, false>(Dg + cb_off(a, k), mi_blk(k, b), p, 1.0);
      double* dst = mi_blk(a, b);
      store16(dst, p);
      __builtin_amdgcn_wave_barrier();
      Mfma<double>::acc_t m = {0, 0, 0, 0};
      m = mm16<true, false>(Dg + cb_off(a, a), dst, m, -1.0);                       // -(T_a)^T * P
      __builtin_amdgcn_wave_barrier();
      store16(dst, m);
    }
    __syncthreads();
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0 && s_fail) atomicOr(info, 1);
    // dense copies for the back substitution: Minv (lower, row-major) and the factor's strictly-lower sub-blocks
    double* Mg = Minv_ws + (size_t)j * BB * BB;
    double* Lg = Ld_ws + (size_t)j * BB * BB;
    for (int e = threadIdx.x; e < BB * BB; e += CHOLBIG_THREADS) {
      const int i = e >> 6, k = e & 63, si = i >> 4, sk = k >> 4;
      double mv = 0, lv = 0;
      if (si == sk) mv = Dg[cb_off(si, si) + (k & 15) * CLD + (i & 15)];           // Linv[i][k] = T[k][i]
      else if (si > sk) { mv = mi_blk(si, sk)[(i & 15) * CLD + (k & 15)]; lv = Dg[cb_off(si, sk) + (i & 15) * CLD + (k & 15)]; }
      Mg[e] = mv; Lg[e] = lv;
    }
  }
  if (q == 0) return;
  // ---- 2. panel blocks in place:  Lp(ri, ci) = sum_{k <= ci} Ap(ri,k) Minv(ci,k)^T, ci = 3 .. 0 (one wave per sub-block row)
  {
    double* Ap = (wid < BSUB) ? Ar : Ac;
    const int ri = wid & (BSUB - 1);
    if (wid < BSUB || c != r) {
      for (int ci = BSUB - 1; ci >= 0; --ci) {
        Mfma<double>::acc_t acc = {0, 0, 0, 0};
        for (int k = 0; k < ci; ++k) acc = mm16<false, true>(Ap + (ri * BSUB + k) * CBS, mi_blk(ci, k), acc, 1.0);
        acc = mm16<false, false>(Ap + (ri * BSUB + ci) * CBS, Dg + cb_off(ci, ci), acc, 1.0);
        __builtin_amdgcn_wave_barrier();
        store16(Ap + (ri * BSUB + ci) * CBS, acc);
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __syncthreads();
  const double* Lr = Ar;
  const double* Lc = (c != r) ? Ac : Ar;
  // ---- 3. W(r,c) -= L(r,j) L(c,j)^T : 16 sub-tiles, two per wave, accumulators straight from / to global memory
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int t = wid + 8 * h, ri = t >> 2, ci = t & 3;
    double* Cg = W + (size_t)(r * BB + ri * CB) * npad + c * BB + ci * CB;
    Mfma<double>::acc_t acc;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) acc[rg] = Cg[(size_t)((lane >> 4) + 4 * rg) * npad + (lane & 15)];
#pragma unroll
    for (int k = 0; k < BSUB; ++k) acc = mm16<false, true>(Lr + (ri * BSUB + k) * CBS, Lc + (ci * BSUB + k) * CBS, acc, -1.0);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) Cg[(size_t)((lane >> 4) + 4 * rg) * npad + (lane & 15)] = acc[rg];
  }
  // ---- publish L(r,j) into the upper block (j, r)
  if (c == j + 1) {
    for (int e = threadIdx.x; e < BB * BB; e += CHOLBIG_THREADS) {
      const int i = e >> 6, k = e & 63;
      W[(size_t)(j * BB + i) * npad + r * BB + k] = Lr[((i >> 4) * BSUB + (k >> 4)) * CBS + (i & 15) * CLD + (k & 15)];
    }
  }
}

// ------------------------------------------------------------------ back substitution
// y = row R of L, columns < n (zero beyond)
__global__ void k_chol_big_back_init(const double* __restrict__ W, int npad, int n, const double* __restrict__ Ld_ws,
                                     double* __restrict__ yv, const LMState* __restrict__ st) {
  if (st->status >= 0) return;
  const int R = cholbig_rhs_row(n), Rb = R / BB, Rl = R % BB;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < npad; k += gridDim.x * blockDim.x) {
    double v = 0;
    if (k < n) {
      const int b = k / BB, kl = k % BB;
      v = (b < Rb) ? W[(size_t)(b * BB + Rl) * npad + Rb * BB + kl] : Ld_ws[(size_t)Rb * BB * BB + Rl * BB + kl];
    }
    yv[k] = v;
  }
}

// block b: grid = b + 1 workgroups of 256.  Workgroup t: x_b = Minv_b^T y_b ; t == b writes it out, t < b: y_t -= L(b,t)^T x_b
__global__ __launch_bounds__(256) void k_chol_big_back(const double* __restrict__ W, int npad, int b, int n,
                                                       const double* __restrict__ Minv_ws, double* __restrict__ yv,
                                                       double* __restrict__ sol, const LMState* __restrict__ st) {
  __shared__ double s_y[BB], s_x[BB], s_p[4][BB];
  if (st->status >= 0) return;
  const int i = threadIdx.x & 63, part = threadIdx.x >> 6, t = blockIdx.x;
  if (threadIdx.x < BB) s_y[threadIdx.x] = yv[b * BB + threadIdx.x];
  __syncthreads();
  const double* Mg = Minv_ws + (size_t)b * BB * BB;
  double s = 0;
  for (int k = part; k < BB; k += 4) s += Mg[k * BB + i] * s_y[k];          // Minv is lower: entries k < i are stored zeros
  s_p[part][i] = s;
  __syncthreads();
  if (part == 0) {
    const double x = (s_p[0][i] + s_p[1][i]) + (s_p[2][i] + s_p[3][i]);
    s_x[i] = x;
    if (t == b && b * BB + i < n) sol[b * BB + i] = x;
  }
  __syncthreads();
  if (t == b) return;
  const double* Lb = W + (size_t)(t * BB) * npad + b * BB;                     // L(b,t)[lr][lc] at Lb[lr*npad + lc]... see below
  // L(b,t) was published into block (t, b): element (lr, lc) of L(b,t) sits at W[(t*64 + lr)*npad + b*64 + lc]
  double u = 0;
  for (int k = part; k < BB; k += 4) u += Lb[(size_t)k * npad + i] * s_x[k];
  __syncthreads();
  s_p[part][i] = u;
  __syncthreads();
  if (part == 0) yv[t * BB + i] -= (s_p[0][i] + s_p[1][i]) + (s_p[2][i] + s_p[3][i]);
}

// ------------------------------------------------------------------ back substitution in ONE launch (round 4)
// The per-block launches above cost ~10 us each for a 64 x 64 matrix-vector product (launch + three dependent round trips to memory):
// 11 of them at n = 704, 26 at n = 1664.  Here block row t is ONE workgroup for the whole substitution,
//     y_t -= L(b,t)^T x_b   for b = last .. t + 1, as the x_b arrive;      x_t = Minv_t^T y_t,   published with a flag,
// so the chain through the blocks is a flag hand-over between resident workgroups (at most 26 of them) instead of a launch
// boundary: the L(b,t) block a workgroup needs next is already in its registers when x_b arrives.  The waits are bounded
// (CHOLBIG_WAIT_TICKS of the 100 MHz clock, then the solve is flagged as failed: a rejected LM step) and every workgroup of
// the launch is resident at once -- the rule of sba_ipc.hpp.  Flags carry the launch's epoch, so they are never reset.
constexpr long long CHOLBIG_WAIT_TICKS = 200000000LL;      // 2 s
__global__ __launch_bounds__(256) void k_chol_big_back_all(const double* __restrict__ W, int npad, int n, const double* __restrict__ Ld_ws,
                                                           const double* __restrict__ Minv_ws, double* __restrict__ xv /* [npad] */,
                                                           unsigned* __restrict__ flags /* [block rows] */, unsigned epoch,
                                                           double* __restrict__ sol, int* __restrict__ info, const LMState* __restrict__ st) {
  __shared__ double s_y[BB], s_x[BB], s_p[4][BB], s_M[BB * BB];
  __shared__ int s_late;
  if (st->status >= 0) return;                      // (the same record on every workgroup: nobody is left waiting)
  const int nbx = (n + BB - 1) / BB;
  const int t = blockIdx.x, i = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int R = cholbig_rhs_row(n), Rb = R / BB, Rl = R % BB;
  // y_t = row R of the factor (columns of block t), Minv_t, and the first L block: all requested before anything is waited for
  double yk = 0;
  if (threadIdx.x < BB) {
    const int k = t * BB + threadIdx.x;
    if (k < n) yk = (t < Rb) ? W[(size_t)(t * BB + Rl) * npad + Rb * BB + threadIdx.x] : Ld_ws[(size_t)Rb * BB * BB + Rl * BB + threadIdx.x];
  }
  const double* Mg = Minv_ws + (size_t)t * BB * BB;
  double mreg[BB * BB / 256];
#pragma unroll
  for (int u = 0; u < BB * BB / 256; ++u) mreg[u] = Mg[threadIdx.x + 256 * u];
  double lreg[BB / 4];                               // L(b,t)[k][i], k = part + 4 u
  auto fetch_L = [&](int b) {
    const double* Lb = W + (size_t)(t * BB) * npad + b * BB;
#pragma unroll
    for (int u = 0; u < BB / 4; ++u) lreg[u] = Lb[(size_t)(part + 4 * u) * npad + i];
  };
  if (nbx - 1 > t) fetch_L(nbx - 1);
  if (threadIdx.x == 0) s_late = 0;
  if (threadIdx.x < BB) s_y[threadIdx.x] = yk;
#pragma unroll
  for (int u = 0; u < BB * BB / 256; ++u) s_M[threadIdx.x + 256 * u] = mreg[u];
  __syncthreads();
  for (int b = nbx - 1; b > t; --b) {
    if (threadIdx.x < BB) {                          // wave 0 waits for x_b and brings it in
      const long long t0 = wall_clock64();
      bool late = false;
      while (__hip_atomic_load(flags + b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        if (wall_clock64() - t0 > CHOLBIG_WAIT_TICKS) { late = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (late) s_late = 1;
      s_x[threadIdx.x] = late ? 0.0 : __builtin_nontemporal_load(xv + b * BB + threadIdx.x);
    }
    __syncthreads();
    if (s_late) break;
    double u_ = 0;
#pragma unroll
    for (int u = 0; u < BB / 4; ++u) u_ += lreg[u] * s_x[part + 4 * u];
    if (b - 1 > t) fetch_L(b - 1);                   // the next block travels while this one is folded
    s_p[part][i] = u_;
    __syncthreads();
    if (part == 0) s_y[i] -= (s_p[0][i] + s_p[1][i]) + (s_p[2][i] + s_p[3][i]);
    __syncthreads();
  }
  if (s_late) { if (threadIdx.x == 0) atomicOr(info, 2); return; }     // (this workgroup's flag stays down: the rows above time out too)
  // x_t = Minv_t^T y_t   (Minv is lower: entries k < i are stored zeros)
  double s = 0;
  for (int k = part; k < BB; k += 4) s += s_M[k * BB + i] * s_y[k];
  s_p[part][i] = s;
  __syncthreads();
  if (part == 0) {
    const double x = (s_p[0][i] + s_p[1][i]) + (s_p[2][i] + s_p[3][i]);
    __builtin_nontemporal_store(x, xv + t * BB + i);
    if (t * BB + i < n) sol[t * BB + i] = x;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_store(flags + t, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace SBA_NS
