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

// ------------------------------------------------------------------ the 64 x 64 diagonal block, by one workgroup of 512 threads
// Dg: its 10 lower sub-blocks (packed, cb_off) -> the factor's sub-blocks below the diagonal and T = Linv^T of the diagonal sub-blocks;
// Mi: the 6 strictly-lower sub-blocks of the inverse of the factor.  *s_fail is set when a pivot is not positive.
__device__ __forceinline__ double* chol_big_mi_blk(double* Mi, int a, int b) { return Mi + (a * (a - 1) / 2 + b) * CBS; }
struct CholBigNoHook { __device__ __forceinline__ void operator()() const {} };
// The 64 pivots are the chain (wave 0: chol16_wave four times); everything else hangs off it.  The inverse's sub-blocks are formed
// as soon as their inputs are final, by waves the panel / downdate phases leave idle, so that after the last pivot only the
// bottom row Minv(3, 0..2) is left -- three independent blocks on three waves.  11 barriers (15 with the inverse as a
// separate pass).  `hook` is called by waves 4..7 while wave 0 works on the last diagonal sub-block (nothing else runs then):
// k_chol_big_dag's walker fetches its next two tiles there.
template <typename Hook = CholBigNoHook>
__device__ __forceinline__ void chol_big_factor64(double* __restrict__ Dg, double* __restrict__ Mi, int* __restrict__ s_fail,
                                                  Hook&& hook = Hook()) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  // Minv(a,b) = -T_a^T * sum_{k=b}^{a-1} L(a,k) Minv(k,b)   (the diagonal sub-blocks hold T = Linv^T; Minv(b,b) = T_b^T)
  auto inv_blk = [&](int a, int b) {
    Mfma<double>::acc_t p = {0, 0, 0, 0};
    p = mm16<false, true>(Dg + cb_off(a, b), Dg + cb_off(b, b), p, 1.0);          // L(a,b) * T_b^T
    for (int k = b + 1; k < a; ++k) p = mm16<false, false>(Dg + cb_off(a, k), chol_big_mi_blk(Mi, k, b), p, 1.0);
    double* dst = chol_big_mi_blk(Mi, a, b);
    store16(dst, p);
    __builtin_amdgcn_wave_barrier();
    Mfma<double>::acc_t m = {0, 0, 0, 0};
    m = mm16<true, false>(Dg + cb_off(a, a), dst, m, -1.0);                       // -(T_a)^T * P
    __builtin_amdgcn_wave_barrier();
    store16(dst, m);
  };
  auto pivots = [&](int jb) { if (!chol16_wave(Dg + cb_off(jb, jb))) { if (lane == 0) *s_fail = 1; } };
  auto upd = [&](int jb, int a, int b) { chol_update_tile(Dg + cb_off(jb + 1 + a, jb + 1 + b), Dg + cb_off(jb + 1 + a, jb), Dg + cb_off(jb + 1 + b, jb)); };
  // ---- sub-block column 0
  if (wid == 0) pivots(0);
  __syncthreads();
  if (wid < 3) chol_panel_block(Dg + cb_off(1 + wid, 0), Dg + cb_off(0, 0));
  __syncthreads();
  if (wid < 6) {
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= wid) ++a;
    upd(0, a, wid - a * (a + 1) / 2);
  }
  __syncthreads();
  // ---- sub-block column 1
  if (wid == 0) pivots(1);
  __syncthreads();
  if (wid < 2) chol_panel_block(Dg + cb_off(2 + wid, 1), Dg + cb_off(1, 1));
  else if (wid == 2) inv_blk(1, 0);
  __syncthreads();
  if (wid < 3) {
    const int a = (wid == 0) ? 0 : 1;
    upd(1, a, wid - a * (a + 1) / 2);
  }
  __syncthreads();
  // ---- sub-block column 2
  if (wid == 0) pivots(2);
  __syncthreads();
  if (wid == 0) chol_panel_block(Dg + cb_off(3, 2), Dg + cb_off(2, 2));
  else if (wid == 1) inv_blk(2, 1);
  __syncthreads();
  if (wid == 0) upd(2, 0, 0);
  else if (wid == 1) inv_blk(2, 0);
  __syncthreads();
  // ---- sub-block column 3
  if (wid == 0) pivots(3);
  else if (wid >= 4) hook();
  __syncthreads();
  if (wid < 3) inv_blk(3, wid);
  __syncthreads();
}
// dense copies for the consumers and the back substitution: Minv (lower, row-major) and the factor's strictly-lower sub-blocks
template <bool MINV = true, bool LD = true>
__device__ __forceinline__ void chol_big_publish_factor(const double* __restrict__ Dg, double* __restrict__ Mi,
                                                        double* __restrict__ Mg, double* __restrict__ Lg) {
  for (int e = threadIdx.x; e < BB * BB; e += CHOLBIG_THREADS) {
    const int i = e >> 6, k = e & 63, si = i >> 4, sk = k >> 4;
    double mv = 0, lv = 0;
    if (si == sk) mv = Dg[cb_off(si, si) + (k & 15) * CLD + (i & 15)];           // Linv[i][k] = T[k][i]
    else if (si > sk) { mv = chol_big_mi_blk(Mi, si, sk)[(i & 15) * CLD + (k & 15)]; lv = Dg[cb_off(si, sk) + (i & 15) * CLD + (k & 15)]; }
    if (MINV) Mg[e] = mv;
    if (LD) Lg[e] = lv;
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
  chol_big_factor64(Dg, Mi, &s_fail);
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0 && s_fail) atomicOr(info, 1);
    chol_big_publish_factor(Dg, Mi, Minv_ws + (size_t)j * BB * BB, Ld_ws + (size_t)j * BB * BB);
  }
  auto mi_blk = [&](int a, int b) { return Mi + (a * (a - 1) / 2 + b) * CBS; };
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

// ------------------------------------------------------------------ the factorisation in ONE launch: the tile DAG (round 4)
// One launch per block column pays, per column, a launch boundary, the cold loads behind it and a factorisation of the diagonal
// block repeated by every workgroup: 19.4 us per column at n = 704, of which the chain that cannot be avoided -- the 64 pivots of
// the diagonal block, its inverse, one panel block and one downdate -- is about 13.  Here the whole factorisation is one launch:
//   workgroup 0, the WALKER, goes down the diagonal: it factors block (c,c) in LDS, inverts the factor, publishes both as one
//     image (Mimg_c: the 10 + 6 sub-blocks exactly as they lie in LDS), then takes W(c+1,c) and the partial W(c+1,c+1), forms
//     L(c+1,c) and the downdate itself -- the serial chain of the factorisation never leaves its LDS;
//   workgroup 1 + t OWNS tile t = (r, c), c <= r, of the lower block triangle (numbered column by column).  It keeps the tile in
//     its accumulators and applies the columns j as they become available (wait W(r,j), W(c,j) final -> wait Mimg_j ->
//     L(r,j) = W(r,j) Minv_j^T, L(c,j) likewise -> tile -= L(r,j) L(c,j)^T): j < c below the diagonal, after which it publishes
//     the final W(r,c) and, once Mimg_c is there, L(r,c) into the upper block (c, r) for the back substitution;  j < c - 1 on the
//     diagonal, after which it hands the partial tile to the walker.
// Hand-overs are epoch flags (one per diagonal block, one per tile) set by a release after a barrier; every wait is bounded
// (CHOLBIG_WAIT_TICKS, then info |= 2: a rejected LM step) and a timed-out workgroup raises an abort flag the others watch.
// Progress does not need the whole launch to be resident (a card shared with other processes), only in-order dispatch: the
// walker is dispatched first, and what it needs to publish Mimg_j are tiles (j, j-1) and (j, j), which are numbered below every
// tile that waits for Mimg_j.  Up to CHOLDAG_MAX_NBR block rows the launch is resident as a whole (254 workgroups, one per CU).
constexpr int CHOLDAG_MAX_NBR = 22;
constexpr int CHOLDAG_IMG = 16 * CBS;                      // doubles per published diagonal block: Dg (10 sub-blocks) + Mi (6)
constexpr long long CHOLBIG_WAIT_TICKS = 200000000LL;      // 2 s of the 100 MHz clock
__host__ __device__ inline int choldag_nflags(int nbr) { return nbr + nbr * nbr + 1; }
// entries of the image: Minv[i][k] (lower) and the factor's strictly-lower sub-blocks L[i][k]
__device__ __forceinline__ double choldag_img_minv(const double* __restrict__ img, int i, int k) {
  const int si = i >> 4, sk = k >> 4;
  if (si == sk) return img[cb_off(si, si) + (k & 15) * CLD + (i & 15)];                              // Linv[i][k] = T[k][i]
  if (si > sk) return img[10 * CBS + (si * (si - 1) / 2 + sk) * CBS + (i & 15) * CLD + (k & 15)];
  return 0.0;
}
__device__ __forceinline__ double choldag_img_l(const double* __restrict__ img, int i, int k) {
  const int si = i >> 4, sk = k >> 4;
  return (si > sk) ? img[cb_off(si, sk) + (i & 15) * CLD + (k & 15)] : 0.0;
}

// wave 0 polls up to two flags (lanes 0 and 1) until both carry the epoch; everybody leaves through a barrier and an acquire fence
__device__ __forceinline__ void choldag_wait(const unsigned* f0, const unsigned* f1, unsigned epoch, unsigned* abortf, int* s_late) {
  if (threadIdx.x < 64) {
    const unsigned* f = (threadIdx.x == 0) ? f0 : f1;
    const long long t0 = wall_clock64();
    int it = 0;
    bool late = false;
    while (true) {
      const bool ready = threadIdx.x >= 2 || __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
      if (__all(ready)) break;
      if ((++it & 31) == 0) {
        const bool out = (wall_clock64() - t0 > CHOLBIG_WAIT_TICKS) ||
                         __hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
        if (__any(out)) { late = true; break; }
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (late && threadIdx.x == 0) { *s_late = 1; __hip_atomic_store(abortf, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// (the barrier orders every wave's stores before thread 0's release, which is cumulative: one L2 write-back instead of eight)
__device__ __forceinline__ void choldag_publish(unsigned* flag, unsigned epoch) {
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// one output block of a panel:  Lp(ri, CI) = sum_{k < CI} Ap(ri,k) Minv(CI,k)^T + Ap(ri,CI) T_CI, operands fetched ahead of the MFMAs.
// F is the image of the diagonal block: T = Linv^T on the diagonal of Dg, the inverse's lower sub-blocks in Mi
template <int CI>
__device__ __forceinline__ Mfma<double>::acc_t choldag_panel_out(const double* __restrict__ Aprow, const double* __restrict__ F) {
  const int lane = threadIdx.x & 63, rc = lane & 15, kq = lane >> 4;
  const double* Mi = F + 10 * CBS;
  double a[(CI + 1) * 4], b[(CI + 1) * 4];
#pragma unroll
  for (int k = 0; k <= CI; ++k) {
    const double* Bk = (k < CI) ? Mi + (CI * (CI - 1) / 2 + k) * CBS : F + cb_off(CI, CI);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      a[4 * k + ks] = Aprow[k * CBS + rc * CLD + 4 * ks + kq];
      b[4 * k + ks] = (k < CI) ? Bk[rc * CLD + 4 * ks + kq] : Bk[(4 * ks + kq) * CLD + rc];   // Minv(CI,k) read transposed, T_CI as it is
    }
  }
  Mfma<double>::acc_t acc = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < (CI + 1) * 4; ++i) acc = Mfma<double>::mma(a[i], b[i], acc);
  return acc;
}
// a whole panel, in place: the calling wave forms all four blocks of sub-block row ri in registers; the caller puts a barrier between
// this and choldag_panel_store (other waves may still be reading the row)
struct CholdagRow { Mfma<double>::acc_t o[BSUB]; };
__device__ __forceinline__ CholdagRow choldag_panel_row(const double* __restrict__ Ap, const double* __restrict__ F, int ri) {
  CholdagRow r;
  const double* row = Ap + ri * BSUB * CBS;
  r.o[3] = choldag_panel_out<3>(row, F);
  r.o[2] = choldag_panel_out<2>(row, F);
  r.o[1] = choldag_panel_out<1>(row, F);
  r.o[0] = choldag_panel_out<0>(row, F);
  return r;
}
__device__ __forceinline__ void choldag_panel_store(double* __restrict__ Ap, int ri, const CholdagRow& r) {
#pragma unroll
  for (int ci = 0; ci < BSUB; ++ci) store16(Ap + (ri * BSUB + ci) * CBS, r.o[ci]);
}
// the same with the four blocks of a row shared by two waves (half 0: blocks 3 and 0, half 1: blocks 2 and 1 -- five products each)
struct CholdagHalfRow { Mfma<double>::acc_t o[2]; };
__device__ __forceinline__ CholdagHalfRow choldag_panel_half(const double* __restrict__ Ap, const double* __restrict__ F, int ri, int half) {
  CholdagHalfRow r;
  const double* row = Ap + ri * BSUB * CBS;
  if (half == 0) { r.o[0] = choldag_panel_out<3>(row, F); r.o[1] = choldag_panel_out<0>(row, F); }
  else { r.o[0] = choldag_panel_out<2>(row, F); r.o[1] = choldag_panel_out<1>(row, F); }
  return r;
}
__device__ __forceinline__ void choldag_panel_half_store(double* __restrict__ Ap, int ri, int half, const CholdagHalfRow& r) {
  store16(Ap + (ri * BSUB + (half == 0 ? 3 : 2)) * CBS, r.o[0]);
  store16(Ap + (ri * BSUB + (half == 0 ? 0 : 1)) * CBS, r.o[1]);
}
// sum_k A(ri,k) B(ci,k)^T over the four sub-blocks of a row, operands fetched ahead of the 16 MFMAs
__device__ __forceinline__ Mfma<double>::acc_t choldag_abt(const double* __restrict__ Arow, const double* __restrict__ Brow) {
  const int lane = threadIdx.x & 63, rc = lane & 15, kq = lane >> 4;
  double a[16], b[16];
#pragma unroll
  for (int k = 0; k < BSUB; ++k)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      a[4 * k + ks] = Arow[k * CBS + rc * CLD + 4 * ks + kq];
      b[4 * k + ks] = Brow[k * CBS + rc * CLD + 4 * ks + kq];
    }
  Mfma<double>::acc_t p = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; ++i) p = Mfma<double>::mma(a[i], b[i], p);
  return p;
}
__device__ __forceinline__ void choldag_copy_img(double* __restrict__ dst, const double* __restrict__ src) {
  for (int e = threadIdx.x; e < CHOLDAG_IMG / 2; e += CHOLBIG_THREADS)
    reinterpret_cast<double2*>(dst)[e] = reinterpret_cast<const double2*>(src)[e];
}
// 64 x 64 block in LDS (4 x 4 sub-blocks) -> W block (br, bc)
__device__ __forceinline__ void choldag_store_block(double* __restrict__ W, int npad, int br, int bc, const double* __restrict__ A) {
  for (int e = threadIdx.x; e < BB * BB; e += CHOLBIG_THREADS) {
    const int i = e >> 6, k = e & 63;
    W[(size_t)(br * BB + i) * npad + bc * BB + k] = A[((i >> 4) * BSUB + (k >> 4)) * CBS + (i & 15) * CLD + (k & 15)];
  }
}

// grid = 1 + nbr (nbr + 1) / 2; 512 threads; dynamic LDS = 48 sub-blocks (the same as k_chol_big_step)
__global__ __launch_bounds__(CHOLBIG_THREADS) void k_chol_big_dag(double* __restrict__ W, int npad, double* __restrict__ Mimg_ws,
                                                                  unsigned* __restrict__ flags, unsigned epoch,
                                                                  int* __restrict__ info, const LMState* __restrict__ st,
                                                                  long long* __restrict__ dbg) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (st->status >= 0) return;                           // (the same record on every workgroup: nobody is left waiting)
  double* F = reinterpret_cast<double*>(smem);           // image of a diagonal block: Dg (10 sub-blocks) + Mi (6)
  double* Ar = F + 16 * CBS;                             // W(r,j) -> L(r,j)
  double* Ac = Ar + 16 * CBS;                            // W(c,j) -> L(c,j);  the walker: the next diagonal tile
  __shared__ int s_fail, s_late, s_pref[4];
  const int nbr = npad / BB;
  unsigned* flagM = flags;
  unsigned* flagW = flags + nbr;
  unsigned* abortf = flags + nbr + nbr * nbr;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) { s_fail = 0; s_late = 0; }
  if (blockIdx.x == 0) {
    // ================================================================ the walker
    // SBA_CHOL_DEBUG: stamps of the 100 MHz clock at the links of the chain
    auto stamp = [&](int c, int k) { if (dbg && threadIdx.x == 0) dbg[c * 8 + k] = wall_clock64(); };
    stamp(0, 0);
    double* Mi = F + 10 * CBS;
    load_block64<true>(W, npad, 0, 0, F);
    __syncthreads();
    for (int c = 0; c < nbr; ++c) {
      stamp(c, 4);
      // waves 4..7, while the last 16 pivots run: if both tiles of the next column are there already, fetch them now
      const bool more = c + 1 < nbr;
      const unsigned* fa = flagW + (c + 1) * nbr + c;
      const unsigned* fb = flagW + (c + 1) * nbr + c + 1;
      if (threadIdx.x < 4) s_pref[threadIdx.x] = 0;
      chol_big_factor64(F, Mi, &s_fail, [&]() {
        if (!more) return;
        bool ready = true;
        if (c >= 1 && lane < 2) ready = __hip_atomic_load(lane == 0 ? fa : fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
        if (!__all(ready)) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int tid = threadIdx.x - 256;
        double2 va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = tid + u * 256, row = e >> 5, col = (e & 31) * 2;
          va[u] = *reinterpret_cast<const double2*>(W + (size_t)((c + 1) * BB + row) * npad + c * BB + col);
          vb[u] = *reinterpret_cast<const double2*>(W + (size_t)((c + 1) * BB + row) * npad + (c + 1) * BB + col);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = tid + u * 256, row = e >> 5, col = (e & 31) * 2;
          const int o = ((row >> 4) * BSUB + (col >> 4)) * CBS + (row & 15) * CLD + (col & 15);
          Ar[o] = va[u].x; Ar[o + 1] = va[u].y;
          Ac[o] = vb[u].x; Ac[o + 1] = vb[u].y;
        }
        if (lane == 0) s_pref[wid - 4] = 1;
      });
      stamp(c, 5);
      choldag_copy_img(Mimg_ws + (size_t)c * CHOLDAG_IMG, F);
      stamp(c, 6);
      choldag_publish(flagM + c, epoch);
      stamp(c, 7);
      if (!more) break;
      const bool fetched = s_pref[0] && s_pref[1] && s_pref[2] && s_pref[3];     // (read after the barrier inside choldag_publish)
      if (!fetched) {
        if (c >= 1) {                                     // (column 0 and tile (1,1) are final as k_chol_big_prepare wrote them)
          choldag_wait(fa, fb, epoch, abortf, &s_late);
          if (s_late) break;
        }
        stamp(c + 1, 1);
        load_block64<false>(W, npad, (c + 1) * BB, c * BB, Ar);
        load_block64<false>(W, npad, (c + 1) * BB, (c + 1) * BB, Ac);
        __syncthreads();
      } else stamp(c + 1, 1);
      stamp(c + 1, 2);
      {
        const CholdagHalfRow pr = choldag_panel_half(Ar, F, wid & 3, wid >> 2);
        __syncthreads();
        choldag_panel_half_store(Ar, wid & 3, wid >> 2, pr);
      }
      __syncthreads();
      stamp(c + 1, 3);
      // the next diagonal block, lower sub-blocks only, straight into the layout the factorisation works on (10 sub-tiles, 8 waves)
      {
        int ri = 0;
        while ((ri + 1) * (ri + 2) / 2 <= wid) ++ri;
        const int ci = wid - ri * (ri + 1) / 2;
        const Mfma<double>::acc_t p0 = choldag_abt(Ar + ri * BSUB * CBS, Ar + ci * BSUB * CBS);
        const int t1 = wid + 8, ri1 = 3, ci1 = t1 - 6;     // sub-tiles 8, 9 = (3,2), (3,3): waves 0 and 1
        Mfma<double>::acc_t p1 = {0, 0, 0, 0};
        if (wid < 2) p1 = choldag_abt(Ar + ri1 * BSUB * CBS, Ar + ci1 * BSUB * CBS);
        Mfma<double>::acc_t a;
        const double* Nt = Ac + (ri * BSUB + ci) * CBS;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) a[rg] = Nt[((lane >> 4) + 4 * rg) * CLD + (lane & 15)] - p0[rg];
        store16(F + cb_off(ri, ci), a);
        if (wid < 2) {
          const double* Nt1 = Ac + (ri1 * BSUB + ci1) * CBS;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) a[rg] = Nt1[((lane >> 4) + 4 * rg) * CLD + (lane & 15)] - p1[rg];
          store16(F + cb_off(ri1, ci1), a);
        }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0 && s_fail) atomicOr(info, 1);
    if (s_late && threadIdx.x == 0) atomicOr(info, 2);
    return;
  }
  // ================================================================ a tile
  int c = 0, t = blockIdx.x - 1;
  while (t >= nbr - c) { t -= nbr - c; ++c; }
  const int r = c + t;
  const int ncol = (r == c) ? c - 1 : c;                  // columns this workgroup applies (the walker applies the last one on the diagonal)
  if (r == c && c < 2) return;
  // the own tile: 16 sub-tiles, two per wave, in the accumulators from here to the end
  Mfma<double>::acc_t acc[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int tt = wid + 8 * h, ri = tt >> 2, ci = tt & 3;
    const double* Cg = W + (size_t)(r * BB + ri * CB) * npad + c * BB + ci * CB;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) acc[h][rg] = Cg[(size_t)((lane >> 4) + 4 * rg) * npad + (lane & 15)];
  }
  __syncthreads();
  for (int j = 0; j < ncol; ++j) {
    if (j > 0) {                                          // (column 0 is final as k_chol_big_prepare wrote it)
      choldag_wait(flagW + r * nbr + j, flagW + c * nbr + j, epoch, abortf, &s_late);
      if (s_late) break;
    }
    load_block64<false>(W, npad, r * BB, j * BB, Ar);
    if (c != r) load_block64<false>(W, npad, c * BB, j * BB, Ac);
    choldag_wait(flagM + j, flagM + j, epoch, abortf, &s_late);
    if (s_late) break;
    choldag_copy_img(F, Mimg_ws + (size_t)j * CHOLDAG_IMG);
    __syncthreads();
    if (c != r) {
      double* Ap = (wid < BSUB) ? Ar : Ac;
      const CholdagRow pr = choldag_panel_row(Ap, F, wid & 3);
      __syncthreads();
      choldag_panel_store(Ap, wid & 3, pr);
    } else {
      const CholdagHalfRow pr = choldag_panel_half(Ar, F, wid & 3, wid >> 2);
      __syncthreads();
      choldag_panel_half_store(Ar, wid & 3, wid >> 2, pr);
    }
    __syncthreads();
    const double* Lr = Ar;
    const double* Lc = (c != r) ? Ac : Ar;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int tt = wid + 8 * h, ri = tt >> 2, ci = tt & 3;
      const Mfma<double>::acc_t p = choldag_abt(Lr + ri * BSUB * CBS, Lc + ci * BSUB * CBS);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) acc[h][rg] -= p[rg];
    }
    __syncthreads();
  }
  if (s_late) { if (threadIdx.x == 0) atomicOr(info, 2); return; }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int tt = wid + 8 * h, ri = tt >> 2, ci = tt & 3;
    store16(Ar + (ri * BSUB + ci) * CBS, acc[h]);
  }
  __syncthreads();
  if (r == c) {                                           // the partial diagonal tile goes to the walker
    choldag_store_block(W, npad, r, c, Ar);
    choldag_publish(flagW + r * nbr + c, epoch);
    return;
  }
  // ---- below the diagonal: the final W(r,c) for the tiles to the right, then L(r,c) for the back substitution
  if (c > 0) {
    choldag_store_block(W, npad, r, c, Ar);
    choldag_publish(flagW + r * nbr + c, epoch);
  }
  choldag_wait(flagM + c, flagM + c, epoch, abortf, &s_late);
  if (s_late) { if (threadIdx.x == 0) atomicOr(info, 2); return; }
  choldag_copy_img(F, Mimg_ws + (size_t)c * CHOLDAG_IMG);
  __syncthreads();
  {
    const CholdagHalfRow pr = choldag_panel_half(Ar, F, wid & 3, wid >> 2);
    __syncthreads();
    choldag_panel_half_store(Ar, wid & 3, wid >> 2, pr);
  }
  __syncthreads();
  choldag_store_block(W, npad, c, r, Ar);
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
__global__ __launch_bounds__(256) void k_chol_big_back_all(const double* __restrict__ W, int npad, int n, const double* __restrict__ Ld_ws,
                                                           const double* __restrict__ Minv_ws, double* __restrict__ xv /* [npad] */,
                                                           unsigned* __restrict__ flags /* [block rows] */, unsigned epoch,
                                                           double* __restrict__ sol, int* __restrict__ info, const LMState* __restrict__ st,
                                                           const double* __restrict__ Mimg_ws /* k_chol_big_dag's images, or null: the dense copies */) {
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
    if (k < n) yk = (t < Rb) ? W[(size_t)(t * BB + Rl) * npad + Rb * BB + threadIdx.x]
                             : (Mimg_ws ? choldag_img_l(Mimg_ws + (size_t)Rb * CHOLDAG_IMG, Rl, threadIdx.x) : Ld_ws[(size_t)Rb * BB * BB + Rl * BB + threadIdx.x]);
  }
  const double* Mg = Minv_ws + (size_t)t * BB * BB;
  double mreg[BB * BB / 256];
#pragma unroll
  for (int u = 0; u < BB * BB / 256; ++u) {
    const int e = threadIdx.x + 256 * u;
    mreg[u] = Mimg_ws ? choldag_img_minv(Mimg_ws + (size_t)t * CHOLDAG_IMG, e >> 6, e & 63) : Mg[e];
  }
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
