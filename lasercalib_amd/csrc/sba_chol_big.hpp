// sba_chol_big.hpp -- Cholesky + solve of a LARGE reduced camera system (n > 256: 24 .. 128 cameras) on many CUs.
//
//   A = S + lam*diag(D2c)   (n = 11*C or 13*C, up to 1664),  A = L L^T,  delta_c = A^-1 rhs
//
// Right-looking, 64-wide block columns, no library call.  The right-hand side is appended as one more ROW (index R =
// 16*ceil(n/16), diagonal entry BIG): the forward substitution y = L^-1 rhs then simply falls out as row R of the factor.
// Padding rows are the identity.  Two forms of the factorisation, one back substitution:
//   k_chol_big_dag<S>    (round 4, the default) ONE launch: a walker workgroup factors the diagonal blocks, one workgroup per tile
//                        of the lower block triangle applies the columns behind epoch flags; the system is formed from the exchange
//                        buffer inside the launch; S = float for the fp32 engine, with the f64 instance launched behind it for
//                        systems the f32 one refuses.  Details at the kernel.
//   k_chol_big_prepare + k_chol_big_step(j)   (rounds 1-3; SBA_CHOL_BIG=launches, ranks that share a card): the damped system
//                        copied into a padded workspace W, then ONE launch per block column, no waits between workgroups: one
//                        workgroup per trailing tile (r, c), j < c <= r, each of which
//                          1. factors the diagonal block W(j,j) itself, in LDS, and inverts the factor -- redundant work, but it
//                             runs in parallel and spares a launch boundary plus a hand-off per block column;
//                          2. forms its two panel blocks L(r,j) = W(r,j) Linv^T, L(c,j) = W(c,j) Linv^T (f64 MFMA);
//                          3. downdates its tile W(r,c) -= L(r,j) L(c,j)^T.
//                        The workgroups of tile column c = j+1 also publish L(r,j), into the UPPER block triangle of W (block
//                        (j,r)): the lower block (r,j) is still being read by the other workgroups of the launch.  Workgroup 0
//                        publishes the inverse of the diagonal factor (Minv) and the factor's sub-blocks (Ld).
//   k_chol_big_back_all<T>   b = last .. 0:  x_b = Minv_b^T y_b, y_t -= L(b,t)^T x_b, block row = workgroup, in one launch, and the LM
//                        epilogue (trial cameras, predicted reduction, failure flag) by the block row that finishes last
//                        (k_chol_big_back_init / k_chol_big_back: one launch per block, SBA_CHOL_BIG_BACK=launches).
// Both forms share the 64 x 64 building block chol_big_factor64 (16 x 16 sub-blocks, the pivot chains of sba_chol_blocked.hpp).
#pragma once
#include "sba_chol_blocked.hpp"

namespace SBA_NS {

constexpr int BB = 64;                      // block edge of the big factorisation
constexpr int BSUB = BB / CB;               // 4 sub-blocks of 16 per edge
constexpr int CHOLBIG_THREADS = 512;
constexpr int CHOLBIG_LDS_BLOCKS = 54;       // k_chol_big_step / k_chol_big_dag: diagonal block + inverse (16), two tiles (32), chol_big_factor64's scratch (6)
constexpr double CHOLBIG_RHS_DIAG = 1e300;  // diagonal entry of the appended rhs row: keeps the augmented matrix PD
constexpr float CHOLBIG_RHS_DIAG_F32 = 1e30f;   // ... on f32 lanes (k_chol_big_dag<float>)

__host__ __device__ inline int cholbig_rhs_row(int n) { return ((n + CB - 1) / CB) * CB; }
__host__ __device__ inline int cholbig_npad(int n) { return ((cholbig_rhs_row(n) + 1 + BB - 1) / BB) * BB; }

// Everything below works on 16 x 16 sub-blocks in LDS in either scalar type: S = double (17-double rows, v_mfma_f64_16x16x4: 64
// cycles on gfx950) or S = float (20-float rows, v_mfma_f32_16x16x4: 32 cycles; the fp32 engine's k_chol_big_dag).  CholLay<S>
// (sba_chol_blocked.hpp) carries the row stride LD, the block size BS and the packed lower-triangle numbering off(r, c).

// one 16x16x16 product on LDS sub-blocks: acc += sign * op(A) * op(B)
//   AT: A is read transposed (A[k][row]),  BT: B is read transposed (B[col][k])
template <bool AT, bool BT, typename S = double>
__device__ __forceinline__ typename Mfma<S>::acc_t mm16(const S* __restrict__ A, const S* __restrict__ B, typename Mfma<S>::acc_t acc,
                                                        S sign = (S)1) {
  using L = CholLay<S>;
  const int lane = threadIdx.x & 63;
  const int rc = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int k = 4 * ks + kq;
    const S a = AT ? A[k * L::LD + rc] : A[rc * L::LD + k];
    const S b = BT ? B[rc * L::LD + k] : B[k * L::LD + rc];
    acc = Mfma<S>::mma(sign * a, b, acc);
  }
  return acc;
}
template <typename S = double>
__device__ __forceinline__ void store16(S* __restrict__ blk, const typename Mfma<S>::acc_t& acc) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) blk[Mfma<S>::row_of(lane, rg) * CholLay<S>::LD + (lane & 15)] = acc[rg];
}
template <typename S = double>
__device__ __forceinline__ typename Mfma<S>::acc_t load16(const S* __restrict__ blk) {
  const int lane = threadIdx.x & 63;
  typename Mfma<S>::acc_t acc;
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) acc[rg] = blk[Mfma<S>::row_of(lane, rg) * CholLay<S>::LD + (lane & 15)];
  return acc;
}
// D -= Pa Pb^T for one 16x16 tile by the calling wave (see chol_update_tile)
template <typename S>
__device__ __forceinline__ void chol_update_tile_t(S* __restrict__ Dt, const S* __restrict__ Pa_blk, const S* __restrict__ Pb_blk) {
  typename Mfma<S>::acc_t prod = {0, 0, 0, 0};
  const typename Mfma<S>::acc_t acc = load16<S>(Dt);
  prod = mm16<false, true, S>(Pa_blk, Pb_blk, prod);
  store16<S>(Dt, acc - prod);
  __builtin_amdgcn_wave_barrier();
}

// 64x64 block of W (rows r0.., columns c0..) -> 4x4 sub-blocks in LDS ((sr*4 + sc) * BS); lower_only: sub-blocks with
// sc > sr are skipped and the destination uses the packed lower numbering off(sr, sc)
template <bool LOWER_ONLY, typename S = double>
__device__ __forceinline__ void load_block64(const S* __restrict__ W, int npad, int r0, int c0, S* __restrict__ dst) {
  using L = CholLay<S>;
  using V2 = typename Vec2<S>::type;
  V2 v[BB * BB / 2 / CHOLBIG_THREADS];
#pragma unroll
  for (int u = 0; u < BB * BB / 2 / CHOLBIG_THREADS; ++u) {
    const int e = threadIdx.x + u * CHOLBIG_THREADS;
    const int row = e >> 5, col = (e & 31) * 2;
    v[u] = *reinterpret_cast<const V2*>(W + (size_t)(r0 + row) * npad + c0 + col);
  }
#pragma unroll
  for (int u = 0; u < BB * BB / 2 / CHOLBIG_THREADS; ++u) {
    const int e = threadIdx.x + u * CHOLBIG_THREADS;
    const int row = e >> 5, col = (e & 31) * 2;
    const int sr = row >> 4, sc = col >> 4;
    if (LOWER_ONLY && sc > sr) continue;
    S* d = dst + (LOWER_ONLY ? L::off(sr, sc) : (sr * BSUB + sc) * L::BS) + (row & 15) * L::LD + (col & 15);
    d[0] = v[u].x; d[1] = v[u].y;
  }
}

// ------------------------------------------------------------------ the 64 x 64 diagonal block, by one workgroup of 512 threads
// Dg: its 10 lower sub-blocks (packed, off) -> the factor's sub-blocks below the diagonal and T = Linv^T of the diagonal sub-blocks;
// Mi: the 6 strictly-lower sub-blocks of the inverse of the factor.  *s_fail is set when a pivot is not positive (S = float: or has
// sunk below tau * a0: chol16_wave_t).
template <typename S>
__device__ __forceinline__ S* chol_big_mi_blk(S* Mi, int a, int b) { return Mi + (a * (a - 1) / 2 + b) * CholLay<S>::BS; }
struct CholBigNoHook { __device__ __forceinline__ void operator()(int) const {} };
// (A T)^T for one sub-block as MFMA accumulators -- exactly the operand layout of the products that follow (chol_panel_update_diag_t)
template <typename S>
__device__ __forceinline__ typename Mfma<S>::acc_t chol_big_panel_T(const S* __restrict__ Ablk, const S* __restrict__ LinvT) {
  using L = CholLay<S>;
  const int lane = threadIdx.x & 63;
  const S* Pa = Ablk + (lane & 15) * L::LD + (lane >> 4);
  const S* Pl = LinvT + (lane >> 4) * L::LD + (lane & 15);
  typename Mfma<S>::acc_t pt = {0, 0, 0, 0};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) pt = Mfma<S>::mma(Pl[4 * ks * L::LD], Pa[4 * ks], pt);
  return pt;
}
template <typename S>
__device__ __forceinline__ void chol_big_store_T(S* __restrict__ blk, const typename Mfma<S>::acc_t& pt) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) blk[(lane & 15) * CholLay<S>::LD + Mfma<S>::row_of(lane, rg)] = pt[rg];
}
template <typename S>
__device__ __forceinline__ typename Mfma<S>::acc_t chol_big_abT(const typename Mfma<S>::acc_t& pa, const typename Mfma<S>::acc_t& pb) {
  typename Mfma<S>::acc_t pp = {0, 0, 0, 0};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) pp = Mfma<S>::mma(pa[ks], pb[ks], pp);
  return pp;
}
// The 64 pivots are the chain: wave 0 runs chol16_wave four times and, between two of them, only forms the next sub-diagonal
// block L(jb+1,jb) and downdates the next diagonal sub-block (accumulators of the one product are the operands of the other: no
// LDS round trip) -- ONE barrier per sub-block column.  The other waves downdate the rest of the trailing sub-blocks behind it,
// each forming the two panel blocks it needs itself (nothing to wait for but T_jb), and build the inverse's sub-blocks as their
// inputs become final: after the last pivot only the bottom row Minv(3, 0..2) is left, three independent blocks on three waves.
// The factor's sub-blocks are collected in Lo (6 sub-blocks of scratch: the blocks of A they come from stay readable for the
// other waves) and copied into Dg at the end.  5 barriers (15 in round 3's version).
// Waves 4..7 have nothing to do from sub-block column 1 on: they call hook(1), hook(2), hook(3) in the three phases that follow
// (wave 0: third diagonal sub-block, last one, bottom row of the inverse) -- k_chol_big_dag's walker requests its next two tiles
// in one phase and puts them into LDS in the next.  The barriers wait for LDS only (lds_barrier): requests stay in flight across them.
template <typename S = double, typename Hook = CholBigNoHook>
__device__ __forceinline__ void chol_big_factor64(S* __restrict__ Dg, S* __restrict__ Mi, S* __restrict__ Lo, int* __restrict__ s_fail,
                                                  const S* __restrict__ a0 = nullptr, S tau = (S)0, Hook&& hook = Hook()) {
  using L = CholLay<S>;
  using acc_t = typename Mfma<S>::acc_t;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  auto lo = [&](int a, int b) { return Lo + (a * (a - 1) / 2 + b) * L::BS; };
  // Minv(a,b) = -T_a^T * sum_{k=b}^{a-1} L(a,k) Minv(k,b)   (the diagonal sub-blocks hold T = Linv^T; Minv(b,b) = T_b^T)
  auto inv_blk = [&](int a, int b) {
    acc_t p = {0, 0, 0, 0};
    p = mm16<false, true, S>(lo(a, b), Dg + L::off(b, b), p);                     // L(a,b) * T_b^T
    for (int k = b + 1; k < a; ++k) p = mm16<false, false, S>(lo(a, k), chol_big_mi_blk<S>(Mi, k, b), p);
    S* dst = chol_big_mi_blk<S>(Mi, a, b);
    store16<S>(dst, p);
    __builtin_amdgcn_wave_barrier();
    acc_t m = {0, 0, 0, 0};
    m = mm16<true, false, S>(Dg + L::off(a, a), dst, m, (S)-1);                   // -(T_a)^T * P
    __builtin_amdgcn_wave_barrier();
    store16<S>(dst, m);
  };
  auto pivots = [&](int jb) {
    const bool ok = chol16_wave_t<S, sizeof(S) == 8, L>(Dg + L::off(jb, jb), a0 ? a0 + jb * CB : nullptr, tau);
    if (!ok && lane == 0) *s_fail = 1;
  };
  // wave 0 between two diagonal sub-blocks: L(jb+1,jb) -> Lo, A(jb+1,jb+1) -= L L^T
  auto next_diag = [&](int jb) {
    const acc_t pt = chol_big_panel_T<S>(Dg + L::off(jb + 1, jb), Dg + L::off(jb, jb));
    const acc_t d = load16<S>(Dg + L::off(jb + 1, jb + 1)) - chol_big_abT<S>(pt, pt);
    chol_big_store_T<S>(lo(jb + 1, jb), pt);
    store16<S>(Dg + L::off(jb + 1, jb + 1), d);
    __builtin_amdgcn_wave_barrier();
  };
  // another wave: A(a,b) -= L(a,jb) L(b,jb)^T with both panel blocks formed here; keep: L(a,jb) -> Lo
  auto tile = [&](int jb, int a, int b, bool keep) {
    const acc_t pa = chol_big_panel_T<S>(Dg + L::off(a, jb), Dg + L::off(jb, jb));
    const acc_t pb = (a == b) ? pa : chol_big_panel_T<S>(Dg + L::off(b, jb), Dg + L::off(jb, jb));
    const acc_t d = load16<S>(Dg + L::off(a, b)) - chol_big_abT<S>(pa, pb);
    store16<S>(Dg + L::off(a, b), d);
    if (keep) chol_big_store_T<S>(lo(a, jb), pa);
  };
  if (wid == 0) pivots(0);
  lds_barrier();
  // ---- behind T_0
  if (wid == 0) { next_diag(0); pivots(1); }
  else if (wid == 1) tile(0, 2, 1, true);
  else if (wid == 2) tile(0, 2, 2, false);
  else if (wid == 3) tile(0, 3, 1, true);
  else if (wid == 4) tile(0, 3, 2, false);
  else if (wid == 5) tile(0, 3, 3, false);
  lds_barrier();
  // ---- behind T_1
  if (wid == 0) { next_diag(1); pivots(2); }
  else if (wid == 1) tile(1, 3, 2, true);
  else if (wid == 2) tile(1, 3, 3, false);
  else if (wid == 3) inv_blk(1, 0);
  else hook(1);
  lds_barrier();
  // ---- behind T_2
  if (wid == 0) { next_diag(2); pivots(3); }
  else if (wid == 1) inv_blk(2, 1);
  else if (wid == 2) inv_blk(2, 0);
  else if (wid >= 4) hook(2);
  lds_barrier();
  // ---- behind T_3: the inverse's bottom row; the factor's sub-blocks go to their places
  if (wid < 3) inv_blk(3, wid);
  else if (wid >= 4) hook(3);
  else {
    for (int q = 0; q < 6; ++q) {
      int a = 1;
      while (a * (a + 1) / 2 <= q) ++a;
      const int b = q - a * (a - 1) / 2;
      const S* src = Lo + q * L::BS;
      S* dst = Dg + L::off(a, b);
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int e = lane + 64 * u; dst[(e >> 4) * L::LD + (e & 15)] = src[(e >> 4) * L::LD + (e & 15)]; }
    }
  }
  lds_barrier();
}
// dense copies for the consumers and the back substitution: Minv (lower, row-major) and the factor's strictly-lower sub-blocks
template <bool MINV = true, bool LD = true>
__device__ __forceinline__ void chol_big_publish_factor(const double* __restrict__ Dg, double* __restrict__ Mi,
                                                        double* __restrict__ Mg, double* __restrict__ Lg) {
  for (int e = threadIdx.x; e < BB * BB; e += CHOLBIG_THREADS) {
    const int i = e >> 6, k = e & 63, si = i >> 4, sk = k >> 4;
    double mv = 0, lv = 0;
    if (si == sk) mv = Dg[cb_off(si, si) + (k & 15) * CLD + (i & 15)];           // Linv[i][k] = T[k][i]
    else if (si > sk) { mv = chol_big_mi_blk<double>(Mi, si, sk)[(i & 15) * CLD + (k & 15)]; lv = Dg[cb_off(si, sk) + (i & 15) * CLD + (k & 15)]; }
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
  chol_big_factor64(Dg, Mi, Ac + 16 * CBS, &s_fail);
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
// block repeated by every workgroup: 19.4 us per column at n = 704.  Here the whole factorisation -- and k_chol_big_prepare's work
// in front of it -- is one launch:
//   workgroup 0, the WALKER, goes down the diagonal: it factors block (c,c) in LDS, inverts the factor, publishes both as one
//     image (Mimg_c: the 10 + 6 sub-blocks exactly as they lie in LDS), then takes W(c+1,c) and the partial W(c+1,c+1) (fetched
//     by its idle waves while the last 16 pivots run), forms L(c+1,c) and the downdate itself -- the serial chain of the
//     factorisation never leaves its LDS;
//   workgroup 1 + t OWNS tile t = (r, c), c <= r, of the lower block triangle (numbered column by column).  It forms the tile of
//     the damped system from E, keeps it in its accumulators and applies the columns j as they become available (wait W(r,j),
//     W(c,j) final -> wait Mimg_j -> L(r,j) = W(r,j) Minv_j^T, L(c,j) likewise -> tile -= L(r,j) L(c,j)^T): j < c below the
//     diagonal, after which it publishes the final W(r,c) and, once Mimg_c is there, L(r,c) into the upper block (c, r) for the
//     back substitution;  j < c - 1 on the diagonal, after which it hands the partial tile to the walker.
// Tiles nobody has written yet (column 0, tile (1,1)) are formed from E by whoever needs them.
// Scalar type S: double, or float for the fp32 engine (f32 pivots: ~155 cycles per pivot instead of ~275, f32 MFMAs: 32 cycles
// instead of 64, half the bytes handed over): a factorisation that fails or whose pivots sink below tau * (their diagonal entry)
// raises LMState::chol_retry, and the f64 instance launched behind it (only_if_retry) does the solve again from E.
// Hand-overs are epoch flags (one per diagonal block, one per tile) set by a release after a barrier; every wait is bounded
// (CHOLBIG_WAIT_TICKS, then info |= 2: a rejected LM step) and a timed-out workgroup raises an abort flag the others watch.
// Progress does not need the whole launch to be resident (a card shared with other processes), only in-order dispatch: the
// walker is dispatched first, and what it needs to publish Mimg_j are tiles (j, j-1) and (j, j), which are numbered below every
// tile that waits for Mimg_j.  Up to 22 block rows the launch is resident as a whole (254 workgroups, one per CU in f64).
constexpr int CHOLDAG_MAX_NBR = 27;         // 128 cameras x 13 parameters; from 23 block rows on (277 .. 379 tiles) the launch is not resident as a whole
constexpr long long CHOLBIG_WAIT_TICKS = 200000000LL;      // 2 s of the 100 MHz clock
constexpr long long CHOLBIG_X_EMPTY = 0x7ff8dead5ba0e111LL; // k_chol_big_back_all: "x_b not there yet" (a NaN payload no computation produces)
template <typename S> __host__ __device__ constexpr int choldag_img() { return 16 * CB * (sizeof(S) == 8 ? CLD : 20); }   // Dg (10 sub-blocks) + Mi (6)
__host__ __device__ inline int choldag_nflags(int nbr) { return nbr + nbr * nbr + 1; }      // Mimg_j, W(r,c), abort
// entries of the image: Minv[i][k] (lower) and the factor's strictly-lower sub-blocks L[i][k]
// (returned in the image's own type: a conversion at the load would make the caller wait for the data there and then)
template <typename S>
__device__ __forceinline__ S choldag_img_minv(const S* __restrict__ img, int i, int k) {
  using L = CholLay<S>;
  const int si = i >> 4, sk = k >> 4;
  if (si == sk) return img[L::off(si, si) + (k & 15) * L::LD + (i & 15)];                              // Linv[i][k] = T[k][i]
  if (si > sk) return img[10 * L::BS + (si * (si - 1) / 2 + sk) * L::BS + (i & 15) * L::LD + (k & 15)];
  return (S)0;
}
template <typename S>
__device__ __forceinline__ S choldag_img_l(const S* __restrict__ img, int i, int k) {
  using L = CholLay<S>;
  const int si = i >> 4, sk = k >> 4;
  return (si > sk) ? img[L::off(si, sk) + (i & 15) * L::LD + (k & 15)] : (S)0;
}

// the damped, augmented system the factorisation works on (what k_chol_big_prepare writes into W)
struct CholDagSys {
  const double* E; int n, R; double lam; bool fresh; const double* D2c; const double* dU; const double* rhs; double big;
  // lam * (camera scaling): monotone max of the squared column norms (x_scale='jac', scipy trf.py:424,545)
  __device__ __forceinline__ double damping(int i) const {
    const double d = D2c[i];
    return lam * fmax_pos(fresh ? fmax(d, dU[i]) : d);
  }
  // entry (i, j), j <= i, outside the camera system proper: the rhs row R, its diagonal entry, identity padding
  __device__ __forceinline__ double border(int i, int j) const {
    if (i == R) return (j == R) ? big : (j < n ? rhs[j] : 0.0);
    return (i == j) ? 1.0 : 0.0;
  }
  __device__ __forceinline__ double diag(int i) const { return i < n ? E[(size_t)i * n + i] + damping(i) : border(i, i); }
};
// block (br, bc), bc <= br, of the system -> 4 x 4 sub-blocks in LDS (LOWER_ONLY: the packed lower sub-blocks), by NTHR threads
// numbered tid.  Everything a thread needs from memory is requested in one batch (clamped addresses): its entries of E and, on a
// diagonal block, the damping of its column (a thread owns one column, so at most one diagonal entry); borders are patched in after.
template <bool LOWER_ONLY, typename S, int NTHR>
__device__ __noinline__ void choldag_block_from_sys(const CholDagSys sys, int br, int bc, S* __restrict__ dst, int tid) {
  using L = CholLay<S>;
  constexpr int U = BB * BB / NTHR;
  double v[U];
  const int col = tid & 63, j = bc * BB + col;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = br * BB + ((tid + u * NTHR) >> 6);
    v[u] = sys.E[(i < sys.n && j < sys.n) ? (size_t)i * sys.n + j : 0];
  }
  double damp = 0;
  if (br == bc) damp = sys.damping(j < sys.n ? j : 0);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int row = (tid + u * NTHR) >> 6, sr = row >> 4, sc = col >> 4;
    const int i = br * BB + row;
    double x = v[u];
    if (!(i < sys.n && j < sys.n)) x = (j <= i) ? sys.border(i, j) : 0.0;
    else if (i == j) x += damp;
    if (LOWER_ONLY && sc > sr) continue;
    dst[(LOWER_ONLY ? L::off(sr, sc) : (sr * BSUB + sc) * L::BS) + (row & 15) * L::LD + (col & 15)] = (S)x;
  }
}

// the walker's start: blocks (0,0) (packed lower), (1,0) and (1,1) at once -- the three batches of loads in flight together
template <typename S>
__device__ __noinline__ void choldag_walker_start(const CholDagSys sys, S* __restrict__ F, S* __restrict__ Ar, S* __restrict__ Ac, int tid, bool three,
                                                  S* __restrict__ a0out /* [64]: the damped diagonal of block (0,0), the pivot test's reference */) {
  using L = CholLay<S>;
  constexpr int U = BB * BB / CHOLBIG_THREADS;
  double v[3][U];
  const int col = tid & 63;
  const double ediag = (tid < BB && tid < sys.n) ? sys.E[(size_t)tid * sys.n + tid] : 0.0;      // (requested with the batch below)
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int br = b == 0 ? 0 : 1, bc = b == 2 ? 1 : 0, j = bc * BB + col;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = br * BB + ((tid + u * CHOLBIG_THREADS) >> 6);
      v[b][u] = (b == 0 || three) ? sys.E[(i < sys.n && j < sys.n) ? (size_t)i * sys.n + j : 0] : 0.0;
    }
  }
  const double damp0 = sys.damping(col < sys.n ? col : 0);
  const double damp1 = three ? sys.damping(BB + col < sys.n ? BB + col : 0) : 0.0;
  if (tid < BB) a0out[tid] = (S)(tid < sys.n ? ediag + damp0 : sys.border(tid, tid));
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    if (b > 0 && !three) break;
    const int br = b == 0 ? 0 : 1, bc = b == 2 ? 1 : 0, j = bc * BB + col;
    S* dst = b == 0 ? F : b == 1 ? Ar : Ac;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = (tid + u * CHOLBIG_THREADS) >> 6, sr = row >> 4, sc = col >> 4;
      const int i = br * BB + row;
      double x = v[b][u];
      if (!(i < sys.n && j < sys.n)) x = (j <= i) ? sys.border(i, j) : 0.0;
      else if (i == j) x += (b == 0 ? damp0 : damp1);
      if (b == 0 && sc > sr) continue;
      dst[(b == 0 ? L::off(sr, sc) : (sr * BSUB + sc) * L::BS) + (row & 15) * L::LD + (col & 15)] = (S)x;
    }
  }
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
template <int CI, typename S>
__device__ __forceinline__ typename Mfma<S>::acc_t choldag_panel_out(const S* __restrict__ Aprow, const S* __restrict__ F) {
  using L = CholLay<S>;
  const int lane = threadIdx.x & 63, rc = lane & 15, kq = lane >> 4;
  const S* Mi = F + 10 * L::BS;
  S a[(CI + 1) * 4], b[(CI + 1) * 4];
#pragma unroll
  for (int k = 0; k <= CI; ++k) {
    const S* Bk = (k < CI) ? Mi + (CI * (CI - 1) / 2 + k) * L::BS : F + L::off(CI, CI);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      a[4 * k + ks] = Aprow[k * L::BS + rc * L::LD + 4 * ks + kq];
      b[4 * k + ks] = (k < CI) ? Bk[rc * L::LD + 4 * ks + kq] : Bk[(4 * ks + kq) * L::LD + rc];   // Minv(CI,k) read transposed, T_CI as it is
    }
  }
  typename Mfma<S>::acc_t acc = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < (CI + 1) * 4; ++i) acc = Mfma<S>::mma(a[i], b[i], acc);
  return acc;
}
// a whole panel, in place: the calling wave forms all four blocks of sub-block row ri in registers; the caller puts a barrier between
// this and choldag_panel_store (other waves may still be reading the row)
template <typename S> struct CholdagRow { typename Mfma<S>::acc_t o[BSUB]; };
template <typename S>
__device__ __forceinline__ CholdagRow<S> choldag_panel_row(const S* __restrict__ Ap, const S* __restrict__ F, int ri) {
  CholdagRow<S> r;
  const S* row = Ap + ri * BSUB * CholLay<S>::BS;
  r.o[3] = choldag_panel_out<3, S>(row, F);
  r.o[2] = choldag_panel_out<2, S>(row, F);
  r.o[1] = choldag_panel_out<1, S>(row, F);
  r.o[0] = choldag_panel_out<0, S>(row, F);
  return r;
}
template <typename S>
__device__ __forceinline__ void choldag_panel_store(S* __restrict__ Ap, int ri, const CholdagRow<S>& r) {
#pragma unroll
  for (int ci = 0; ci < BSUB; ++ci) store16<S>(Ap + (ri * BSUB + ci) * CholLay<S>::BS, r.o[ci]);
}
// the same with the four blocks of a row shared by two waves (half 0: blocks 3 and 0, half 1: blocks 2 and 1 -- five products each)
template <typename S> struct CholdagHalfRow { typename Mfma<S>::acc_t o[2]; };
template <typename S>
__device__ __forceinline__ CholdagHalfRow<S> choldag_panel_half(const S* __restrict__ Ap, const S* __restrict__ F, int ri, int half) {
  CholdagHalfRow<S> r;
  const S* row = Ap + ri * BSUB * CholLay<S>::BS;
  if (half == 0) { r.o[0] = choldag_panel_out<3, S>(row, F); r.o[1] = choldag_panel_out<0, S>(row, F); }
  else { r.o[0] = choldag_panel_out<2, S>(row, F); r.o[1] = choldag_panel_out<1, S>(row, F); }
  return r;
}
template <typename S>
__device__ __forceinline__ void choldag_panel_half_store(S* __restrict__ Ap, int ri, int half, const CholdagHalfRow<S>& r) {
  store16<S>(Ap + (ri * BSUB + (half == 0 ? 3 : 2)) * CholLay<S>::BS, r.o[0]);
  store16<S>(Ap + (ri * BSUB + (half == 0 ? 0 : 1)) * CholLay<S>::BS, r.o[1]);
}
// sum_k A(ri,k) B(ci,k)^T over the four sub-blocks of a row, operands fetched ahead of the 16 MFMAs
template <typename S>
__device__ __forceinline__ typename Mfma<S>::acc_t choldag_abt(const S* __restrict__ Arow, const S* __restrict__ Brow) {
  using L = CholLay<S>;
  const int lane = threadIdx.x & 63, rc = lane & 15, kq = lane >> 4;
  S a[16], b[16];
#pragma unroll
  for (int k = 0; k < BSUB; ++k)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      a[4 * k + ks] = Arow[k * L::BS + rc * L::LD + 4 * ks + kq];
      b[4 * k + ks] = Brow[k * L::BS + rc * L::LD + 4 * ks + kq];
    }
  typename Mfma<S>::acc_t p = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; ++i) p = Mfma<S>::mma(a[i], b[i], p);
  return p;
}
template <typename S>
__device__ __forceinline__ void choldag_copy_img(S* __restrict__ dst, const S* __restrict__ src) {
  using V2 = typename Vec2<S>::type;
  for (int e = threadIdx.x; e < choldag_img<S>() / 2; e += CHOLBIG_THREADS)
    reinterpret_cast<V2*>(dst)[e] = reinterpret_cast<const V2*>(src)[e];
}
// 64 x 64 block in LDS (4 x 4 sub-blocks) -> W block (br, bc)
template <typename S>
__device__ __forceinline__ void choldag_store_block(S* __restrict__ W, int npad, int br, int bc, const S* __restrict__ A) {
  using L = CholLay<S>;
  for (int e = threadIdx.x; e < BB * BB; e += CHOLBIG_THREADS) {
    const int i = e >> 6, k = e & 63;
    W[(size_t)(br * BB + i) * npad + bc * BB + k] = A[((i >> 4) * BSUB + (k >> 4)) * L::BS + (i & 15) * L::LD + (k & 15)];
  }
}

template <typename S>
__device__ __forceinline__ void choldag_body(const double* __restrict__ E, int n, LMState* __restrict__ st,
                                             double* __restrict__ D2c, S* __restrict__ W, int npad,
                                             S* __restrict__ Mimg_ws, unsigned* __restrict__ flags, unsigned epoch,
                                             int* __restrict__ info, S tau, long long* __restrict__ dbg) {
  using L = CholLay<S>;
  using acc_t = typename Mfma<S>::acc_t;
  constexpr bool F32 = sizeof(S) == 4;
  constexpr int IMG = choldag_img<S>();
  extern __shared__ __align__(16) unsigned char smem[];
  S* F = reinterpret_cast<S*>(smem);                     // image of a diagonal block: Dg (10 sub-blocks) + Mi (6)
  S* Ar = F + 16 * L::BS;                                // W(r,j) -> L(r,j)
  S* Ac = Ar + 16 * L::BS;                               // W(c,j) -> L(c,j);  the walker: the next diagonal tile
  __shared__ int s_fail, s_late, s_pref[4];
  __shared__ S s_a0[2][BB];                               // the damped diagonal of the walker's block (pivot test on f32 lanes), this block / the next
  const int nbr = npad / BB;
  unsigned* flagM = flags;
  unsigned* flagW = flags + nbr;
  unsigned* abortf = flags + nbr + nbr * nbr;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  CholDagSys sys;
  sys.E = E; sys.n = n; sys.R = cholbig_rhs_row(n); sys.lam = st->lam; sys.fresh = st->fresh != 0; sys.D2c = D2c;
  sys.rhs = E + (size_t)n * n; sys.dU = sys.rhs + n; sys.big = F32 ? (double)CHOLBIG_RHS_DIAG_F32 : CHOLBIG_RHS_DIAG;
  if (threadIdx.x == 0) { s_fail = 0; s_late = 0; }
  if (blockIdx.x == 0) {
    // ================================================================ the walker
    // SBA_CHOL_DEBUG: stamps of the 100 MHz clock at the links of the chain
    auto stamp = [&](int c, int k) { if (dbg && threadIdx.x == 0) dbg[c * 8 + k] = wall_clock64(); };
    stamp(0, 0);
    if (threadIdx.x == 0) *info = 0;
    S* Mi = F + 10 * L::BS;
    // (nobody writes column 0 or tile (1,1): the walker's first two tiles come from E as well, together with the first diagonal block
    //  and the reference diagonal of the pivot test -- one batch of loads)
    choldag_walker_start<S>(sys, F, Ar, Ac, threadIdx.x, nbr > 1, s_a0[0]);
    __syncthreads();
    for (int c = 0; c < nbr; ++c) {
      stamp(c, 4);
      // waves 4..7, while the last 16 pivots run: if both tiles of the next column are there already, fetch them now
      const bool more = c + 1 < nbr;
      const unsigned* fa = flagW + (c + 1) * nbr + c;
      const unsigned* fb = flagW + (c + 1) * nbr + c + 1;
      if (threadIdx.x < 4) s_pref[threadIdx.x] = 0;
      using V2 = typename Vec2<S>::type;
      V2 va[8], vb[8];
      bool got = false;
      const int ftid = threadIdx.x - 256;
      bool stored = false;
      S a0n = (S)0;
      auto hook = [&](int phase) {
        if (!more) return;
        // the damped diagonal of the next block (the pivot test's reference): requested in phase 1, into LDS in phase 2
        if (F32 && ftid < BB) {
          if (phase == 1) a0n = (S)sys.diag((c + 1) * BB + ftid);
          else if (phase == 2) s_a0[(c + 1) & 1][ftid] = a0n;
        }
        if (c == 0) { if (phase == 3 && lane == 0) s_pref[wid - 4] = 1; return; }     // (fetched in front of the loop)
        if (got && !stored) {                               // requested in an earlier phase: into LDS now
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int e = ftid + u * 256, row = e >> 5, col = (e & 31) * 2;
            const int o = ((row >> 4) * BSUB + (col >> 4)) * L::BS + (row & 15) * L::LD + (col & 15);
            Ar[o] = va[u].x; Ar[o + 1] = va[u].y;
            Ac[o] = vb[u].x; Ac[o + 1] = vb[u].y;
          }
          stored = true;
          if (lane == 0) s_pref[wid - 4] = 1;
          return;
        }
        if (got || phase == 3) return;
        if (!F32 && phase == 1) return;                     // (f64: phases are long and registers scarce: one look, in phase 2)
        bool ready = true;
        if (lane < 2) ready = __hip_atomic_load(lane == 0 ? fa : fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
        if (!__all(ready)) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        got = true;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = ftid + u * 256, row = e >> 5, col = (e & 31) * 2;
          va[u] = *reinterpret_cast<const V2*>(W + (size_t)((c + 1) * BB + row) * npad + c * BB + col);
          vb[u] = *reinterpret_cast<const V2*>(W + (size_t)((c + 1) * BB + row) * npad + (c + 1) * BB + col);
        }
        if (!F32) {                                         // (f64: straight into LDS, nothing kept in registers across a barrier)
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int e = ftid + u * 256, row = e >> 5, col = (e & 31) * 2;
            const int o = ((row >> 4) * BSUB + (col >> 4)) * L::BS + (row & 15) * L::LD + (col & 15);
            Ar[o] = va[u].x; Ar[o + 1] = va[u].y;
            Ac[o] = vb[u].x; Ac[o + 1] = vb[u].y;
          }
          stored = true;
          if (lane == 0) s_pref[wid - 4] = 1;
        }
      };
      chol_big_factor64<S>(F, Mi, Ac + 16 * L::BS, &s_fail, F32 ? s_a0[c & 1] : (const S*)nullptr, tau, hook);
      stamp(c, 5);
      choldag_copy_img<S>(Mimg_ws + (size_t)c * IMG, F);
      stamp(c, 6);
      choldag_publish(flagM + c, epoch);
      stamp(c, 7);
      if (!more) break;
      const bool fetched = s_pref[0] && s_pref[1] && s_pref[2] && s_pref[3];     // (read after the barrier inside choldag_publish)
      if (!fetched) {
        choldag_wait(fa, fb, epoch, abortf, &s_late);     // (c >= 1 here: at c = 0 the fetch cannot fail)
        if (s_late) break;
        stamp(c + 1, 1);
        load_block64<false, S>(W, npad, (c + 1) * BB, c * BB, Ar);
        load_block64<false, S>(W, npad, (c + 1) * BB, (c + 1) * BB, Ac);
        __syncthreads();
      } else stamp(c + 1, 1);
      stamp(c + 1, 2);
      {
        const CholdagHalfRow<S> pr = choldag_panel_half<S>(Ar, F, wid & 3, wid >> 2);
        __syncthreads();
        choldag_panel_half_store<S>(Ar, wid & 3, wid >> 2, pr);
      }
      __syncthreads();
      stamp(c + 1, 3);
      // the next diagonal block, lower sub-blocks only, straight into the layout the factorisation works on (10 sub-tiles, 8 waves)
      {
        int ri = 0;
        while ((ri + 1) * (ri + 2) / 2 <= wid) ++ri;
        const int ci = wid - ri * (ri + 1) / 2;
        const acc_t p0 = choldag_abt<S>(Ar + ri * BSUB * L::BS, Ar + ci * BSUB * L::BS);
        const int ri1 = 3, ci1 = wid + 2;                  // sub-tiles 8, 9 = (3,2), (3,3): waves 0 and 1
        acc_t p1 = {0, 0, 0, 0};
        if (wid < 2) p1 = choldag_abt<S>(Ar + ri1 * BSUB * L::BS, Ar + ci1 * BSUB * L::BS);
        store16<S>(F + L::off(ri, ci), load16<S>(Ac + (ri * BSUB + ci) * L::BS) - p0);
        if (wid < 2) store16<S>(F + L::off(ri1, ci1), load16<S>(Ac + (ri1 * BSUB + ci1) * L::BS) - p1);
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      st->cost = E[(size_t)n * n + 3 * n];              // (k_chol_big_prepare's other duty; here, off the start of the chain)
      if (F32) { st->chol_retry = s_fail ? 1 : 0; if (s_fail) st->chol_f64_retries += 1; }
      else if (s_fail) atomicOr(info, 1);
      if (s_late) atomicOr(info, 2);
    }
    // the camera scaling the damping used, for k_chol_epilogue and the next trial (monotone max: writing it twice is harmless)
    if (sys.fresh) for (int i = threadIdx.x; i < n; i += CHOLBIG_THREADS) D2c[i] = fmax(D2c[i], sys.dU[i]);
    return;
  }
  // ================================================================ a tile
  int c = 0, t = blockIdx.x - 1;
  while (t >= nbr - c) { t -= nbr - c; ++c; }
  const int r = c + t;
  const int ncol = (r == c) ? c - 1 : c;                  // columns this workgroup applies (the walker applies the last one on the diagonal)
  if (r == c && c < 2) return;
  // the own tile: 16 sub-tiles, two per wave, in the accumulators from here to the end
  acc_t acc[2];
  choldag_block_from_sys<false, S, CHOLBIG_THREADS>(sys, r, c, Ar, threadIdx.x);
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int tt = wid + 8 * h, ri = tt >> 2, ci = tt & 3;
    acc[h] = load16<S>(Ar + (ri * BSUB + ci) * L::BS);
  }
  __syncthreads();
  for (int j = 0; j < ncol; ++j) {
    if (j > 0) {
      choldag_wait(flagW + r * nbr + j, flagW + c * nbr + j, epoch, abortf, &s_late);
      if (s_late) break;
      load_block64<false, S>(W, npad, r * BB, j * BB, Ar);
      if (c != r) load_block64<false, S>(W, npad, c * BB, j * BB, Ac);
    } else {                                              // (column 0 comes from E)
      choldag_block_from_sys<false, S, CHOLBIG_THREADS>(sys, r, 0, Ar, threadIdx.x);
      if (c != r) choldag_block_from_sys<false, S, CHOLBIG_THREADS>(sys, c, 0, Ac, threadIdx.x);
    }
    choldag_wait(flagM + j, flagM + j, epoch, abortf, &s_late);
    if (s_late) break;
    choldag_copy_img<S>(F, Mimg_ws + (size_t)j * IMG);
    __syncthreads();
    if (c != r) {
      S* Ap = (wid < BSUB) ? Ar : Ac;
      const CholdagRow<S> pr = choldag_panel_row<S>(Ap, F, wid & 3);
      __syncthreads();
      choldag_panel_store<S>(Ap, wid & 3, pr);
    } else {
      const CholdagHalfRow<S> pr = choldag_panel_half<S>(Ar, F, wid & 3, wid >> 2);
      __syncthreads();
      choldag_panel_half_store<S>(Ar, wid & 3, wid >> 2, pr);
    }
    __syncthreads();
    const S* Lr = Ar;
    const S* Lc = (c != r) ? Ac : Ar;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int tt = wid + 8 * h, ri = tt >> 2, ci = tt & 3;
      acc[h] -= choldag_abt<S>(Lr + ri * BSUB * L::BS, Lc + ci * BSUB * L::BS);
    }
    __syncthreads();
  }
  if (s_late) { if (threadIdx.x == 0) atomicOr(info, 2); return; }
  // the tile is final (below the diagonal) or ready for the walker (on it): straight from the accumulators to W, then the flag
  if (c > 0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int tt = wid + 8 * h, ri = tt >> 2, ci = tt & 3;
      S* Cg = W + (size_t)(r * BB + ri * CB) * npad + c * BB + ci * CB;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) Cg[(size_t)Mfma<S>::row_of(lane, rg) * npad + (lane & 15)] = acc[h][rg];
    }
    choldag_publish(flagW + r * nbr + c, epoch);
  }
  if (r == c) return;
  // ---- below the diagonal: L(r,c) for the back substitution, once Mimg_c is there
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int tt = wid + 8 * h, ri = tt >> 2, ci = tt & 3;
    store16<S>(Ar + (ri * BSUB + ci) * L::BS, acc[h]);
  }
  choldag_wait(flagM + c, flagM + c, epoch, abortf, &s_late);
  if (s_late) { if (threadIdx.x == 0) atomicOr(info, 2); return; }
  choldag_copy_img<S>(F, Mimg_ws + (size_t)c * IMG);
  __syncthreads();
  {
    const CholdagHalfRow<S> pr = choldag_panel_half<S>(Ar, F, wid & 3, wid >> 2);
    __syncthreads();
    choldag_panel_half_store<S>(Ar, wid & 3, wid >> 2, pr);
  }
  __syncthreads();
  choldag_store_block<S>(W, npad, c, r, Ar);
}

// grid = 1 + nbr (nbr + 1) / 2; 512 threads; dynamic LDS = CHOLBIG_LDS_BLOCKS sub-blocks of S.
// only_if_retry: the f64 instance the fp32 engine launches behind the f32 one -- it leaves at once unless that one refused the system.
// (Both in one launch, every workgroup waiting for the walker's verdict, was measured: no faster at 64 cameras, 2-4 us at 24 -- and
//  a workgroup that waits instead of leaving keeps its CU, which breaks the progress argument above on a shared card: the 8-rank
//  rehearsal on one card ran into the bounded waits.)
template <typename S>
__global__ __launch_bounds__(CHOLBIG_THREADS) void k_chol_big_dag(const double* __restrict__ E, int n, LMState* __restrict__ st,
                                                                  double* __restrict__ D2c, S* __restrict__ W, int npad,
                                                                  S* __restrict__ Mimg_ws, unsigned* __restrict__ flags, unsigned epoch,
                                                                  int* __restrict__ info, int only_if_retry, S tau,
                                                                  long long* __restrict__ dbg) {
  if (st->status >= 0 || (only_if_retry && !st->chol_retry)) return;     // (the same record on every workgroup: nobody is left waiting)
  choldag_body<S>(E, n, st, D2c, W, npad, Mimg_ws, flags, epoch, info, tau, dbg);
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
//     y_t -= L(b,t)^T x_b   for b = last .. t + 1, as the x_b arrive;      x_t = Minv_t^T y_t,   published as its own flag (below),
// so the chain through the blocks is a hand-over between resident workgroups (at most 27 of them) instead of a launch
// boundary: the L(b,t) block a workgroup needs next is already in its registers when x_b arrives.  The waits are bounded
// (CHOLBIG_WAIT_TICKS of the 100 MHz clock, then the solve is flagged as failed: a rejected LM step) and every workgroup of
// the launch is resident at once -- the rule of sba_ipc.hpp.
template <typename S>
__device__ __forceinline__ void chol_big_back_all_body(const S* __restrict__ W, int npad, int n, const double* __restrict__ Ld_ws,
                                                       const double* __restrict__ Minv_ws, double* __restrict__ xv,
                                                       unsigned epoch, double* __restrict__ sol, int* __restrict__ info,
                                                       const S* __restrict__ Mimg_ws, double* __restrict__ s_y, double* __restrict__ s_x,
                                                       double (*__restrict__ s_p)[BB], double* __restrict__ s_M, int& s_late,
                                                       double* __restrict__ s_sol /* block row 0 keeps the whole solution (for the epilogue) */) {
  constexpr int IMG = choldag_img<S>();
  const int nbx = (n + BB - 1) / BB;
  const int t = blockIdx.x, i = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int R = cholbig_rhs_row(n), Rb = R / BB, Rl = R % BB;
  // y_t = row R of the factor (columns of block t), Minv_t, and the first L block: all requested before anything is waited for
  // (values stay in the type they were loaded in until they are used: the requests of this prologue are all in flight together)
  S yk = (S)0;
  double yk_dense = 0;                               // (the dense copies of the per-column launches are doubles whatever S is)
  if (threadIdx.x < BB) {
    const int k = t * BB + threadIdx.x;
    if (k < n) {
      if (t < Rb) yk = W[(size_t)(t * BB + Rl) * npad + Rb * BB + threadIdx.x];
      else if (Mimg_ws) yk = choldag_img_l<S>(Mimg_ws + (size_t)Rb * IMG, Rl, threadIdx.x);
      else yk_dense = Ld_ws[(size_t)Rb * BB * BB + Rl * BB + threadIdx.x];
    }
  }
  const double* Mg = Minv_ws + (size_t)t * BB * BB;
  S mreg[BB * BB / 256];
  double mreg_dense[BB * BB / 256];
#pragma unroll
  for (int u = 0; u < BB * BB / 256; ++u) {
    const int e = threadIdx.x + 256 * u;
    mreg[u] = (S)0; mreg_dense[u] = 0;
    if (Mimg_ws) mreg[u] = choldag_img_minv<S>(Mimg_ws + (size_t)t * IMG, e >> 6, e & 63);
    else mreg_dense[u] = Mg[e];
  }
  S lreg[BB / 4];                                    // L(b,t)[k][i], k = part + 4 u (kept as loaded: a conversion here would wait for the data)
  auto fetch_L = [&](int b) {
    const S* Lb = W + (size_t)(t * BB) * npad + b * BB;
#pragma unroll
    for (int u = 0; u < BB / 4; ++u) lreg[u] = Lb[(size_t)(part + 4 * u) * npad + i];
  };
  if (nbx - 1 > t) fetch_L(nbx - 1);
  // x travels as its own flag: xv has one copy per parity of the epoch, filled with a NaN nobody computes (CHOLBIG_X_EMPTY) until the
  // owner of a block row stores its x there; a waiting thread polls the very word it needs -- one trip through memory per hand-over
  // instead of two (flag, then data).  Every workgroup empties its slice of the OTHER copy for the next launch.
  double* xcur = xv + (size_t)(epoch & 1) * npad;
  double* xnext = xv + (size_t)((epoch + 1) & 1) * npad;
  if (threadIdx.x < BB) xnext[t * BB + threadIdx.x] = __longlong_as_double(CHOLBIG_X_EMPTY);
  if (threadIdx.x == 0) s_late = 0;
  if (threadIdx.x < BB) s_y[threadIdx.x] = (double)yk + yk_dense;
#pragma unroll
  for (int u = 0; u < BB * BB / 256; ++u) s_M[threadIdx.x + 256 * u] = (double)mreg[u] + mreg_dense[u];
  __syncthreads();
  for (int b = nbx - 1; b > t; --b) {
    if (threadIdx.x < BB) {                          // wave 0 waits for x_b and brings it in
      const long long t0 = wall_clock64();
      bool late = false;
      long long bits;
      int it = 0;
      while (true) {
        bits = __hip_atomic_load(reinterpret_cast<const long long*>(xcur + b * BB + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(bits != CHOLBIG_X_EMPTY)) break;
        if ((++it & 31) == 0 && __any(wall_clock64() - t0 > CHOLBIG_WAIT_TICKS)) { late = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (late) s_late = 1;
      s_x[threadIdx.x] = late ? 0.0 : __longlong_as_double(bits);
      if (t == 0) s_sol[b * BB + threadIdx.x] = s_x[threadIdx.x];
    }
    __syncthreads();
    if (s_late) break;
    double u_ = 0;
#pragma unroll
    for (int u = 0; u < BB / 4; ++u) u_ += (double)lreg[u] * s_x[part + 4 * u];
    if (b - 1 > t) fetch_L(b - 1);                   // the next block travels while this one is folded
    s_p[part][i] = u_;
    __syncthreads();
    if (part == 0) s_y[i] -= (s_p[0][i] + s_p[1][i]) + (s_p[2][i] + s_p[3][i]);
    __syncthreads();
  }
  if (s_late) { if (threadIdx.x == 0) atomicOr(info, 2); return; }     // (this workgroup's x stays empty: the rows above time out too)
  // x_t = Minv_t^T y_t   (Minv is lower: entries k < i are stored zeros)
  double s = 0;
  for (int k = part; k < BB; k += 4) s += s_M[k * BB + i] * s_y[k];
  s_p[part][i] = s;
  __syncthreads();
  if (part == 0) {
    double x = (s_p[0][i] + s_p[1][i]) + (s_p[2][i] + s_p[3][i]);
    if (__double_as_longlong(x) == CHOLBIG_X_EMPTY) x = __longlong_as_double(0x7ff8000000000000LL);
    __hip_atomic_store(reinterpret_cast<long long*>(xcur + t * BB + i), __double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t * BB + i < n) sol[t * BB + i] = x;
    if (t == 0) s_sol[i] = x;
  }
}
// W / Mimg_ws: doubles (the per-column launches, k_chol_big_dag<double>) or floats (f32_data: k_chol_big_dag<float>, unless its
// factorisation was refused and the f64 instance behind it did the work: LMState::chol_retry).
// Block row 0 is the last to finish and has seen every x_b on its way: it keeps them in LDS and runs the LM epilogue on them
// (chol_epilogue_body: the trial cameras, the predicted reduction, the failure flag) -- one launch less per trial.
constexpr int CHOLBIG_MAX_NBX = 27;         // 1728 unknowns: 128 cameras x 13 parameters + the padding
template <typename T>
__global__ __launch_bounds__(256) void k_chol_big_back_all(const void* __restrict__ W, int npad, int n, const double* __restrict__ Ld_ws,
                                                           const double* __restrict__ Minv_ws,
                                                           double* __restrict__ xv /* [2][npad], CHOLBIG_X_EMPTY before the first launch */,
                                                           unsigned epoch, double* __restrict__ sol, int* __restrict__ info,
                                                           LMState* __restrict__ st,
                                                           const void* __restrict__ Mimg_ws /* k_chol_big_dag's images, or null: the dense copies */,
                                                           int f32_data, const double* __restrict__ E, int C, const double* __restrict__ D2c,
                                                           const ParamSets<T> ps, double* __restrict__ delta_c,
                                                           const int32_t* __restrict__ tie, const int32_t* __restrict__ first) {
  __shared__ double s_y[BB], s_x[BB], s_p[4][BB], s_M[BB * BB], s_sol[CHOLBIG_MAX_NBX * BB], s_scr[16];
  __shared__ int s_late, s_flag;
  const int st_status = st->status, st_retry = st->chol_retry;      // (both requested at once: two dependent trips to memory otherwise)
  if (st_status >= 0) {                             // (the same record on every workgroup: nobody is left waiting)
    // a launch that has nothing to do (the solve has ended, the iterations enqueued behind it drain) still empties its slice of the
    // other copy of x: the next launch -- of the next solve on this handle -- counts on finding it empty
    if (threadIdx.x < BB) xv[(size_t)((epoch + 1) & 1) * npad + blockIdx.x * BB + threadIdx.x] = __longlong_as_double(CHOLBIG_X_EMPTY);
    return;
  }
  if (f32_data && !st_retry)
    chol_big_back_all_body<float>(static_cast<const float*>(W), npad, n, Ld_ws, Minv_ws, xv, epoch, sol, info,
                                  static_cast<const float*>(Mimg_ws), s_y, s_x, s_p, s_M, s_late, s_sol);
  else
    chol_big_back_all_body<double>(static_cast<const double*>(W), npad, n, Ld_ws, Minv_ws, xv, epoch, sol, info,
                                   static_cast<const double*>(Mimg_ws), s_y, s_x, s_p, s_M, s_late, s_sol);
  if (blockIdx.x != 0) return;
  __syncthreads();                                  // (a timed-out substitution has raised info: the epilogue then rejects the step)
  chol_epilogue_body<T>(E, C, n, st, D2c, ps, delta_c, s_sol, info, tie, first, s_scr, &s_flag);
}

}  // namespace SBA_NS
