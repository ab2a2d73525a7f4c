// sba_chol_blocked.hpp -- blocked Cholesky + solve of the reduced camera system, one workgroup, all in LDS.
//
//   A = S + lam*diag(D2c)  (n = 11*C <= 176, padded to n16 = 16*nb with an identity tail)
//   A = L L^T ;  delta_c = A^-1 rhs
//
// Block size 16 (= one f64 MFMA 16x16x4 tile); the lower triangle is kept as 16x16 blocks with a 17-double
// row stride (conflict-free ds_read_b64 for the MFMA operand pattern lane -> [row lane&15][col lane>>4]).
// Per block column jb (two barriers):
//   B   the diagonal block already holds Linv^T (below); every wave forms its row blocks of L21 = A21 Linv^T with
//       4 f64 MFMAs per 16x16 block, one wave forward-solves the rhs block (y = Linv rhs), and wave 0 -- which owns the
//       block right below the diagonal -- downdates the next diagonal tile with it (chol_panel_update_diag: the panel product
//       is formed transposed, so its accumulator registers are the operands of the downdate).
//   C   wave 0 factors the next diagonal tile (look-ahead) while wave 1 downdates the rhs tail and waves 1..7 apply the
//       rank-16 trailing update tile by tile (4 MFMAs per tile).
//   The diagonal factorisation (chol16_wave) is the serial chain of the kernel (one cross-lane broadcast per pivot: the row's own
//   diagonal entry is downdated with the lane's own l_ik): lanes 0..15 own the rows of the tile,
//   lanes 16..31 run the rows of the identity through the same column operations and so end up with Linv^T, which is
//   all that is stored (L11 itself is never needed again: B' and the back substitution both multiply by Linv).
// After the last column the blocked back substitution runs right-looking over all waves.
#pragma once
#include "sba_lm_kernels.hpp"

namespace SBA_NS {

constexpr int CB = 16;                 // block edge
constexpr int CLD = 17;                // row stride inside a block (doubles)
constexpr int CBS = CB * CLD;          // doubles per block
constexpr int CHOLB_THREADS = 512;          // k_cholesky_stream
constexpr int CHOLB_LDS_THREADS = 512;      // k_cholesky_blocked (1024 threads were measured twice: same time, the look-ahead phase is bound per SIMD, not per wave)
constexpr int CHOLB_MAX_NB = 11;       // 176 rows

__device__ inline int cb_off(int r, int c) { return (r * (r + 1) / 2 + c) * CBS; }

// wave-uniform broadcast of lane `src` (compile-time constant) through SGPRs: v_readlane_b32 x2, no LDS round trip
__device__ inline double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// 1/sqrt(x): hardware estimate (5e-8) + one Newton step -> 4e-15 relative (tools/micro/rsq_acc.hip).  The pivot only
// has to be used consistently (L_kk = a_kk * piv and the column scaled by the same piv), which perturbs the factored
// diagonal by ~8e-15 relative -- below the rounding error the 176 column operations accumulate anyway.
__device__ inline double rsqrt_nr(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  return y * (1.5 - 0.5 * x * y * y);
}

// Cholesky of one 16x16 diagonal block by the calling wave.  Lane i < 16 owns row i of the tile, lane 16+j the row e_j
// of the identity; both kinds go through the same right-looking column steps (x_k = a_k / L_kk ; a_c -= x_k L[c][k]),
// whose broadcasts of column k go through v_readlane (SGPRs) instead of LDS round trips.  The identity rows end up as
// the rows of Linv^T, which replace the tile.  Returns false (wave-uniform) when the block is not positive definite.
template <bool NEWTON = true>
__device__ __forceinline__ bool chol16_wave(double* __restrict__ blk) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15;
  const bool ident = lane >= 16;
  double a[CB];
#pragma unroll
  for (int j = 0; j < CB; ++j) { const double v = blk[i * CLD + j]; a[j] = ident ? ((j == i) ? 1.0 : 0.0) : v; }
  double dg = blk[i * CLD + i];            // the row's own diagonal entry (lanes 0..15), downdated with the row's own l_ik
  __builtin_amdgcn_wave_barrier();
  // The serial chain of the whole factorisation runs through here.  Pivot k+1 is row k+1's diagonal entry after column step
  // k, and that downdate needs nothing from another lane (d -= l_ik^2 with the lane's own l_ik), so one link of the chain
  //     l_ik = a_ik * piv_k  ->  d -= l_ik^2  ->  v_readlane x2 (lane k+1)  ->  v_rsq_f64 (+ Newton)  -> piv_k+1
  // has a single cross-lane broadcast in it; the positivity test accumulates beside the chain (a non-positive pivot turns
  // the tile into NaNs, which the failure flag discards), and the broadcasts that update the other columns are nobody's
  // critical path.
  auto pivot = [](double x) { return NEWTON ? rsqrt_nr(x) : __builtin_amdgcn_rsq(x); };
  double akk = readlane_f64(dg, 0);
  // positivity: a pivot a_kk <= 0 (or non-finite) makes 1/sqrt(a_kk) NaN or infinite; the pivots are summed beside the chain (one
  // add per link instead of two compares and two ANDs -- the chain is issue-bound) and the sum is tested once at the end
  double piv = pivot(akk);
  double chk = piv;
#pragma unroll
  for (int k = 0; k < CB; ++k) {
    const double lik = a[k] * piv;
    a[k] = lik;
    if (k + 1 < CB) {
      dg = __builtin_fma(-lik, lik, dg);
      akk = readlane_f64(dg, k + 1);
      piv = pivot(akk);
      chk += piv;
    }
#pragma unroll
    for (int j = k + 1; j < CB; ++j) a[j] -= lik * readlane_f64(lik, j);
  }
  if (lane >= 16 && lane < 32) {
#pragma unroll
    for (int j = 0; j < CB; ++j) blk[i * CLD + j] = a[j];     // row i of Linv^T (zero left of the diagonal)
  }
  return isfinite(chk) && chk > 0.0;
}

// one 16x16 block of L21 = A21 * Linv^T, in place, by the calling wave (4 chained f64 MFMAs)
__device__ __forceinline__ void chol_panel_block(double* __restrict__ Ablk, const double* __restrict__ LinvT) {
  const int lane = threadIdx.x & 63;
  const double* Pa = Ablk + (lane & 15) * CLD + (lane >> 4);          // A[row lane&15][k = 4ks + lane>>4]
  const double* Pl = LinvT + (lane >> 4) * CLD + (lane & 15);         // LinvT[k = 4ks + lane>>4][col lane&15]
  Mfma<double>::acc_t acc = {0, 0, 0, 0};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) acc = Mfma<double>::mma(Pa[4 * ks], Pl[4 * ks * CLD], acc);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) Ablk[((lane >> 4) + 4 * rg) * CLD + (lane & 15)] = acc[rg];
  __builtin_amdgcn_wave_barrier();
}

// D -= P P^T (or Pa Pb^T) for one 16x16 tile by the calling wave
__device__ __forceinline__ void chol_update_tile(double* __restrict__ Dt, const double* __restrict__ Pa_blk, const double* __restrict__ Pb_blk) {
  const int lane = threadIdx.x & 63;
  const double* Pa = Pa_blk + (lane & 15) * CLD + (lane >> 4);
  const double* Pb = Pb_blk + (lane & 15) * CLD + (lane >> 4);
  // (the product is accumulated with its own sign and subtracted at the end: negating the operand is a VALU instruction per
  //  k-step on the f64 lanes the MFMAs are waiting for)
  Mfma<double>::acc_t acc, prod = {0, 0, 0, 0};
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) acc[rg] = Dt[((lane >> 4) + 4 * rg) * CLD + (lane & 15)];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) prod = Mfma<double>::mma(Pa[4 * ks], Pb[4 * ks], prod);
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) Dt[((lane >> 4) + 4 * rg) * CLD + (lane & 15)] = acc[rg] - prod[rg];
  __builtin_amdgcn_wave_barrier();
}

// Two tiles at once: a wave's block product is 4 dependent MFMAs (256 cycles of issue, ~130 of latency each) behind an LDS round trip,
// ~850 cycles during which the matrix pipe of its SIMD is busy 256 -- two independent products interleaved keep it busy.
__device__ __forceinline__ void chol_update_tile2(double* __restrict__ Dt0, const double* __restrict__ Pa0_blk, const double* __restrict__ Pb0_blk,
                                                  double* __restrict__ Dt1, const double* __restrict__ Pa1_blk, const double* __restrict__ Pb1_blk) {
  const int lane = threadIdx.x & 63;
  const int po = (lane & 15) * CLD + (lane >> 4);
  double a0[4], b0[4], a1[4], b1[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { a0[ks] = Pa0_blk[po + 4 * ks]; b0[ks] = Pb0_blk[po + 4 * ks]; }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { a1[ks] = Pa1_blk[po + 4 * ks]; b1[ks] = Pb1_blk[po + 4 * ks]; }
  Mfma<double>::acc_t acc0, acc1, p0 = {0, 0, 0, 0}, p1 = {0, 0, 0, 0};
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) { acc0[rg] = Dt0[((lane >> 4) + 4 * rg) * CLD + (lane & 15)]; acc1[rg] = Dt1[((lane >> 4) + 4 * rg) * CLD + (lane & 15)]; }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { p0 = Mfma<double>::mma(a0[ks], b0[ks], p0); p1 = Mfma<double>::mma(a1[ks], b1[ks], p1); }
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) Dt0[((lane >> 4) + 4 * rg) * CLD + (lane & 15)] = acc0[rg] - p0[rg];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) Dt1[((lane >> 4) + 4 * rg) * CLD + (lane & 15)] = acc1[rg] - p1[rg];
  __builtin_amdgcn_wave_barrier();
}

// Wave 0's share of phase B in one go: P = A21 Linv^T for the block right below the diagonal, then D -= P P^T for the next
// diagonal tile.  The panel product is formed TRANSPOSED (operands swapped: P^T = Linv A21^T), because the accumulator
// layout of P^T -- lane holds P[lane & 15][(lane >> 4) + 4 reg] -- is exactly the A/B operand layout the downdate needs:
// the four accumulator registers go straight back into the MFMAs and the serial chain loses an LDS write/read round trip.
// P itself is still written to LDS for the other waves' trailing updates (not on the chain).
__device__ __forceinline__ void chol_panel_update_diag(double* __restrict__ Pblk, const double* __restrict__ LinvT, double* __restrict__ Dt) {
  const int lane = threadIdx.x & 63;
  const double* Pa = Pblk + (lane & 15) * CLD + (lane >> 4);          // A21[row lane&15][k = 4ks + lane>>4]  (B operand of the swapped product)
  const double* Pl = LinvT + (lane >> 4) * CLD + (lane & 15);         // LinvT[k = 4ks + lane>>4][lane&15] = Linv[lane&15][k]  (A operand)
  Mfma<double>::acc_t d;
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) d[rg] = Dt[((lane >> 4) + 4 * rg) * CLD + (lane & 15)];
  Mfma<double>::acc_t pt = {0, 0, 0, 0};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) pt = Mfma<double>::mma(Pl[4 * ks * CLD], Pa[4 * ks], pt);
  __builtin_amdgcn_wave_barrier();
  Mfma<double>::acc_t pp = {0, 0, 0, 0};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) pp = Mfma<double>::mma(pt[ks], pt[ks], pp);
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) d[rg] -= pp[rg];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) Pblk[(lane & 15) * CLD + (lane >> 4) + 4 * rg] = pt[rg];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) Dt[((lane >> 4) + 4 * rg) * CLD + (lane & 15)] = d[rg];
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------ the same building blocks for either scalar type
// S = double: the kernel as it has been since round 1 (f64 VALU chain, v_mfma_f64_16x16x4).  S = float (round 4, fp32 engine only):
// the tile factorisation on f32 lanes (v_rsq_f32 is accurate to 1 ulp: no Newton step; one v_readlane per broadcast instead of two;
// f32 VALU instructions issue in half the cycles) and the panel / trailing products on v_mfma_f32_16x16x4_f32 (32 cycles instead of
// 64, half the LDS bytes).  The two MFMAs place their accumulators differently (Mfma<S>::row_of), everything else is shared.
template <typename S> struct CholNum;
template <> struct CholNum<double> {
  static constexpr int LD = CLD, BS = CB * LD;                 // row stride inside a 16 x 16 block: 17 doubles (conflict-free 8-byte MFMA operand reads)
  static __device__ __forceinline__ int off(int r, int c) { return (r * (r + 1) / 2 + c) * BS; }
  static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
  static __device__ __forceinline__ double bcast(double v, int src) { return readlane_f64(v, src); }
  static __device__ __forceinline__ double rsq(double x) { return __builtin_amdgcn_rsq(x); }
  static __device__ __forceinline__ double xrow(double v) { return xrow_sum(v); }
};
template <> struct CholNum<float> {
  // 20 floats: operand reads (row lane & 15, k = lane >> 4) and accumulator accesses (row 4 (lane >> 4) + reg, column lane & 15) both
  // fall on 32 different banks per half wave (17 floats: up to four lanes per bank on the accumulator pattern)
  static constexpr int LD = 20, BS = CB * LD;
  static __device__ __forceinline__ int off(int r, int c) { return (r * (r + 1) / 2 + c) * BS; }
  static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
  static __device__ __forceinline__ float bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
  static __device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
  static __device__ __forceinline__ float xrow(float v) {       // sum over lanes l, l^16, l^32, l^48 (see xrow_sum)
    unsigned a0 = __float_as_uint(v), a1 = a0;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a0), "+v"(a1));
    const float p = __uint_as_float(a0) + __uint_as_float(a1);
    a0 = __float_as_uint(p); a1 = a0;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a0), "+v"(a1));
    return __uint_as_float(a0) + __uint_as_float(a1);
  }
};

// block layout in LDS: LD = row stride inside a 16 x 16 block, in elements of S
template <typename S, int LDV = CholNum<S>::LD> struct CholLay {
  static constexpr int LD = LDV, BS = CB * LDV;
  static __device__ __forceinline__ int off(int r, int c) { return (r * (r + 1) / 2 + c) * BS; }
};

// chol16_wave for either scalar type.  a0 (S = float): the 16 diagonal entries of the DAMPED system as they were loaded; the
// factorisation is refused (returns false) when a pivot L_ii^2 has sunk below tau * a0_i -- at tau ~ 2^-23 the pivot is the rounding
// noise of its own diagonal entry and the rows behind it carry no information (the caller then factors in f64).
template <typename S, bool NEWTON, typename L = CholLay<S>>
__device__ __forceinline__ bool chol16_wave_t(S* __restrict__ blk, const S* __restrict__ a0 = nullptr, S tau = (S)0) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15;
  const bool ident = lane >= 16;
  S a[CB];
#pragma unroll
  for (int j = 0; j < CB; ++j) { const S v = blk[i * L::LD + j]; a[j] = ident ? ((j == i) ? (S)1 : (S)0) : v; }
  S dg = blk[i * L::LD + i];
  __builtin_amdgcn_wave_barrier();
  auto pivot = [](S x) -> S {
    if constexpr (NEWTON) return (S)rsqrt_nr((double)x);
    else return CholNum<S>::rsq(x);
  };
  S akk = CholNum<S>::bcast(dg, 0);
  S piv = pivot(akk);
  S chk = piv;
#pragma unroll
  for (int k = 0; k < CB; ++k) {
    const S lik = a[k] * piv;
    a[k] = lik;
    if (k + 1 < CB) {
      dg = CholNum<S>::fma(-lik, lik, dg);
      akk = CholNum<S>::bcast(dg, k + 1);
      piv = pivot(akk);
      chk += piv;
    }
#pragma unroll
    for (int j = k + 1; j < CB; ++j) a[j] -= lik * CholNum<S>::bcast(lik, j);
  }
  if (lane >= 16 && lane < 32) {
#pragma unroll
    for (int j = 0; j < CB; ++j) blk[i * L::LD + j] = a[j];     // row i of Linv^T (zero left of the diagonal)
  }
  bool ok = isfinite(chk) && chk > (S)0;
  if (a0) {                                                    // pivot growth test (wave-uniform pointer)
    __builtin_amdgcn_wave_barrier();
    const S li = blk[i * L::LD + i];                             // 1 / L_ii
    const bool sunk = lane < 16 && !(((S)1 / (li * li)) >= tau * a0[i]);
    ok = ok && !__any(sunk);
  }
  return ok;
}

template <typename S, typename L = CholLay<S>>
__device__ __forceinline__ void chol_panel_block_t(S* __restrict__ Ablk, const S* __restrict__ LinvT) {
  const int lane = threadIdx.x & 63;
  const S* Pa = Ablk + (lane & 15) * L::LD + (lane >> 4);
  const S* Pl = LinvT + (lane >> 4) * L::LD + (lane & 15);
  typename Mfma<S>::acc_t acc = {0, 0, 0, 0};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) acc = Mfma<S>::mma(Pa[4 * ks], Pl[4 * ks * L::LD], acc);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) Ablk[Mfma<S>::row_of(lane, rg) * L::LD + (lane & 15)] = acc[rg];
  __builtin_amdgcn_wave_barrier();
}

// (see chol_panel_update_diag: the transposed panel product's accumulators are the operands of the downdate; with the f32 MFMA the
//  accumulator rows are 4 (lane >> 4) + reg instead of (lane >> 4) + 4 reg -- another order of the same sixteen k, used on both sides)
template <typename S, typename L = CholLay<S>>
__device__ __forceinline__ void chol_panel_update_diag_t(S* __restrict__ Pblk, const S* __restrict__ LinvT, S* __restrict__ Dt) {
  const int lane = threadIdx.x & 63;
  const S* Pa = Pblk + (lane & 15) * L::LD + (lane >> 4);
  const S* Pl = LinvT + (lane >> 4) * L::LD + (lane & 15);
  typename Mfma<S>::acc_t d;
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) d[rg] = Dt[Mfma<S>::row_of(lane, rg) * L::LD + (lane & 15)];
  typename Mfma<S>::acc_t pt = {0, 0, 0, 0};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) pt = Mfma<S>::mma(Pl[4 * ks * L::LD], Pa[4 * ks], pt);
  __builtin_amdgcn_wave_barrier();
  typename Mfma<S>::acc_t pp = {0, 0, 0, 0};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) pp = Mfma<S>::mma(pt[ks], pt[ks], pp);
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) d[rg] -= pp[rg];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) Pblk[(lane & 15) * L::LD + Mfma<S>::row_of(lane, rg)] = pt[rg];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) Dt[Mfma<S>::row_of(lane, rg) * L::LD + (lane & 15)] = d[rg];
  __builtin_amdgcn_wave_barrier();
}

// What k_cholesky_blocked shares between its scalar types (static LDS).  MAXNB = block rows the kernel is built for: 11 (176 unknowns,
// both scalar types fit the LDS) or 16 (256 unknowns: the f32 triangle only -- round 4, fp32 rigs of 17 .. 23 cameras).
template <int MAXNB> struct CholbShared {
  int fail;
  short rc[MAXNB * (MAXNB + 1) / 2];                  // block index -> (r << 8 | c)
  double dd[MAXNB * CB];                              // lam * D2c (diagonal damping); 0 in the padded tail
  double x[MAXNB * CB];                               // rhs in, solution out
};
template <int MAXNB> struct CholbPre {
  static constexpr int BPR = CHOLB_LDS_THREADS / 128;                      // blocks of the first column per load round (128 threads each)
  static constexpr int U0 = (MAXNB + BPR - 1) / BPR;                       // rounds for the first block column
  static constexpr int NREM = MAXNB * (MAXNB - 1) / 2;                     // the other blocks
  static constexpr int TREM = CHOLB_LDS_THREADS - 64;                      // loaded by waves 1..7
  static constexpr int U1_ALL = (NREM * 128 + TREM - 1) / TREM;            // rounds that would cover them all (16 at 11 block rows, 35 at 16)
  static constexpr int U1 = U1_ALL < 16 ? U1_ALL : 16;                     // ... of which this many are prefetched into registers (2 doubles each);
                                                                           // the blocks beyond them are fetched after the prefetched ones are in LDS
};

// Factorisation + both substitutions in scalar type S on the block triangle in dynamic LDS.  PREFETCHED: the system's entries are in
// the c0 / c1 registers the kernel requested in its first instructions; otherwise (second attempt after a refused f32 factorisation)
// they are read from E here.  Returns the failure flag; the solution is left in sh.x.
template <typename S, bool NEWTON, bool PREFETCHED, int MAXNB, typename L = CholLay<S>>
__device__ __forceinline__ bool cholb_core(unsigned char* __restrict__ smem, CholbShared<MAXNB>& sh, const double* __restrict__ E, const int n,
                                           double (&c0)[CholbPre<MAXNB>::U0][2], double (&c1)[CholbPre<MAXNB>::U1][2], const int (&rc1)[CholbPre<MAXNB>::U1],
                                           const S tau, long long* __restrict__ dbg, int& nstamp) {
#define CHOL_STAMP() do { if (dbg && threadIdx.x == 0) dbg[nstamp] = clock64(); ++nstamp; } while (0)
  using Pre = CholbPre<MAXNB>;
  constexpr int BPR = Pre::BPR, U0 = Pre::U0, TREM = Pre::TREM, U1 = Pre::U1;
  constexpr bool F32 = std::is_same<S, float>::value;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nb = (n + CB - 1) / CB;
  const int n16 = nb * CB;
  const int nblk = nb * (nb + 1) / 2;
  S* Lb = reinterpret_cast<S*>(smem);                          // nblk blocks
  S* s_y = Lb + nblk * L::BS;                                    // [n16]  rhs -> y -> x
  S* s_a0 = s_y + n16;                                         // [n16]  damped diagonal as loaded (growth test of the f32 factorisation)
  const int s4 = tid >> 7, ii0 = (tid >> 3) & 15, jp0 = tid & 7;
  auto fix = [&](int I, int J, double& v0, double& v1) {       // padded tail = identity
    if (I >= n || J >= n) v0 = (I == J) ? 1.0 : 0.0;
    if (I >= n || J + 1 >= n) v1 = (I == J + 1) ? 1.0 : 0.0;
  };
  if (tid < n16) s_y[tid] = (S)sh.x[tid];
  if constexpr (PREFETCHED) {
#pragma unroll
    for (int u = 0; u < U0; ++u) fix((BPR * u + s4) * CB + ii0, 2 * jp0, c0[u][0], c0[u][1]);
    {
      const int ii = ii0, jp = jp0;
#pragma unroll
      for (int u = 0; u < U0; ++u) {
        const int r = BPR * u + s4;
        if (r < nb) {
          const int I = r * CB + ii, J = 2 * jp;
          S* dst = Lb + L::off(r, 0) + ii * L::LD + J;
          const double v0 = c0[u][0] + ((I == J && I < n) ? sh.dd[I] : 0.0), v1 = c0[u][1] + ((I == J + 1 && I < n) ? sh.dd[I] : 0.0);
          dst[0] = (S)v0;
          dst[1] = (S)v1;
          if (F32 && I == J) s_a0[I] = (S)v0;
          if (F32 && I == J + 1) s_a0[I] = (S)v1;
        }
      }
    }
    __syncthreads();
    CHOL_STAMP();
    if (wid > 0) {
#pragma unroll
      for (int u = 0; u < U1; ++u) {
        const int e = (tid - 64) + TREM * u;
        if (e < (nblk - nb) * 128) {
          // (row, column) from the load loop's registers: as a table lookup in LDS in front of every store, sixteen dependent
          // LDS round trips per thread made this phase 9k cycles, twice the first tile's factorisation it runs beside
          const int rc = rc1[u];
          const int br = rc >> 8, bc = rc & 255;
          const int ii = (e >> 3) & 15, I = br * CB + ii, J = bc * CB + 2 * (e & 7);
          S* dst = Lb + L::off(br, bc) + ii * L::LD + 2 * (e & 7);
          fix(I, J, c1[u][0], c1[u][1]);
          double d0 = 0.0, d1 = 0.0;
          if (br == bc) {              // only a diagonal block carries damping (uniform for the 128 threads of a block)
            d0 = (I == J && I < n) ? sh.dd[I] : 0.0;
            d1 = (I == J + 1 && I < n) ? sh.dd[I] : 0.0;
          }
          dst[0] = (S)(c1[u][0] + d0);
          dst[1] = (S)(c1[u][1] + d1);
          if (F32 && br == bc && I == J) s_a0[I] = (S)(c1[u][0] + d0);
          if (F32 && br == bc && I == J + 1) s_a0[I] = (S)(c1[u][1] + d1);
        }
      }
      if constexpr (Pre::U1 < Pre::U1_ALL) {
        // the blocks the prefetch registers did not cover (more than 11 block rows): two doubles per thread and round, four rounds in flight
        const int first = U1 * TREM, last = (nblk - nb) * 128;
        for (int e0 = first + (tid - 64); e0 < last; e0 += 4 * TREM) {
          double v[4][2];
          int rcq[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int e = min(e0 + q * TREM, last - 1);
            const int k = e >> 7;
            int cc = 1, base = 0;
            while (k >= base + (nb - cc) && cc < nb) { base += nb - cc; ++cc; }
            rcq[q] = ((cc + (k - base)) << 8) | cc;
            const int I = (cc + (k - base)) * CB + ((e >> 3) & 15), J = cc * CB + 2 * (e & 7);
            const double* qp = E + (size_t)min(I, n - 1) * n + min(J, n - 1);
            v[q][0] = qp[0]; v[q][1] = qp[(J + 1 < n && I < n) ? 1 : 0];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int e = e0 + q * TREM;
            if (e < last) {
              const int br = rcq[q] >> 8, bc = rcq[q] & 255;
              const int ii = (e >> 3) & 15, I = br * CB + ii, J = bc * CB + 2 * (e & 7);
              S* dst = Lb + L::off(br, bc) + ii * L::LD + 2 * (e & 7);
              fix(I, J, v[q][0], v[q][1]);
              double d0 = 0.0, d1 = 0.0;
              if (br == bc) { d0 = (I == J && I < n) ? sh.dd[I] : 0.0; d1 = (I == J + 1 && I < n) ? sh.dd[I] : 0.0; }
              dst[0] = (S)(v[q][0] + d0);
              dst[1] = (S)(v[q][1] + d1);
              if (F32 && br == bc && I == J) s_a0[I] = (S)(v[q][0] + d0);
              if (F32 && br == bc && I == J + 1) s_a0[I] = (S)(v[q][1] + d1);
            }
          }
        }
      }
    }
  } else {
    // second attempt: everything once more from E (L2 by now), no overlap with the first tile
    for (int e = tid; e < nblk * CB * CB; e += CHOLB_LDS_THREADS) {
      const int k = e >> 8, rc = sh.rc[k], br = rc >> 8, bc = rc & 255;
      const int ii = (e >> 4) & 15, jj = e & 15, I = br * CB + ii, J = bc * CB + jj;
      double v = (I < n && J < n) ? E[(size_t)I * n + J] : ((I == J) ? 1.0 : 0.0);
      if (I == J && I < n) v += sh.dd[I];
      Lb[L::off(br, bc) + ii * L::LD + jj] = (S)v;
      if (F32 && I == J) s_a0[I] = (S)v;
    }
    __syncthreads();
    CHOL_STAMP();
  }
  if (wid == 0) {
    if (!chol16_wave_t<S, NEWTON, L>(Lb + L::off(0, 0), F32 ? s_a0 : nullptr, tau)) { if (lane == 0) sh.fail = 1; }
  }
  __syncthreads();
  CHOL_STAMP();

  constexpr int NW = CHOLB_LDS_THREADS / 64;
  for (int jb = 0; jb < nb && !sh.fail; ++jb) {
    const int m = (nb - jb - 1) * CB;              // rows below the diagonal block
    const S* LinvT = Lb + L::off(jb, jb);
    // ---- B: L21 = A21 Linv^T block by block; wave 0 takes the block right below the diagonal and downdates the next
    //      diagonal tile with it, the last wave forward-solves the rhs block
    if (wid == 0) {
      if (jb + 1 < nb) chol_panel_update_diag_t<S, L>(Lb + L::off(jb + 1, jb), LinvT, Lb + L::off(jb + 1, jb + 1));
    } else {
      for (int r = jb + 2 + (wid - 1); r < nb; r += NW - 1) chol_panel_block_t<S, L>(Lb + L::off(r, jb), LinvT);
      if (wid == NW - 1) {
        // y_blk = Linv rhs_blk :  y[i] = sum_k LinvT[k][i] rhs[k]   (lane = (part, i): 4 terms each, then 2 swaps)
        const int i = lane & 15, part = lane >> 4;
        S x = 0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { const int k = part + 4 * kk; x += LinvT[k * L::LD + i] * s_y[jb * CB + k]; }
        x = CholNum<S>::xrow(x);
        __builtin_amdgcn_wave_barrier();
        if (part == 0) s_y[jb * CB + i] = x;
      }
    }
    __syncthreads();
    CHOL_STAMP();
    // ---- C: look-ahead factorisation of the next diagonal tile (wave 0) | rhs tail + trailing update (waves 1..7)
    if (wid == 0) {
      if (jb + 1 < nb) {
        if (!chol16_wave_t<S, NEWTON, L>(Lb + L::off(jb + 1, jb + 1), F32 ? s_a0 + (jb + 1) * CB : nullptr, tau)) { if (lane == 0) sh.fail = 1; }
      }
    } else {
      // The wave that shares its SIMD with wave 0 (wave 4: waves go round the four SIMDs, tools/micro/hw_id.hip) takes the light
      // part, the right-hand side's tail, and stays out of the update: an f64 MFMA holds the SIMD's issue for 64 cycles and the pivot
      // chain (~30 dependent f64 VALU instructions per pivot) gets three instructions in per MFMA (tools/micro/rate_f64.hip).
      if (wid == 4) {
        // rhs tail: y_i -= sum_k L21[i][k] y_blk[k]
        for (int t = lane; t < m; t += 64) {
          const S* row = Lb + L::off(jb + 1 + (t >> 4), jb) + (t & 15) * L::LD;
          S s0 = 0, s1 = 0;
#pragma unroll
          for (int k = 0; k < CB; k += 2) { s0 += row[k] * s_y[jb * CB + k]; s1 += row[k + 1] * s_y[jb * CB + k + 1]; }
          s_y[(jb + 1) * CB + t] -= s0 + s1;
        }
      } else {
        // tiles (r,c), jb < c <= r < nb, except (jb+1,jb+1): the trailing block triangle in row-major order from index 1, a CONTIGUOUS
        // run per worker: consecutive tiles share their row, so the row block's fragment (r, jb) stays in registers, the target and
        // the column block's fragment advance by constant strides, and a tile costs 8 LDS reads, 4 MFMAs, 4 subtractions and 4 LDS
        // writes.  (Measured, n = 176, f64, sum of the ten look-ahead phases: table-driven tiles dealt round robin to seven workers 49.0k
        //  cycles; six workers taking two tiles at a time with interleaved MFMA chains 49.7k; these strips 48.3k; sixteen waves 49k.
        //  A tile costs a wave ~850 cycles in every variant and a SIMD ~450: docs/EXPERIMENTS.md.)
        constexpr int NWORK = NW - 2;
        const int wk = wid - 1 - (wid > 4 ? 1 : 0);
        const int q = nb - jb - 1;
        const int ntile = q * (q + 1) / 2;
        const int per = (ntile - 1 + NWORK - 1) / NWORK;
        int t = 1 + wk * per;
        const int t_hi = min(ntile, t + per);
        if (t < t_hi) {
          const int rc = sh.rc[t];
          int r = jb + 1 + (rc >> 8), c = jb + 1 + (rc & 255);
          const int po = (lane & 15) * L::LD + (lane >> 4);              // operand pattern: [row lane & 15][k = 4 ks + (lane >> 4)]
          const int to = Mfma<S>::row_of(lane, 0) * L::LD + (lane & 15);   // accumulator pattern: [row_of(lane, reg)][col lane & 15]
          constexpr int trs = F32 ? L::LD : 4 * L::LD;                      // ... whose rows are `trs` apart from register to register
          while (t < t_hi) {
            const S* Ap = Lb + L::off(r, jb) + po;
            S a[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[ks] = Ap[4 * ks];
            S* D = Lb + L::off(r, c) + to;
            const S* Bp = Lb + L::off(c, jb) + po;
            for (; c <= r && t < t_hi; ++c, ++t) {
              S bq[4];
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) bq[ks] = Bp[4 * ks];
              typename Mfma<S>::acc_t acc, prod = {0, 0, 0, 0};
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) acc[rg] = D[rg * trs];
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) prod = Mfma<S>::mma(a[ks], bq[ks], prod);
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) D[rg * trs] = acc[rg] - prod[rg];
              D += L::BS;
              Bp += (c + 1) * L::BS;                                      // L::off(c + 1, jb) - L::off(c, jb)
            }
            ++r;
            c = jb + 1;
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    if (dbg && jb == 0 && lane == 0) dbg[40 + wid] = clock64();      // diagnostic: when each wave finished its share of C0
    __syncthreads();
    CHOL_STAMP();
  }
  const bool fail = sh.fail != 0;
  // ---- back substitution, right-looking:  x_b = Linv_b^T y_b  (wave 0), then every earlier row t < 16 b takes
  //      y_t -= sum_i L[16b+i][t] x_b[i]  (all waves).  The diagonal blocks hold Linv^T.
  if (!fail) {
    for (int b = nb - 1; b >= 0; --b) {
      if (wid == 0) {
        const int j = lane & 15, part = lane >> 4;
        const S* LiT = Lb + L::off(b, b) + j * L::LD;    // row j of Linv^T
        S x = 0;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) { const int i = part + 4 * ii; x += LiT[i] * s_y[b * CB + i]; }
        x = CholNum<S>::xrow(x);
        __builtin_amdgcn_wave_barrier();
        if (part == 0) s_y[b * CB + j] = x;
      }
      __syncthreads();
      if (tid < b * CB) {
        const S* col = Lb + L::off(b, tid >> 4) + (tid & 15);
        S s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < CB; i += 2) { s0 += col[i * L::LD] * s_y[b * CB + i]; s1 += col[(i + 1) * L::LD] * s_y[b * CB + i + 1]; }
        s_y[tid] -= s0 + s1;
      }
      __syncthreads();
    }
    if (tid < n16) sh.x[tid] = (double)s_y[tid];
  }
  __syncthreads();
  CHOL_STAMP();
  return fail;
#undef CHOL_STAMP
}

// use_f32 (fp32 engine only): 0 = the f64 factorisation, as every round before; 1 = factor on f32 lanes first and repeat in f64 when that
// is refused (a non-positive or non-finite pivot, or a pivot below tau32 times its diagonal entry).  In the LM loop a refused f32
// factorisation is almost always followed by a refused f64 one (S carries the 1e-7 rounding of its f32 products either way,
// tools/chol_f32_model.py), so the repeat costs little; for a caller that hands sba_lm_solve_trial an arbitrary system it is what
// keeps the answer right (tests/test_gpu_cholesky.py: condition 1e9).
template <typename T, int MAXNB = CHOLB_MAX_NB, int LD32 = 20>
__global__ __launch_bounds__(CHOLB_LDS_THREADS) void k_cholesky_blocked(
    const double* __restrict__ E /* summed exchange buffer [S | rhs | diagU | gc | cost] */, int C,
    LMState* __restrict__ st, double* __restrict__ D2c, const ParamSets<T> ps,
    double* __restrict__ delta_c, int n_sys /* size of the system in E: 11*C, or fewer when cameras share parameters */,
    const int32_t* __restrict__ tie /* [11*C] camera parameter -> system row, or NULL = identity */,
    const int32_t* __restrict__ first /* [n_sys] system row -> one camera parameter mapped to it, or NULL */,
    long long* __restrict__ dbg /* optional cycle stamps (diagnostic runs only) */, int use_f32, float tau32) {
  extern __shared__ __align__(16) unsigned char smem[];
  // fp32 engine, f64 factorisation: the pivot 1/sqrt(a_kk) is the hardware estimate (5e-8 relative) without the Newton step.  L_kk = a_kk piv
  // and the column scaled by the same piv factor a matrix whose row/column k differ from A's by 1e-7 relative -- the size
  // of the rounding error every entry of S already carries there -- and four dependent f64 operations leave each of the
  // 176 links of the pivot chain.
  constexpr bool PIV_NEWTON = !std::is_same<T, float>::value;
  // The first block column of the system, the right-hand side and the scalings are requested BEFORE the state record is looked at:
  // they depend on kernel arguments only, and the record's round trip (the workgroup's first, ~4.7k cycles) would otherwise sit
  // in front of them.  (A finished solve returns below with these loads in flight: harmless.)
  const int n = n_sys;           // rows of the system being solved
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const double* rhs = E + (size_t)n * n;
  const double* dU = rhs + n;
  const double* gct = dU + n;
  const bool pair_ok = (n & 1) == 0;
  using Pre = CholbPre<MAXNB>;
  using L32 = CholLay<float, LD32>;
  constexpr int BPR = Pre::BPR, U0 = Pre::U0, TREM = Pre::TREM, U1 = Pre::U1;
  double c0[U0][2];
  const int nlast = n - 1;
  auto addr = [&](int I, int J) { return E + (size_t)min(I, nlast) * n + min(J, nlast - 1 + (pair_ok ? 0 : 1)); };
  const int s4 = tid >> 7, ii0 = (tid >> 3) & 15, jp0 = tid & 7;
  const int tclamp = min(tid, nlast);
  if (pair_ok) {
#pragma unroll
    for (int u = 0; u < U0; ++u) {
      const double2 t = *reinterpret_cast<const double2*>(addr((BPR * u + s4) * CB + ii0, 2 * jp0));
      c0[u][0] = t.x; c0[u][1] = t.y;
    }
  } else {
#pragma unroll
    for (int u = 0; u < U0; ++u) {
      const double* q = addr((BPR * u + s4) * CB + ii0, 2 * jp0);
      c0[u][0] = q[0]; c0[u][1] = q[(2 * jp0 + 1 < n) ? 1 : 0];
    }
  }
  double in_d = D2c[tclamp], in_u = dU[tclamp], in_r = rhs[tclamp];
  const double my_g = (tid < n) ? gct[tid] : 0.0;
  // ... and so are the other 55 blocks, column by column (waves 1..7; the factorisation of the first tile does not wait for them).
  // Block k of that list sits in column cc, row cc + (k - base); k grows with the load round, so the search continues where it was.
  double c1[U1][2];
  int rc1[U1];                     // (row << 8) | column of the round's block: the LDS stores use it without a table lookup
  {
    const int nb_ = (n + CB - 1) / CB;
    const int last = (nb_ * (nb_ + 1) / 2 - nb_) * 128 - 1;
    int cc = 1, base = 0;
#pragma unroll
    for (int u = 0; u < U1; ++u) {
      const int e = min(max(tid - 64, 0) + TREM * u, last);
      const int k = e >> 7;
      while (k >= base + (nb_ - cc) && cc < nb_) { base += nb_ - cc; ++cc; }
      rc1[u] = ((cc + (k - base)) << 8) | cc;
      const int I = (cc + (k - base)) * CB + ((e >> 3) & 15), J = cc * CB + 2 * (e & 7);
      if (pair_ok) {
        const double2 t = *reinterpret_cast<const double2*>(addr(I, J));
        c1[u][0] = t.x; c1[u][1] = t.y;
      } else {
        const double* q = addr(I, J);
        c1[u][0] = q[0]; c1[u][1] = q[(J + 1 < n) ? 1 : 0];
      }
    }
  }
  if (st->status >= 0) return;
  const int cur_ = ps_cur(ps, st);
  const double* __restrict__ cams = ps.cams[cur_];
  double* __restrict__ cams_new = ps.cams[cur_ ^ 1];
  T* __restrict__ campre_new = ps.campre[cur_ ^ 1];
  int nstamp = 0;
#define CHOL_STAMP() do { if (dbg && threadIdx.x == 0) dbg[nstamp] = clock64(); ++nstamp; } while (0)
  CHOL_STAMP();
  const int ncam = C * NCP;      // camera parameters
  const int nb = (n + CB - 1) / CB;
  const int n16 = nb * CB;
  const int nblk = nb * (nb + 1) / 2;
  __shared__ CholbShared<MAXNB> sh;
  __shared__ double s_scr[4][CHOLB_LDS_THREADS / 64];
  const double lam = st->lam;
  const bool fresh = st->fresh != 0;
  // operands of the epilogue, requested now so that their latency is hidden behind the factorisation
  const double my_cam = (tid < ncam) ? cams[tid] : 0.0;                       // camera parameter tid
  const double my_xs = (tid < n) ? cams[first ? first[tid] : tid] : 0.0;      // system unknown tid (a shared one counts once)

  if (tid < nblk) {
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= tid) ++r;
    sh.rc[tid] = (short)((r << 8) | (tid - r * (r + 1) / 2));
  }
  if (tid == 0) { sh.fail = 0; st->cost = E[(size_t)n * n + 3 * n]; }
  // ---- the lower block triangle.  E was written by other CUs, so every access is a fabric round trip and one CU
  // sustains few of them: two doubles per lane per load (16 B, E rows are 16-byte aligned when n is even), all loads
  // of a thread issued before any is used, the first block column first -- its 11 blocks are all the first
  // factorisation step needs, so wave 0 starts on it while waves 1..7 are still receiving the other 55 blocks.
  // The padded tail is the identity.
  // All loads are unconditional (out-of-range ones are clamped to a valid address and replaced afterwards) and sit in
  // straight-line code: only then can the compiler wait with vmcnt(N) for the first block column alone.
  // camera scaling: monotone max of the squared column norms (x_scale='jac', scipy trf.py:424,545)
  if (tid < n16) {                                 // n16 <= 176 < CHOLB_LDS_THREADS
    double dd = 0;
    if (tid >= n) in_r = 0;
    if (tid < n) {
      double d = in_d;
      if (fresh) { d = fmax(d, in_u); D2c[tid] = d; }
      dd = lam * fmax_pos(d);
    }
    sh.x[tid] = in_r;
    sh.dd[tid] = dd;
  }
  __syncthreads();
  bool fail;
  if constexpr (MAXNB > CHOLB_MAX_NB) {
    // more than 176 unknowns: only the f32 triangle fits the LDS.  A refused factorisation raises LMState::chol_retry and leaves the
    // step to the f64 kernel launched behind this one (k_cholesky_ll with only_if_retry), which otherwise returns at once.
    static_assert(std::is_same<T, float>::value, "the 16-block-row kernel exists for the fp32 engine only");
    fail = cholb_core<float, false, true, MAXNB, L32>(smem, sh, E, n, c0, c1, rc1, tau32, dbg, nstamp);
    if (tid == 0) { st->chol_retry = fail ? 1 : 0; if (fail) st->chol_f64_retries += 1; }
    if (fail) return;                     // (block-uniform) nothing of the step has been written
  } else if constexpr (std::is_same<T, float>::value) {
    if (use_f32) {
      fail = cholb_core<float, false, true, MAXNB, L32>(smem, sh, E, n, c0, c1, rc1, tau32, dbg, nstamp);
      if (fail) {                                 // refused: once more in f64, from E (sh.x still holds the right-hand side)
        __syncthreads();
        if (tid == 0) sh.fail = 0;
        __syncthreads();
        fail = cholb_core<double, true, false, MAXNB>(smem, sh, E, n, c0, c1, rc1, 0.0, dbg, nstamp);      // (refined pivots: this pass is about the answer, not the time)
        if (tid == 0) atomicAdd(&st->chol_f64_retries, 1);
      }
    } else {
      fail = cholb_core<double, PIV_NEWTON, true, MAXNB>(smem, sh, E, n, c0, c1, rc1, 0.0, dbg, nstamp);
    }
  } else {
    fail = cholb_core<double, PIV_NEWTON, true, MAXNB>(smem, sh, E, n, c0, c1, rc1, 0.0, dbg, nstamp);
  }
  double pred = 0, dx2 = 0, x2 = 0, gm = 0;
  __shared__ double s_cnew[CHOLB_LDS_THREADS]; // the trial cameras once more in LDS (one per thread: with shared intrinsics there are more camera
                                               // parameters than system rows -- 24 cameras tie to 195 unknowns but keep 264 parameters): the CamPre rebuild below reads them
  if (tid < ncam) {                                  // from there instead of waiting for its own global stores to come back
    const double d = fail ? 0.0 : sh.x[tie ? tie[tid] : tid];
    delta_c[tid] = d;
    cams_new[tid] = my_cam + d;
    s_cnew[tid] = my_cam + d;
  }
  if (tid < n) {                                     // scalars of the step live in the system's own unknowns
    const double d = fail ? 0.0 : sh.x[tid];
    pred = 0.5 * d * (sh.dd[tid] * d - my_g);
    dx2 = d * d;
    x2 = my_xs * my_xs;
    gm = fabs(my_g);
  }
  pred = wave_sum(pred); dx2 = wave_sum(dx2); x2 = wave_sum(x2); gm = wave_max(gm);
  if (lane == 0) { s_scr[0][wid] = pred; s_scr[1][wid] = dx2; s_scr[2][wid] = x2; s_scr[3][wid] = gm; }
  __syncthreads();      // s_cnew and the wave partials are in LDS
  if (tid == 0) {
    double p = 0, d2 = 0, xx = 0, g = 0;
    for (int w = 0; w < CHOLB_LDS_THREADS / 64; ++w) { p += s_scr[0][w]; d2 += s_scr[1][w]; xx += s_scr[2][w]; g = fmax(g, s_scr[3][w]); }
    st->pred_c = p; st->dx2_c = d2; st->x2_c = xx; st->gmax_c = g;
    st->chol_fail = fail ? 1 : 0;
    st->fresh = 0;
  }
  if (tid >= 64 && tid < 64 + C)          // one camera per lane of wave 1 (wave 0 is busy writing the state record)
    campre_build<T>(s_cnew + (size_t)(tid - 64) * NCP, campre_new + (size_t)(tid - 64) * CAMPRE);
  CHOL_STAMP();
#undef CHOL_STAMP
}

// ------------------------------------------------------------------ streamed Cholesky for 176 < n <= 512
// The same 16x16 building blocks, left-looking, for systems whose block triangle does not fit the LDS (17 .. 46
// cameras): only the current block column (panel) lives in LDS; the finished columns are written to a global block
// workspace W (L2-resident: one workgroup wrote them) and streamed back, as many at a time as the staging buffer
// holds, to downdate the next panel.  A = damped matrix prepared by k_chol_prepare (full, symmetric, row-major),
// sol = rhs in / solution out, so the kernel sits between the same prepare / epilogue kernels as the library path.
constexpr int CS_MAX_NB = 32;
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt (its release fence cannot tell loads
// from stores on gfx9), which would expose the latency of the global loads deliberately left in flight across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ inline size_t cs_blk(int r, int c) { return ((size_t)r * (r + 1) / 2 + c) * (CB * CB); }

__global__ __launch_bounds__(CHOLB_THREADS) void k_cholesky_stream(
    const double* __restrict__ A, int n, double* __restrict__ W /* nb(nb+1)/2 blocks of 16x16 */,
    double* __restrict__ sol, int* __restrict__ info, const LMState* __restrict__ st,
    int scap /* blocks the staging buffer holds (>= nb): the launch gives it whatever LDS is left */,
    long long* __restrict__ dbg /* optional: cycles per phase [update, factor, solve, write-back, back substitution] */) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (st->status >= 0) return;
  const int nb = (n + CB - 1) / CB, n16 = nb * CB;
  double* P = reinterpret_cast<double*>(smem);          // [nb][CBS] panel: block i <-> block row j + i
  double* S = P + (size_t)nb * CBS;                     // [scap][CBS] staging of finished columns
  double* s_y = S + (size_t)scap * CBS;                 // [n16] rhs -> y -> x
  __shared__ int s_fail;
  __shared__ double s_part[32][CB];
  __shared__ int s_src[72];                              // source block offsets of a staging round (scap <= 64)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int NW = CHOLB_THREADS / 64;
  for (int i = tid; i < n16; i += CHOLB_THREADS) s_y[i] = (i < n) ? sol[i] : 0.0;
  if (tid == 0) s_fail = 0;
  __syncthreads();
  // the next panel's entries of A (and, on the way back, the next column of L) are requested one step ahead into
  // registers, so the fabric / L2 latency of a panel sits behind the factorisation of the previous one
  constexpr int MAXPRE = CS_MAX_NB * CB * CB / CHOLB_THREADS;     // 16 values per thread at 32 block rows
  double pre[MAXPRE];
  auto fetch_panel = [&](int jc) {
    const int mm = nb - jc;
#pragma unroll
    for (int u = 0; u < MAXPRE; ++u) {
      const int e = tid + u * CHOLB_THREADS;
      pre[u] = 0;
      if (e < mm * CB * CB) {
        const int i = e >> 8, ii = (e >> 4) & 15, jj = e & 15;
        const int I = (jc + i) * CB + ii, J = jc * CB + jj;
        pre[u] = (I < n && J < n) ? A[(size_t)I * n + J] : ((I == J) ? 1.0 : 0.0);     // padded tail = identity
      }
    }
  };
  auto fetch_column = [&](int bc) {
    const int mm = nb - bc;
#pragma unroll
    for (int u = 0; u < MAXPRE; ++u) {
      const int e = tid + u * CHOLB_THREADS;
      pre[u] = 0;
      if (e < mm * CB * CB) pre[u] = W[cs_blk(bc + (e >> 8), bc) + (e & 255)];
    }
  };
  auto commit = [&](double* dst, int mm) {
#pragma unroll
    for (int u = 0; u < MAXPRE; ++u) {
      const int e = tid + u * CHOLB_THREADS;
      if (e < mm * CB * CB) dst[(e >> 8) * CBS + ((e >> 4) & 15) * CLD + (e & 15)] = pre[u];
    }
  };
  long long ph[5] = {0, 0, 0, 0, 0}, t_prev = clock64();
  auto lap = [&](int k) { const long long t = clock64(); ph[k] += t - t_prev; t_prev = t; };
  fetch_panel(0);
  for (int j = 0; j < nb && !s_fail; ++j) {
    const int m = nb - j;
    // (1) the panel: blocks (j+i, j) of A, requested during the previous step
    commit(P, m);
    // (2) downdate with the finished block columns, g of them per staging round
    const int gmax = max(1, scap / m);
    for (int k0 = 0; k0 < j; k0 += gmax) {
      const int g = min(gmax, j - k0);
      __syncthreads();
      if (tid < g * m) { const int kk = tid / m, i = tid - kk * m; s_src[tid] = (int)cs_blk(j + i, k0 + kk); }
      __syncthreads();
      {
        // eight loads in flight per thread before the first LDS store: one L2 round trip per batch, not per element
        const int tot = g * m * CB * CB;
        for (int e0 = tid; e0 < tot; e0 += 8 * CHOLB_THREADS) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * CHOLB_THREADS;
            v[u] = (e < tot) ? W[s_src[e >> 8] + (e & 255)] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * CHOLB_THREADS;
            if (e < tot) S[(e >> 8) * CBS + ((e >> 4) & 15) * CLD + (e & 15)] = v[u];
          }
        }
      }
      __syncthreads();
      for (int i = wid; i < m; i += NW) {
        double* Dt = P + i * CBS;
        Mfma<double>::acc_t acc;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) acc[rg] = Dt[((lane >> 4) + 4 * rg) * CLD + (lane & 15)];
        // two independent accumulation chains (even / odd staged columns): the operand reads of one pair are in
        // flight while the other pair's MFMAs issue, and the dependent-MFMA latency is halved
        Mfma<double>::acc_t acc1 = {0, 0, 0, 0};
        const int po = (lane & 15) * CLD + (lane >> 4);
        int kk = 0;
        for (; kk + 1 < g; kk += 2) {
          const double* Pa0 = S + (kk * m + i) * CBS + po;
          const double* Pb0 = S + (kk * m) * CBS + po;
          const double* Pa1 = S + ((kk + 1) * m + i) * CBS + po;
          const double* Pb1 = S + ((kk + 1) * m) * CBS + po;
          double a0[4], b0[4], a1[4], b1[4];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) { a0[ks] = Pa0[4 * ks]; b0[ks] = Pb0[4 * ks]; a1[ks] = Pa1[4 * ks]; b1[ks] = Pb1[4 * ks]; }
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            acc = Mfma<double>::mma(-a0[ks], b0[ks], acc);
            acc1 = Mfma<double>::mma(-a1[ks], b1[ks], acc1);
          }
        }
        if (kk < g) {
          const double* Pa = S + (kk * m + i) * CBS + po;
          const double* Pb = S + (kk * m) * CBS + po;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) acc = Mfma<double>::mma(-Pa[4 * ks], Pb[4 * ks], acc);
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) acc[rg] += acc1[rg];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) Dt[((lane >> 4) + 4 * rg) * CLD + (lane & 15)] = acc[rg];
      }
    }
    __syncthreads();
    lap(0);
    // behind the staging loads in program order (vmcnt is in-order), ahead of factor + solve + write-back
    if (j + 1 < nb) fetch_panel(j + 1);
    // (3) diagonal tile -> Linv^T
    if (wid == 0) {
      if (!chol16_wave(P)) { if (lane == 0) s_fail = 1; }
    }
    lds_barrier();
    lap(1);
    if (s_fail) break;
    // (4) panel solve (MFMA) and forward substitution of the rhs block
    for (int i = 1 + wid; i < m; i += NW) chol_panel_block(P + i * CBS, P);
    if (wid == NW - 1) {       // the wave with the fewest panel blocks forward-solves the rhs block: y = Linv rhs
      const int i = lane & 15, part = lane >> 4;
      double x = 0;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { const int k = part + 4 * kk; x += P[k * CLD + i] * s_y[j * CB + k]; }
      x = xrow_sum(x);
      __builtin_amdgcn_wave_barrier();
      if (part == 0) s_y[j * CB + i] = x;
    }
    lds_barrier();
    lap(2);
    // rhs tail and write-back of the finished column
    for (int t = tid; t < (m - 1) * CB; t += CHOLB_THREADS) {
      const double* row = P + (1 + (t >> 4)) * CBS + (t & 15) * CLD;
      double s0 = 0, s1 = 0;
#pragma unroll
      for (int k = 0; k < CB; k += 2) { s0 += row[k] * s_y[j * CB + k]; s1 += row[k + 1] * s_y[j * CB + k + 1]; }
      s_y[(j + 1) * CB + t] -= s0 + s1;
    }
    for (int e = tid; e < m * CB * CB; e += CHOLB_THREADS) {
      const int i = e >> 8, ii = (e >> 4) & 15, jj = e & 15;
      W[cs_blk(j + i, j) + ii * CB + jj] = P[i * CBS + ii * CLD + jj];
    }
    // read back only by this workgroup: the barrier's workgroup-scope release is all the ordering it needs
    __syncthreads();
    lap(3);
  }
  __syncthreads();
  const bool fail = s_fail != 0;
  // back substitution, left-looking: x_b = Linv_b^T ( y_b - sum_{r>b} L(r,b)^T x_r ), column b streamed from W
  if (!fail) {
    fetch_column(nb - 1);
    for (int b = nb - 1; b >= 0; --b) {
      const int m = nb - b;
      commit(S, m);
      __syncthreads();
      if (b > 0) fetch_column(b - 1);
      {
        const int jcol = tid & 15, part = tid >> 4;          // 32 parts over the (m-1)*16 rows below the diagonal block
        double s0 = 0;
        for (int rr = part; rr < (m - 1) * CB; rr += 32)
          s0 += S[(1 + (rr >> 4)) * CBS + (rr & 15) * CLD + jcol] * s_y[(b + 1) * CB + rr];
        s_part[part][jcol] = s0;
      }
      __syncthreads();
      if (wid == 0) {
        const int jcol = lane & 15, part = lane >> 4;
        if (lane < CB) {
          double v = 0;
#pragma unroll
          for (int q = 0; q < 32; ++q) v += s_part[q][lane];
          s_y[b * CB + lane] -= v;
        }
        __builtin_amdgcn_wave_barrier();
        const double* LiT = S + jcol * CLD;                  // row jcol of Linv^T (block 0 of the staged column)
        double x = 0;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) { const int i = part + 4 * ii; x += LiT[i] * s_y[b * CB + i]; }
        x = xrow_sum(x);
        __builtin_amdgcn_wave_barrier();
        if (part == 0) s_y[b * CB + jcol] = x;
      }
      __syncthreads();
    }
  }
  lap(4);
  if (dbg && tid == 0) { for (int k = 0; k < 5; ++k) dbg[k] = ph[k]; }
  for (int i = tid; i < n; i += CHOLB_THREADS) sol[i] = fail ? 0.0 : s_y[i];
  if (tid == 0) *info = fail ? 1 : 0;
}

}  // namespace SBA_NS
