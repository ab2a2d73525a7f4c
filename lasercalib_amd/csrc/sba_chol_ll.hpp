// sba_chol_ll.hpp -- Cholesky + solve of the reduced camera system for n <= 256 (up to 23 cameras of 11 parameters), one
// workgroup, the factor kept on chip.
//
//   A = S + lam*diag(D2c)  (n16 = 16*nb rows, nb <= 16, padded with an identity tail);  A = L L^T ;  delta_c = A^-1 rhs
//
// k_cholesky_blocked (right-looking) keeps the WHOLE lower block triangle in LDS and therefore stops at nb = 11 (176 rows).
// This kernel is left-looking with a look-ahead of one block column, and the blocks that are still being updated never
// touch LDS at all:
//   * every sub-diagonal block (i, c) belongs to ONE wave (its row's owner, waves 1..7) for its whole life and lives in
//     that wave's f64-MFMA accumulator registers -- TRANSPOSED (the registers hold A(c, i), read from the upper triangle
//     of E), because then
//         - updates          G -= L(c,k) L(i,k)^T      take both operands in the plain row-major operand pattern,
//         - the panel solve  L(i,c)^T = Linv_c G       takes G straight from the accumulator registers as its B operand,
//         - and its result, again in accumulator layout, IS the operand pattern of L(i,c) for the next update;
//     a wave holds at most two rows x (current column, next column) = four blocks = 32 VGPRs;
//   * LDS holds finished L blocks only, and only the rows that later columns still need: at step j rows > j, columns <= j,
//     at most 64 blocks for nb = 16.  A tiny slot allocator (one lane) hands out 2176-byte slots and takes the slots of
//     finished rows back when it runs short; every L block and every Linv^T also goes to a global workspace (written and
//     read by this workgroup only: L2-resident), from where the back substitution fetches what is no longer in LDS, one
//     row ahead of its use.  Up to nb = 12 (17 cameras) nothing is ever evicted.
//   * wave 0 runs the serial chain: panel product of the block right below the diagonal (formed transposed, see
//     chol_panel_update_diag) -> downdate of the next diagonal tile -> chol16_wave, while the other waves apply
//         (a) the last update (k = j) to column j+1,   (b) all updates k <= j to column j+2,
//     so the look-ahead work of a step is (nb-j-2)(j+2) block products spread over 7 waves instead of the (nb-j-1)^2/2 of
//     the right-looking form, whose first steps were three times longer than the pivot chain they run beside.
// Per block column: barrier B1 after the panel solves, barrier B2 after the factorisation / updates.
#pragma once
#include "sba_chol_blocked.hpp"

namespace SBA_NS {

constexpr int CLL_MAX_NB = 16;
constexpr int CLL_THREADS = 512;
constexpr int CLL_NSLOT = 67;                       // L block slots (64 live at most + headroom)
constexpr int CLL_NBLK = CLL_NSLOT + 4;             // + Linv^T of the current column, the sub-diagonal staging block, two diagonal staging blocks
constexpr size_t CLL_LDS_BYTES = ((size_t)CLL_NBLK * CBS + 2 * (size_t)CLL_MAX_NB * CB) * sizeof(double);
static_assert(CLL_LDS_BYTES + 2 * 1024 <= 160 * 1024, "k_cholesky_ll: dynamic + static LDS must fit 160 KB");

// chol16_wave with separate source and destination (the factorisation of the tile at src leaves Linv^T at dst)
template <bool NEWTON>
__device__ __forceinline__ bool chol16_wave_to(const double* __restrict__ src, double* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15;
  const bool ident = lane >= 16;
  double a[CB];
#pragma unroll
  for (int j = 0; j < CB; ++j) { const double v = src[i * CLD + j]; a[j] = ident ? ((j == i) ? 1.0 : 0.0) : v; }
  double dg = src[i * CLD + i];
  __builtin_amdgcn_wave_barrier();
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
    for (int j = 0; j < CB; ++j) dst[i * CLD + j] = a[j];     // row i of Linv^T (zero left of the diagonal)
  }
  return isfinite(chk) && chk > 0.0;
}

template <typename T>
__global__ __launch_bounds__(CLL_THREADS) void k_cholesky_ll(
    const double* __restrict__ E /* summed exchange buffer [S | rhs | diagU | gc | cost], S full and symmetric */, int C,
    LMState* __restrict__ st, double* __restrict__ D2c, const ParamSets<T> ps,
    double* __restrict__ delta_c, int n_sys, const int32_t* __restrict__ tie, const int32_t* __restrict__ first,
    double* __restrict__ W /* nb(nb+1)/2 blocks of 16x16 (cs_blk order): L blocks, Linv^T on the diagonal */,
    long long* __restrict__ dbg,
    int only_if_retry /* 1: run only when the f32-lane factorisation in front of this launch refused the system (LMState::chol_retry) */) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr bool PIV_NEWTON = !std::is_same<T, float>::value;
  constexpr int NW = CLL_THREADS / 64;
  using acc_t = Mfma<double>::acc_t;
  const int n = n_sys;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nb = (n + CB - 1) / CB, n16 = nb * CB;
  const double* rhs = E + (size_t)n * n;
  const double* dU = rhs + n;
  const double* gct = dU + n;
  const int nlast = n - 1;
  const int l15 = lane & 15, lq = lane >> 4;
  // ---- everything that depends on kernel arguments only is requested before the state record is looked at
  // block (br, bc) of E in accumulator layout: register rg <- E[16 br + lq + 4 rg][16 bc + l15]   (clamped, fixed up later)
  auto load_acc = [&](int br, int bc, acc_t& a) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) a[rg] = E[(size_t)min(br * CB + lq + 4 * rg, nlast) * n + min(bc * CB + l15, nlast)];
  };
  // (the padded tail is the identity: an off-diagonal block is zero there, a diagonal one has ones)
  auto fix_acc = [&](int br, int bc, acc_t& a) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int I = br * CB + lq + 4 * rg, J = bc * CB + l15;
      if (I >= n || J >= n) a[rg] = (I == J) ? 1.0 : 0.0;
    }
  };
  // Roles.  Wave 0 runs the pivot chain.  Waves 1..7 are the workers: worker x = wave - 1 owns block rows 2 + x and 9 + x (rows retire
  // from the top, so a worker's two rows are rarely active together once the first seven are done).  Wave 4 -- it shares wave 0's SIMD,
  // a workgroup's waves go to the SIMDs in cyclic order -- is also the helper: forward substitution, right-hand-side tail, slot
  // allocator, copies to the workspace.
  constexpr int NR = 2, HELPER = 4;
  const bool helper = wid == HELPER;
  const int widx = wid - 1;
  int rowi[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) rowi[r] = (widx >= 0) ? 2 + widx + 7 * r : CLL_MAX_NB + 1;
  acc_t cur[NR], nxt[NR], nx2[NR], Lr[NR];    // block columns j, j+1 (j+1, j+2 after the panel solve), the one requested two steps ahead, L(i,j)
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    cur[r] = acc_t{0, 0, 0, 0}; nxt[r] = acc_t{0, 0, 0, 0}; nx2[r] = acc_t{0, 0, 0, 0}; Lr[r] = acc_t{0, 0, 0, 0};
    if (rowi[r] < nb) { load_acc(0, rowi[r], cur[r]); load_acc(1, rowi[r], nxt[r]); if (nb > 2) load_acc(2, rowi[r], nx2[r]); }
  }
  // staging blocks D0 = A(0,0), S0 = A(1,0), D1 = A(1,1): 128 threads each, two doubles per thread
  const int sblk = tid >> 7, sii = (tid >> 3) & 15, sjp = tid & 7;
  double s0v[2] = {0, 0};
  {
    const int br = (sblk == 0) ? 0 : 1, bc = (sblk == 2) ? 1 : 0;
    if (sblk < 3) {
      const int I = min(br * CB + sii, nlast), J = bc * CB + 2 * sjp;
      s0v[0] = E[(size_t)I * n + min(J, nlast)];
      s0v[1] = E[(size_t)I * n + min(J + 1, nlast)];
    }
  }
  const int tclamp = min(tid, nlast);
  double in_d = D2c[tclamp], in_u = dU[tclamp], in_r = rhs[tclamp];
  const double my_g = (tid < n) ? gct[tid] : 0.0;
  if (st->status >= 0 || (only_if_retry && !st->chol_retry)) return;
  const int cur_ = ps_cur(ps, st);
  const double* __restrict__ cams = ps.cams[cur_];
  double* __restrict__ cams_new = ps.cams[cur_ ^ 1];
  T* __restrict__ campre_new = ps.campre[cur_ ^ 1];
  int nstamp = 0;
#define CLL_STAMP() do { if (dbg && threadIdx.x == 0 && nstamp < 60) dbg[nstamp] = clock64(); ++nstamp; } while (0)
  CLL_STAMP();
  const int ncam = C * NCP;
  double* Lb = reinterpret_cast<double*>(smem);                   // CLL_NSLOT slots
  double* s_linv = Lb + (size_t)CLL_NSLOT * CBS;                  // Linv^T of the current block column
  double* s_stS = s_linv + CBS;                                   // A'(j+1, j): the block right below the diagonal, fully updated
  double* s_stD = s_stS + CBS;                                    // [2] next diagonal tiles (parity of the block row)
  double* s_y = s_stD + 2 * CBS;                                  // [n16] rhs -> y -> x
  double* s_d = s_y + CLL_MAX_NB * CB;                            // [n16] lam * D2c
  __shared__ int s_fail, s_nfree;
  __shared__ unsigned char s_free[CLL_NSLOT];
  __shared__ unsigned char s_slot[CLL_MAX_NB][CLL_MAX_NB];       // (row, column) -> slot
  __shared__ unsigned char s_nfreed[CLL_MAX_NB];                 // leading blocks of the row whose slots were taken back (0 = row resident)
  __shared__ double s_scr[4][NW];
  double* s_cnew = Lb;                                            // [ncam] trial cameras for the CamPre rebuild (the slots are free by then)
  const double lam = st->lam;
  const bool fresh = st->fresh != 0;
  const double my_cam = (tid < ncam) ? cams[tid] : 0.0;
  const double my_xs = (tid < n) ? cams[first ? first[tid] : tid] : 0.0;
  if (tid == 0) { s_fail = 0; st->cost = E[(size_t)n * n + 3 * n]; }
  if (tid < CLL_MAX_NB) s_nfreed[tid] = 0;
  if (tid < CLL_NSLOT) s_free[tid] = (unsigned char)(CLL_NSLOT - 1 - tid);
  if (tid < n16) {
    double dd = 0;
    if (tid >= n) in_r = 0;
    if (tid < n) {
      double d = in_d;
      if (fresh) { d = fmax(d, in_u); D2c[tid] = d; }
      dd = lam * fmax_pos(d);
    }
    s_y[tid] = in_r;
    s_d[tid] = dd;
  }
  __syncthreads();
  // slots of block column 0 (rows 1 .. nb-1)
  if (tid == 0) {
    int nf = CLL_NSLOT;
    for (int i = 1; i < nb; ++i) s_slot[i][0] = s_free[--nf];
    s_nfree = nf;
  }
  if (sblk < 3) {
    const int br = (sblk == 0) ? 0 : 1, bc = (sblk == 2) ? 1 : 0;
    const int I = br * CB + sii, J = bc * CB + 2 * sjp;
    if (I >= n || J >= n) s0v[0] = (I == J) ? 1.0 : 0.0;
    if (I >= n || J + 1 >= n) s0v[1] = (I == J + 1) ? 1.0 : 0.0;
    if (br == bc) {
      if (I == J && I < n) s0v[0] += s_d[I];
      if (I == J + 1 && I < n) s0v[1] += s_d[I];
    }
    double* dst = (sblk == 0) ? s_linv : (sblk == 1) ? s_stS : s_stD + CBS;      // D1 has odd parity
    dst[sii * CLD + 2 * sjp] = s0v[0];
    dst[sii * CLD + 2 * sjp + 1] = s0v[1];
  }
#pragma unroll
  for (int r = 0; r < NR; ++r)
    if (rowi[r] < nb) { fix_acc(0, rowi[r], cur[r]); fix_acc(1, rowi[r], nxt[r]); }
  __syncthreads();
  CLL_STAMP();
  // every L block and every Linv^T also goes to the workspace (the back substitution reads from there what LDS no longer holds):
  // copied by the helper wave from the LDS image, 32 bytes per lane and store, never by the waves that produce them
  auto block_to_global = [&](const double* src /* LDS, row stride CLD */, double* dst /* global, row stride CB */) {
    const int rr = lane >> 2, c4 = (lane & 3) * 4;
    double4 v;
    v.x = src[rr * CLD + c4]; v.y = src[rr * CLD + c4 + 1]; v.z = src[rr * CLD + c4 + 2]; v.w = src[rr * CLD + c4 + 3];
    *reinterpret_cast<double4*>(dst + rr * CB + c4) = v;
  };
  if (wid == 0) {
    if (!chol16_wave_to<PIV_NEWTON>(s_linv, s_linv)) { if (lane == 0) s_fail = 1; }
  }
  lds_barrier();
  CLL_STAMP();

  const int po = l15 * CLD + lq;                                // operand pattern X[l15][lq + 4 ks] of a row-major LDS block
  for (int j = 0; j < nb && !s_fail; ++j) {
    // ------------------------------------------------------------ X_j: panel solves of block column j
    if (wid == 0) {
      if (j + 1 < nb) {
        // P^T = Linv A'(j+1,j)^T (accumulator layout = operand pattern of P), D(j+1) -= P P^T
        const double* Pa = s_stS + po;
        const double* Pl = s_linv + lq * CLD + l15;
        double* Dt = s_stD + ((j + 1) & 1) * CBS;
        acc_t d;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) d[rg] = Dt[(lq + 4 * rg) * CLD + l15];
        acc_t pt;
        {
          acc_t h[4];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) h[ks] = Mfma<double>::mma(Pl[4 * ks * CLD], Pa[4 * ks], acc_t{0, 0, 0, 0});
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) pt[rg] = (h[0][rg] + h[1][rg]) + (h[2][rg] + h[3][rg]);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) h[ks] = Mfma<double>::mma(pt[ks], pt[ks], acc_t{0, 0, 0, 0});
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) d[rg] -= (h[0][rg] + h[1][rg]) + (h[2][rg] + h[3][rg]);
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) Dt[(lq + 4 * rg) * CLD + l15] = d[rg];
        double* Ls = Lb + (size_t)s_slot[j + 1][j] * CBS;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) Ls[l15 * CLD + lq + 4 * rg] = pt[rg];
      }
    }
    if (helper) {
      // y_j = Linv_j rhs_j ; Linv_j^T to the workspace
      const int i = l15, part = lq;
      double x = 0;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { const int k = part + 4 * kk; x += s_linv[k * CLD + i] * s_y[j * CB + k]; }
      x = xrow_sum(x);
      __builtin_amdgcn_wave_barrier();
      if (part == 0) s_y[j * CB + i] = x;
      block_to_global(s_linv, W + cs_blk(j, j));
    }
    if (wid != 0) {
      const double* Pl = s_linv + lq * CLD + l15;
      // The panel solves come first and request nothing; then the register sets move up -- the copies wait for loads requested
      // a whole step ago (E was written by other CUs: every request is a fabric round trip of 2-4k cycles, so a column is
      // requested two steps before its first update) while nothing younger is in flight (vmcnt counts in order: a wait behind
      // this step's own requests would sit those out as well) --, and only then the next column is requested.
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int i = rowi[r];
        if (i >= j + 2 && i < nb) {
          // L(i,j)^T = Linv_j G with G = A'(i,j)^T in the accumulator registers (B operand as it stands)
          acc_t R;
          {
            acc_t h[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) h[ks] = Mfma<double>::mma(Pl[4 * ks * CLD], cur[r][ks], acc_t{0, 0, 0, 0});
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) R[rg] = (h[0][rg] + h[1][rg]) + (h[2][rg] + h[3][rg]);
          }
          Lr[r] = R;
          double* Ls = Lb + (size_t)s_slot[i][j] * CBS;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) Ls[l15 * CLD + lq + 4 * rg] = R[rg];
        }
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) { cur[r] = nxt[r]; nxt[r] = nx2[r]; }
    }
    if (dbg && lane == 0 && j == 3) dbg[64 + wid] = clock64();
    lds_barrier();                                                                   // B1
    CLL_STAMP();
    // ------------------------------------------------------------ Y_j: factor diagonal j+1 | updates of columns j+1, j+2
    if (wid == 0) {
      if (j + 1 < nb) {
        if (!chol16_wave_to<PIV_NEWTON>(s_stD + ((j + 1) & 1) * CBS, s_linv)) { if (lane == 0) s_fail = 1; }
      }
    }
    if (helper) {
      if (j + 1 < nb) {
        // rhs tail of block row j+1 (wave 0's panel block), from its LDS copy
        const double* row = Lb + (size_t)s_slot[j + 1][j] * CBS + l15 * CLD;
        double sacc = 0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { const int k = lq + 4 * kk; sacc += row[k] * s_y[j * CB + k]; }
        sacc = xrow_sum(sacc);
        if (lq == 0) s_y[(j + 1) * CB + l15] -= sacc;
        // slots for block column j+1 (rows j+2 .. nb-1).  Rows <= j are finished and may give theirs back, lowest row first (the
        // back substitution needs those last); of row j+1 only block (j+1, j) is still being read in this interval.  What must stay:
        // rows >= j+2 x columns <= j, block (j+1, j), the new column = (nb-j-2)(j+2) + 1 <= 65 slots at nb = 16.
        if (lane == 0) {
          int nf = s_nfree;
          const int need = nb - j - 2;
          for (int rr = 1; rr <= j + 1 && nf < need; ++rr) {
            const int lim = (rr <= j) ? rr : j;
            int f = s_nfreed[rr];
            while (nf < need && f < lim) s_free[nf++] = s_slot[rr][f++];
            s_nfreed[rr] = (unsigned char)f;
          }
          for (int i = j + 2; i < nb; ++i) s_slot[i][j + 1] = s_free[--nf];
          s_nfree = nf;
        }
        // block column j of L to the workspace (none of these slots is given back before the next step)
        for (int i = j + 1; i < nb; ++i) block_to_global(Lb + (size_t)s_slot[i][j] * CBS, W + cs_blk(i, j));
      }
    }
    if (wid != 0) {
      // block column j+3 is requested here, behind the barrier: its address arithmetic does not lengthen the panel phase every wave
      // waits for (the register sets moved up in that phase, two intervals after their loads were issued)
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int i = rowi[r];
        if (j + 3 < nb && i >= j + 3 && i < nb) load_acc(j + 3, i, nx2[r]);      // (the diagonal block when i == j+3)
      }
      double yv[4];                                             // y_j[lq + 4 ks]
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) yv[ks] = s_y[j * CB + lq + 4 * ks];
      // (a) column j+1, last update k = j:  G -= L(j+1,j) L(i,j)^T   (B operand = this wave's own panel result)
      if (j + 1 < nb) {
        const double* Pa = Lb + (size_t)s_slot[j + 1][j] * CBS + po;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int i = rowi[r];
          if (i >= j + 2 && i < nb) {
            double a4[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a4[ks] = Pa[4 * ks];
            {
              acc_t h[4];
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) h[ks] = Mfma<double>::mma(a4[ks], Lr[r][ks], acc_t{0, 0, 0, 0});
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) cur[r][rg] -= (h[0][rg] + h[1][rg]) + (h[2][rg] + h[3][rg]);
            }
            // rhs tail of this block row: y_i -= L(i,j) y_j  (the lane holds L(i,j)[l15][lq + 4 rg])
            double s = 0;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s += Lr[r][ks] * yv[ks];
            s = xrow_sum(s);
            if (lq == 0) s_y[i * CB + l15] -= s;
            if (i == j + 2) {
              // the block right below the next diagonal is complete: hand it to wave 0 (row-major A'(j+2, j+1))
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) s_stS[l15 * CLD + lq + 4 * rg] = cur[r][rg];
            }
          }
        }
      }
      // (b) column j+2, all updates k <= j:  G -= L(j+2,k) L(i,k)^T   (both operands from LDS).  FOUR accumulators per block, one per
      // k-step of the 16-deep product: an f64 MFMA that accumulates onto the result of the one right before it waits out that one's
      // whole latency (about two issue slots), so consecutive MFMAs never share an accumulator; the four are added up at the end.
      if (j + 2 < nb) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int i = rowi[r];
          if (i >= j + 2 && i < nb) {
            fix_acc(j + 2, i, nxt[r]);
            if (i == j + 2) {
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) {
                const int I = i * CB + lq + 4 * rg;
                if (lq + 4 * rg == l15 && I < n) nxt[r][rg] += s_d[I];
              }
            }
            // (the products are accumulated with their own sign and subtracted once at the end: negating an operand is four VALU
            //  instructions per product on the f64 lanes the MFMAs of the SIMD's two waves are waiting for)
            acc_t g[4] = {acc_t{0, 0, 0, 0}, acc_t{0, 0, 0, 0}, acc_t{0, 0, 0, 0}, acc_t{0, 0, 0, 0}};
            // the slots of the operand blocks: lane l looks up column l once, the loop takes them through v_readlane (scalar
            // address arithmetic instead of a dependent LDS read per step); the operands of step k+1 are requested before the
            // MFMAs of step k issue (two register sets, loop unrolled by two)
            const int sA = s_slot[j + 2][l15], sB = s_slot[i][l15];
            auto ld = [&](int k, double (&a)[4], double (&b)[4]) {
              const double* A0 = Lb + (size_t)__builtin_amdgcn_readlane(sA, k) * CBS + po;
              const double* B0 = Lb + (size_t)__builtin_amdgcn_readlane(sB, k) * CBS + po;
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) { a[ks] = A0[4 * ks]; b[ks] = B0[4 * ks]; }
            };
            auto mm = [&](const double (&a)[4], const double (&b)[4]) {
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) g[ks] = Mfma<double>::mma(a[ks], b[ks], g[ks]);
            };
            double aX[4], bX[4], aY[4], bY[4];
            ld(0, aX, bX);
            for (int k = 0; k <= j; k += 2) {
              if (k + 1 <= j) ld(k + 1, aY, bY);
              mm(aX, bX);
              if (k + 1 <= j) {
                if (k + 2 <= j) ld(k + 2, aX, bX);
                mm(aY, bY);
              }
            }
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) nxt[r][rg] -= (g[0][rg] + g[1][rg]) + (g[2][rg] + g[3][rg]);
            if (i == j + 2) {
              double* Dt = s_stD + ((j + 2) & 1) * CBS;
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) Dt[(lq + 4 * rg) * CLD + l15] = nxt[r][rg];
            }
          }
        }
      }
    }
    if (dbg && lane == 0 && j == 3) dbg[72 + wid] = clock64();
    lds_barrier();                                                                   // B2
    CLL_STAMP();
  }
  __syncthreads();                  // the helper's copies to the workspace are complete and visible to the other waves
  const bool fail = s_fail != 0;
  // ---- back substitution, right-looking over block rows:  x_b = Linv_b^T y_b, then y_t -= sum_i L[16b+i][t] x_b[i] for t < 16 b.
  //      Every Linv^T comes from the workspace and is requested NOW, all at once (four doubles per lane of wave 0 and block row;
  //      the loop over the block rows is unrolled so that they sit in registers): one round trip instead of one per row.  The row
  //      of L comes from LDS while the row is resident, else from the workspace, requested one row ahead.
  if (!fail) {
    double liv[CLL_MAX_NB][4];
    double pre[2][4];
#pragma unroll
    for (int b = 0; b < CLL_MAX_NB; ++b) {
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) liv[b][ii] = 0;
      if (wid == 0 && b < nb) {
        const double* src = W + cs_blk(b, b) + l15 * CB;             // row l15 of Linv_b^T
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) liv[b][ii] = src[lq + 4 * ii];
      }
    }
    // the update of block row b: wave w takes the blocks (b, w) and (b, w + 8); lane (l15, lq) the four terms i = lq + 4 ii of
    // output column l15, folded over lq by the row-swap sum
    auto fetch_row = [&](int b) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = wid + 8 * h;
        if (c < b) {
          const double* src = W + cs_blk(b, c) + l15;
#pragma unroll
          for (int ii = 0; ii < 4; ++ii) pre[h][ii] = src[(lq + 4 * ii) * CB];
        }
      }
    };
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) pre[h][ii] = 0;
    if (s_nfreed[nb - 1]) fetch_row(nb - 1);
#pragma unroll
    for (int bb = CLL_MAX_NB - 1; bb >= 0; --bb) {
      if (bb < nb) {
        const int b = bb;
        if (wid == 0) {
          double x = 0;
#pragma unroll
          for (int ii = 0; ii < 4; ++ii) x += liv[bb][ii] * s_y[b * CB + lq + 4 * ii];
          x = xrow_sum(x);
          __builtin_amdgcn_wave_barrier();
          if (lq == 0) s_y[b * CB + l15] = x;
        }
        if (dbg && lane == 0 && b == 5) dbg[80 + wid] = clock64();
        lds_barrier();
        if (dbg && tid == 0 && b == 5) dbg[96] = clock64();
        const bool res = s_nfreed[b] == 0;
        double xb[4];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) xb[ii] = s_y[b * CB + lq + 4 * ii];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int c = wid + 8 * h;
          if (c < b) {
            double sacc = 0;
            if (res) {
              const double* col = Lb + (size_t)s_slot[b][c] * CBS + l15;
#pragma unroll
              for (int ii = 0; ii < 4; ++ii) sacc += col[(lq + 4 * ii) * CLD] * xb[ii];
            } else {
#pragma unroll
              for (int ii = 0; ii < 4; ++ii) sacc += pre[h][ii] * xb[ii];
            }
            sacc = xrow_sum(sacc);
            if (lq == 0) s_y[c * CB + l15] -= sacc;
          }
        }
        if (b > 0 && s_nfreed[b - 1]) fetch_row(b - 1);
        if (dbg && lane == 0 && b == 5) dbg[88 + wid] = clock64();
        lds_barrier();
        if (dbg && tid == 0 && b == 5) dbg[97] = clock64();
      }
    }
  }
  __syncthreads();
  CLL_STAMP();
  double pred = 0, dx2 = 0, x2 = 0, gm = 0;
  if (tid < ncam) {
    const double d = fail ? 0.0 : s_y[tie ? tie[tid] : tid];
    delta_c[tid] = d;
    cams_new[tid] = my_cam + d;
    s_cnew[tid] = my_cam + d;
  }
  if (tid < n) {
    const double d = fail ? 0.0 : s_y[tid];
    pred = 0.5 * d * (s_d[tid] * d - my_g);
    dx2 = d * d;
    x2 = my_xs * my_xs;
    gm = fabs(my_g);
  }
  pred = wave_sum(pred); dx2 = wave_sum(dx2); x2 = wave_sum(x2); gm = wave_max(gm);
  if (lane == 0) { s_scr[0][wid] = pred; s_scr[1][wid] = dx2; s_scr[2][wid] = x2; s_scr[3][wid] = gm; }
  __syncthreads();
  if (tid == 0) {
    double p = 0, d2 = 0, xx = 0, g = 0;
    for (int w = 0; w < NW; ++w) { p += s_scr[0][w]; d2 += s_scr[1][w]; xx += s_scr[2][w]; g = fmax(g, s_scr[3][w]); }
    st->pred_c = p; st->dx2_c = d2; st->x2_c = xx; st->gmax_c = g;
    st->chol_fail = fail ? 1 : 0;
    st->fresh = 0;
  }
  if (tid >= 64 && tid < 64 + C)
    campre_build<T>(s_cnew + (size_t)(tid - 64) * NCP, campre_new + (size_t)(tid - 64) * CAMPRE);
  CLL_STAMP();
#undef CLL_STAMP
}

}  // namespace SBA_NS
