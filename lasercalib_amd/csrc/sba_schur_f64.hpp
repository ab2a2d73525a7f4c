// fp64 fused linearise + Schur kernels: k_schur_fused_f64 for one camera group (<= 16 cameras of the 11-parameter model, dense or
// masked visibility) and k_schur_fused_wide_f64 for 17 .. 23 cameras (compact rows; further down).
// Test/bench infrastructure never includes this file directly: it is part of the engine translation unit (sba_engine.hpp).
//
// What k_schur_fused_f64 replaces (round 3): the fp64 default path ran k_linearize_cams<double> (Jacobians + U_c on the f64 MFMA,
// 35 us at 16 x 50k), k_reduce_cams (5 us), k_schur_sym<double, LIN> (141 us: all eight waves build a 32-point panel, then all eight
// consume it -- profiles/r3_sq_counters_16x50k_f64.json: matrix pipe 42-48 % busy, VALU and MFMA phases alternate in lockstep) and
// k_decide (5 us).  Here the fp32 kernels' structure is used instead (k_schur_fused, sba_kernels.hpp): the previous step's LM
// decision in the prologue; four producer waves evaluate every observation's Jacobian ONCE per iteration (lane = (point of a
// 16-point chunk, camera)), accumulate U_c / g_c, reduce V_p / g_p over the DPP row, factor the damped 3x3 block and write the
// 48 x 176 panel Ytilde of the chunk into one of two LDS buffers; four consumer waves (one per SIMD, 17/17/16/16 of the 66
// upper-triangle tiles each, 8 VGPRs per tile) form panel^T panel with v_mfma_f64_16x16x4_f64 from the other buffer.  On gfx950 the
// f64 MFMA holds the SIMD's issue for 64 cycles and a VALU wave beside it gets about three instructions in per MFMA
// (tools/micro/rate_f64.hip): the time of a chunk is, to first order, the SUM of the two instruction streams (12.8k MFMA cycles +
// ~760 VALU instructions per observation = 18.6k cycles per 16 points), so what the kernel gains over its predecessors is the
// second Jacobian evaluation, three launches, and every instruction it does not issue -- and it only gains as long as it runs
// without scratch (see SchurF64Cfg::KREG).
//   LDS: 2 x (48 x 176 + 48) doubles of panel + z (135,936 B), camera table 16 x 25 doubles, folded U_c 16 x 77 doubles, the
//   accumulator sets 4 x 16 x 25 doubles: 161,792 B.  The register accumulators are handed over through the panel buffers.
//   Outputs are those of k_schur_fused in T = double: slab [121 tile slots][64 lanes][4] per workgroup, bpart, gdpart, pf, gp, D2p.
#pragma once
#include "sba_kernels.hpp"
#include "sba_schur_wide.hpp"

namespace SBA_NS {
#if SBA_NCP == 11

struct SchurF64Cfg {
  using elem = double;
  static constexpr bool diag = true;
  static constexpr int THREADS = SCHUR_THREADS;
  static constexpr int NPROD = THREADS / 2;
  static constexpr int NCW = THREADS / 128;                  // consumer waves: one per SIMD
  static constexpr int NTILE = (GROUP_TILES * (GROUP_TILES + 1)) / 2;
  static constexpr int TS = 1;
  static constexpr int NV = NCW;
  static constexpr int TPW = (NTILE + NV - 1) / NV;          // 17 tiles = 136 accumulator VGPRs
  static constexpr int PTS = 16;
  static constexpr int K = 3 * PTS;
  static constexpr int BUF = K * GROUP_ROWS + K;             // panel [K][176] + z [K], in doubles
  // U_c / g_c accumulators of a producer lane: the first KREG of the UPK = 77 live in registers, the last NL in LDS, one set per
  // producer wave and camera, added to with ds_add_f64 (the four points of a wave meet in the same slot; with all 77 in registers
  // the producers ran 180 bytes of scratch per lane, ten accumulators read-modify-written in memory every chunk)
  static constexpr int KREG = 52, NL = UPK - KREG;
  static constexpr int KS = KREG + 1;                        // hand-over stride of a lane (53 doubles = 106 words: 64 lanes, 32 even bank pairs, no conflict)
  static constexpr size_t CAM_OFF = (size_t)2 * BUF * sizeof(double);
  static constexpr size_t URED_OFF = CAM_OFF + (size_t)GROUP_CAMS * CAMPRE * sizeof(double);
  static constexpr size_t ACC_OFF = URED_OFF + (size_t)GROUP_CAMS * UPK * sizeof(double);
  static constexpr size_t LDS_BYTES = ACC_OFF + (size_t)(NPROD / 64) * GROUP_CAMS * NL * sizeof(double);
};
static_assert((size_t)SchurF64Cfg::NPROD * SchurF64Cfg::KS * sizeof(double) <= SchurF64Cfg::CAM_OFF, "the hand-over area is the two panel buffers");
static_assert(SchurF64Cfg::LDS_BYTES + 1024 <= 160 * 1024, "LDS budget of a CU (static record, log row and scratch on top)");

// L^-1 of the 3x3 SPD matrix (v00,v01,v02,v11,v12,v22) in double without sqrt / divide sequences: v_rsq_f64 (5e-8, tools/micro/
// rsq_acc.hip) + two Newton steps (4e-15, then rounding level); straight-line like chol3_inv_fast(float): a failing pivot is
// replaced by 1 and the verdict is one flag at the end
__device__ __forceinline__ double rsqrt_nr2(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = __builtin_fma(0.5 * y, __builtin_fma(-x * y, y, 1.0), y);
  y = __builtin_fma(0.5 * y, __builtin_fma(-x * y, y, 1.0), y);
  return y;
}
__device__ __forceinline__ bool chol3_inv_fast(const double v[6], double li[6]) {
  const bool ok0 = v[0] > 0.0;
  const double i00 = rsqrt_nr2(ok0 ? v[0] : 1.0);
  const double l10 = v[1] * i00, l20 = v[2] * i00;
  const double d11 = v[3] - l10 * l10;
  const bool ok1 = d11 > 0.0;
  const double i11 = rsqrt_nr2(ok1 ? d11 : 1.0);
  const double l21 = (v[4] - l20 * l10) * i11;
  const double d22 = v[5] - l20 * l20 - l21 * l21;
  const bool ok2 = d22 > 0.0;
  const double i22 = rsqrt_nr2(ok2 ? d22 : 1.0);
  li[0] = i00;
  li[1] = -l10 * i00 * i11;
  li[2] = i11;
  li[3] = (-l20 * i00 - l21 * li[1]) * i22;
  li[4] = -l21 * i11 * i22;
  li[5] = i22;
  return ok0 && ok1 && ok2 && isfinite(i22) && isfinite(i11) && isfinite(i00);
}

// schur_consume (sba_kernels.hpp) with a runtime number of k-steps, for the last chunk of a workgroup's slice: it may hold fewer than
// 16 points, and only the producer waves that have a point write their 12 panel rows (3 k-steps each)
template <int V>
__device__ inline void schur_f64_consume_part(const double* __restrict__ pl /* panel + lane offset */, int nks,
                                              Mfma<double>::acc_t (&acc)[SchurF64Cfg::TPW]) {
  using Cfg = SchurF64Cfg;
  using M_ = Mfma<double>;
  constexpr int LO = schur_lo(Cfg::NTILE, Cfg::NV, V), HI = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
  constexpr int RMIN = schur_tile_R(true, LO);
#pragma unroll 1
  for (int ks = 0; ks < nks; ++ks) {                             // (a rolled loop: at most 9 k-steps once per kernel)
    double fa[GROUP_TILES];
#pragma unroll
    for (int b = RMIN; b < GROUP_TILES; ++b) fa[b] = pl[ks * 4 * GROUP_ROWS + 16 * b];
    static_for<LO, HI>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int R = schur_tile_R(true, t), Tc = schur_tile_T(true, t);
      acc[t - LO] = M_::mma(fa[R], fa[Tc], acc[t - LO]);
    });
  }
}
template <int V = 0>
__device__ inline void schur_f64_consume_part_v(int v, const double* pl, int nks, Mfma<double>::acc_t (&acc)[SchurF64Cfg::TPW]) {
  if constexpr (V < SchurF64Cfg::NV) {
    if (v == V) schur_f64_consume_part<V>(pl, nks, acc);
    else schur_f64_consume_part_v<V + 1>(v, pl, nks, acc);
  }
}

__global__ __launch_bounds__(SCHUR_THREADS) void k_schur_fused_f64(
    const ParamSets<double> ps, const FusedDecide fd, int C,
    const double2* __restrict__ uv /* observations of a point in camera order; dense rigs: (p, c) at p*C + c */,
    const double* __restrict__ w, const int32_t* __restrict__ pt_start, const uint16_t* __restrict__ vis /* per point: bit c = camera c
    sees it; NULL = every camera sees every point */,
    int N, int ksplit, double* __restrict__ D2p, double* __restrict__ gp, double* __restrict__ pf, double* __restrict__ slabs,
    double* __restrict__ bpart, double* __restrict__ gdpart /* [ksplit][2][176]: g_c and diag U_c partials */,
    double* __restrict__ cost_part, double* __restrict__ gmax_part, long long* __restrict__ dbg) {
  extern __shared__ __align__(16) unsigned char smem[];
  using T = double;
  using Cfg = SchurF64Cfg;
  constexpr int THREADS = Cfg::THREADS, NPROD = Cfg::NPROD, TPW = Cfg::TPW, PTS = Cfg::PTS, K = Cfg::K, BUF = Cfg::BUF, KREG = Cfg::KREG, NL = Cfg::NL, KS = Cfg::KS;
  const bool stamp_wg = dbg && blockIdx.x == 0;
  if (stamp_wg && threadIdx.x == 0) dbg[48] = clock64();
  __shared__ LMState s_st;
  __shared__ LMLogRow s_row;
  __shared__ int s_have_row;
  __shared__ double s_scr[2][NPROD / 64];
  T* s_buf = reinterpret_cast<T*>(smem);                          // [2][BUF]
  T* s_cam = reinterpret_cast<T*>(smem + Cfg::CAM_OFF);           // [16][CAMPRE]
  T* s_Ured = reinterpret_cast<T*>(smem + Cfg::URED_OFF);         // [C][UPK]
  T* s_acc = reinterpret_cast<T*>(smem + Cfg::ACC_OFF);           // [4 producer waves][16][NL]
  T* s_U = s_buf;                                                 // [256][KS] once the panels are done with
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool producer = threadIdx.x < NPROD;
  // the slice of this workgroup: a multiple of 4 points (one producer wave's share of a chunk), NOT of the 16-point chunk -- at
  // 50 000 points and 256 workgroups that is 196 points = 12 chunks + 4 points everywhere instead of 13 chunks on 240 workgroups;
  // the last chunk then costs a quarter (one producer wave works, the consumers take 3 of the 12 k-steps)
  int per = (N + ksplit - 1) / ksplit;
  per = ((per + 3) / 4) * 4;
  const int pbeg = min(N, (int)blockIdx.x * per), pend = min(N, pbeg + per);
  const int nchunk = (pend - pbeg + PTS - 1) / PTS;
  // ---- prologue: the previous trial step's accept / reject decision (see FusedDecide), the panel buffers zeroed meanwhile
  DecidePartials dp;
  {
    constexpr int NWORD = sizeof(LMState) / 4;
    if ((int)threadIdx.x < NWORD) reinterpret_cast<int*>(&s_st)[threadIdx.x] = reinterpret_cast<const int*>(fd.st_in)[threadIdx.x];
    if (fd.do_decide) decide_gather(dp, fd.scal_all, fd.trial_part, fd.gmax_in, fd.n_trial, fd.n_gmax);
    {
      // panel rows of cameras >= C are never written and must read as zero; with 16 cameras every row the consumers read has been
      // written by the producers of the same chunk
      uint4* z4 = reinterpret_cast<uint4*>(smem);
      if (C < GROUP_CAMS)
        for (int i = threadIdx.x; i < (int)(Cfg::CAM_OFF / 16); i += THREADS) z4[i] = make_uint4(0, 0, 0, 0);
      for (int i = threadIdx.x; i < (NPROD / 64) * GROUP_CAMS * NL; i += THREADS) s_acc[i] = 0.0;
    }
    asm volatile("" : "+v"(dp.a), "+v"(dp.b), "+v"(dp.c), "+v"(dp.d), "+v"(dp.g));
    __syncthreads();
    if (fd.do_decide) {
      const bool running = s_st.status < 0;
      bool have_row = false;
      static_assert(THREADS == DECIDE_THREADS, "decide_fold is written for the thread count of this kernel");
      if (running) have_row = decide_core(&s_st, dp, fd.scal_all, fd.n_ranks, &s_row, fd.log_cap, reinterpret_cast<double*>(smem), nullptr);
      if (threadIdx.x == 0) s_have_row = have_row ? 1 : 0;
      __syncthreads();
      if (running && fd.scal_all == nullptr && C < GROUP_CAMS) {
        uint4* z4 = reinterpret_cast<uint4*>(smem);
        for (int i = threadIdx.x; i < 5 * THREADS * (int)sizeof(double) / 16; i += THREADS) z4[i] = make_uint4(0, 0, 0, 0);
      }
      if (blockIdx.x == 0) {
        if ((int)threadIdx.x < NWORD) reinterpret_cast<int*>(fd.st_out)[threadIdx.x] = reinterpret_cast<const int*>(&s_st)[threadIdx.x];
        if (threadIdx.x == 0 && s_have_row && fd.log) fd.log[s_st.iter - 1] = s_row;
      }
    }
  }
  if (s_st.status >= 0) return;
  const LMState* st = &s_st;
  const int cur_ = ps_cur(ps, st);
  const T* __restrict__ ptsT = ps.ptsT[cur_];
  for (int i = threadIdx.x; i < C * CAMPRE; i += THREADS) s_cam[i] = ps.campre[cur_][i];
  const T lam = st->lam;
  // fold the 16 point-lanes of every camera (register accumulators, handed over through s_U) and the four waves' LDS sets
  auto fold_u = [&]() {
    for (int o = threadIdx.x; o < C * UPK; o += THREADS) {
      const int c = o / UPK, k = o - c * UPK;
      T sum = 0;
      if (k < KREG) {
#pragma unroll
        for (int q = 0; q < 16; ++q) sum += s_U[(q * 16 + c) * KS + k];
      } else {
#pragma unroll
        for (int wv = 0; wv < NPROD / 64; ++wv) sum += s_acc[(wv * GROUP_CAMS + c) * NL + (k - KREG)];
      }
      s_Ured[o] = sum;
    }
  };
  if (stamp_wg && threadIdx.x == 0) dbg[49] = clock64();
  __syncthreads();

  if (producer) {
    const int q = threadIdx.x >> 4, c = threadIdx.x & 15;
    const bool cam_ok = c < C;
    const T* cp_safe = s_cam + (cam_ok ? c : 0) * CAMPRE;
    T Uacc[KREG];
    static_for<0, KREG>([&](auto kc) { Uacc[decltype(kc)::value] = (T)0; });
    T* acc_lane = s_acc + (wid * GROUP_CAMS + c) * NL;
    auto lds_add = [](T* slot, T v) { (void)__hip_atomic_fetch_add(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    T sq = 0, gmx = 0;
    __builtin_amdgcn_s_setprio(3);
    for (int it = 0; it <= nchunk; ++it) {
      if (it < nchunk && pbeg + it * PTS + 4 * wid < pend) {        // (a wave without a point in the last chunk sits the round out)
        T* panel = s_buf + (it & 1) * BUF;
        T* s_z = panel + K * GROUP_ROWS;
        // (no operand prefetch: the 18 registers it takes put accumulators into scratch, and the matrix pipe keeps the SIMD busy
        //  while this wave waits for its loads)
        const int p = pbeg + it * PTS + q;
        const bool have_pt = p < pend;
        const size_t pp = (size_t)(have_pt ? p : pbeg);
        unsigned mask = 0xffffu;
        size_t o = pp * C + c;
        if (vis) { mask = vis[pp]; o = (size_t)pt_start[pp] + __builtin_popcount(mask & ((1u << c) - 1u)); }
        const bool valid = have_pt && cam_ok && ((mask >> c) & 1u);
        double2 m = make_double2(0., 0.);
        T ww = 1;
        if (valid) { m = uv[o]; if (w) ww = w[o]; }
        const T X0 = ptsT[3 * pp], X1 = ptsT[3 * pp + 1], X2 = ptsT[3 * pp + 2];
        T r[2], Jc[2][NCP], Jp[2][3];
        obs_resjac<T>(cp_safe, X0, X1, X2, m.x, m.y, valid ? ww : (T)0, r, Jc, Jp, valid);
        sq += robust_apply<T>(ps.loss(), r, Jc, Jp);
        T v6[6], g3[3];
        v6[0] = row16_sum(Jp[0][0] * Jp[0][0] + Jp[1][0] * Jp[1][0]);
        v6[1] = row16_sum(Jp[0][0] * Jp[0][1] + Jp[1][0] * Jp[1][1]);
        v6[2] = row16_sum(Jp[0][0] * Jp[0][2] + Jp[1][0] * Jp[1][2]);
        v6[3] = row16_sum(Jp[0][1] * Jp[0][1] + Jp[1][1] * Jp[1][1]);
        v6[4] = row16_sum(Jp[0][1] * Jp[0][2] + Jp[1][1] * Jp[1][2]);
        v6[5] = row16_sum(Jp[0][2] * Jp[0][2] + Jp[1][2] * Jp[1][2]);
        g3[0] = row16_sum(Jp[0][0] * r[0] + Jp[1][0] * r[1]);
        g3[1] = row16_sum(Jp[0][1] * r[0] + Jp[1][1] * r[1]);
        g3[2] = row16_sum(Jp[0][2] * r[0] + Jp[1][2] * r[1]);
        const bool fixedp = have_pt && pt_fixed(ps, (size_t)p);
        if (!fixedp) gmx = fmax(gmx, fmax(fabs(g3[0]), fmax(fabs(g3[1]), fabs(g3[2]))));
        // point scaling: monotone max of the squared column norms (x_scale='jac', scipy trf.py:424,545)
        const double E0 = fmax(D2p[3 * pp], v6[0]), E1 = fmax(D2p[3 * pp + 1], v6[3]), E2 = fmax(D2p[3 * pp + 2], v6[5]);
        T f[PF];
        T li[6];
        const T vd[6] = {v6[0] + lam * fmax_pos(E0), v6[1], v6[2], v6[3] + lam * fmax_pos(E1), v6[4], v6[5] + lam * fmax_pos(E2)};
        const bool okp = have_pt && !fixedp && chol3_inv_fast(vd, li);
#pragma unroll
        for (int k = 0; k < PF; ++k) f[k] = (T)0;
        if (okp) {
#pragma unroll
          for (int k = 0; k < 6; ++k) f[k] = li[k];
          f[6] = li[0] * g3[0];
          f[7] = li[1] * g3[0] + li[2] * g3[1];
          f[8] = li[3] * g3[0] + li[4] * g3[1] + li[5] * g3[2];
          f[9] = (T)1;
        }
        if (have_pt && c == 0) {
          D2p[3 * (size_t)p] = E0; D2p[3 * (size_t)p + 1] = E1; D2p[3 * (size_t)p + 2] = E2;
          gp[3 * (size_t)p] = g3[0]; gp[3 * (size_t)p + 1] = g3[1]; gp[3 * (size_t)p + 2] = g3[2];
          double2* o2 = reinterpret_cast<double2*>(pf + (size_t)p * PF);
#pragma unroll
          for (int k = 0; k < PF / 2; ++k) o2[k] = make_double2(f[2 * k], f[2 * k + 1]);
        }
        // Jc[1][cx] = Jc[0][cy] = 0 structurally (obs_resjac; the robust scaling keeps zeros): their products are left out
        constexpr auto nz0 = [](int e) { return e != CP_CY; };      // row 0 (u) of column e can be non-zero
        constexpr auto nz1 = [](int e) { return e != CP_CX; };
        if (cam_ok) {
          // panel block of the observation: Ytilde = Jc^T (Jp L^-T) (11 x 3), rows 3q .. 3q+2 of the panel, columns 11 c ..; a
          // degenerate or absent point (f[9] == 0: L^-1 = 0) writes zeros
          T Jt[2][3];
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            Jt[rr][0] = Jp[rr][0] * f[0];
            Jt[rr][1] = Jp[rr][0] * f[1] + Jp[rr][1] * f[2];
            Jt[rr][2] = Jp[rr][0] * f[3] + Jp[rr][1] * f[4] + Jp[rr][2] * f[5];
          }
          T* pan = panel + 3 * q * GROUP_ROWS + c * NCP;
          const bool degenerate = f[9] == (T)0;                       // exact zeros then, whatever the Jacobian holds
          static_for<0, NCP>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              T y;
              if constexpr (!nz1(e)) y = Jc[0][e] * Jt[0][d];
              else if constexpr (!nz0(e)) y = Jc[1][e] * Jt[1][d];
              else y = Jc[0][e] * Jt[0][d] + Jc[1][e] * Jt[1][d];
              pan[d * GROUP_ROWS + e] = degenerate ? (T)0 : y;
            }
          });
          if (c == 0) { s_z[3 * q + 0] = f[6]; s_z[3 * q + 1] = f[7]; s_z[3 * q + 2] = f[8]; }
        }
        static_for<0, NCP>([&](auto ac) {
          constexpr int a = decltype(ac)::value;
          static_for<a, NCP>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            constexpr int k = a * NCP - (a * (a - 1)) / 2 + (b - a);
            constexpr bool t0 = nz0(a) && nz0(b), t1 = nz1(a) && nz1(b);
            if constexpr (k < KREG) {
              if constexpr (t0 && t1) Uacc[k] = __builtin_fma(Jc[1][a], Jc[1][b], __builtin_fma(Jc[0][a], Jc[0][b], Uacc[k]));
              else if constexpr (t0) Uacc[k] = __builtin_fma(Jc[0][a], Jc[0][b], Uacc[k]);
              else if constexpr (t1) Uacc[k] = __builtin_fma(Jc[1][a], Jc[1][b], Uacc[k]);
            } else {
              if constexpr (t0 && t1) lds_add(acc_lane + (k - KREG), __builtin_fma(Jc[1][a], Jc[1][b], Jc[0][a] * Jc[0][b]));
              else if constexpr (t0) lds_add(acc_lane + (k - KREG), Jc[0][a] * Jc[0][b]);
              else if constexpr (t1) lds_add(acc_lane + (k - KREG), Jc[1][a] * Jc[1][b]);
            }
          });
          constexpr int kg = NCP * (NCP + 1) / 2 + a;
          static_assert(kg >= KREG, "g_c lives in LDS");
          if constexpr (nz0(a) && nz1(a)) lds_add(acc_lane + (kg - KREG), __builtin_fma(Jc[1][a], r[1], Jc[0][a] * r[0]));
          else if constexpr (nz0(a)) lds_add(acc_lane + (kg - KREG), Jc[0][a] * r[0]);
          else lds_add(acc_lane + (kg - KREG), Jc[1][a] * r[1]);
        });
      }
      if (stamp_wg && threadIdx.x == 0 && it < 20) dbg[2 * it] = clock64();
      __syncthreads();
    }
    if (stamp_wg && threadIdx.x == 0) dbg[50] = clock64();
    // hand the register accumulators over (the panel buffers are free now)
    static_for<0, KREG>([&](auto kc) { constexpr int k = decltype(kc)::value; s_U[threadIdx.x * KS + k] = Uacc[k]; });
    const double cs = wave_sum(sq), gm = wave_max(gmx);
    if (lane == 0) { s_scr[0][wid] = cs; s_scr[1][wid] = gm; }
    __syncthreads();
    fold_u();
    __syncthreads();
  } else {
    const int cw = wid - NPROD / 64;
    typename Mfma<T>::acc_t acc[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) acc[s] = typename Mfma<T>::acc_t{0, 0, 0, 0};
    const int ct = threadIdx.x - NPROD;
    const int lane_off = (lane >> 4) * GROUP_ROWS + (lane & 15);
    double bacc = 0;
    // chunk it - 1 while the producers build chunk it; every chunk but the last is full (16 points, 12 k-steps).  The last one goes
    // through the rolled loop with the k-steps of the producer waves that had a point (12 panel rows = 3 k-steps per wave)
    auto rhs_part = [&](const T* panel, int nk) {
      const T* s_z = panel + K * GROUP_ROWS;
      if (ct < GROUP_ROWS) {
        T s0 = 0, s1 = 0;
#pragma unroll 4
        for (int k = 0; k < nk; k += 2) { s0 += panel[k * GROUP_ROWS + ct] * s_z[k]; s1 += panel[(k + 1) * GROUP_ROWS + ct] * s_z[k + 1]; }
        bacc += s0 + s1;
      }
    };
    for (int it = 0; it < nchunk; ++it) {
      if (it >= 1) {
        const T* panel = s_buf + ((it - 1) & 1) * BUF;
        rhs_part(panel, K);
        schur_consume_v<Cfg, K>(cw, panel + lane_off, panel + lane_off, acc);
      }
      if (stamp_wg && threadIdx.x == NPROD && it < 20) dbg[2 * it + 1] = clock64();
      __syncthreads();
    }
    if (nchunk >= 1) {
      const T* panel = s_buf + ((nchunk - 1) & 1) * BUF;
      const int nact = min(4, (pend - (pbeg + (nchunk - 1) * PTS) + 3) >> 2);
      rhs_part(panel, 12 * nact);
      schur_f64_consume_part_v(cw, panel + lane_off, 3 * nact, acc);
    }
    if (stamp_wg && threadIdx.x == NPROD && nchunk < 20) dbg[2 * nchunk + 1] = clock64();
    __syncthreads();
    __syncthreads();
    fold_u();
    __syncthreads();
    // take the camera blocks out of the tiles on and next to the diagonal: the slab then holds this workgroup's share of
    // sum Ytilde Ytilde^T - U, and k_build_exchange's plain sum of the slabs is -S
    static_for<0, Cfg::NV>([&](auto vc) {
      constexpr int V = decltype(vc)::value;
      if (cw == V) {
        constexpr int T0 = schur_lo(Cfg::NTILE, Cfg::NV, V), T1 = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
        static_for<T0, T1>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          constexpr int R = schur_tile_R(true, t), Tc = schur_tile_T(true, t);
          if constexpr (Tc - R <= 1) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const int i = 16 * R + Mfma<T>::row_of(lane, rg), j = 16 * Tc + (lane & 15);
              const int ci_ = i / NCP, cj_ = j / NCP;
              if (ci_ == cj_ && ci_ < C) {
                const int a = min(i - ci_ * NCP, j - cj_ * NCP), b = max(i - ci_ * NCP, j - cj_ * NCP);
                acc[t - T0][rg] -= s_Ured[ci_ * UPK + (a * NCP - (a * (a - 1)) / 2 + (b - a))];
              }
            }
          }
        });
      }
    });
    if (stamp_wg && threadIdx.x == NPROD) dbg[51] = clock64();
    T* slab = slabs + (size_t)blockIdx.x * (size_t)(GROUP_TILES * GROUP_TILES) * 256;
    schur_store_v<Cfg>(cw, slab, lane, acc);
    if (stamp_wg && threadIdx.x == NPROD) dbg[52] = clock64();
    if (ct < GROUP_ROWS) {
      const int c = ct / NCP, e = ct - c * NCP;
      const double gpart = (c < C) ? s_Ured[c * UPK + NCP * (NCP + 1) / 2 + e] : 0.0;
      const double dpart = (c < C) ? s_Ured[c * UPK + (e * NCP - (e * (e - 1)) / 2)] : 0.0;
      bpart[(size_t)blockIdx.x * GROUP_ROWS + ct] = bacc - gpart;
      gdpart[((size_t)blockIdx.x * 2 + 0) * GROUP_ROWS + ct] = gpart;
      gdpart[((size_t)blockIdx.x * 2 + 1) * GROUP_ROWS + ct] = dpart;
    }
  }
  if (threadIdx.x == 0) {
    double cs = 0, gm = 0;
    for (int wv = 0; wv < NPROD / 64; ++wv) { cs += s_scr[0][wv]; gm = fmax(gm, s_scr[1][wv]); }
    cost_part[blockIdx.x] = 0.5 * cs;
    gmax_part[blockIdx.x] = gm;
    if (stamp_wg) dbg[53] = clock64();
  }
}

// ------------------------------------------------------------------ 17 .. 23 cameras in fp64: the same kernel on compact rows
// The reference's own example rig has 17 cameras (example/config.json:24-42) and PySBA computes in float64: above one camera group
// the fp64 engine fell back to two linearisation launches + k_point_factor + three pair launches of k_schur_sym (541 us per step at
// 17 x 50k against 196 at 16 x 50k).  k_schur_fused_wide's row scheme (sba_schur_wide.hpp) -- COMPACT parameter-major rows
// row = e C + c, n = 11 C rows in NTW = ceil(n / 16) tiles: 12 tiles for 17 cameras, 13 for 18 -- with k_schur_fused_f64's arithmetic:
//   producers  PW = 3 (the default): a wave holds THREE points, packed -- lane l serves point l / C, camera l % C, 51 or 54 of the 64
//              lanes work --, 12 points per chunk = 36 panel rows = 9 k-steps of the f64 MFMA; per-point sums = segment differences
//              of a wave-wide DPP prefix scan read at the segment ends.  PW = 2 (SBA_WIDE_PW2=1): two points, one per 32-lane half,
//              8 points per chunk = 6 k-steps, per-point sums = DPP row sum + one v_permlane16_swap per dword;
//              52 of the 77 U_c / g_c accumulators in registers, 25 in per-wave LDS sets (ds_add_f64), as above;
//   panel      [24][16 NTW] doubles + z, double-buffered: a producer store is 32 consecutive doubles per row;
//   consumers  4 waves, contiguous ranges of the NTW (NTW + 1) / 2 upper-triangular tiles (20 of 78, 23 of 91: 160 / 184
//              accumulator VGPRs), the fragments of one k-step in registers;
//   outputs    k_schur_fused_wide's: slab [WIDE_SLOTS][64 lanes][4] per workgroup (k_build_exchange, emajor_mode 3), bpart /
//              gdpart rows in the exchange buffer's own order with a stride of WIDE_ROWS.
// 19 .. 23 cameras (14 .. 16 tiles: 27 .. 34 tiles = 216+ accumulator VGPRs per consumer wave) run with TS = 2: two workgroups per
// slice (grid.y), each building the whole panel and consuming half of the tiles; only the first stores the per-point and per-row results.
template <int NTW, int PW = 2, int TS = 1> struct SchurWide64Cfg {
  using elem = double;
  static constexpr int THREADS = SCHUR_THREADS, NPROD = 256, NV = 4 * TS;       // TS workgroups per slice share its tiles (grid.y)
  static constexpr int ROWS = 16 * NTW;
  static constexpr int NTILE = NTW * (NTW + 1) / 2;
  static constexpr int TPW = (NTILE + NV - 1) / NV;
  static constexpr int PTS = 4 * PW, K = 3 * PTS;                // points per chunk: 8 (two per producer wave) or 12 (three, packed)
  static constexpr int BUF = K * ROWS + K;                       // doubles: panel [K][ROWS] + z [K]
  static constexpr int MAXC = TS == 1 ? 18 : 23;
  static constexpr int KREG = SchurF64Cfg::KREG, NL = SchurF64Cfg::NL;
  static constexpr int KH = KREG / 2, KS = KH + 1;               // register accumulators handed over per pass, lane stride (odd)
  static constexpr size_t CAM_OFF = (size_t)2 * BUF * sizeof(double);
  static constexpr size_t URED_OFF = CAM_OFF + (size_t)MAXC * CAMPRE * sizeof(double);
  static constexpr size_t ACC_OFF = URED_OFF + (size_t)MAXC * UPK * sizeof(double);
  static constexpr size_t LDS_BYTES = ACC_OFF + (size_t)(NPROD / 64) * MAXC * NL * sizeof(double);
  static_assert(KREG % 2 == 0, "two hand-over passes");
  static_assert((size_t)NPROD * KS * sizeof(double) <= CAM_OFF, "the hand-over area is the two panel buffers");
  static_assert(CAM_OFF >= 5 * THREADS * sizeof(double), "the decision's scratch lives in the panel buffers");
  static_assert(LDS_BYTES + 1024 <= 160 * 1024, "LDS budget of a CU");
  static_assert(ROWS <= WIDE_ROWS && NTILE <= WIDE_SLOTS, "slab / row-partial strides of the wide path");
  static_assert(PW == 2 || PW == 3, "two points per wave (32-lane halves) or three (packed: lane l -> point l / C, camera l % C)");
  static_assert(K % 4 == 0, "whole k-steps per chunk");
};



template <int NTW, int PW, int TS>
__global__ __launch_bounds__(SCHUR_THREADS) void k_schur_fused_wide_f64(
    const ParamSets<double> ps, const FusedDecide fd, int C,
    const double2* __restrict__ uv, const double* __restrict__ w,
    const uint16_t* __restrict__ gmask /* [2][N] visibility masks of the two 16-camera groups, or NULL: dense */,
    const int32_t* __restrict__ gstart /* [2][N] first observation of the point in the group */,
    int N, int ksplit, double* __restrict__ D2p, double* __restrict__ gp, double* __restrict__ pf, double* __restrict__ slabs,
    double* __restrict__ bpart, double* __restrict__ gdpart, double* __restrict__ cost_part, double* __restrict__ gmax_part,
    long long* __restrict__ dbg) {
  extern __shared__ __align__(16) unsigned char smem[];
  using T = double;
  using Cfg = SchurWide64Cfg<NTW, PW, TS>;
  const int z = TS == 1 ? 0 : (int)blockIdx.y;      // which share of the slice's tiles this workgroup consumes (both build the whole panel)
  constexpr int THREADS = Cfg::THREADS, NPROD = Cfg::NPROD, TPW = Cfg::TPW, PTS = Cfg::PTS, K = Cfg::K, BUF = Cfg::BUF, ROWS = Cfg::ROWS;
  constexpr int KREG = Cfg::KREG, NL = Cfg::NL, KH = Cfg::KH, KS = Cfg::KS, MAXC = Cfg::MAXC;
  const bool stamp_wg = dbg && blockIdx.x == 0 && blockIdx.y == 0;
  if (stamp_wg && threadIdx.x == 0) dbg[48] = clock64();
  __shared__ LMState s_st;
  __shared__ LMLogRow s_row;
  __shared__ int s_have_row;
  __shared__ double s_scr[2][NPROD / 64];
  T* s_buf = reinterpret_cast<T*>(smem);                          // [2][BUF]
  T* s_cam = reinterpret_cast<T*>(smem + Cfg::CAM_OFF);           // [C][CAMPRE]
  T* s_Ured = reinterpret_cast<T*>(smem + Cfg::URED_OFF);         // [C][UPK]
  T* s_acc = reinterpret_cast<T*>(smem + Cfg::ACC_OFF);           // [4 producer waves][MAXC][NL]
  T* s_U = s_buf;                                                 // [256][KS] once the panels are done with
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool producer = threadIdx.x < NPROD;
  const int n = C * NCP;
  int per = (N + ksplit - 1) / ksplit;
  per = ((per + PTS - 1) / PTS) * PTS;
  const int pbeg = min(N, (int)blockIdx.x * per), pend = min(N, pbeg + per);
  const int nchunk = (pend - pbeg + PTS - 1) / PTS;
  // ---- prologue: state record, the pending LM decision, zeroed panels (rows n .. ROWS - 1 are never written)
  DecidePartials dp;
  {
    constexpr int NWORD = sizeof(LMState) / 4;
    if ((int)threadIdx.x < NWORD) reinterpret_cast<int*>(&s_st)[threadIdx.x] = reinterpret_cast<const int*>(fd.st_in)[threadIdx.x];
    if (fd.do_decide) decide_gather(dp, fd.scal_all, fd.trial_part, fd.gmax_in, fd.n_trial, fd.n_gmax);
    {
      uint4* z4 = reinterpret_cast<uint4*>(smem);
      for (int i = threadIdx.x; i < (int)(Cfg::CAM_OFF / 16); i += THREADS) z4[i] = make_uint4(0, 0, 0, 0);
      for (int i = threadIdx.x; i < (NPROD / 64) * MAXC * NL; i += THREADS) s_acc[i] = 0.0;
    }
    asm volatile("" : "+v"(dp.a), "+v"(dp.b), "+v"(dp.c), "+v"(dp.d), "+v"(dp.g));
    __syncthreads();
    if (fd.do_decide) {
      const bool running = s_st.status < 0;
      bool have_row = false;
      static_assert(THREADS == DECIDE_THREADS, "decide_fold is written for the thread count of this kernel");
      if (running) have_row = decide_core(&s_st, dp, fd.scal_all, fd.n_ranks, &s_row, fd.log_cap, reinterpret_cast<double*>(smem), nullptr);
      if (threadIdx.x == 0) s_have_row = have_row ? 1 : 0;
      __syncthreads();
      if (running && fd.scal_all == nullptr) {
        uint4* z4 = reinterpret_cast<uint4*>(smem);
        for (int i = threadIdx.x; i < 5 * THREADS * (int)sizeof(double) / 16; i += THREADS) z4[i] = make_uint4(0, 0, 0, 0);
      }
      if (blockIdx.x == 0 && z == 0) {
        if ((int)threadIdx.x < NWORD) reinterpret_cast<int*>(fd.st_out)[threadIdx.x] = reinterpret_cast<const int*>(&s_st)[threadIdx.x];
        if (threadIdx.x == 0 && s_have_row && fd.log) fd.log[s_st.iter - 1] = s_row;
      }
    }
  }
  if (s_st.status >= 0) return;
  const LMState* st = &s_st;
  const int cur_ = ps_cur(ps, st);
  const T* __restrict__ ptsT = ps.ptsT[cur_];
  for (int i = threadIdx.x; i < C * CAMPRE; i += THREADS) s_cam[i] = ps.campre[cur_][i];
  const T lam = st->lam;
  // fold the 4 PW lanes that served camera c (4 waves x PW points) for the register accumulators [k0, k0 + KH) handed over through
  // s_U, and -- in the second pass -- the four waves' LDS sets
  auto fold_u = [&](int k0) {
    for (int o = threadIdx.x; o < C * KH; o += THREADS) {
      const int c = o / KH, k = o - c * KH;
      T sum = 0;
      if constexpr (PW == 2) {
#pragma unroll
        for (int q = 0; q < 8; ++q) sum += s_U[(q * 32 + c) * KS + k];
      } else {
#pragma unroll
        for (int wv = 0; wv < NPROD / 64; ++wv)
#pragma unroll
          for (int hs = 0; hs < 3; ++hs) sum += s_U[(wv * 64 + hs * C + c) * KS + k];
      }
      s_Ured[c * UPK + k0 + k] = sum;
    }
    if (k0 != 0)
      for (int o = threadIdx.x; o < C * NL; o += THREADS) {
        const int c = o / NL, k = o - c * NL;
        T sum = 0;
#pragma unroll
        for (int wv = 0; wv < NPROD / 64; ++wv) sum += s_acc[(wv * MAXC + c) * NL + k];
        s_Ured[c * UPK + KREG + k] = sum;
      }
  };
  if (stamp_wg && threadIdx.x == 0) dbg[49] = clock64();
  __syncthreads();

  if (producer) {
    // point of the chunk, camera slot.  PW = 2: one point per 32-lane half; PW = 3: lane l of the wave -> point l / C, camera l % C
    const int hseg = PW == 2 ? (lane >> 5) : ((lane >= C) + (lane >= 2 * C) + (lane >= 3 * C));
    const int q = PW * wid + min(hseg, PW - 1);
    const int c = PW == 2 ? (lane & 31) : lane - hseg * C;
    const bool cam_ok = PW == 2 ? c < C : hseg < 3;
    auto point_sum = [&](T v) -> T {
      if constexpr (PW == 2) return half32_sum(v);
      else {
        // segment sums = differences of the wave-wide prefix scan read at the segment ends
        const T sc = wave_scan(v);
        const T e0 = lane_f64(sc, C - 1), e1 = lane_f64(sc, 2 * C - 1), e2 = lane_f64(sc, 3 * C - 1);
        return hseg == 0 ? e0 : hseg == 1 ? e1 - e0 : e2 - e1;
      }
    };
    const T* cp_safe = s_cam + (cam_ok ? c : 0) * CAMPRE;
    const int cs_ = cam_ok ? c : 0;
    const int grp = cs_ >> 4, cc = cs_ & 15;
    const uint16_t* __restrict__ gm = gmask ? gmask + (size_t)grp * N : nullptr;
    const int32_t* __restrict__ gs = gstart ? gstart + (size_t)grp * N : nullptr;
    T Uacc[KREG];
    static_for<0, KREG>([&](auto kc) { Uacc[decltype(kc)::value] = (T)0; });
    T* acc_lane = s_acc + (wid * MAXC + (cam_ok ? c : 0)) * NL;
    auto lds_add = [](T* slot, T v) { (void)__hip_atomic_fetch_add(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    T sq = 0, gmx = 0;
    const int row0 = cam_ok ? c : 0, rstep = cam_ok ? C : 0;        // row of parameter e: e C + c
    for (int it = 0; it <= nchunk; ++it) {
      if (it < nchunk) {
        T* panel = s_buf + (it & 1) * BUF;
        T* s_z = panel + K * ROWS;
        const int p = pbeg + it * PTS + q;
        const bool have_pt = p < pend;
        const size_t pp = (size_t)(have_pt ? p : pbeg);
        unsigned mask = 0xffffu;
        size_t o = pp * C + c;
        if (gm && cam_ok) { mask = gm[pp]; o = (size_t)gs[pp] + __builtin_popcount(mask & ((1u << cc) - 1u)); }
        const bool valid = have_pt && cam_ok && ((mask >> cc) & 1u);
        double2 m = make_double2(0., 0.);
        T ww = 1;
        if (valid) { m = uv[o]; if (w) ww = w[o]; }
        const T X0 = ptsT[3 * pp], X1 = ptsT[3 * pp + 1], X2 = ptsT[3 * pp + 2];
        T r[2], Jc[2][NCP], Jp[2][3];
        obs_resjac<T>(cp_safe, X0, X1, X2, m.x, m.y, valid ? ww : (T)0, r, Jc, Jp, valid);
        sq += robust_apply<T>(ps.loss(), r, Jc, Jp);
        T v6[6], g3[3];
        v6[0] = point_sum(Jp[0][0] * Jp[0][0] + Jp[1][0] * Jp[1][0]);
        v6[1] = point_sum(Jp[0][0] * Jp[0][1] + Jp[1][0] * Jp[1][1]);
        v6[2] = point_sum(Jp[0][0] * Jp[0][2] + Jp[1][0] * Jp[1][2]);
        v6[3] = point_sum(Jp[0][1] * Jp[0][1] + Jp[1][1] * Jp[1][1]);
        v6[4] = point_sum(Jp[0][1] * Jp[0][2] + Jp[1][1] * Jp[1][2]);
        v6[5] = point_sum(Jp[0][2] * Jp[0][2] + Jp[1][2] * Jp[1][2]);
        g3[0] = point_sum(Jp[0][0] * r[0] + Jp[1][0] * r[1]);
        g3[1] = point_sum(Jp[0][1] * r[0] + Jp[1][1] * r[1]);
        g3[2] = point_sum(Jp[0][2] * r[0] + Jp[1][2] * r[1]);
        const bool fixedp = have_pt && pt_fixed(ps, (size_t)p);
        if (!fixedp) gmx = fmax(gmx, fmax(fabs(g3[0]), fmax(fabs(g3[1]), fabs(g3[2]))));
        const double E0 = fmax(D2p[3 * pp], v6[0]), E1 = fmax(D2p[3 * pp + 1], v6[3]), E2 = fmax(D2p[3 * pp + 2], v6[5]);
        T f[PF];
        T li[6];
        const T vd[6] = {v6[0] + lam * fmax_pos(E0), v6[1], v6[2], v6[3] + lam * fmax_pos(E1), v6[4], v6[5] + lam * fmax_pos(E2)};
        const bool okp = have_pt && !fixedp && chol3_inv_fast(vd, li);
#pragma unroll
        for (int k = 0; k < PF; ++k) f[k] = (T)0;
        if (okp) {
#pragma unroll
          for (int k = 0; k < 6; ++k) f[k] = li[k];
          f[6] = li[0] * g3[0];
          f[7] = li[1] * g3[0] + li[2] * g3[1];
          f[8] = li[3] * g3[0] + li[4] * g3[1] + li[5] * g3[2];
          f[9] = (T)1;
        }
        if (have_pt && cam_ok && c == 0 && z == 0) {      // (the other workgroup of the slice computes the same values)
          D2p[3 * (size_t)p] = E0; D2p[3 * (size_t)p + 1] = E1; D2p[3 * (size_t)p + 2] = E2;
          gp[3 * (size_t)p] = g3[0]; gp[3 * (size_t)p + 1] = g3[1]; gp[3 * (size_t)p + 2] = g3[2];
          double2* o2 = reinterpret_cast<double2*>(pf + (size_t)p * PF);
#pragma unroll
          for (int k = 0; k < PF / 2; ++k) o2[k] = make_double2(f[2 * k], f[2 * k + 1]);
        }
        constexpr auto nz0 = [](int e) { return e != CP_CY; };      // row 0 (u) of column e can be non-zero
        constexpr auto nz1 = [](int e) { return e != CP_CX; };
        if (cam_ok) {
          T Jt[2][3];
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            Jt[rr][0] = Jp[rr][0] * f[0];
            Jt[rr][1] = Jp[rr][0] * f[1] + Jp[rr][1] * f[2];
            Jt[rr][2] = Jp[rr][0] * f[3] + Jp[rr][1] * f[4] + Jp[rr][2] * f[5];
          }
          T* pan = panel + 3 * q * ROWS + row0;
          const bool degenerate = f[9] == (T)0;
          static_for<0, NCP>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              T y;
              if constexpr (!nz1(e)) y = Jc[0][e] * Jt[0][d];
              else if constexpr (!nz0(e)) y = Jc[1][e] * Jt[1][d];
              else y = Jc[0][e] * Jt[0][d] + Jc[1][e] * Jt[1][d];
              pan[d * ROWS + e * rstep] = degenerate ? (T)0 : y;
            }
          });
          if (c == 0) { s_z[3 * q + 0] = f[6]; s_z[3 * q + 1] = f[7]; s_z[3 * q + 2] = f[8]; }
          static_for<0, NCP>([&](auto ac) {
            constexpr int a = decltype(ac)::value;
            static_for<a, NCP>([&](auto bc) {
              constexpr int b = decltype(bc)::value;
              constexpr int k = a * NCP - (a * (a - 1)) / 2 + (b - a);
              constexpr bool t0 = nz0(a) && nz0(b), t1 = nz1(a) && nz1(b);
              if constexpr (k < KREG) {
                if constexpr (t0 && t1) Uacc[k] = __builtin_fma(Jc[1][a], Jc[1][b], __builtin_fma(Jc[0][a], Jc[0][b], Uacc[k]));
                else if constexpr (t0) Uacc[k] = __builtin_fma(Jc[0][a], Jc[0][b], Uacc[k]);
                else if constexpr (t1) Uacc[k] = __builtin_fma(Jc[1][a], Jc[1][b], Uacc[k]);
              } else {
                if constexpr (t0 && t1) lds_add(acc_lane + (k - KREG), __builtin_fma(Jc[1][a], Jc[1][b], Jc[0][a] * Jc[0][b]));
                else if constexpr (t0) lds_add(acc_lane + (k - KREG), Jc[0][a] * Jc[0][b]);
                else if constexpr (t1) lds_add(acc_lane + (k - KREG), Jc[1][a] * Jc[1][b]);
              }
            });
            constexpr int kg = NCP * (NCP + 1) / 2 + a;
            static_assert(kg >= KREG, "g_c lives in LDS");
            if constexpr (nz0(a) && nz1(a)) lds_add(acc_lane + (kg - KREG), __builtin_fma(Jc[1][a], r[1], Jc[0][a] * r[0]));
            else if constexpr (nz0(a)) lds_add(acc_lane + (kg - KREG), Jc[0][a] * r[0]);
            else lds_add(acc_lane + (kg - KREG), Jc[1][a] * r[1]);
          });
        }
      }
      if (stamp_wg && threadIdx.x == 0 && it < 20) dbg[2 * it] = clock64();
      __syncthreads();
    }
    if (stamp_wg && threadIdx.x == 0) dbg[50] = clock64();
    // hand the register accumulators over in two passes (the panel buffers are free now)
    static_for<0, KH>([&](auto kc) { constexpr int k = decltype(kc)::value; s_U[threadIdx.x * KS + k] = Uacc[k]; });
    const double cs = wave_sum(sq), gm_ = wave_max(gmx);
    if (lane == 0) { s_scr[0][wid] = cs; s_scr[1][wid] = gm_; }
    __syncthreads();
    fold_u(0);
    __syncthreads();
    static_for<0, KH>([&](auto kc) { constexpr int k = decltype(kc)::value; s_U[threadIdx.x * KS + k] = Uacc[KH + k]; });
    __syncthreads();
    fold_u(KH);
    __syncthreads();
  } else {
    const int cw = wid - NPROD / 64, vw = 4 * z + cw;            // virtual consumer wave of the slice
    typename Mfma<T>::acc_t acc[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) acc[s] = typename Mfma<T>::acc_t{0, 0, 0, 0};
    const int ct = threadIdx.x - NPROD;
    const int lane_off = (lane >> 4) * ROWS + (lane & 15);
    double bacc = 0;
    for (int it = 0; it <= nchunk; ++it) {
      if (it >= 1) {
        const T* panel = s_buf + ((it - 1) & 1) * BUF;
        const T* s_z = panel + K * ROWS;
        if (ct < ROWS) {                  // right-hand side: b_row += sum_k panel[k][row] z[k]
          T s0 = 0, s1 = 0;
#pragma unroll 4
          for (int k = 0; k < K; k += 2) { s0 += panel[k * ROWS + ct] * s_z[k]; s1 += panel[(k + 1) * ROWS + ct] * s_z[k + 1]; }
          bacc += s0 + s1;
        }
        const T* pl = panel + lane_off;
        static_for<0, Cfg::NV>([&](auto vc) {
          constexpr int V = decltype(vc)::value;
          if (vw == V) {
            constexpr int LO = schur_lo(Cfg::NTILE, Cfg::NV, V), HI = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
            constexpr int RMIN = wide_tile_R<NTW>(LO);
#pragma unroll
            for (int ks = 0; ks < K / 4; ++ks) {
              T fa[NTW];
#pragma unroll
              for (int b = RMIN; b < NTW; ++b) fa[b] = pl[ks * 4 * ROWS + 16 * b];
              __builtin_amdgcn_sched_barrier(0);
              static_for<LO, HI>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int R = wide_tile_R<NTW>(t), Tc = wide_tile_T<NTW>(t);
                acc[t - LO] = Mfma<T>::mma(fa[R], fa[Tc], acc[t - LO]);
              });
            }
          }
        });
      }
      if (stamp_wg && threadIdx.x == NPROD && it < 20) dbg[2 * it + 1] = clock64();
      __syncthreads();
    }
    __syncthreads();
    fold_u(0);
    __syncthreads();
    __syncthreads();
    fold_u(KH);
    __syncthreads();
    // the camera's own block U_c sits wherever row and column belong to the same camera (row = e C + c, column = e' C + c): of the
    // lane's four rows rho0 + 4 reg at most one is congruent to its column modulo C (C > 12)
    const float invC = 1.0f / (float)C;
    static_for<0, Cfg::NV>([&](auto vc) {
      constexpr int V = decltype(vc)::value;
      if (vw == V) {
        constexpr int T0 = schur_lo(Cfg::NTILE, Cfg::NV, V), T1 = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
        static_for<T0, T1>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          constexpr int R = wide_tile_R<NTW>(t), Tc = wide_tile_T<NTW>(t);
          const int kap = 16 * Tc + (lane & 15), rho0 = 16 * R + (lane >> 4);
          const int dif = kap - rho0 + 16 * C;                                    // > 0, same residue modulo C
          const int dq = (int)(((float)dif + 0.5f) * invC), d = dif - dq * C;     // (kap - rho0) mod C
          if ((d & 3) == 0 && d < 16 && kap < n && rho0 + d < n) {
            const int rho = rho0 + d;
            const int er = (int)(((float)rho + 0.5f) * invC), cr = rho - er * C;
            const int ek = (int)(((float)kap + 0.5f) * invC);
            const int a = min(er, ek), b = max(er, ek);
            const T u = s_Ured[cr * UPK + (a * NCP - (a * (a - 1)) / 2 + (b - a))];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) acc[t - T0][rg] -= (4 * rg == d) ? u : (T)0;
          }
        });
      }
    });
    if (stamp_wg && threadIdx.x == NPROD) dbg[51] = clock64();
    T* slab = slabs + (size_t)blockIdx.x * (size_t)WIDE_SLOTS * 256;
    static_for<0, Cfg::NV>([&](auto vc) {
      constexpr int V = decltype(vc)::value;
      if (vw == V) {
        constexpr int LO = schur_lo(Cfg::NTILE, Cfg::NV, V), HI = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
#pragma unroll
        for (int t = LO; t < HI; ++t) {
          double2* dst = reinterpret_cast<double2*>(slab + (size_t)t * 256 + lane * 4);       // [tile][lane][reg]
          dst[0] = make_double2(acc[t - LO][0], acc[t - LO][1]);
          dst[1] = make_double2(acc[t - LO][2], acc[t - LO][3]);
        }
      }
    });
    if (stamp_wg && threadIdx.x == NPROD) dbg[52] = clock64();
    if (ct < ROWS) {
      // this thread's compact row rho = e C + c goes out in the exchange buffer's order o = c * 11 + e
      const int er = (int)(((float)ct + 0.5f) * invC), cr = ct - er * C;
      if (ct < n && z == 0) {
        const int o = cr * NCP + er;
        const double gpart = s_Ured[cr * UPK + NCP * (NCP + 1) / 2 + er];
        const double dpart = s_Ured[cr * UPK + (er * NCP - (er * (er - 1)) / 2)];
        bpart[(size_t)blockIdx.x * WIDE_ROWS + o] = bacc - gpart;             // rhs = sum (b - g_c) over the workgroups
        gdpart[((size_t)blockIdx.x * 2 + 0) * WIDE_ROWS + o] = gpart;
        gdpart[((size_t)blockIdx.x * 2 + 1) * WIDE_ROWS + o] = dpart;
      }
    }
  }
  if (threadIdx.x == 0 && z == 0) {
    double cs = 0, gm_ = 0;
    for (int wv = 0; wv < NPROD / 64; ++wv) { cs += s_scr[0][wv]; gm_ = fmax(gm_, s_scr[1][wv]); }
    cost_part[blockIdx.x] = 0.5 * cs;
    gmax_part[blockIdx.x] = gm_;
    if (stamp_wg) dbg[53] = clock64();
  }
}

#endif  // SBA_NCP == 11
}  // namespace SBA_NS
