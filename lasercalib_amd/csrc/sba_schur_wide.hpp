// sba_schur_wide.hpp -- fused linearise + Schur kernel for up to 256 reduced-system rows (f32 engine): 17 .. 23 cameras of the
// 11-parameter model, and every rig of up to 19 cameras of the 13-parameter model (which has no one-group kernel of its own).
//
// k_schur_fused_bf3 stops at one camera group (16 cameras = 176 rows = 11 MFMA tiles).  Right above it the rig used to fall back to
// the three-pass path with group PAIRS (two linearisation kernels + k_point_factor + a diagonal and an off-diagonal pair kernel):
// every Jacobian evaluated four times, a last group with one camera paying two full pair launches.  The reference's own example rig
// has 17 cameras (example/config.json:24-42).  This kernel keeps the one-launch structure for up to 16 row tiles (256 rows):
//
//   rows      COMPACT parameter-major: row = e*C + c (parameter e of camera c), n = 11 C rows in NTW = ceil(n / 16) tiles --
//             17 cameras are 12 tiles (78 upper-triangular ones), not two groups of 11 (253);
//   producers lane = (point, camera slot): a wave holds TWO points, one per 32-lane half, cameras c < C of the half active;
//             8 points per round = ONE 32-deep bf16 k-step.  Per-point sums (V_p, g_p) = DPP row sum + one row swap
//             (v_permlane16_swap); everything else -- r, Jc, Jp in registers, damped 3x3 factor, Ytilde = Jc^T Jp L^-T split
//             into three bf16 planes, the U_c / g_c / b_c register accumulators of the lane's camera -- is k_schur_fused_bf3's;
//   LDS       per k-step buffer 3 planes x [NTW tiles][2 halves][16 rows][4 slots] of 8-byte point slots (the layout of
//             k_schur_offdiag_bf3: point q -> half q >> 2, slot (q & 3) ^ ((row >> 2) & 3)), double-buffered;
//   consumers 4 waves, contiguous ranges of the NTW (NTW + 1) / 2 upper-triangular tiles; the fragments of the wave's own row
//             tiles stay in registers for the k-step, the column tiles beyond them stream through a double-buffered 3-plane
//             fragment set; six bf16 MFMAs per tile (h h', m h', l h', h m', m m', h l');
//   prologue  the LM decision of the previous step (FusedDecide), as in k_schur_fused_bf3.
// PW = 3 (17 and 18 cameras, <= 13 tiles): THREE points per wave, packed -- lane l serves point l / C, camera l % C -- so that 51
// or 54 of the 64 lanes work instead of 34 or 36.  A round then covers 12 points = one and a half k-steps: the k-step buffers
// form a ring of four, a round writes wherever its points fall, and the consumers take whatever k-steps the rounds so far have
// completed (one or two per barrier interval).  The per-point sums become segment sums of a wave-wide DPP prefix scan.
// Outputs: slabs [workgroup][tile][lane][reg] (k_build_exchange, emajor_mode 3 maps row -> (camera, parameter)), and
// bpart / gdpart rows in the exchange buffer's own order with a row stride of WIDE_ROWS.
#pragma once
#include "sba_kernels.hpp"

namespace SBA_NS {

template <int NTW, int PW = 2> struct SchurWideCfg {
  using elem = float;
  static constexpr int THREADS = SCHUR_THREADS, NPROD = 256, NV = 4;
  static constexpr int NTILE = NTW * (NTW + 1) / 2;
  static constexpr int TPW = (NTILE + NV - 1) / NV;
  static constexpr int PTS = 4 * PW;                              // points per round: 8 = one k-step, 12 = one and a half
  static constexpr int NBUF = PW == 3 ? 4 : 2;                    // k-step buffers (ring)
  static constexpr int HALF_BYTES = 16 * 32, TILE_BYTES = 2 * HALF_BYTES, PLANE_BYTES = NTW * TILE_BYTES;
  static constexpr int BUF_BYTES = 3 * PLANE_BYTES;
  static constexpr int UPKB = UPK + NCP, UPKS = UPKB | 1;
  static constexpr int MAXC = 24;                                 // cameras the camera table and the hand-over area are sized for (<= 23 used)
  static constexpr size_t HAND_BYTES = (size_t)(NPROD + MAXC) * UPKS * sizeof(float);          // accumulator hand-over area
  static constexpr size_t PANEL_BYTES = NBUF * (size_t)BUF_BYTES > HAND_BYTES ? NBUF * (size_t)BUF_BYTES : HAND_BYTES;
  static constexpr size_t LDS_BYTES = PANEL_BYTES + (size_t)MAXC * CAMPRE * sizeof(float);
  static_assert(LDS_BYTES + 1024 <= 160 * 1024, "panels + camera table must fit the LDS");
  static_assert(PW == 2 || PW == 3, "two points per wave (32-lane halves) or three (packed)");
};
// row-major enumeration of the upper triangle of an NTW x NTW tile grid
template <int NTW> __host__ __device__ constexpr int wide_tile_R(int t) {
  int R = 0, rem = t;
  while (rem >= NTW - R) { rem -= NTW - R; ++R; }
  return R;
}
template <int NTW> __host__ __device__ constexpr int wide_tile_T(int t) {
  int R = 0, rem = t;
  while (rem >= NTW - R) { rem -= NTW - R; ++R; }
  return R + rem;
}


template <int NTW, int PW>
__global__ __launch_bounds__(SCHUR_THREADS) void k_schur_fused_wide(
    const ParamSets<float> ps, const FusedDecide fd, int C,
    const float2* __restrict__ uv, const float* __restrict__ w,
    const uint16_t* __restrict__ gmask /* [2][N] visibility masks of the two 16-camera groups, or NULL: dense */,
    const int32_t* __restrict__ gstart /* [2][N] first observation of the point in the group */,
    int N, int ksplit, double* __restrict__ D2p, double* __restrict__ gp,
    float* __restrict__ pf, float* __restrict__ slabs, double* __restrict__ bpart, double* __restrict__ gdpart,
    double* __restrict__ cost_part, double* __restrict__ gmax_part, long long* __restrict__ dbg) {
  extern __shared__ __align__(16) unsigned char smem[];
  using T = float;
  using Cfg = SchurWideCfg<NTW, PW>;
  constexpr int THREADS = Cfg::THREADS, NPROD = Cfg::NPROD, TPW = Cfg::TPW, PTS = Cfg::PTS, UPKB = Cfg::UPKB, UPKS = Cfg::UPKS;
  const bool stamp_wg = dbg && blockIdx.x == 0;
  if (stamp_wg && threadIdx.x == 0) dbg[48] = clock64();
  __shared__ LMState s_st;
  __shared__ LMLogRow s_row;
  __shared__ int s_have_row;
  __shared__ double s_scr[2][NPROD / 64];
  T* s_cam = reinterpret_cast<T*>(smem + Cfg::PANEL_BYTES);              // [C][CAMPRE]
  T* s_U = reinterpret_cast<T*>(smem);                                   // [256][UPKS] once the panels are done with
  T* s_Ured = s_U + NPROD * UPKS;                                        // [C][UPKB]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool producer = threadIdx.x < NPROD;
  const int n = C * NCP;
  int per = (N + ksplit - 1) / ksplit;
  per = ((per + PTS - 1) / PTS) * PTS;
  const int pbeg = min(N, (int)blockIdx.x * per), pend = min(N, pbeg + per);
  const int nchunk = (pend - pbeg + PTS - 1) / PTS;
  // ---- prologue: state record, the pending LM decision, zeroed panels (k_schur_fused_bf3's, see there)
  DecidePartials dp;
  {
    constexpr int NWORD = sizeof(LMState) / 4;
    if ((int)threadIdx.x < NWORD) reinterpret_cast<int*>(&s_st)[threadIdx.x] = reinterpret_cast<const int*>(fd.st_in)[threadIdx.x];
    if (fd.do_decide) decide_gather(dp, fd.scal_all, fd.trial_part, fd.gmax_in, fd.n_trial, fd.n_gmax);
    {   // zero both k-step buffers meanwhile: padding rows and the slots of absent points are never written
      uint4* z4 = reinterpret_cast<uint4*>(smem);
      for (int i = threadIdx.x; i < Cfg::NBUF * Cfg::BUF_BYTES / 16; i += THREADS) z4[i] = make_uint4(0, 0, 0, 0);
    }
    asm volatile("" : "+v"(dp.a), "+v"(dp.b), "+v"(dp.c), "+v"(dp.d), "+v"(dp.g));
    __syncthreads();
    if (fd.do_decide) {
      const bool running = s_st.status < 0;
      bool have_row = false;
      static_assert(THREADS == DECIDE_THREADS, "decide_fold is written for the thread count of this kernel");
      static_assert(2 * Cfg::BUF_BYTES >= 5 * THREADS * (int)sizeof(double), "the decision's scratch lives in the zeroed panel buffers");
      if (running) have_row = decide_core(&s_st, dp, fd.scal_all, fd.n_ranks, &s_row, fd.log_cap, reinterpret_cast<double*>(smem), nullptr);
      if (threadIdx.x == 0) s_have_row = have_row ? 1 : 0;
      __syncthreads();
      if (running && fd.scal_all == nullptr) {
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
  const T lam = (T)st->lam;
  // the 4 PW lanes that served camera c (4 waves x PW points) fold their accumulators
  auto fold_u = [&]() {
    for (int o = threadIdx.x; o < C * UPKB; o += THREADS) {
      const int c = o / UPKB, k = o - c * UPKB;
      T sum = 0;
#pragma unroll
      for (int q = 0; q < 4 * PW; ++q) {
        const int t = PW == 2 ? q * 32 + c : (q / 3) * 64 + (q % 3) * C + c;
        sum += s_U[t * UPKS + k];
      }
      s_Ured[o] = sum;
    }
  };
  if (stamp_wg && threadIdx.x == 0) dbg[49] = clock64();
  __syncthreads();

  if (producer) {
    // point of the round, camera slot.  PW = 2: one point per 32-lane half; PW = 3: lane l of the wave -> point l / C, camera l % C
    const int hseg = PW == 2 ? (lane >> 5) : ((lane >= C) + (lane >= 2 * C) + (lane >= 3 * C));
    const int q = PW * wid + min(hseg, PW - 1);
    const int c = PW == 2 ? (lane & 31) : lane - hseg * C;
    const bool cam_ok = PW == 2 ? c < C : hseg < 3;
    auto point_sum = [&](T v) -> T {
      if constexpr (PW == 2) return half32_sum(v);
      else {
        const T sc = wave_scan(v);
        const int si = __builtin_bit_cast(int, sc);
        const T e0 = __builtin_bit_cast(T, __builtin_amdgcn_readlane(si, C - 1));
        const T e1 = __builtin_bit_cast(T, __builtin_amdgcn_readlane(si, 2 * C - 1));
        const T e2 = __builtin_bit_cast(T, __builtin_amdgcn_readlane(si, 3 * C - 1));
        return hseg == 0 ? e0 : hseg == 1 ? e1 - e0 : e2 - e1;
      }
    };
    const T* cp_safe = s_cam + (cam_ok ? c : 0) * CAMPRE;
    const int grp = c >> 4, cc = c & 15;
    const uint16_t* __restrict__ gm = gmask ? gmask + (size_t)grp * N : nullptr;
    const int32_t* __restrict__ gs = gstart ? gstart + (size_t)grp * N : nullptr;
    T Uacc[UPKB];
    static_for<0, UPKB>([&](auto kc) { Uacc[decltype(kc)::value] = (T)0; });
    T sq = 0, gmx = 0;
    float2 n_uv = make_float2(0.f, 0.f);
    T n_w = 1, n_X[3] = {0, 0, 0};
    double n_D[3] = {0, 0, 0};
    bool n_valid = false, n_pt = false;
    unsigned i_mask = 0; int i_start = 0; bool i_pt = false;
    auto request_index = [&](int chunk) {
      const int p = pbeg + chunk * PTS + q;
      i_pt = chunk < nchunk && p < pend;
      i_mask = 0xffffu; i_start = 0;
      if (i_pt && gm && cam_ok) { i_mask = gm[p]; i_start = gs[p]; }
    };
    auto request = [&](int chunk) {
      const int p = pbeg + chunk * PTS + q;
      n_pt = i_pt;
      n_valid = i_pt && cam_ok && ((i_mask >> cc) & 1u);
      if (n_pt) {
        n_X[0] = ptsT[3 * (size_t)p]; n_X[1] = ptsT[3 * (size_t)p + 1]; n_X[2] = ptsT[3 * (size_t)p + 2];
        n_D[0] = D2p[3 * (size_t)p]; n_D[1] = D2p[3 * (size_t)p + 1]; n_D[2] = D2p[3 * (size_t)p + 2];
      }
      if (n_valid) {
        const size_t o = gm ? (size_t)i_start + __builtin_popcount(i_mask & ((1u << cc) - 1u)) : (size_t)p * C + c;
        n_uv = uv[o];
        n_w = w ? w[o] : (T)1;
      }
      request_index(chunk + 1);
    };
    request_index(0);
    request(0);
    // byte offset of this lane's 8-byte point slot inside a plane, for parameter e: row e C + c -> tile, row of the tile; the
    // point of the round is the lane's own for the whole kernel (half q >> 2, slot (q & 3) ^ ((row >> 2) & 3))
    // (PW = 3: the point's place inside its k-step changes from round to round; slot_off then holds the row part with the row's
    //  swizzle bits, and the round adds buffer, half and slot)
    int slot_off[NCP];
#pragma unroll
    for (int e = 0; e < NCP; ++e) {
      const int row = e * (cam_ok ? C : 0) + (cam_ok ? c : 0), i = row & 15;
      if constexpr (PW == 2) slot_off[e] = (row >> 4) * Cfg::TILE_BYTES + (q >> 2) * Cfg::HALF_BYTES + i * 32 + (((q & 3) ^ ((i >> 2) & 3)) << 3);
      else slot_off[e] = (row >> 4) * Cfg::TILE_BYTES + i * 32 + (((i >> 2) & 3) << 3);
    }
    for (int it = 0; it <= nchunk; ++it) {
      if (it < nchunk) {
        unsigned char* pbuf;
        int sxor = 0;
        if constexpr (PW == 2) pbuf = smem + (it & 1) * Cfg::BUF_BYTES;
        else {
          const int g = it * PTS + q, kk = g >> 3, sp = g & 7;             // k-step of the point, its place in it
          pbuf = smem + (kk & 3) * Cfg::BUF_BYTES + (sp >> 2) * Cfg::HALF_BYTES;
          sxor = (sp & 3) << 3;
          if ((nchunk & 1) && it == nchunk - 1) {
            // an odd number of rounds leaves the last k-step half filled: its second half (points 4..7) still holds what the ring
            // slot carried four k-steps ago -- zero it now (nobody reads that slot during this round)
            const int kl = (nchunk * PTS) >> 3;
            unsigned char* zb = smem + (kl & 3) * Cfg::BUF_BYTES + Cfg::HALF_BYTES;
            for (int i = threadIdx.x; i < 3 * NTW * (Cfg::HALF_BYTES / 16); i += NPROD) {
              const int blk = i / (Cfg::HALF_BYTES / 16), off = i - blk * (Cfg::HALF_BYTES / 16);
              *reinterpret_cast<uint4*>(zb + blk * Cfg::TILE_BYTES + off * 16) = make_uint4(0, 0, 0, 0);
            }
          }
        }
        const bool valid = n_valid, have_pt = n_pt;
        const float2 m = n_uv;
        const T ww = n_w, X0 = n_X[0], X1 = n_X[1], X2 = n_X[2];
        const double D0 = n_D[0], D1 = n_D[1], D2 = n_D[2];
        const int p = pbeg + it * PTS + q;
        request(it + 1);
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
        if (!fixedp) gmx = fmaxf(gmx, fmaxf(fabsf(g3[0]), fmaxf(fabsf(g3[1]), fabsf(g3[2]))));
        const double E0 = fmax(D0, (double)v6[0]), E1 = fmax(D1, (double)v6[3]), E2 = fmax(D2, (double)v6[5]);
        T f[PF];
        T li[6];
        T vd[6] = {v6[0] + lam * (T)fmax_pos(E0), v6[1], v6[2], v6[3] + lam * (T)fmax_pos(E1), v6[4], v6[5] + lam * (T)fmax_pos(E2)};
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
        if (have_pt && cam_ok && c == 0) {
          D2p[3 * (size_t)p] = E0; D2p[3 * (size_t)p + 1] = E1; D2p[3 * (size_t)p + 2] = E2;
          gp[3 * (size_t)p] = (double)g3[0]; gp[3 * (size_t)p + 1] = (double)g3[1]; gp[3 * (size_t)p + 2] = (double)g3[2];
          float4* o4 = reinterpret_cast<float4*>(pf + (size_t)p * PF);
          o4[0] = make_float4(f[0], f[1], f[2], f[3]);
          o4[1] = make_float4(f[4], f[5], f[6], f[7]);
          o4[2] = make_float4(f[8], f[9], f[10], f[11]);
        }
        T Jt[2][3];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          Jt[rr][0] = Jp[rr][0] * f[0];
          Jt[rr][1] = Jp[rr][0] * f[1] + Jp[rr][1] * f[2];
          Jt[rr][2] = Jp[rr][0] * f[3] + Jp[rr][1] * f[4] + Jp[rr][2] * f[5];
        }
        if (cam_ok) {
          static_for<0, NCP>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            T y[3];
#pragma unroll
            for (int d = 0; d < 3; ++d)      // (Jc[1][cx] = Jc[0][cy] = 0 structurally: those products are left out, same bits)
              y[d] = e == CP_CX ? Jc[0][e] * Jt[0][d] : e == CP_CY ? Jc[1][e] * Jt[1][d] : Jc[0][e] * Jt[0][d] + Jc[1][e] * Jt[1][d];
            Uacc[UPK + e] = __builtin_fmaf(y[2], f[8], __builtin_fmaf(y[1], f[7], __builtin_fmaf(y[0], f[6], Uacc[UPK + e])));
            auto pk = [](float lo, float hi) -> unsigned {
              unsigned v;
              asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v) : "v"(lo), "v"(hi));
              return v;
            };
            auto pk1 = [](float lo) -> unsigned {
              unsigned v;
              asm("v_cvt_pk_bf16_f32 %0, %1, 0" : "=v"(v) : "v"(lo));
              return v;
            };
            auto lo_f = [](unsigned pkd) -> float { return __builtin_bit_cast(float, pkd << 16); };
            auto hi_f = [](unsigned pkd) -> float { return __builtin_bit_cast(float, pkd & 0xffff0000u); };
            const unsigned h01 = pk(y[0], y[1]), h2 = pk1(y[2]);
            const float r0 = y[0] - lo_f(h01), r1 = y[1] - hi_f(h01), r2 = y[2] - lo_f(h2);
            const unsigned m01 = pk(r0, r1), m2 = pk1(r2);
            const float s0 = r0 - lo_f(m01), s1 = r1 - hi_f(m01), s2 = r2 - lo_f(m2);
            const unsigned l01 = pk(s0, s1), l2 = pk1(s2);
            unsigned char* dst = pbuf + (PW == 2 ? slot_off[e] : (slot_off[e] ^ sxor));
            *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h2);
            *reinterpret_cast<uint2*>(dst + Cfg::PLANE_BYTES) = make_uint2(m01, m2);
            *reinterpret_cast<uint2*>(dst + 2 * Cfg::PLANE_BYTES) = make_uint2(l01, l2);
          });
        }
        static_for<0, NCP>([&](auto ac) {
          constexpr int a = decltype(ac)::value;
          static_for<a, NCP>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            constexpr int k = a * NCP - (a * (a - 1)) / 2 + (b - a);
            constexpr bool t0 = a != CP_CY && b != CP_CY, t1 = a != CP_CX && b != CP_CX;      // row 0 (u) / row 1 (v) can be non-zero
            if constexpr (t0 && t1) Uacc[k] = __builtin_fmaf(Jc[1][a], Jc[1][b], __builtin_fmaf(Jc[0][a], Jc[0][b], Uacc[k]));
            else if constexpr (t0) Uacc[k] = __builtin_fmaf(Jc[0][a], Jc[0][b], Uacc[k]);
            else if constexpr (t1) Uacc[k] = __builtin_fmaf(Jc[1][a], Jc[1][b], Uacc[k]);
          });
          constexpr int kg = NCP * (NCP + 1) / 2 + a;
          if constexpr (a == CP_CX) Uacc[kg] = __builtin_fmaf(Jc[0][a], r[0], Uacc[kg]);
          else if constexpr (a == CP_CY) Uacc[kg] = __builtin_fmaf(Jc[1][a], r[1], Uacc[kg]);
          else Uacc[kg] = __builtin_fmaf(Jc[1][a], r[1], __builtin_fmaf(Jc[0][a], r[0], Uacc[kg]));
        });
      }
      if (stamp_wg && threadIdx.x == 0 && it < 20) dbg[2 * it] = clock64();
      __syncthreads();
    }
    if (stamp_wg && threadIdx.x == 0) dbg[50] = clock64();
    __syncthreads();                       // the consumers have read the last k-step: the buffers become the hand-over area
    static_for<0, UPKB>([&](auto kc) { constexpr int k = decltype(kc)::value; s_U[threadIdx.x * UPKS + k] = Uacc[k]; });
    const double cs = wave_sum((double)sq), gm_ = wave_max((double)gmx);
    if (lane == 0) { s_scr[0][wid] = cs; s_scr[1][wid] = gm_; }
    __syncthreads();
    fold_u();
    __syncthreads();
  } else {
    const int cw = wid - NPROD / 64;
    typename Mfma<T>::acc_t acc[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) acc[s] = typename Mfma<T>::acc_t{0, 0, 0, 0};
    const int ct = threadIdx.x - NPROD;
    // fragment of a tile: points g and g + 4 of row i = lane & 15: slot g ^ ((i >> 2) & 3) of the two halves
    const int frag_off = (lane & 15) * 32 + (((lane >> 4) ^ (((lane & 15) >> 2) & 3)) << 3);
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    auto process = [&](const unsigned char* pbuf) {
        static_for<0, Cfg::NV>([&](auto vc) {
          constexpr int V = decltype(vc)::value;
          if (cw == V) {
            constexpr int LO = schur_lo(Cfg::NTILE, Cfg::NV, V), HI = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
            constexpr int RMIN = wide_tile_R<NTW>(LO), RMAX = wide_tile_R<NTW>(HI - 1);
            constexpr int NRW = RMAX - RMIN + 1;
            auto frag = [&](int plane, int b) -> bf16x8_t {
              const unsigned char* rowp = pbuf + plane * Cfg::PLANE_BYTES + b * Cfg::TILE_BYTES;
              const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(rowp);
              const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(rowp + Cfg::HALF_BYTES);
              return __builtin_bit_cast(bf16x8_t, u32x4_t{lo[0], lo[1], hi[0], hi[1]});
            };
            // the wave's own row tiles RMIN..RMAX: all three planes stationary for the k-step; column tiles beyond RMAX stream
            // through fb (double-buffered), column tiles inside the row range use the stationary fragments
            bf16x8_t fa[3][NRW], fb[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
              for (int rr = 0; rr < NRW; ++rr) fa[pl][rr] = frag(pl, RMIN + rr);
            if constexpr (RMAX + 1 < NTW) {
#pragma unroll
              for (int pl = 0; pl < 3; ++pl) fb[(RMAX + 1) & 1][pl] = frag(pl, RMAX + 1);
            }
            static_for<RMIN, NTW>([&](auto cc_) {
              constexpr int Tc = decltype(cc_)::value;
              constexpr int cur = Tc & 1;
              if constexpr (Tc > RMAX && Tc + 1 < NTW) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fb[cur ^ 1][pl] = frag(pl, Tc + 1);
              }
              __builtin_amdgcn_sched_barrier(0);
              static_for<0, 6>([&](auto pc) {
                constexpr int pr = decltype(pc)::value;
                constexpr int pa = (pr == 0 || pr == 3 || pr == 5) ? 0 : (pr == 1 || pr == 4) ? 1 : 2;     // h h' | m h' | l h' | h m' | m m' | h l'
                constexpr int pb = pr < 3 ? 0 : pr < 5 ? 1 : 2;
                static_for<0, NRW>([&](auto rc) {
                  constexpr int R = RMIN + decltype(rc)::value;
                  if constexpr (R <= Tc) {
                    constexpr int t = R * NTW - (R * (R - 1)) / 2 + (Tc - R);
                    if constexpr (t >= LO && t < HI) {
                      if constexpr (Tc <= RMAX)
                        acc[t - LO] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[pa][R - RMIN], fa[pb][Tc - RMIN], acc[t - LO], 0, 0, 0);
                      else
                        acc[t - LO] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[pa][R - RMIN], fb[cur][pb], acc[t - LO], 0, 0, 0);
                    }
                  }
                });
              });
              __builtin_amdgcn_sched_barrier(0);
            });
          }
        });
    };
    int kprev = 0;
    for (int it = 0; it <= nchunk; ++it) {
      if (it >= 1) {
        if constexpr (PW == 2) process(smem + ((it - 1) & 1) * Cfg::BUF_BYTES + frag_off);
        else {
          // k-steps the rounds before this one have completed (the last interval also takes the half-filled one)
          const int kd = it == nchunk ? (nchunk * PTS + 7) >> 3 : (it * PTS) >> 3;
          for (int kk = kprev; kk < kd; ++kk) process(smem + (kk & 3) * Cfg::BUF_BYTES + frag_off);
          kprev = kd;
        }
      }
      if (stamp_wg && threadIdx.x == NPROD && it < 20) dbg[2 * it + 1] = clock64();
      __syncthreads();
    }
    __syncthreads();                       // matches the producers' barrier in front of the hand-over
    __syncthreads();                       // accumulators are in s_U
    fold_u();
    __syncthreads();
    const float invC = 1.0f / (float)C;
    static_for<0, Cfg::NV>([&](auto vc) {
      constexpr int V = decltype(vc)::value;
      if (cw == V) {
        constexpr int T0 = schur_lo(Cfg::NTILE, Cfg::NV, V), T1 = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
        static_for<T0, T1>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          constexpr int R = wide_tile_R<NTW>(t), Tc = wide_tile_T<NTW>(t);
          // the camera's own block U_c sits wherever row and column belong to the same camera (row = e C + c, column = e' C + c):
          // of the lane's four rows rho0 .. rho0 + 3 at most one is congruent to its column modulo C (C > 4)
          const int kap = 16 * Tc + (lane & 15), rho0 = 16 * R + 4 * (lane >> 4);
          const int dif = kap - rho0 + 16 * C;                                    // > 0 (C >= 16), same residue modulo C
          const int dq = (int)(((float)dif + 0.5f) * invC), d = dif - dq * C;     // (kap - rho0) mod C
          if (d < 4 && kap < n && rho0 + d < n) {
            const int rho = rho0 + d;
            const int er = (int)(((float)rho + 0.5f) * invC), cr = rho - er * C;
            const int ek = (int)(((float)kap + 0.5f) * invC);
            const int a = min(er, ek), b = max(er, ek);
            const T u = s_Ured[cr * UPKB + (a * NCP - (a * (a - 1)) / 2 + (b - a))];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) acc[t - T0][rg] -= (rg == d) ? u : (T)0;
          }
        });
      }
    });
    if (stamp_wg && threadIdx.x == NPROD) dbg[51] = clock64();
    float4* slab4 = reinterpret_cast<float4*>(slabs + (size_t)blockIdx.x * (size_t)WIDE_SLOTS * 256) + lane;
    static_for<0, Cfg::NV>([&](auto vc) {
      constexpr int V = decltype(vc)::value;
      if (cw == V) {
        constexpr int LO = schur_lo(Cfg::NTILE, Cfg::NV, V), HI = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
        float4* pt = slab4 + LO * 64;
        static_for<LO, HI>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          *pt = make_float4(acc[t - LO][0], acc[t - LO][1], acc[t - LO][2], acc[t - LO][3]);     // [tile][lane][reg]
          pt += 64;
          asm volatile("" : "+v"(pt));
        });
      }
    });
    if (stamp_wg && threadIdx.x == NPROD) dbg[52] = clock64();
    for (int o = ct; o < n; o += THREADS - NPROD) {                      // rows in the exchange buffer's order: o = c * 11 + e
      const int c = o / NCP, e = o - c * NCP;
      const double gpart = (double)s_Ured[c * UPKB + NCP * (NCP + 1) / 2 + e];
      const double dpart = (double)s_Ured[c * UPKB + (e * NCP - (e * (e - 1)) / 2)];
      const double bsum = (double)s_Ured[c * UPKB + UPK + e];
      bpart[(size_t)blockIdx.x * WIDE_ROWS + o] = bsum - gpart;             // rhs = sum (b - g_c) over the workgroups
      gdpart[((size_t)blockIdx.x * 2 + 0) * WIDE_ROWS + o] = gpart;
      gdpart[((size_t)blockIdx.x * 2 + 1) * WIDE_ROWS + o] = dpart;
    }
  }
  if (threadIdx.x == 0) {
    double cs = 0, gm_ = 0;
    for (int wv = 0; wv < NPROD / 64; ++wv) { cs += s_scr[0][wv]; gm_ = fmax(gm_, s_scr[1][wv]); }
    cost_part[blockIdx.x] = 0.5 * cs;
    gmax_part[blockIdx.x] = gm_;
    if (stamp_wg) dbg[53] = clock64();
  }
}

}  // namespace SBA_NS
