// sba_kernels.hpp -- HIP kernels of the bundle-adjustment hot path (gfx950 / CDNA4 only).
//
// Data layout in HBM (all static per problem, built once by sba_upload):
//   point-major (pm) observation arrays   uv_pm[M] (T2), w_pm[M] (T), ci_pm[M] (i32), pi_pm[M] (i32)
//       observations of one point are contiguous (the order scripts/get_points3d.py:78-86 emits);
//       pt_start[N+1] is the CSR offset, blk_pt[B+1] cuts the list into POINT-ALIGNED blocks of
//       <= 256 observations so no point straddles a workgroup (no global atomics, deterministic sums).
//   camera-major (cm) copies               uv_cm[M], w_cm[M], pi_cm[M]; chunk table {cam, begin, end}
//       every chunk holds observations of ONE camera, so a wave's J^T J is a single 16x16 MFMA
//       accumulation with no cross-lane traffic.
//   parameters: cams (C x 11, f64 master), pts (N x 3, f64 master) + T-typed shadow of the points,
//       CamPre table (C x 25, T) rebuilt by k_cam_prep whenever the cameras change.
//   dense rigs (every camera sees every point once) are stored in canonical order, observation (p, c) at p*C + c: with
//       <= 16 cameras a 16-lane DPP row then holds all observations of one point and lane % 16 is the camera.
//
// wavefront = 64 everywhere; workgroup sizes are multiples of 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>
#include "sba_model.hpp"

namespace SBA_NS {

constexpr int PM_BLOCK = 256;      // threads (= max observations) of a point-major workgroup
constexpr int CM_WAVE_OBS = 64;    // observations a wave turns into one 128-row MFMA tile
constexpr int CM_CHUNK = 1024;     // observations per camera-major workgroup (256 threads x 4)
constexpr int SCHUR_PTS = 16;      // points per panel chunk  (K = 48 panel rows)
constexpr int SCHUR_K = 3 * SCHUR_PTS;
constexpr int GROUP_CAMS = 16;     // cameras per Schur camera group: 16*11 = 176 = 11 MFMA tiles exactly (16*13 = 208 = 13 tiles)
constexpr int GROUP_ROWS = GROUP_CAMS * NCP;   // 176 (208)
constexpr int GROUP_TILES = GROUP_ROWS / 16;   // 11 (13)
constexpr int NSCAL = 8;
// k_schur_fused_wide (sba_schur_wide.hpp: 17 .. 23 cameras in one launch): compact parameter-major rows e*C + c in NTW = ceil(11 C / 16)
// tiles, slabs [workgroup][WIDE_SLOTS tiles][64 lanes][4], row partials with a stride of WIDE_ROWS
constexpr int WIDE_ROWS = 256;
constexpr int WIDE_MAX_NTW = 16;
constexpr int WIDE_SLOTS = WIDE_MAX_NTW * (WIDE_MAX_NTW + 1) / 2;
// (tile counts the kernel is instantiated for: 8, 12, 13, 14, 15, 16 -- a smaller rig runs the next larger one, its surplus tiles stay zero)
__host__ __device__ constexpr int wide_ntw(int C) { const int t = (C * NCP + 15) / 16; return t <= 8 ? 8 : t <= 12 ? 12 : t; }
__host__ __device__ inline void wide_tile_rc(int ntw, int t, int& R, int& Tc) {      // row-major enumeration of the upper tile triangle
  int r = 0, rem = t;
  while (rem >= ntw - r) { rem -= ntw - r; ++r; }
  R = r; Tc = r + rem;
}

// Current / trial parameter buffers: both sets travel BY VALUE as a kernel argument (no extra dependent load), and
// a kernel picks its side from LMState::cur, which decide_core flips when a step is accepted -- so the host never has
// to learn the outcome of an iteration before enqueueing the next one.  `base` is the side that was current when
// the solve began (or simply the current side for launches outside the LM loop, where st == nullptr).
template <typename T> struct ParamSets {
  double* cams[2]; double* pts[2]; T* ptsT[2]; T* campre[2];
  int base;
  T loss_delta;                     // (the engine's real type: the same rounded value everywhere, launch_residual included) > 0: robust loss with this f_scale on every residual component (sba_set_robust_loss); 0: linear
  int loss_kind;                    // sba_loss: which rho (Huber, soft-l1, Cauchy) when loss_delta > 0
  __host__ __device__ RLoss<T> loss() const { return RLoss<T>{loss_delta, loss_kind}; }
  const unsigned char* fixed;       // per point: 1 = held fixed (gauge anchor, sba_set_fixed_points); NULL: every point is free
};
template <typename T> __device__ inline bool pt_fixed(const ParamSets<T>& ps, size_t p) { return ps.fixed != nullptr && ps.fixed[p] != 0; }

struct LMState {
  double lam, nu;
  double cost, cost_new, pred, rho, actual;
  double step_norm, x_norm, gnorm;
  double ftol, xtol, gtol;
  double pred_c, dx2_c, x2_c, gmax_c;
  double lam_min, lam_max;
  long long nfev, njev, max_nfev;
  int status;        // -1: keep iterating; else scipy status code.  Every LM kernel is a no-op once it is >= 0.
  int accepted;      // decision of the last trial
  int chol_fail;
  int fresh;         // a new linearization is waiting to be absorbed into the camera scaling
  int iter, n_accepted;
  int free_cams;     // 0: points-only mode
  int need_lin;      // the next linearize launches do work (set by k_decide: accepted, or bench's always_relinearize)
  int always_relin;
  int max_iter;      // > 0: stop (status 0) after this many trial steps
  int cur;           // parity of accepted steps (which host-side buffer pair is "current")
  int comm_fail;     // set by k_ipc_gate when a peer rank did not reach an exchange in time (sba_ipc.hpp): the solve stops, the host reports it
  int chol_f64_retries;   // fp32 engine: reduced systems whose f32 factorisation was refused and that were factored again in f64 (k_cholesky_blocked)
  int chol_retry;         // the f32-lane factorisation of 177 .. 256 unknowns refused this system: the f64 kernel launched behind it takes over
};
__device__ inline bool lm_done(const LMState* st) { return st != nullptr && st->status >= 0; }
template <typename T> __device__ inline int ps_cur(const ParamSets<T>& ps, const LMState* st) { return (ps.base ^ (st ? st->cur : 0)) & 1; }

template <typename T> struct Vec2;
template <> struct Vec2<double> { using type = double2; };
template <> struct Vec2<float> { using type = float2; };

// ------------------------------------------------------------------ small helpers
__device__ inline double fmax_pos(double d) { return d > 0.0 ? d : 1.0; }   // scale 0 -> 1 (scipy common.py:598-610)
// Wave-wide sum / maximum of a double, result in every lane.  Inside a row of 16 lanes: four DPP steps (two v_mov_dpp + one f64
// op each); across the four rows: v_readlane of the row totals and three more operations.  (__shfl_xor on a double is two
// ds_bpermute_b32 per step and six dependent steps: ~800 cycles per value against ~150.)
template <int CTRL> __device__ __forceinline__ double dpp_mov_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_f64(double v, int src) {          // wave-uniform copy of lane `src` (compile-time constant)
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// sum over the four 16-lane rows (lanes l, l^16, l^32, l^48), result in all four: the gfx950 row-swap instructions, two per dword
// and step, instead of __shfl_xor's ds_bpermute round trips (tools/micro/permlane_swap.hip: 113 vs 189 cycles per dependent sum)
__device__ __forceinline__ double xrow_sum(double v) {
  // Written as inline assembly with explicit wait states around every swap: the compiler (ROCm 7.2 clang) does not insert the ones
  // these instructions need -- 88 swaps generated from the builtins in a row produced wrong sums in k_schur_fused_bf3's hand-over
  // experiment and right ones with the s_nops; a single sum from the builtins happened to be right here.
  unsigned lo0 = __double2loint(v), hi0 = __double2hiint(v), lo1 = lo0, hi1 = hi0;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 1"
               : "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1));
  const double p = __hiloint2double(hi0, lo0) + __hiloint2double(hi1, lo1);
  lo0 = __double2loint(p); hi0 = __double2hiint(p); lo1 = lo0; hi1 = hi0;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1"
               : "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1));
  return __hiloint2double(hi0, lo0) + __hiloint2double(hi1, lo1);
}
__device__ inline double wave_sum(double v) {
  v += dpp_mov_f64<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_mov_f64<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_mov_f64<0x141>(v);     // row_half_mirror
  v += dpp_mov_f64<0x140>(v);     // row_mirror: every lane of a row holds the row total
  return (lane_f64(v, 0) + lane_f64(v, 16)) + (lane_f64(v, 32) + lane_f64(v, 48));
}
__device__ inline double wave_max(double v) {
  v = fmax(v, dpp_mov_f64<0xB1>(v));
  v = fmax(v, dpp_mov_f64<0x4E>(v));
  v = fmax(v, dpp_mov_f64<0x141>(v));
  v = fmax(v, dpp_mov_f64<0x140>(v));
  return fmax(fmax(lane_f64(v, 0), lane_f64(v, 16)), fmax(lane_f64(v, 32), lane_f64(v, 48)));
}
// block-wide sum of a double; result valid in thread 0.  scratch: >= blockDim/64 doubles of LDS.
__device__ inline double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) s += scratch[i];
  return s;
}

// ------------------------------------------------------------------ accept / reject / terminate (device function)
// One iteration-log row per trial step, same columns scipy prints with verbose=2 (layout = sba_lm_iter_log).
struct LMLogRow { int iteration; int accepted; long long nfev; double cost, cost_reduction, step_norm, optimality, lambda, rho; };
static_assert(sizeof(LMState) % 4 == 0, "LMState is copied word by word");

// per-thread partial sums of the trial scalars (single rank: folded here from the per-workgroup partials of the trial and
// linearisation kernels; multi-rank: the gathered scalars are read by thread 0 in decide_core)
struct DecidePartials { double a = 0, b = 0, c = 0, d = 0, g = 0; };
__device__ inline void decide_gather(DecidePartials& p, const double* __restrict__ scal_all, const double* __restrict__ trial_part,
                                     const double* __restrict__ gmax_part, int nblk, int n_gmax) {
  if (scal_all != nullptr) return;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
    const double t0 = trial_part[i], t1 = trial_part[nblk + i], t2 = trial_part[2 * nblk + i], t3 = trial_part[3 * nblk + i];
    p.a += t0; p.b += t1; p.c += t2; p.d += t3;
  }
  for (int i = threadIdx.x; i < n_gmax; i += blockDim.x) p.g = fmax(p.g, gmax_part[i]);
}

// The accept/reject/terminate decision of one trial step on a copy `st` of the state record in LDS.  Called by every thread
// of the workgroup (it contains one barrier); thread 0 updates the record and fills *row; the caller adds a barrier before
// other threads read either.  Returns (thread 0) whether the row belongs into the log.  Fixed summation order: every
// workgroup that runs this on the same inputs reaches the same decision bit for bit.
// Five block-wide reductions (four sums, one maximum) of the per-thread partials through LDS: every thread deposits its partials,
// wave k folds array k 8 : 1 three times (512 -> 64 -> 8 -> 1) in a fixed order; result in out[0..4] of thread 0.  Called by all
// DECIDE_THREADS threads of the workgroup (two barriers).  Every caller -- k_decide, k_trial_scalars, the prologue of
// k_schur_fused_bf3 -- gathers with the same stride and folds in the same order, so they produce the same bits from the same
// partials.  (Wave-level shuffles of doubles -- two ds_bpermute per step and value, six steps, five values -- took 4.2k cycles.)
constexpr int DECIDE_THREADS = 512;
__device__ inline void decide_fold(const DecidePartials& p, double* __restrict__ scr /* LDS, 5 x DECIDE_THREADS doubles */, double (&out)[5]) {
  __shared__ double s_red[5];
  constexpr int nt = DECIDE_THREADS;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  scr[tid] = p.a; scr[nt + tid] = p.b; scr[2 * nt + tid] = p.c; scr[3 * nt + tid] = p.d; scr[4 * nt + tid] = p.g;
  __syncthreads();
  if (wv < 5) {
    double* a = scr + wv * nt;
    const bool mx = wv == 4;
    double v = 0;
#pragma unroll
    for (int j = 0; j < nt / 64; ++j) { const double x = a[lane + 64 * j]; v = mx ? fmax(v, x) : v + x; }
    a[lane] = v;
    __builtin_amdgcn_wave_barrier();
    if (lane < 8) {
      double u = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const double x = a[lane * 8 + j]; u = mx ? fmax(u, x) : u + x; }
      a[lane * 8] = u;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      double u = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const double x = a[j * 8]; u = mx ? fmax(u, x) : u + x; }
      s_red[wv] = u;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) out[k] = s_red[k];
  }
}

__device__ inline bool decide_core(LMState* st, const DecidePartials& p, const double* __restrict__ scal_all, int n_ranks,
                                   LMLogRow* row, int log_cap, double* __restrict__ scr /* LDS scratch, 5 x DECIDE_THREADS doubles */,
                                   long long* dbgp = nullptr /* diagnostic cycle stamps */) {
  double cost_new = 0, pred = 0, dx2 = 0, x2 = 0, gmax = 0, failv = 0;
  if (scal_all == nullptr) {
    double f5[5] = {0, 0, 0, 0, 0};
    decide_fold(p, scr, f5);
    cost_new = f5[0]; pred = f5[1]; dx2 = f5[2]; x2 = f5[3]; gmax = f5[4];
    failv = (double)st->chol_fail;
  }
  if (threadIdx.x != 0) return false;
  if (dbgp) dbgp[0] = clock64();
  // thread 0 works on a register copy of the record: as a chain of dependent LDS reads and writes the same logic took 6.5k cycles
  LMState* const st_lds = st;
  LMState L = *st_lds;
  st = &L;
  if (scal_all != nullptr) {
    for (int r = 0; r < n_ranks; ++r) {
      const double* s = scal_all + (size_t)r * NSCAL;
      cost_new += s[0]; pred += s[1]; dx2 += s[2]; x2 += s[3];
      gmax = fmax(gmax, s[4]); failv = fmax(failv, s[5]);
    }
  }
  pred += st->pred_c; dx2 += st->dx2_c; x2 += st->x2_c; gmax = fmax(gmax, st->gmax_c);
  st->gnorm = gmax;
  st->step_norm = sqrt(dx2);
  st->x_norm = sqrt(x2);
  st->cost_new = cost_new;
  st->pred = pred;
  st->iter += 1;
  int status = -1;
  int accepted = 0;
  double actual = 0, rho = 0;
  if (gmax < st->gtol) {                      // scipy trf.py:452 tests this before taking a step
    status = 1;
  } else {
    st->nfev += 1;
    const bool ok = !(failv > 0) && isfinite(cost_new) && pred > 0;
    actual = ok ? st->cost - cost_new : -1.0;
    rho = ok ? actual / pred : -1.0;
    if (ok) {                                  // scipy common.py:705-717
      const bool f_ok = actual < st->ftol * st->cost && rho > 0.25;
      const bool x_ok = sqrt(dx2) < st->xtol * (st->xtol + sqrt(x2));
      status = (f_ok && x_ok) ? 4 : f_ok ? 2 : x_ok ? 3 : -1;
    }
    if (actual > 0) {
      const double t = 2.0 * rho - 1.0;
      const double l = st->lam * fmax(1.0 / 3.0, 1.0 - t * t * t);
      st->lam = fmin(fmax(l, st->lam_min), st->lam_max);
      st->nu = 2.0;
      accepted = 1;
      st->n_accepted += 1;
      st->cost = cost_new;        // refreshed again from the exchange buffer after the next linearization
      st->njev += 1;              // the accepted point gets a new Jacobian (scipy counts it the same way)
      st->fresh = 1;
      st->cur ^= 1;               // the trial point becomes the current one
    } else {
      st->lam = fmin(st->lam * st->nu, st->lam_max);
      st->nu *= 2.0;
    }
    if (status < 0 && st->nfev >= st->max_nfev) status = 0;
  }
  if (status < 0 && st->max_iter > 0 && st->iter >= st->max_iter) status = 0;
  st->accepted = accepted; st->actual = actual; st->rho = rho;
  st->need_lin = (accepted || st->always_relin) ? 1 : 0;
  st->status = status;
  row->iteration = st->iter; row->accepted = accepted; row->nfev = st->nfev; row->cost = st->cost;
  row->cost_reduction = actual; row->step_norm = st->step_norm; row->optimality = gmax; row->lambda = st->lam; row->rho = rho;
  *st_lds = L;
  if (dbgp) dbgp[1] = clock64();
  return st->iter <= log_cap;
}
__device__ inline double block_max(double v, double* scratch) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) s = fmax(s, scratch[i]);
  return s;
}

// 3x3 SPD:  V' = V + lam*D  ->  Linv (lower, 6 values: l00,l10,l11,l20,l21,l22 of L^-1).  ok=false if not PD.
template <typename T>
__device__ inline bool chol3_inv(const T v[6] /*v00,v01,v02,v11,v12,v22*/, T li[6]) {
  const T a00 = v[0], a10 = v[1], a20 = v[2], a11 = v[3], a21 = v[4], a22 = v[5];
  if (!(a00 > (T)0)) return false;
  const T l00 = sqrt(a00);
  const T i00 = (T)1 / l00;
  const T l10 = a10 * i00, l20 = a20 * i00;
  const T d11 = a11 - l10 * l10;
  if (!(d11 > (T)0)) return false;
  const T l11 = sqrt(d11);
  const T i11 = (T)1 / l11;
  const T l21 = (a21 - l20 * l10) * i11;
  const T d22 = a22 - l20 * l20 - l21 * l21;
  if (!(d22 > (T)0)) return false;
  const T l22 = sqrt(d22);
  const T i22 = (T)1 / l22;
  // inverse of lower-triangular L
  li[0] = i00;
  li[1] = -l10 * i00 * i11;
  li[2] = i11;
  li[3] = (-l20 * i00 - l21 * li[1]) * i22;   // -(l20*li00 + l21*li10)/l22
  li[4] = -l21 * i11 * i22;
  li[5] = i22;
  return true;
}

// ------------------------------------------------------------------ K0: CamPre table
template <typename T>
__global__ void k_cam_prep(const double* __restrict__ cams, T* __restrict__ campre, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) campre_build<T>(cams + (size_t)c * NCP, campre + (size_t)c * CAMPRE);
}

template <typename T>
__device__ inline void stage_campre(const T* __restrict__ campre, T* __restrict__ s_cam, int C) {
  for (int i = threadIdx.x; i < C * CAMPRE; i += blockDim.x) s_cam[i] = campre[i];
}

// ------------------------------------------------------------------ stateless rotate / project (gathered rows)
template <typename T>
__global__ void k_rotate_rows(const double* __restrict__ pts, const double* __restrict__ rv,
                              double* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T P0, P1, P2;
  rotate_raw<T>((T)rv[3 * i], (T)rv[3 * i + 1], (T)rv[3 * i + 2], (T)pts[3 * i], (T)pts[3 * i + 1],
                (T)pts[3 * i + 2], P0, P1, P2);
  out[3 * i] = P0; out[3 * i + 1] = P1; out[3 * i + 2] = P2;
}

template <typename T>
__global__ void k_project_rows(const double* __restrict__ pts, const double* __restrict__ cam,
                               double* __restrict__ uv, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* c = cam + NCP * i;
  T P0, P1, P2;
  rotate_raw<T>((T)c[0], (T)c[1], (T)c[2], (T)pts[3 * i], (T)pts[3 * i + 1], (T)pts[3 * i + 2], P0, P1, P2);
  const T p0 = P0 + (T)c[3], p1 = P1 + (T)c[4], p2 = P2 + (T)c[5];
  const T x = p0 / p2, y = p1 / p2;
  const T nn = x * x + y * y;
  const T r = (T)1 + (T)c[7] * nn + (T)c[8] * nn * nn;
  if constexpr (TANGENTIAL) {
    const T tp1 = (T)c[CP_P1], tp2 = (T)c[CP_P2], xy2 = (T)2 * x * y;
    uv[2 * i] = (x * r + tp1 * xy2 + tp2 * (nn + (T)2 * x * x)) * (T)c[6] + (T)c[CP_CX];
    uv[2 * i + 1] = (y * r + tp1 * (nn + (T)2 * y * y) + tp2 * xy2) * (T)c[6] + (T)c[CP_CY];
  } else {
    uv[2 * i] = x * (r * (T)c[6]) + (T)c[9];
    uv[2 * i + 1] = y * (r * (T)c[6]) + (T)c[10];
  }
}

// ------------------------------------------------------------------ device-side layout of a dense observation list (sba_upload)
// The caller's raw arrays (float64 pixels / weights, int64 indices; scripts/get_points3d.py:73-86 order) -> device layout.
// Observation i must be (point i / C, camera i % C); any deviation raises the flag and the host path takes over.
// Several camera groups: where a point's observations of one group sit in the point-major list.  For every (group g, point p)
// gmask[g*N + p] has bit c set when camera 16 g + c sees p and gstart[g*N + p] is the index of the first such observation, so
// that a Schur producer lane (point, camera) finds its observation at gstart + popcount(mask below c) instead of scanning the
// point's whole observation list for members of its group.  Requires strictly ascending cameras inside every point (no
// duplicate (point, camera) pair); *bad is set otherwise and the kernels keep scanning.
__global__ void k_group_index(const int32_t* __restrict__ ci, const int32_t* __restrict__ pt_start, int N, int ngroups,
                              uint16_t* __restrict__ gmask, int32_t* __restrict__ gstart, int* __restrict__ bad) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const int a = pt_start[p], b = pt_start[p + 1];
  int g = 0, prev = -1;
  unsigned m = 0;
  int first = a;
  bool ok = true;
  auto flush = [&](int upto) {          // close groups g .. upto-1
    for (; g < upto; ++g) { gmask[(size_t)g * N + p] = (uint16_t)m; gstart[(size_t)g * N + p] = first; m = 0; }
  };
  for (int o = a; o < b; ++o) {
    const int c = ci[o];
    if (c <= prev) ok = false;
    prev = c;
    const int cg = c / GROUP_CAMS;
    if (cg > g) { flush(cg); first = o; }
    if (cg == g) m |= 1u << (c - cg * GROUP_CAMS);
  }
  flush(ngroups);
  if (!ok) *bad = 1;
}

template <typename T>
__global__ void k_upload_dense(const double2* __restrict__ uv, const long long* __restrict__ ci, const long long* __restrict__ pi,
                               const double* __restrict__ w, int C, int N, long long M,
                               typename Vec2<T>::type* __restrict__ uv_pm, int32_t* __restrict__ ci_pm, int32_t* __restrict__ pi_pm,
                               T* __restrict__ w_pm, typename Vec2<T>::type* __restrict__ uv_cm, int32_t* __restrict__ pi_cm,
                               T* __restrict__ w_cm, int32_t* __restrict__ pt_start, int* __restrict__ flag) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= N) pt_start[i] = (int32_t)(i * C);
  if (i >= M) return;
  const int p = (int)(i / C), c = (int)(i - (long long)p * C);
  if (pi[i] != p || ci[i] != c) *flag = 1;
  const double2 m = uv[i];
  typename Vec2<T>::type v; v.x = (T)m.x; v.y = (T)m.y;
  uv_pm[i] = v; ci_pm[i] = c; pi_pm[i] = p;
  const size_t d = (size_t)c * N + p;                 // camera-major position: points ascending inside a camera
  uv_cm[d] = v; pi_cm[d] = p;
  if (w) { const T ww = (T)w[i]; w_pm[i] = ww; w_cm[d] = ww; }
}

// ------------------------------------------------------------------ K1: residual (+ cost partial)
// One thread per observation in pm order; grid-strided is unnecessary: grid = ceil(M/256).
// r_out (if non-null) is written in pm order as T2 -> fully coalesced.
template <typename T>
__global__ __launch_bounds__(PM_BLOCK) void k_residual(
    const T* __restrict__ campre, int C, const T* __restrict__ ptsT,
    const typename Vec2<T>::type* __restrict__ uv, const T* __restrict__ w,
    const int32_t* __restrict__ ci, const int32_t* __restrict__ pi, int64_t M,
    typename Vec2<T>::type* __restrict__ r_out, double* __restrict__ cost_part, RLoss<T> loss) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* s_cam = reinterpret_cast<T*>(smem);
  __shared__ double s_red[PM_BLOCK / 64];
  stage_campre(campre, s_cam, C);
  __syncthreads();
  const int64_t o = (int64_t)blockIdx.x * PM_BLOCK + threadIdx.x;
  double sq = 0;
  if (o < M) {
    const int c = ci[o];
    const int p = pi[o];
    const auto m = uv[o];
    const T ww = w ? w[o] : (T)1;
    T u, v;
    obs_project<T>(s_cam + c * CAMPRE, ptsT[3 * (size_t)p], ptsT[3 * (size_t)p + 1], ptsT[3 * (size_t)p + 2], u, v);
    const T r0 = ww * (u - m.x), r1 = ww * (v - m.y);
    if (r_out) { typename Vec2<T>::type rr; rr.x = r0; rr.y = r1; r_out[o] = rr; }
    sq = (loss.delta > (T)0) ? (double)robust_cost<T>(loss, r0, r1) : (double)r0 * (double)r0 + (double)r1 * (double)r1;
  }
  const double s = block_sum(sq, s_red);
  if (threadIdx.x == 0) cost_part[blockIdx.x] = 0.5 * s;
}

// ------------------------------------------------------------------ K2: materialising residual + Jacobian
// Writes r (T2), Jc (22 T) and Jp (6 T) per observation in pm order.  The 28 Jacobian values of a
// lane are staged through LDS so that global stores are contiguous across the wave.
template <typename T>
__global__ __launch_bounds__(PM_BLOCK) void k_resjac(
    const T* __restrict__ campre, int C, const T* __restrict__ ptsT,
    const typename Vec2<T>::type* __restrict__ uv, const T* __restrict__ w,
    const int32_t* __restrict__ ci, const int32_t* __restrict__ pi, int64_t M,
    typename Vec2<T>::type* __restrict__ r_out, T* __restrict__ Jc_out, T* __restrict__ Jp_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* s_cam = reinterpret_cast<T*>(smem);
  constexpr int JW = 2 * NCP + 6, JS = JW + 1;        // 28 (32) Jacobian values per observation, odd LDS stride 29 (33)
  T* s_j = s_cam + ((C * CAMPRE + 3) & ~3);          // [256][JS] (odd stride: conflict-free lane writes)
  stage_campre(campre, s_cam, C);
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * PM_BLOCK;
  const int64_t o = base + threadIdx.x;
  const int nvalid = (int)min((int64_t)PM_BLOCK, M - base);
  if (o < M) {
    const int c = ci[o];
    const int p = pi[o];
    const auto m = uv[o];
    const T ww = w ? w[o] : (T)1;
    T r[2], Jc[2][NCP], Jp[2][3];
    obs_resjac<T>(s_cam + c * CAMPRE, ptsT[3 * (size_t)p], ptsT[3 * (size_t)p + 1], ptsT[3 * (size_t)p + 2],
                  m.x, m.y, ww, r, Jc, Jp);
    if (r_out) { typename Vec2<T>::type rr; rr.x = r[0]; rr.y = r[1]; r_out[o] = rr; }
    T* dst = s_j + threadIdx.x * JS;
#pragma unroll
    for (int k = 0; k < NCP; ++k) { dst[k] = Jc[0][k]; dst[NCP + k] = Jc[1][k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { dst[2 * NCP + k] = Jp[0][k]; dst[2 * NCP + 3 + k] = Jp[1][k]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nvalid * 2 * NCP; i += PM_BLOCK) {
    const int l = i / (2 * NCP), k = i - l * 2 * NCP;
    Jc_out[base * 2 * NCP + i] = s_j[l * JS + k];
  }
  for (int i = threadIdx.x; i < nvalid * 6; i += PM_BLOCK) {
    const int l = i / 6, k = i - l * 6;
    Jp_out[base * 6 + i] = s_j[l * JS + 2 * NCP + k];
  }
}

// ------------------------------------------------------------------ K3a: linearize, point side
// Workgroup b owns points [blk_pt[b], blk_pt[b+1]) and all their observations (<= 256, one per thread).
// Outputs per point: V (6: v00 v01 v02 v11 v12 v22), gp (3), D2p <- max(D2p, diag V); per block:
// cost partial and max|gp| partial.
template <typename T>
__global__ __launch_bounds__(PM_BLOCK) void k_linearize_points(
    const ParamSets<T> ps, const LMState* __restrict__ st, int C,
    const typename Vec2<T>::type* __restrict__ uv, const T* __restrict__ w,
    const int32_t* __restrict__ ci, const int32_t* __restrict__ pi, const int32_t* __restrict__ pt_start,
    const int4* __restrict__ blk_desc /* {p_lo, p_hi, o_lo, o_hi} per block */, double* __restrict__ V,
    double* __restrict__ gp, double* __restrict__ D2p, double* __restrict__ cost_part, double* __restrict__ gmax_part) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (st && (st->status >= 0 || !st->need_lin)) return;
  const int cur_ = ps_cur(ps, st);
  const T* __restrict__ campre = ps.campre[cur_];
  const T* __restrict__ ptsT = ps.ptsT[cur_];
  double* s_red = reinterpret_cast<double*>(smem);              // [256][9]
  T* s_cam = reinterpret_cast<T*>(s_red + PM_BLOCK * 9);
  __shared__ double s_scr[PM_BLOCK / 64];
  const int4 bd = blk_desc[blockIdx.x];
  stage_campre(campre, s_cam, C);
  const int p_lo = bd.x, p_hi = bd.y;
  const int o_lo = bd.z, o_hi = bd.w;
  const int nobs = o_hi - o_lo;
  __syncthreads();
  double sq = 0;
  if ((int)threadIdx.x < nobs) {
    const int o = o_lo + threadIdx.x;
    const int p = pi[o];
    const int c = ci[o];
    const auto m = uv[o];
    const T ww = w ? w[o] : (T)1;
    T r[2], Jc[2][NCP], Jp[2][3];
    obs_resjac<T>(s_cam + c * CAMPRE, ptsT[3 * (size_t)p], ptsT[3 * (size_t)p + 1], ptsT[3 * (size_t)p + 2],
                  m.x, m.y, ww, r, Jc, Jp);
    if (ps.loss_delta > 0.f) sq = (double)robust_apply<T>(ps.loss(), r, Jc, Jp);
    else sq = (double)r[0] * r[0] + (double)r[1] * r[1];
    double* d = s_red + threadIdx.x * 9;
    d[0] = (double)Jp[0][0] * Jp[0][0] + (double)Jp[1][0] * Jp[1][0];
    d[1] = (double)Jp[0][0] * Jp[0][1] + (double)Jp[1][0] * Jp[1][1];
    d[2] = (double)Jp[0][0] * Jp[0][2] + (double)Jp[1][0] * Jp[1][2];
    d[3] = (double)Jp[0][1] * Jp[0][1] + (double)Jp[1][1] * Jp[1][1];
    d[4] = (double)Jp[0][1] * Jp[0][2] + (double)Jp[1][1] * Jp[1][2];
    d[5] = (double)Jp[0][2] * Jp[0][2] + (double)Jp[1][2] * Jp[1][2];
    d[6] = (double)Jp[0][0] * r[0] + (double)Jp[1][0] * r[1];
    d[7] = (double)Jp[0][1] * r[0] + (double)Jp[1][1] * r[1];
    d[8] = (double)Jp[0][2] * r[0] + (double)Jp[1][2] * r[1];
  }
  __syncthreads();
  double gmax = 0;
  const int npts = p_hi - p_lo;
  for (int item = threadIdx.x; item < npts * 9; item += PM_BLOCK) {
    const int q = item / 9, e = item - q * 9;
    const int a = pt_start[p_lo + q] - o_lo, b = pt_start[p_lo + q + 1] - o_lo;
    double s = 0;
    for (int k = a; k < b; ++k) s += s_red[k * 9 + e];
    const size_t p = (size_t)(p_lo + q);
    if (e < 6) {
      V[p * 6 + e] = s;
      const int dsel = (e == 0) ? 0 : (e == 3) ? 1 : (e == 5) ? 2 : -1;
      if (dsel >= 0) D2p[p * 3 + dsel] = fmax(D2p[p * 3 + dsel], s);
    } else {
      gp[p * 3 + (e - 6)] = s;
      if (!pt_fixed(ps, p)) gmax = fmax(gmax, fabs(s));      // a fixed point is not an unknown: its gradient does not count
    }
  }
  const double cs = block_sum(sq, s_scr);
  const double gm = block_max(gmax, s_scr);
  if (threadIdx.x == 0) { cost_part[blockIdx.x] = 0.5 * cs; gmax_part[blockIdx.x] = gm; }
}

// ------------------------------------------------------------------ MFMA wrappers (16x16x4, A and B one value per lane)
//   f64: v_mfma_f64_16x16x4_f64, D rows = (lane>>4) + 4*reg, col = lane&15
//   f32: v_mfma_f32_16x16x4_f32, D rows = 4*(lane>>4) + reg, col = lane&15
template <typename T> struct Mfma;
template <> struct Mfma<double> {
  using acc_t = __attribute__((ext_vector_type(4))) double;
  __device__ static inline acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  __device__ static inline int row_of(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct Mfma<float> {
  using acc_t = __attribute__((ext_vector_type(4))) float;
  __device__ static inline acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  __device__ static inline int row_of(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

// ------------------------------------------------------------------ K3b: linearize, camera side
// Chunk = up to CM_CHUNK observations of ONE camera (cm order).  Each wave processes 64 observations
// at a time: every lane writes the 2 rows [Jc(11) | r | 0 0 0 0] of its observation into a 128x16
// LDS tile, then 32 MFMAs (A = B = tile fragment) accumulate tile^T * tile = [[U, g],[g^T, r.r]]
// (16x16) in 4 accumulator registers.  No shuffles, no atomics; one 16x16 partial per wave.
template <typename T>
__global__ __launch_bounds__(256) void k_linearize_cams(
    const ParamSets<T> ps, const LMState* __restrict__ st,
    const typename Vec2<T>::type* __restrict__ uv_cm, const T* __restrict__ w_cm,
    const int32_t* __restrict__ pi_cm, const int32_t* __restrict__ chunk_cam,
    const int32_t* __restrict__ chunk_begin, const int32_t* __restrict__ chunk_end,
    double* __restrict__ Upart /* [n_chunks][256] */) {
  constexpr int LD = 130;                       // [16 params][128 rows + 2]: conflict-free lane writes and MFMA reads
  __shared__ T s_tile[4][16 * LD];
  __shared__ T s_cam[CAMPRE];
  using M_ = Mfma<T>;
  if (st && (st->status >= 0 || !st->need_lin)) return;
  const int cur_ = ps_cur(ps, st);
  const T* __restrict__ campre = ps.campre[cur_];
  const T* __restrict__ ptsT = ps.ptsT[cur_];
  const int chunk = blockIdx.x;
  const int cam = chunk_cam[chunk];
  const int beg = chunk_begin[chunk], end = chunk_end[chunk];
  if (threadIdx.x < CAMPRE) s_cam[threadIdx.x] = campre[(size_t)cam * CAMPRE + threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  T* tile = s_tile[wid];
  typename M_::acc_t acc = {0, 0, 0, 0};
  for (int o0 = beg + wid * 64; o0 < end; o0 += 256) {
    const int o = o0 + lane;
    T row0[16], row1[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { row0[k] = 0; row1[k] = 0; }
    if (o < end) {
      const int p = pi_cm[o];
      const auto m = uv_cm[o];
      const T ww = w_cm ? w_cm[o] : (T)1;
      T r[2], Jc[2][NCP], Jp[2][3];
      obs_resjac<T>(s_cam, ptsT[3 * (size_t)p], ptsT[3 * (size_t)p + 1], ptsT[3 * (size_t)p + 2], m.x, m.y, ww, r, Jc, Jp);
      if (ps.loss_delta > 0.f) (void)robust_apply<T>(ps.loss(), r, Jc, Jp);
#pragma unroll
      for (int k = 0; k < NCP; ++k) { row0[k] = Jc[0][k]; row1[k] = Jc[1][k]; }
      row0[NCP] = r[0]; row1[NCP] = r[1];
    }
    // tile[param][row], row = lane (u rows) and 64 + lane (v rows): row order is irrelevant to J^T J
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
  // fold the four waves' 16x16 partials through LDS (fixed order => deterministic), one partial per chunk
  __syncthreads();
  double* s_acc = reinterpret_cast<double*>(&s_tile[0][0]);      // reuse the tile storage: [4][256] doubles
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) s_acc[wid * 256 + M_::row_of(lane, rg) * 16 + (lane & 15)] = (double)acc[rg];
  __syncthreads();
  Upart[(size_t)chunk * 256 + threadIdx.x] =
      (s_acc[threadIdx.x] + s_acc[256 + threadIdx.x]) + (s_acc[512 + threadIdx.x] + s_acc[768 + threadIdx.x]);
}

// Sum the per-chunk 16x16 partials of each camera: U (C x 121), gc (C x 11).  grid = C, block = 1024
// (4 groups x 256 entries; group g takes chunks g, g+4, ... ; groups are folded through LDS in a fixed order).
__global__ __launch_bounds__(1024) void k_reduce_cams(const double* __restrict__ Upart,
                                                      const int32_t* __restrict__ cam_chunk_start /*C+1*/,
                                                      double* __restrict__ U, double* __restrict__ gc,
                                                      const LMState* __restrict__ st) {
  __shared__ double s_p[4][256];
  if (st && (st->status >= 0 || !st->need_lin)) return;
  const int c = blockIdx.x;
  const int e = threadIdx.x & 255, g = threadIdx.x >> 8;
  const int a = cam_chunk_start[c], b = cam_chunk_start[c + 1];
  double s0 = 0, s1 = 0;
  int k = a + g;
  for (; k + 4 < b; k += 8) { s0 += Upart[(size_t)k * 256 + e]; s1 += Upart[(size_t)(k + 4) * 256 + e]; }
  if (k < b) s0 += Upart[(size_t)k * 256 + e];
  s_p[g][e] = s0 + s1;
  __syncthreads();
  if (g == 0) {
    const double s = (s_p[0][e] + s_p[1][e]) + (s_p[2][e] + s_p[3][e]);
    const int i = e >> 4, j = e & 15;
    if (i < NCP && j < NCP) U[(size_t)c * NCP * NCP + i * NCP + j] = s;
    if (i < NCP && j == NCP) gc[(size_t)c * NCP + i] = s;
  }
}

// ------------------------------------------------------------------ per-point damped factor (once per LM trial)
// pf[p] = { L^-1 (6: l00 l10 l11 l20 l21 l22) of V_p + lam*D_p, z = L^-1 g_p (3), ok, 0, 0 }  in T, 12 values per point.
// Evaluated in double, stored narrowed: the Schur producers (one lane per observation) then read 12 T per lane
// instead of 12 doubles and run no sqrt/divide of their own.
constexpr int PF = 12;
template <typename T>
__device__ __forceinline__ void point_factor_one(const double* __restrict__ V, const double* __restrict__ gp, const double* __restrict__ D2p,
                                                 double lam, size_t p, T* __restrict__ pf, const unsigned char* __restrict__ fixed) {
  double v6[6], li[6];
  v6[0] = V[p * 6 + 0] + lam * fmax_pos(D2p[p * 3 + 0]);
  v6[1] = V[p * 6 + 1];
  v6[2] = V[p * 6 + 2];
  v6[3] = V[p * 6 + 3] + lam * fmax_pos(D2p[p * 3 + 1]);
  v6[4] = V[p * 6 + 4];
  v6[5] = V[p * 6 + 5] + lam * fmax_pos(D2p[p * 3 + 2]);
  T* o = pf + p * PF;
  // degenerate point, or a point held fixed: contributes nothing to the Schur complement, its step is zero in the back substitution
  if ((fixed && fixed[p]) || !chol3_inv<double>(v6, li)) {
#pragma unroll
    for (int k = 0; k < PF; ++k) o[k] = (T)0;
    return;
  }
  const double g0 = gp[p * 3], g1 = gp[p * 3 + 1], g2 = gp[p * 3 + 2];
#pragma unroll
  for (int k = 0; k < 6; ++k) o[k] = (T)li[k];
  o[6] = (T)(li[0] * g0);
  o[7] = (T)(li[1] * g0 + li[2] * g1);
  o[8] = (T)(li[3] * g0 + li[4] * g1 + li[5] * g2);
  o[9] = (T)1; o[10] = (T)0; o[11] = (T)0;
}
template <typename T>
__global__ void k_point_factor(const double* __restrict__ V, const double* __restrict__ gp, const double* __restrict__ D2p,
                               const LMState* __restrict__ st, int N, T* __restrict__ pf, const unsigned char* __restrict__ fixed) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N || st->status >= 0) return;
  point_factor_one<T>(V, gp, D2p, st->lam, (size_t)p, pf, fixed);
}

// ------------------------------------------------------------------ DPP row reductions (lane = (point, camera) kernels)
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row, result in every lane: xor 1, xor 2 (quad_perm), then mirror inside 8 and inside 16
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);     // row_half_mirror
  v += dpp_mov<0x140>(v);     // row_mirror
  return v;
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}
// sum over the 32 lanes of a wave half (two DPP rows), result in all 32: DPP row sum + one row swap (v_permlane16_swap; inline
// assembly with the wait states the instruction needs, see xrow_sum)
__device__ __forceinline__ float half32_sum(float v) {
  v = row16_sum(v);
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
// sum over all 64 lanes, result in every lane (a point per wave: k_backsub_dense<T, 64>)
__device__ __forceinline__ float wave64_sum(float v) {
  v = half32_sum(v);
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ double wave64_sum(double v) { return wave_sum(v); }
__device__ __forceinline__ double half32_sum(double v) {
  v = row16_sum(v);
  unsigned lo0 = __double2loint(v), hi0 = __double2hiint(v), lo1 = lo0, hi1 = hi0;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 1"
               : "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1));
  return __hiloint2double(hi0, lo0) + __hiloint2double(hi1, lo1);
}
// inclusive prefix sum over the 64 lanes of the wave: Hillis-Steele inside every DPP row (row_shr 1, 2, 4, 8; lanes shifted in from
// outside the row read 0), then the row totals travel on: row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3
__device__ __forceinline__ float wave_scan(float v) {
  auto dpp = [](float x, auto ctrl, auto rmask) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(rmask)::value, 0xf, true));
  };
  using I = std::integral_constant<int, 0xf>;
  v += dpp(v, std::integral_constant<int, 0x111>{}, I{});      // row_shr:1
  v += dpp(v, std::integral_constant<int, 0x112>{}, I{});      // row_shr:2
  v += dpp(v, std::integral_constant<int, 0x114>{}, I{});      // row_shr:4
  v += dpp(v, std::integral_constant<int, 0x118>{}, I{});      // row_shr:8
  v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});      // row_bcast:15 -> rows 1, 3
  v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});      // row_bcast:31 -> rows 2, 3
  return v;
}
// inclusive prefix sum of a double over the 64 lanes of the wave (wave_scan of sba_schur_wide.hpp on both dwords): Hillis-Steele
// inside every DPP row (row_shr 1, 2, 4, 8; lanes shifted in from outside the row read 0), then the row totals travel on
// (row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3)
__device__ __forceinline__ double wave_scan(double v) {
  auto dpp = [](double x, auto ctrl, auto rmask) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), decltype(ctrl)::value, decltype(rmask)::value, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), decltype(ctrl)::value, decltype(rmask)::value, 0xf, true);
    return __hiloint2double(hi, lo);
  };
  using I = std::integral_constant<int, 0xf>;
  v += dpp(v, std::integral_constant<int, 0x111>{}, I{});      // row_shr:1
  v += dpp(v, std::integral_constant<int, 0x112>{}, I{});      // row_shr:2
  v += dpp(v, std::integral_constant<int, 0x114>{}, I{});      // row_shr:4
  v += dpp(v, std::integral_constant<int, 0x118>{}, I{});      // row_shr:8
  v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});      // row_bcast:15 -> rows 1, 3
  v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});      // row_bcast:31 -> rows 2, 3
  return v;
}
// wave-uniform copy of lane `src` (a wave-uniform index)
__device__ __forceinline__ float lane_bcast(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); }
__device__ __forceinline__ double lane_bcast(double v, int src) { return lane_f64(v, src); }
// L^-1 of the 3x3 SPD matrix (v00,v01,v02,v11,v12,v22) with hardware rsq (1 ulp): no sqrt / divide sequences
__device__ __forceinline__ bool chol3_inv_fast(const float v[6], float li[6]) {
  // straight-line: a failing pivot is replaced by 1 so that everything stays finite, and the verdict is one flag at the
  // end (the caller discards li then) -- three nested early returns cost three exec-mask round trips per point
  const bool ok0 = v[0] > 0.f;
  const float i00 = __builtin_amdgcn_rsqf(ok0 ? v[0] : 1.f);
  const float l10 = v[1] * i00, l20 = v[2] * i00;
  const float d11 = v[3] - l10 * l10;
  const bool ok1 = d11 > 0.f;
  const float i11 = __builtin_amdgcn_rsqf(ok1 ? d11 : 1.f);
  const float l21 = (v[4] - l20 * l10) * i11;
  const float d22 = v[5] - l20 * l20 - l21 * l21;
  const bool ok2 = d22 > 0.f;
  const float i22 = __builtin_amdgcn_rsqf(ok2 ? d22 : 1.f);
  li[0] = i00;
  li[1] = -l10 * i00 * i11;
  li[2] = i11;
  li[3] = (-l20 * i00 - l21 * li[1]) * i22;
  li[4] = -l21 * i11 * i22;
  li[5] = i22;
  return ok0 && ok1 && ok2 && isfinite(i22) && isfinite(i11) && isfinite(i00);
}

// ------------------------------------------------------------------ K4: Schur complement partials (MFMA)
// Pair (ga, gb), ga <= gb, of camera groups (16 cameras = 176 rows each).  Each workgroup walks its slice of
// the point list in chunks of PTS points and is split into two roles that overlap through a double-buffered
// LDS panel (one workgroup barrier per chunk):
//   producer waves (lane = observation): residual/Jacobian blocks -> Ytilde_i = W_i L^-T (11x3) written into
//       the [K = 3*PTS][176] panel of the camera group(s), plus z = L^-1 g_p for the right-hand side;
//   consumer waves: rhs += panel^T z, then 16x16x4 MFMAs accumulate panel products into their output tiles.
// Partials go to slabs; k_build_exchange sums them in a fixed order (deterministic, no atomics).
//   DIAG  (ga == gb): one panel, 66 upper-triangular tiles   (grid = (ksplit, ngroups, TS))
//   !DIAG (ga <  gb): two panels, 121 tiles                  (grid = (ksplit, npairs - ngroups, TS))
//   The tiles of a pair are dealt to TS workgroups (grid.z) so that a consumer wave never holds more than ~72 accumulator
//   VGPRs; every workgroup of a split still builds the whole panel.
//   slab layout: [pair][ks][tile (121 slots)][lane 64][reg 4]   (acc type T; a lane's 4 registers are one 16-byte store)
//   bpart layout: [ga][ks][176] doubles (only diagonal pairs contribute)
// Flavours: k_schur (f32: 4 producer + 4 consumer waves), k_schur_sym (f64: all 8 waves produce, then all 8 consume),
// k_schur_fused (f32, dense visibility, one group: the linearisation rides in the producers) -- see DESIGN.md 4.
constexpr int SCHUR_THREADS = 512;                           // 4 producer waves + 4 consumer waves
template <typename T, bool DIAG> struct SchurCfg {
  using elem = T;
  static constexpr bool diag = DIAG;
  static constexpr int THREADS = SCHUR_THREADS;
  static constexpr int NPROD = THREADS / 2;                  // producer threads (first half of the workgroup)
  static constexpr int NCW = THREADS / 128;                  // consumer waves
  static constexpr int NTILE = DIAG ? (GROUP_TILES * (GROUP_TILES + 1)) / 2 : GROUP_TILES * GROUP_TILES;
  // tile split: the pair's tiles are dealt to TS workgroups (grid.z) so that a consumer wave never holds more than
  // ~17 f32 / ~9 f64 accumulator tiles (<= 72 VGPRs); every workgroup of a split still builds the whole panel
  static constexpr int TS = DIAG ? (sizeof(T) == 8 ? 2 : 1) : (sizeof(T) == 8 ? 4 : (NCP > 11 ? 3 : 2));
  static constexpr int NV = NCW * TS;                        // "virtual" consumer waves of a pair
  static constexpr int TPW = (NTILE + NV - 1) / NV;          // tiles per consumer wave: 17 / 9 / 16 / 8
  static constexpr int PTS = (!DIAG && (sizeof(T) == 8 || NCP > 11)) ? 8 : 16;   // points per chunk (LDS budget: 2 buffers x panels)
  static constexpr int K = 3 * PTS;
  static constexpr int NPANEL = DIAG ? 1 : 2;
  static constexpr int BUF = NPANEL * K * GROUP_ROWS + K;    // one buffer: panel(s) + z   (in T)
  static constexpr size_t LDS_BYTES = (size_t)(2 * BUF + 2 * GROUP_CAMS * CAMPRE) * sizeof(T);
};

// tiles [schur_lo(v), schur_lo(v+1)) of virtual wave v when ntile tiles are dealt to nv waves as evenly as possible
// (the first ntile % nv waves take one more); waves v and v + 4 share a SIMD, so the larger shares go to v = 0, 1, ...
__host__ __device__ constexpr int schur_lo(int ntile, int nv, int v) {
  const int base = ntile / nv, extra = ntile % nv;
  return v * base + (v < extra ? v : extra);
}
// row-major enumeration of a pair's tiles: upper triangle (R <= Tc) for a diagonal pair, all 11x11 otherwise
__host__ __device__ constexpr int schur_tile_R(bool diag, int t) {
  if (!diag) return t / GROUP_TILES;
  int R = 0, rem = t;
  while (rem >= GROUP_TILES - R) { rem -= GROUP_TILES - R; ++R; }
  return R;
}
__host__ __device__ constexpr int schur_tile_T(bool diag, int t) {
  if (!diag) return t % GROUP_TILES;
  int R = 0, rem = t;
  while (rem >= GROUP_TILES - R) { rem -= GROUP_TILES - R; ++R; }
  return R + rem;
}
__device__ inline void schur_tile_rc(bool diag, int t, int& R, int& Tc) { R = schur_tile_R(diag, t); Tc = schur_tile_T(diag, t); }
// compile-time loop: f(integral_constant<int, LO>) ... f(integral_constant<int, HI-1>)
template <int LO, int HI, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (LO < HI) { f(std::integral_constant<int, LO>{}); static_for<LO + 1, HI>(f); }
}

// Symmetric variant (used for f64): all 8 waves first build the panel of a chunk, then all 8 consume it.  f32/f64 MFMA
// and VALU instructions of the two waves of a SIMD do not overlap on gfx950 (tools/micro/mix_rate.hip: the times add),
// so specialising waves buys nothing beyond latency hiding, while the f64 accumulators of a pair (66 tiles x 8 VGPRs)
// only fit when they are spread over all 8 waves; v_mfma_f64_16x16x4 runs at 64 cycles per SIMD with LDS-fed
// operands (tools/micro/mfma64_lds.hip), which is the floor of the consume phase.
template <typename T, bool DIAG> struct SchurSymCfg {
  using elem = T;
  static constexpr bool diag = DIAG;
  static constexpr int THREADS = SCHUR_THREADS;
  static constexpr int NCW = THREADS / 64;                   // every wave consumes
  static constexpr int NTILE = DIAG ? (GROUP_TILES * (GROUP_TILES + 1)) / 2 : GROUP_TILES * GROUP_TILES;
  static constexpr int TS = (DIAG ? 1 : 2) * (NCP > 11 ? 2 : 1);
  static constexpr int NV = NCW * TS;
  static constexpr int TPW = (NTILE + NV - 1) / NV;          // 9 / 8 tiles per wave (13-parameter model: 6 / 6)
  static constexpr int PTS = (DIAG ? 32 : 16) / (NCP > 11 ? 2 : 1);   // one single-buffered chunk: 96 (48+48) panel rows; 208-column panels: half
  static constexpr int K = 3 * PTS;
  static constexpr int NPANEL = DIAG ? 1 : 2;
  static constexpr int BUF = NPANEL * K * GROUP_ROWS + K;
  static constexpr size_t LDS_BYTES = (size_t)(BUF + 2 * GROUP_CAMS * CAMPRE) * sizeof(T);
};

// which flavour runs: f64 -> symmetric, f32 -> producer/consumer specialised
template <typename T> constexpr bool SCHUR_SYM = sizeof(T) == 8;
// the point linearisation rides in k_schur_sym (LIN) only when a chunk has one lane per (point, camera): 32 points x 16 cameras
template <typename T> constexpr bool SCHUR_LIN_OK = SCHUR_SYM<T> && SchurSymCfg<T, true>::PTS == 32;
template <typename T, bool DIAG> using SchurSel = std::conditional_t<SCHUR_SYM<T>, SchurSymCfg<T, DIAG>, SchurCfg<T, DIAG>>;

// Consumer wave V of a pair owns the contiguous tile range [V*TPW, V*TPW+TPW): consecutive tiles share their row, so
// with V a compile-time constant one k-step needs only the distinct 16-row fragments (<= 11 for a diagonal pair, whose
// A and B operands are the same panel fragments; <= 3 + 11 otherwise), all MFMAs of a k-step are independent, and the
// next k-step's fragments are fetched while they issue.
template <typename Cfg, int V, int K, bool PARTIAL = false>
__device__ inline void schur_consume(const typename Cfg::elem* __restrict__ pla /* panelA + lane offset */,
                                     const typename Cfg::elem* __restrict__ plb,
                                     typename Mfma<typename Cfg::elem>::acc_t (&acc)[Cfg::TPW],
                                     int usedA = GROUP_TILES, int usedB = GROUP_TILES /* PARTIAL: 16-row tiles that hold cameras */) {
  using T = typename Cfg::elem;
  using M_ = Mfma<T>;
  constexpr bool DIAG = Cfg::diag;
  constexpr int LO = schur_lo(Cfg::NTILE, Cfg::NV, V), HI = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
  if constexpr (LO < HI) {
    constexpr int RMIN = schur_tile_R(DIAG, LO), RMAX = schur_tile_R(DIAG, HI - 1);
    T fa[2][GROUP_TILES], fb[2][GROUP_TILES];
    auto load = [&](int buf, int ks) {
      if constexpr (DIAG) {
#pragma unroll
        for (int b = RMIN; b < GROUP_TILES; ++b) fa[buf][b] = pla[ks * 4 * GROUP_ROWS + 16 * b];
      } else {
#pragma unroll
        for (int b = RMIN; b <= RMAX; ++b) fa[buf][b] = pla[ks * 4 * GROUP_ROWS + 16 * b];
#pragma unroll
        for (int b = 0; b < GROUP_TILES; ++b) fb[buf][b] = plb[ks * 4 * GROUP_ROWS + 16 * b];
      }
    };
    load(0, 0);
#pragma unroll
    for (int ks = 0; ks < K / 4; ++ks) {
      const int cur = ks & 1;
      if (ks + 1 < K / 4) load(cur ^ 1, ks + 1);
      __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of this step's MFMAs (the scheduler sinks it otherwise)
      static_for<LO, HI>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int R = schur_tile_R(DIAG, t), Tc = schur_tile_T(DIAG, t);
        // PARTIAL kernels (camera count not a multiple of 16): the last group leaves whole tiles empty -- skip them
        if (!PARTIAL || (R < usedA && Tc < usedB))
          acc[t - LO] = M_::mma(fa[cur][R], DIAG ? fa[cur][Tc] : fb[cur][Tc], acc[t - LO]);
      });
    }
  }
}

template <typename Cfg, int V>
__device__ inline void schur_store(typename Cfg::elem* __restrict__ slab, int lane,
                                   const typename Mfma<typename Cfg::elem>::acc_t (&acc)[Cfg::TPW]) {
  constexpr int LO = schur_lo(Cfg::NTILE, Cfg::NV, V), HI = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
#pragma unroll
  for (int t = LO; t < HI; ++t) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) slab[(size_t)t * 256 + lane * 4 + rg] = acc[t - LO][rg];      // one dwordx4 store per tile (f32)
  }
}

// dispatch on the (runtime) virtual wave index; NV <= 16
template <typename Cfg, int K, bool PARTIAL = false, int V = 0>
__device__ inline void schur_consume_v(int v, const typename Cfg::elem* pla, const typename Cfg::elem* plb,
                                       typename Mfma<typename Cfg::elem>::acc_t (&acc)[Cfg::TPW],
                                       int usedA = GROUP_TILES, int usedB = GROUP_TILES) {
  if constexpr (V < Cfg::NV) {
    if (v == V) schur_consume<Cfg, V, K, PARTIAL>(pla, plb, acc, usedA, usedB);
    else schur_consume_v<Cfg, K, PARTIAL, V + 1>(v, pla, plb, acc, usedA, usedB);
  }
}
template <typename Cfg, int V = 0>
__device__ inline void schur_store_v(int v, typename Cfg::elem* slab, int lane,
                                     const typename Mfma<typename Cfg::elem>::acc_t (&acc)[Cfg::TPW]) {
  if constexpr (V < Cfg::NV) {
    if (v == V) schur_store<Cfg, V>(slab, lane, acc);
    else schur_store_v<Cfg, V + 1>(v, slab, lane, acc);
  }
}

// Panel block of one observation from its Jacobian blocks: Ytilde = Jc^T (Jp L^-T)  (11x3), rows 3q..3q+2, columns col0..
// f = the point's factor row: L^-1 (6), z (3), ok flag.  A degenerate point (ok == 0) writes zeros when `dense` demands
// that every entry be rewritten.
template <typename T>
__device__ __forceinline__ void schur_emit_block(T* __restrict__ pan, int q, int col0, int dense,
                                                 const T (&Jc)[2][NCP], const T (&Jp)[2][3], const T* f) {
  if (f[9] == (T)0) {
    if (dense) {
#pragma unroll
      for (int e = 0; e < NCP; ++e)
#pragma unroll
        for (int d = 0; d < 3; ++d) pan[(3 * q + d) * GROUP_ROWS + col0 + e] = (T)0;
    }
    return;
  }
  // Jp~ = Jp * L^-T  (2x3):  (L^-T)[k][d] = Linv[d][k]
  T Jt[2][3];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    Jt[rr][0] = Jp[rr][0] * f[0];
    Jt[rr][1] = Jp[rr][0] * f[1] + Jp[rr][1] * f[2];
    Jt[rr][2] = Jp[rr][0] * f[3] + Jp[rr][1] * f[4] + Jp[rr][2] * f[5];
  }
#pragma unroll
  for (int e = 0; e < NCP; ++e) {
#pragma unroll
    for (int d = 0; d < 3; ++d)
      pan[(3 * q + d) * GROUP_ROWS + col0 + e] = Jc[0][e] * Jt[0][d] + Jc[1][e] * Jt[1][d];
  }
}

// One observation -> its 11x3 block of Ytilde in the panel of its camera group (+ z of its point).
template <typename T, bool DIAG>
__device__ __forceinline__ void schur_emit(T* __restrict__ panelA, T* __restrict__ panelB, T* __restrict__ s_z,
                                           const T* __restrict__ s_cam, int camA0, int nA, int camB0, int nB, int dense,
                                           int c, int q, T ux, T uy, T ww, T X0, T X1, T X2, const T* f, RLoss<T> loss = RLoss<T>{}) {
  const bool inA = (c >= camA0 && c < camA0 + nA);
  const bool inB = !DIAG && (c >= camB0 && c < camB0 + nB);
  if (!inA && !inB) return;
  T* pan = inA ? panelA : panelB;
  const int col0 = (inA ? (c - camA0) : (c - camB0)) * NCP;
  T r[2], Jc[2][NCP], Jp[2][3];
  if (f[9] != (T)0) {
    const T* cp = s_cam + (inA ? (c - camA0) : (GROUP_CAMS + c - camB0)) * CAMPRE;
    obs_resjac<T>(cp, X0, X1, X2, ux, uy, ww, r, Jc, Jp);
    if (loss.delta > (T)0) (void)robust_apply<T>(loss, r, Jc, Jp);
  }
  schur_emit_block<T>(pan, q, col0, dense, Jc, Jp, f);
  if (DIAG && f[9] != (T)0) { s_z[3 * q + 0] = f[6]; s_z[3 * q + 1] = f[7]; s_z[3 * q + 2] = f[8]; }   // same 3 values from every observation of the point
}

// ------------------------------------------------------------------ K4: the kernel.  grid = (ksplit, pairs of this kind, TS)
template <typename T, bool DIAG, bool PARTIAL = false>
__global__ __launch_bounds__(SCHUR_THREADS) void k_schur(
    const ParamSets<T> ps, const LMState* __restrict__ st, int C,
    const typename Vec2<T>::type* __restrict__ uv, const T* __restrict__ w,
    const int32_t* __restrict__ ci, const int32_t* __restrict__ pi, const int32_t* __restrict__ pt_start, int N,
    const T* __restrict__ pf, const int32_t* __restrict__ pair_ga, const int32_t* __restrict__ pair_gb,
    int pair0, int ksplit, int dense, T* __restrict__ slabs, double* __restrict__ bpart,
    long long* __restrict__ dbg /* optional cycle stamps of workgroup (0,0): [it][producer done, consumer done, barrier out] */,
    const uint16_t* __restrict__ gmask = nullptr, const int32_t* __restrict__ gstart = nullptr /* k_group_index tables, or NULL: scan */) {
  extern __shared__ __align__(16) unsigned char smem[];
  using Cfg = SchurCfg<T, DIAG>;
  constexpr int THREADS = Cfg::THREADS, NPROD = Cfg::NPROD, NCW = Cfg::NCW, TPW = Cfg::TPW;
  constexpr int PTS = Cfg::PTS, K = Cfg::K, BUF = Cfg::BUF;
  if (st->status >= 0) return;
  const int cur_ = ps_cur(ps, st);
  const T* __restrict__ campre = ps.campre[cur_];
  const T* __restrict__ ptsT = ps.ptsT[cur_];
  const int pair = pair0 + blockIdx.y;
  const int ga = pair_ga[pair], gb = pair_gb[pair];
  const int camA0 = ga * GROUP_CAMS, camB0 = gb * GROUP_CAMS;
  const int nA = min(GROUP_CAMS, C - camA0), nB = min(GROUP_CAMS, C - camB0);
  const int usedA = (nA * NCP + 15) / 16, usedB = DIAG ? usedA : (nB * NCP + 15) / 16;
  T* s_buf = reinterpret_cast<T*>(smem);                          // [2][BUF]: panelA [K][176], (panelB), z [K]
  T* s_cam = s_buf + 2 * BUF;                                     // [32][CAMPRE] : group A then group B
  for (int i = threadIdx.x; i < 2 * BUF; i += THREADS) s_buf[i] = (T)0;
  for (int i = threadIdx.x; i < nA * CAMPRE; i += THREADS) s_cam[i] = campre[(size_t)camA0 * CAMPRE + i];
  if (!DIAG)
    for (int i = threadIdx.x; i < nB * CAMPRE; i += THREADS)
      s_cam[GROUP_CAMS * CAMPRE + i] = campre[(size_t)camB0 * CAMPRE + i];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool producer = threadIdx.x < NPROD;
  const int cw = wid - NPROD / 64;                                // consumer wave index (valid when !producer)
  const int vw = (int)blockIdx.z * NCW + cw;                      // virtual consumer wave of the pair
  const int ct = threadIdx.x - NPROD;                             // consumer thread index
  typename Mfma<T>::acc_t acc[TPW];
#pragma unroll
  for (int s = 0; s < TPW; ++s) acc[s] = typename Mfma<T>::acc_t{0, 0, 0, 0};
  double bacc = 0;   // consumer thread rho < 176 of split 0 accumulates the rhs contribution of row rho (diag pairs only)
  const bool do_rhs = DIAG && blockIdx.z == 0;
  const int lane_off = (lane >> 4) * GROUP_ROWS + (lane & 15);

  // slice of points for this k-split (multiple of PTS)
  int per = (N + ksplit - 1) / ksplit;
  per = ((per + PTS - 1) / PTS) * PTS;
  const int pbeg = min(N, (int)blockIdx.x * per), pend = min(N, pbeg + per);
  const int nchunk = (pend - pbeg + PTS - 1) / PTS;
  // producer software pipeline (diagonal pairs): per-lane operands of the current chunk and indices of the next
  using T2 = typename Vec2<T>::type;
  bool cur_valid = false, n1_valid = false;
  int cur_c = 0, cur_p = 0, n1_c = 0, n1_p = 0;
  T2 cur_uv, n1_uv; cur_uv.x = cur_uv.y = n1_uv.x = n1_uv.y = (T)0;
  T cur_w = (T)1, n1_w = (T)1;
  T cur_X[3] = {0, 0, 0};
  T cur_f[PF];
#pragma unroll
  for (int k = 0; k < PF; ++k) cur_f[k] = (T)0;
  auto load_idx = [&](int chunk, bool& valid, int& c, int& pp, T2& m, T& ww) {
    valid = false;
    if (chunk < nchunk) {
      const int q0 = pbeg + chunk * PTS, q1 = min(pend, q0 + PTS);
      const int o = pt_start[q0] + (int)threadIdx.x;
      if (o < pt_start[q1]) { valid = true; c = ci[o]; pp = pi[o]; m = uv[o]; ww = w ? w[o] : (T)1; }
    }
  };
  // a chunk holds at most PTS*C observations: one per producer lane only when all cameras are in this one group
  const bool piped = DIAG && (C <= GROUP_CAMS);
  if (piped && producer) {
    load_idx(0, cur_valid, cur_c, cur_p, cur_uv, cur_w);
    if (cur_valid) {
      cur_X[0] = ptsT[3 * (size_t)cur_p]; cur_X[1] = ptsT[3 * (size_t)cur_p + 1]; cur_X[2] = ptsT[3 * (size_t)cur_p + 2];
#pragma unroll
      for (int k = 0; k < PF; ++k) cur_f[k] = pf[(size_t)cur_p * PF + k];
    }
    load_idx(1, n1_valid, n1_c, n1_p, n1_uv, n1_w);
  }
  __syncthreads();
  // the consumer waves are the second-dispatched half of the workgroup and lose the per-SIMD issue arbitration against
  // their producer partner by age; their MFMAs pace the kernel, so they get static priority
  if (!producer) __builtin_amdgcn_s_setprio(2);
  for (int it = 0; it <= nchunk; ++it) {
    if (producer) {
      if (it < nchunk) {
        T* buf = s_buf + (it & 1) * BUF;
        T* panelA = buf;
        T* panelB = DIAG ? panelA : panelA + K * GROUP_ROWS;
        T* s_z = buf + Cfg::NPANEL * K * GROUP_ROWS;
        const int p0 = pbeg + it * PTS, p1 = min(pend, p0 + PTS);
        if (dense && p1 - p0 < PTS)      // partial last chunk of a dense problem: clear the rows no observation will write
          for (int i = (p1 - p0) * 3 * GROUP_ROWS + threadIdx.x; i < Cfg::NPANEL * K * GROUP_ROWS; i += NPROD) {
            const int rowi = (i % (K * GROUP_ROWS)) / GROUP_ROWS;
            if (rowi >= 3 * (p1 - p0)) buf[i] = (T)0;
          }
        auto emit = [&](int c, int q, T ux, T uy, T ww, T X0, T X1, T X2, const T* f) {
          schur_emit<T, DIAG>(panelA, panelB, s_z, s_cam, camA0, nA, camB0, nB, dense, c, q, ux, uy, ww, X0, X1, X2, f, ps.loss());
        };
        if (piped) {
          // <= 16 points x 16 cameras = 256 observations = one per producer lane; operands of the NEXT chunk were
          // requested one iteration ago (software pipeline: indices two chunks ahead, point data one chunk ahead)
          if (cur_valid) emit(cur_c, cur_p - p0, cur_uv.x, cur_uv.y, cur_w, cur_X[0], cur_X[1], cur_X[2], cur_f);
        } else if (gmask) {
          // several camera groups, indexed: lane = (point, camera of the group) finds its observation directly -- every lane
          // of a wave has work when the group sees the point (scanning the point's whole list kept 16 of 64 cameras' lanes
          // busy); an off-diagonal pair takes its second group in a second pass (or in the upper lane rows when PTS = 8)
          constexpr int ROWS = NPROD / 16;
          constexpr int PASSES = (PTS * Cfg::NPANEL + ROWS - 1) / ROWS;
          const int r = threadIdx.x >> 4, cl = threadIdx.x & 15;
#pragma unroll
          for (int pass = 0; pass < PASSES; ++pass) {
            const int idx = pass * ROWS + r;
            const int q = idx % PTS, h = idx / PTS;
            const int pp = p0 + q;
            if (h < Cfg::NPANEL && pp < p1) {
              const int g = h ? gb : ga;
              const unsigned m = gmask[(size_t)g * N + pp];
              if ((m >> cl) & 1u) {
                const size_t o = (size_t)gstart[(size_t)g * N + pp] + __builtin_popcount(m & ((1u << cl) - 1u));
                T f[PF];
#pragma unroll
                for (int k = 0; k < PF; ++k) f[k] = pf[(size_t)pp * PF + k];
                const auto mm = uv[o];
                emit(g * GROUP_CAMS + cl, q, mm.x, mm.y, w ? w[o] : (T)1, ptsT[3 * (size_t)pp], ptsT[3 * (size_t)pp + 1], ptsT[3 * (size_t)pp + 2], f);
              }
            }
          }
        } else {
          const int o_lo = pt_start[p0], o_hi = pt_start[p1];
          for (int o = o_lo + threadIdx.x; o < o_hi; o += NPROD) {
            const int c = ci[o];
            // several camera groups: most observations belong to other pairs -- drop them before touching anything else
            if (!((c >= camA0 && c < camA0 + nA) || (!DIAG && c >= camB0 && c < camB0 + nB))) continue;
            const int pp = pi[o];
            T f[PF];
#pragma unroll
            for (int k = 0; k < PF; ++k) f[k] = pf[(size_t)pp * PF + k];
            const auto m = uv[o];
            emit(c, pp - p0, m.x, m.y, w ? w[o] : (T)1, ptsT[3 * (size_t)pp], ptsT[3 * (size_t)pp + 1], ptsT[3 * (size_t)pp + 2], f);
          }
        }
      }
      if (piped) {
        // advance the pipeline: data for chunk it+1 (its indices are already here), indices for chunk it+2
        cur_valid = n1_valid; cur_c = n1_c; cur_p = n1_p; cur_uv = n1_uv; cur_w = n1_w;
        if (cur_valid) {
          cur_X[0] = ptsT[3 * (size_t)cur_p]; cur_X[1] = ptsT[3 * (size_t)cur_p + 1]; cur_X[2] = ptsT[3 * (size_t)cur_p + 2];
#pragma unroll
          for (int k = 0; k < PF; ++k) cur_f[k] = pf[(size_t)cur_p * PF + k];
        }
        load_idx(it + 2, n1_valid, n1_c, n1_p, n1_uv, n1_w);
      }
    } else if (it >= 1) {
      T* buf = s_buf + ((it - 1) & 1) * BUF;
      const T* panelA = buf;
      const T* panelB = DIAG ? panelA : panelA + K * GROUP_ROWS;
      const T* s_z = buf + Cfg::NPANEL * K * GROUP_ROWS;
      // rhs: b[rho] += sum_k panel[k][rho] * z[k]
      if (do_rhs && ct < GROUP_ROWS) {
        T s0 = 0, s1 = 0;
#pragma unroll 8
        for (int k = 0; k < K; k += 2) { s0 += panelA[k * GROUP_ROWS + ct] * s_z[k]; s1 += panelA[(k + 1) * GROUP_ROWS + ct] * s_z[k + 1]; }
        bacc += (double)(s0 + s1);
      }
      schur_consume_v<Cfg, K, PARTIAL>(vw, panelA + lane_off, panelB + lane_off, acc, usedA, usedB);
    }
    if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && it < 20) {
      if (threadIdx.x == 0) dbg[3 * it + 0] = clock64();
      if (threadIdx.x == NPROD) dbg[3 * it + 1] = clock64();
    }
    __syncthreads();
    if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && it < 20 && threadIdx.x == 0) dbg[3 * it + 2] = clock64();
    if (!dense && it >= 1) {
      // sparse visibility: not every panel entry is rewritten by the next chunk, so the buffer that was just
      // consumed is cleared by the whole workgroup before the producers get it back
      T* done = s_buf + ((it - 1) & 1) * BUF;
      for (int i = threadIdx.x; i < Cfg::NPANEL * K * GROUP_ROWS; i += THREADS) done[i] = (T)0;
      __syncthreads();
    }
  }
  // write partial tiles
  if (!producer) {
    T* slab = slabs + ((size_t)pair * ksplit + blockIdx.x) * (size_t)(GROUP_TILES * GROUP_TILES) * 256;
    schur_store_v<Cfg>(vw, slab, lane, acc);
    if (do_rhs && ct < GROUP_ROWS)
      bpart[((size_t)ga * ksplit + blockIdx.x) * GROUP_ROWS + ct] = bacc;
  }
}

// ------------------------------------------------------------------ K4 (symmetric): grid = (ksplit, pairs of this kind, TS)
template <typename T, bool DIAG, bool PARTIAL = false, bool LIN = false>
__global__ __launch_bounds__(SCHUR_THREADS) void k_schur_sym(
    const ParamSets<T> ps, const LMState* __restrict__ st, int C,
    const typename Vec2<T>::type* __restrict__ uv, const T* __restrict__ w,
    const int32_t* __restrict__ ci, const int32_t* __restrict__ pi, const int32_t* __restrict__ pt_start, int N,
    const T* __restrict__ pf, const int32_t* __restrict__ pair_ga, const int32_t* __restrict__ pair_gb,
    int pair0, int ksplit, int dense, T* __restrict__ slabs, double* __restrict__ bpart,
    long long* __restrict__ dbg /* optional cycle stamps of workgroup (0,0,0): [it][produce done, after barrier, consume done] */,
    const uint16_t* __restrict__ vis = nullptr /* LIN: per-point visibility mask, NULL = dense */, double* __restrict__ D2p = nullptr,
    double* __restrict__ gp = nullptr, double* __restrict__ cost_part = nullptr, double* __restrict__ gmax_part = nullptr,
    const uint16_t* __restrict__ gmask = nullptr, const int32_t* __restrict__ gstart = nullptr /* k_group_index tables, or NULL: scan */) {
  extern __shared__ __align__(16) unsigned char smem[];
  using Cfg = SchurSymCfg<T, DIAG>;
  constexpr int THREADS = Cfg::THREADS, NCW = Cfg::NCW, TPW = Cfg::TPW, PTS = Cfg::PTS, K = Cfg::K;
  if (st->status >= 0) return;
  const int cur_ = ps_cur(ps, st);
  const T* __restrict__ campre = ps.campre[cur_];
  const T* __restrict__ ptsT = ps.ptsT[cur_];
  const int pair = pair0 + blockIdx.y;
  const int ga = pair_ga[pair], gb = pair_gb[pair];
  const int camA0 = ga * GROUP_CAMS, camB0 = gb * GROUP_CAMS;
  const int nA = min(GROUP_CAMS, C - camA0), nB = min(GROUP_CAMS, C - camB0);
  const int usedA = (nA * NCP + 15) / 16, usedB = DIAG ? usedA : (nB * NCP + 15) / 16;
  T* panelA = reinterpret_cast<T*>(smem);                         // [K][176]
  T* panelB = DIAG ? panelA : panelA + K * GROUP_ROWS;
  T* s_z = panelA + Cfg::NPANEL * K * GROUP_ROWS;                 // [K]
  T* s_cam = s_z + K;                                             // [32][CAMPRE] : group A then group B
  for (int i = threadIdx.x; i < Cfg::BUF; i += THREADS) panelA[i] = (T)0;
  for (int i = threadIdx.x; i < nA * CAMPRE; i += THREADS) s_cam[i] = campre[(size_t)camA0 * CAMPRE + i];
  if (!DIAG)
    for (int i = threadIdx.x; i < nB * CAMPRE; i += THREADS)
      s_cam[GROUP_CAMS * CAMPRE + i] = campre[(size_t)camB0 * CAMPRE + i];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int vw = (int)blockIdx.z * NCW + wid;                     // virtual consumer wave of the pair
  typename Mfma<T>::acc_t acc[TPW];
#pragma unroll
  for (int s = 0; s < TPW; ++s) acc[s] = typename Mfma<T>::acc_t{0, 0, 0, 0};
  // rhs of a diagonal pair: thread (half, rho) accumulates b[rho] over its half of the panel rows
  const bool do_rhs = DIAG && blockIdx.z == 0 && threadIdx.x < 2 * GROUP_ROWS;
  const int rhs_row = threadIdx.x % GROUP_ROWS, rhs_half = threadIdx.x / GROUP_ROWS;
  double bacc = 0;
  const int lane_off = (lane >> 4) * GROUP_ROWS + (lane & 15);

  int per = (N + ksplit - 1) / ksplit;
  per = ((per + PTS - 1) / PTS) * PTS;
  const int pbeg = min(N, (int)blockIdx.x * per), pend = min(N, pbeg + per);
  const int nchunk = (pend - pbeg + PTS - 1) / PTS;
  // software pipeline of the producer phase: operands of the next chunk and indices of the one after are requested
  // before the consume phase, so their latency is covered by the MFMAs
  using T2 = typename Vec2<T>::type;
  bool cur_valid = false, n1_valid = false;
  int cur_c = 0, cur_p = 0, n1_c = 0, n1_p = 0;
  T2 cur_uv, n1_uv; cur_uv.x = cur_uv.y = n1_uv.x = n1_uv.y = (T)0;
  T cur_w = (T)1, n1_w = (T)1;
  T cur_X[3] = {0, 0, 0};
  T cur_f[PF];
#pragma unroll
  for (int k = 0; k < PF; ++k) cur_f[k] = (T)0;
  auto load_idx = [&](int chunk, bool& valid, int& c, int& pp, T2& m, T& ww) {
    valid = false;
    if (chunk < nchunk) {
      const int q0 = pbeg + chunk * PTS, q1 = min(pend, q0 + PTS);
      const int o = pt_start[q0] + (int)threadIdx.x;
      if (o < pt_start[q1]) { valid = true; c = ci[o]; pp = pi[o]; m = uv[o]; ww = w ? w[o] : (T)1; }
    }
  };
  auto load_point = [&]() {
    if (cur_valid) {
      cur_X[0] = ptsT[3 * (size_t)cur_p]; cur_X[1] = ptsT[3 * (size_t)cur_p + 1]; cur_X[2] = ptsT[3 * (size_t)cur_p + 2];
#pragma unroll
      for (int k = 0; k < PF; ++k) cur_f[k] = pf[(size_t)cur_p * PF + k];
    }
  };
  // LIN (one camera group, no duplicate (point, camera) pairs): lane (q, c) = (point of the 32-point chunk, camera), as
  // in k_schur_fused -- the point blocks V_p, g_p are DPP row sums, and scaling, damped factor and z follow in
  // registers, so k_linearize_points and k_point_factor are not launched (f64 keeps k_linearize_cams: 77 f64
  // accumulators per lane do not fit beside the MFMA tiles).
  static_assert(!LIN || (DIAG && PTS == 32), "the fused point linearisation needs the one-group diagonal pair and 32-point chunks");
  const int lq = threadIdx.x >> 4, lc = threadIdx.x & 15;
  const bool cam_ok = lc < C;
  const double lam = st->lam;
  double l_sq = 0, l_gmx = 0;
  T2 n_uv; n_uv.x = n_uv.y = (T)0;
  T n_w = (T)1, n_X[3] = {0, 0, 0};
  double n_D[3] = {0, 0, 0};
  bool n_valid = false, n_pt = false;
  unsigned i_mask = 0; int i_start = 0; bool i_pt = false;
  auto request_index = [&](int chunk) {
    const int p = pbeg + chunk * PTS + lq;
    i_pt = chunk < nchunk && p < pend;
    i_mask = 0xffffu; i_start = 0;
    if (i_pt && vis) { i_mask = vis[p]; i_start = pt_start[p]; }
  };
  auto request = [&](int chunk) {
    const int p = pbeg + chunk * PTS + lq;
    n_pt = i_pt;
    n_valid = i_pt && cam_ok && ((i_mask >> lc) & 1u);
    if (n_pt) {
      n_X[0] = ptsT[3 * (size_t)p]; n_X[1] = ptsT[3 * (size_t)p + 1]; n_X[2] = ptsT[3 * (size_t)p + 2];
      n_D[0] = D2p[3 * (size_t)p]; n_D[1] = D2p[3 * (size_t)p + 1]; n_D[2] = D2p[3 * (size_t)p + 2];
    }
    if (n_valid) {
      const size_t o = vis ? (size_t)i_start + __builtin_popcount(i_mask & ((1u << lc) - 1u)) : (size_t)p * C + lc;
      n_uv = uv[o];
      n_w = w ? w[o] : (T)1;
    }
    request_index(chunk + 1);
  };
  if constexpr (LIN) { request_index(0); request(0); }
  // a chunk holds at most PTS*C observations: one per lane only when all cameras are in this one group (32 x 16 = 512)
  const bool piped = !LIN && DIAG && (C <= GROUP_CAMS);
  if (piped) {
    load_idx(0, cur_valid, cur_c, cur_p, cur_uv, cur_w);
    load_point();
    load_idx(1, n1_valid, n1_c, n1_p, n1_uv, n1_w);
  }
  __syncthreads();
  for (int it = 0; it < nchunk; ++it) {
    // ---- produce
    const int p0 = pbeg + it * PTS, p1 = min(pend, p0 + PTS);
    if (!LIN && dense && p1 - p0 < PTS)   // partial last chunk of a dense problem: clear the rows no observation will write
      for (int i = (p1 - p0) * 3 * GROUP_ROWS + threadIdx.x; i < Cfg::NPANEL * K * GROUP_ROWS; i += THREADS) {
        const int rowi = (i % (K * GROUP_ROWS)) / GROUP_ROWS;
        if (rowi >= 3 * (p1 - p0)) panelA[i] = (T)0;
      }
    if constexpr (LIN) {
      const bool valid = n_valid, have_pt = n_pt;
      const T2 m = n_uv;
      const T ww = n_w, X0 = n_X[0], X1 = n_X[1], X2 = n_X[2];
      const double D0 = n_D[0], D1 = n_D[1], D2 = n_D[2];
      const int p = p0 + lq;
      request(it + 1);
      T r[2] = {0, 0}, Jc[2][NCP], Jp[2][3];
#pragma unroll
      for (int e = 0; e < NCP; ++e) { Jc[0][e] = 0; Jc[1][e] = 0; }
#pragma unroll
      for (int d = 0; d < 3; ++d) { Jp[0][d] = 0; Jp[1][d] = 0; }
      if (valid) obs_resjac<T>(s_cam + lc * CAMPRE, X0, X1, X2, m.x, m.y, ww, r, Jc, Jp);
      if (ps.loss_delta > 0.f) l_sq += (double)robust_apply<T>(ps.loss(), r, Jc, Jp);
      else l_sq += (double)r[0] * r[0] + (double)r[1] * r[1];
      double v6[6], g3[3];
      v6[0] = row16_sum((double)(Jp[0][0] * Jp[0][0] + Jp[1][0] * Jp[1][0]));
      v6[1] = row16_sum((double)(Jp[0][0] * Jp[0][1] + Jp[1][0] * Jp[1][1]));
      v6[2] = row16_sum((double)(Jp[0][0] * Jp[0][2] + Jp[1][0] * Jp[1][2]));
      v6[3] = row16_sum((double)(Jp[0][1] * Jp[0][1] + Jp[1][1] * Jp[1][1]));
      v6[4] = row16_sum((double)(Jp[0][1] * Jp[0][2] + Jp[1][1] * Jp[1][2]));
      v6[5] = row16_sum((double)(Jp[0][2] * Jp[0][2] + Jp[1][2] * Jp[1][2]));
      g3[0] = row16_sum((double)(Jp[0][0] * r[0] + Jp[1][0] * r[1]));
      g3[1] = row16_sum((double)(Jp[0][1] * r[0] + Jp[1][1] * r[1]));
      g3[2] = row16_sum((double)(Jp[0][2] * r[0] + Jp[1][2] * r[1]));
      const bool fixedp = have_pt && pt_fixed(ps, (size_t)p);
      if (!fixedp) l_gmx = fmax(l_gmx, fmax(fabs(g3[0]), fmax(fabs(g3[1]), fabs(g3[2]))));
      // point scaling: monotone max of the squared column norms (x_scale='jac', scipy trf.py:424,545)
      const double E0 = fmax(D0, v6[0]), E1 = fmax(D1, v6[3]), E2 = fmax(D2, v6[5]);
      double vd[6] = {v6[0] + lam * fmax_pos(E0), v6[1], v6[2], v6[3] + lam * fmax_pos(E1), v6[4], v6[5] + lam * fmax_pos(E2)};
      double li[6];
      const bool okp = have_pt && !fixedp && chol3_inv<double>(vd, li);
      T f[PF];
#pragma unroll
      for (int k = 0; k < PF; ++k) f[k] = (T)0;
      if (okp) {
#pragma unroll
        for (int k = 0; k < 6; ++k) f[k] = (T)li[k];
        f[6] = (T)(li[0] * g3[0]);
        f[7] = (T)(li[1] * g3[0] + li[2] * g3[1]);
        f[8] = (T)(li[3] * g3[0] + li[4] * g3[1] + li[5] * g3[2]);
        f[9] = (T)1;
      }
      if (have_pt && lc == 0) {
        D2p[3 * (size_t)p] = E0; D2p[3 * (size_t)p + 1] = E1; D2p[3 * (size_t)p + 2] = E2;
        gp[3 * (size_t)p] = g3[0]; gp[3 * (size_t)p + 1] = g3[1]; gp[3 * (size_t)p + 2] = g3[2];
        T* o = const_cast<T*>(pf) + (size_t)p * PF;
#pragma unroll
        for (int k = 0; k < PF; ++k) o[k] = f[k];
      }
      if (cam_ok) {
        schur_emit_block<T>(panelA, lq, lc * NCP, 1, Jc, Jp, f);
        if (lc == 0) { s_z[3 * lq + 0] = f[6]; s_z[3 * lq + 1] = f[7]; s_z[3 * lq + 2] = f[8]; }
      }
    } else if (piped) {
      if (cur_valid)
        schur_emit<T, DIAG>(panelA, panelB, s_z, s_cam, camA0, nA, camB0, nB, dense, cur_c, cur_p - p0, cur_uv.x, cur_uv.y, cur_w,
                            cur_X[0], cur_X[1], cur_X[2], cur_f, ps.loss());
      cur_valid = n1_valid; cur_c = n1_c; cur_p = n1_p; cur_uv = n1_uv; cur_w = n1_w;
      load_point();
      load_idx(it + 2, n1_valid, n1_c, n1_p, n1_uv, n1_w);
    } else if (gmask) {
      // several camera groups, indexed (see k_schur): lane = (point, camera of the group)
      constexpr int ROWS = THREADS / 16;
      constexpr int PASSES = (PTS * Cfg::NPANEL + ROWS - 1) / ROWS;
      const int r = threadIdx.x >> 4, cl = threadIdx.x & 15;
#pragma unroll
      for (int pass = 0; pass < PASSES; ++pass) {
        const int idx = pass * ROWS + r;
        const int q = idx % PTS, h = idx / PTS;
        const int pp = p0 + q;
        if (h < Cfg::NPANEL && pp < p1) {
          const int g = h ? gb : ga;
          const unsigned m = gmask[(size_t)g * N + pp];
          if ((m >> cl) & 1u) {
            const size_t o = (size_t)gstart[(size_t)g * N + pp] + __builtin_popcount(m & ((1u << cl) - 1u));
            T f[PF];
#pragma unroll
            for (int k = 0; k < PF; ++k) f[k] = pf[(size_t)pp * PF + k];
            const auto mm = uv[o];
            schur_emit<T, DIAG>(panelA, panelB, s_z, s_cam, camA0, nA, camB0, nB, dense, g * GROUP_CAMS + cl, q, mm.x, mm.y, w ? w[o] : (T)1,
                                ptsT[3 * (size_t)pp], ptsT[3 * (size_t)pp + 1], ptsT[3 * (size_t)pp + 2], f, ps.loss());
          }
        }
      }
    } else {
      const int o_lo = pt_start[p0], o_hi = pt_start[p1];
      for (int o = o_lo + threadIdx.x; o < o_hi; o += THREADS) {
        const int c = ci[o];
        if (!((c >= camA0 && c < camA0 + nA) || (!DIAG && c >= camB0 && c < camB0 + nB))) continue;
        const int pp = pi[o];
        T f[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) f[k] = pf[(size_t)pp * PF + k];
        const auto m = uv[o];
        schur_emit<T, DIAG>(panelA, panelB, s_z, s_cam, camA0, nA, camB0, nB, dense, c, pp - p0, m.x, m.y, w ? w[o] : (T)1,
                            ptsT[3 * (size_t)pp], ptsT[3 * (size_t)pp + 1], ptsT[3 * (size_t)pp + 2], f, ps.loss());
      }
    }
    const bool stamp = dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && it < 20 && threadIdx.x == 0;
    if (stamp) dbg[3 * it + 0] = clock64();
    __syncthreads();
    if (stamp) dbg[3 * it + 1] = clock64();
    // ---- consume
    if (do_rhs) {
      T s0 = 0, s1 = 0;
      const int k0 = rhs_half * (K / 2);
#pragma unroll 8
      for (int k = k0; k < k0 + K / 2; k += 2) {
        s0 += panelA[k * GROUP_ROWS + rhs_row] * s_z[k];
        s1 += panelA[(k + 1) * GROUP_ROWS + rhs_row] * s_z[k + 1];
      }
      bacc += (double)(s0 + s1);
    }
    schur_consume_v<Cfg, K, PARTIAL>(vw, panelA + lane_off, panelB + lane_off, acc, usedA, usedB);
    __syncthreads();
    if (stamp) dbg[3 * it + 2] = clock64();
    if (!LIN && !dense) {
      // sparse visibility: not every panel entry is rewritten by the next chunk
      for (int i = threadIdx.x; i < Cfg::NPANEL * K * GROUP_ROWS; i += THREADS) panelA[i] = (T)0;
      __syncthreads();
    }
  }
  T* slab = slabs + ((size_t)pair * ksplit + blockIdx.x) * (size_t)(GROUP_TILES * GROUP_TILES) * 256;
  schur_store_v<Cfg>(vw, slab, lane, acc);
  if constexpr (LIN) {
    __shared__ double s_lin[SCHUR_THREADS / 64];
    const double cs = block_sum(l_sq, s_lin);
    const double gm = block_max(l_gmx, s_lin);
    if (threadIdx.x == 0) { cost_part[blockIdx.x] = 0.5 * cs; gmax_part[blockIdx.x] = gm; }
  }
  if (DIAG && blockIdx.z == 0) {
    double* s_rhs = reinterpret_cast<double*>(smem);              // the panel is free now
    if (do_rhs && rhs_half == 1) s_rhs[rhs_row] = bacc;
    __syncthreads();
    if (do_rhs && rhs_half == 0)
      bpart[((size_t)ga * ksplit + blockIdx.x) * GROUP_ROWS + rhs_row] = bacc + s_rhs[rhs_row];
  }
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

constexpr int UPK = NCP * (NCP + 1) / 2 + NCP;                 // per-lane U_c (upper triangle) + g_c accumulators of the fused kernels: 77 (104)
// The accept/reject decision of the PREVIOUS trial step can ride in this kernel's prologue (do_decide): every workgroup reads
// the untouched record st_in and the trial partials, runs decide_core on its own LDS copy -- same inputs, same summation
// order, same decision everywhere -- and workgroup 0 publishes the updated record to st_out (the other slot of a two-entry
// buffer: nothing a late workgroup still has to read is overwritten; for the same reason the gradient maxima this kernel
// writes go to a different array than the one the decision reads) and appends the log row.  That takes the k_decide launch
// (6.6 us of mostly launch and memory latency) out of the iteration for one more dependent read in front of the camera load.
struct FusedDecide {
  const LMState* st_in;          // record before the decision (never written by this kernel)
  LMState* st_out;               // where workgroup 0 publishes it afterwards (== st_in when do_decide == 0: nothing is written)
  int do_decide;
  const double* scal_all;        // multi-rank: gathered scalars [n_ranks][NSCAL], else NULL
  int n_ranks;
  const double* trial_part;      // [4][n_trial] partials of the trial kernel
  const double* gmax_in;         // [n_gmax] gradient maxima of the previous linearisation
  int n_trial, n_gmax;
  LMLogRow* log;
  int log_cap;
};
#if SBA_NCP == 11
// ------------------------------------------------------------------ K3+K4 fused (dense visibility, <= 16 cameras, f32)
// When every point is seen by every camera and there is one camera group, the producer lane (q, c) = (point of the
// chunk, camera) meets the same camera in every chunk, and the 16 lanes of a DPP row hold all observations of one
// point.  The whole linearisation then fits into the Schur producer, and each observation's Jacobian is evaluated
// once per LM iteration instead of three times (k_linearize_points, k_linearize_cams, k_schur):
//   * V_p = sum Jp^T Jp and g_p = sum Jp^T r : DPP row reduction over the 16 lanes of the point; column scaling D2p,
//     damped 3x3 factor and z = L^-1 g_p follow in registers (the factor row pf[p] is stored for k_backsub_trial);
//   * U_c = sum Jc^T Jc (66 upper-triangle entries) and g_c = sum Jc^T r (11) : per-lane register accumulators,
//     folded over the 16 points-lanes of a camera through LDS at the end, one partial per workgroup
//     (Upart2[wg][c][77]); k_build_exchange sums the partials in the same fixed-order loop as the slabs.
// The two roles are split at the top level (same number of barriers on both sides), so the producers' 77 accumulators
// and the consumers' 68 MFMA accumulator registers share the register file instead of adding up.
template <typename T> struct SchurFusedCfg : SchurCfg<T, true> {
  static constexpr size_t LDS_BYTES = SchurCfg<T, true>::LDS_BYTES + (size_t)SchurCfg<T, true>::NPROD * UPK * sizeof(T);
};

__global__ __launch_bounds__(SCHUR_THREADS) void k_schur_fused(
    const ParamSets<float> ps, const LMState* __restrict__ st, int C,
    const float2* __restrict__ uv /* observations of a point in camera order; dense rigs: (p, c) at p*C + c */,
    const float* __restrict__ w, const int32_t* __restrict__ pt_start, const uint16_t* __restrict__ vis /* per point: bit c
    = camera c sees it; NULL = every camera sees every point */,
    int N, int ksplit, double* __restrict__ D2p, double* __restrict__ gp, float* __restrict__ pf, float* __restrict__ slabs,
    double* __restrict__ bpart, double* __restrict__ gdpart /* [ksplit][2][176]: g_c and diag U_c partials */,
    double* __restrict__ cost_part, double* __restrict__ gmax_part,
    long long* __restrict__ dbg /* optional cycle stamps of workgroup 0: [it][producer done, consumer done] */) {
  extern __shared__ __align__(16) unsigned char smem[];
  using T = float;
  using Cfg = SchurCfg<T, true>;
  constexpr int THREADS = Cfg::THREADS, NPROD = Cfg::NPROD, TPW = Cfg::TPW, PTS = Cfg::PTS, K = Cfg::K, BUF = Cfg::BUF;
  static_assert(PTS == 16 && NPROD == 256, "lane = (point of the chunk, camera) needs 16 x 16 producer lanes");
  if (st->status >= 0) return;
  const bool stamp_wg = dbg && blockIdx.x == 0;
  if (stamp_wg && threadIdx.x == 0) dbg[48] = clock64();         // kernel entry
  const int cur_ = ps_cur(ps, st);
  const T* __restrict__ campre = ps.campre[cur_];
  const T* __restrict__ ptsT = ps.ptsT[cur_];
  T* s_buf = reinterpret_cast<T*>(smem);                          // [2][BUF]: panel [K][176], z [K]
  T* s_cam = s_buf + 2 * BUF;                                     // [32][CAMPRE] (first 16 used)
  T* s_U = s_cam + 2 * GROUP_CAMS * CAMPRE;                       // [256][UPK] at the end
  __shared__ double s_scr[2][NPROD / 64];
  for (int i = threadIdx.x; i < 2 * BUF; i += THREADS) s_buf[i] = (T)0;
  for (int i = threadIdx.x; i < C * CAMPRE; i += THREADS) s_cam[i] = campre[i];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool producer = threadIdx.x < NPROD;
  int per = (N + ksplit - 1) / ksplit;
  per = ((per + PTS - 1) / PTS) * PTS;
  const int pbeg = min(N, (int)blockIdx.x * per), pend = min(N, pbeg + per);
  const int nchunk = (pend - pbeg + PTS - 1) / PTS;
  const T lam = (T)st->lam;
  T* s_Ured = s_buf;                                              // [C][UPK] once the panels are done with
  // fold the 16 point-lanes of every camera: U_c[k] = sum_q s_U[16 q + c][k]
  auto fold_u = [&]() {
    for (int o = threadIdx.x; o < C * UPK; o += THREADS) {
      const int c = o / UPK, k = o - c * UPK;
      T sum = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) sum += s_U[(q * 16 + c) * UPK + k];
      s_Ured[o] = sum;
    }
  };
  __syncthreads();

  if (producer) {
    const int q = threadIdx.x >> 4, c = threadIdx.x & 15;
    const bool cam_ok = c < C;
    const T* cp_safe = s_cam + (cam_ok ? c : 0) * CAMPRE;          // lanes beyond the last camera read camera 0's row (weight 0)
    T Uacc[UPK];
    static_for<0, UPK>([&](auto kc) { Uacc[decltype(kc)::value] = (T)0; });
    T sq = 0, gmx = 0;
    // operands of the next chunk, requested one chunk ahead; with sparse visibility the observation index of lane
    // (q, c) is pt_start[p] + popcount(vis[p] below bit c), so the point's mask and offset travel two chunks ahead
    float2 n_uv = make_float2(0.f, 0.f);
    T n_w = 1, n_X[3] = {0, 0, 0};
    double n_D[3] = {0, 0, 0};
    bool n_valid = false, n_pt = false;
    unsigned i_mask = 0; int i_start = 0; bool i_pt = false;      // index stage (chunk + 2)
    auto request_index = [&](int chunk) {
      const int p = pbeg + chunk * PTS + q;
      i_pt = chunk < nchunk && p < pend;
      i_mask = 0xffffu; i_start = 0;
      if (i_pt && vis) { i_mask = vis[p]; i_start = pt_start[p]; }
    };
    auto request = [&](int chunk) {              // consumes the index stage of `chunk`, then refills it for chunk + 1
      const int p = pbeg + chunk * PTS + q;
      n_pt = i_pt;
      n_valid = i_pt && cam_ok && ((i_mask >> c) & 1u);
      if (n_pt) {                                // every lane of the row needs the point (row sums are taken by all 16)
        n_X[0] = ptsT[3 * (size_t)p]; n_X[1] = ptsT[3 * (size_t)p + 1]; n_X[2] = ptsT[3 * (size_t)p + 2];
        n_D[0] = D2p[3 * (size_t)p]; n_D[1] = D2p[3 * (size_t)p + 1]; n_D[2] = D2p[3 * (size_t)p + 2];
      }
      if (n_valid) {
        const size_t o = vis ? (size_t)i_start + __builtin_popcount(i_mask & ((1u << c) - 1u)) : (size_t)p * C + c;
        n_uv = uv[o];
        n_w = w ? w[o] : (T)1;
      }
      request_index(chunk + 1);
    };
    request_index(0);
    request(0);
    if (stamp_wg && threadIdx.x == 0) dbg[49] = clock64();       // prologue done
    for (int it = 0; it <= nchunk; ++it) {
      if (it < nchunk) {
        T* panel = s_buf + (it & 1) * BUF;
        T* s_z = panel + K * GROUP_ROWS;
        const bool valid = n_valid, have_pt = n_pt;
        const float2 m = n_uv;
        const T ww = n_w, X0 = n_X[0], X1 = n_X[1], X2 = n_X[2];
        const double D0 = n_D[0], D1 = n_D[1], D2 = n_D[2];
        const int p = pbeg + it * PTS + q;
        request(it + 1);
        // a lane without an observation runs the same code with weight 0 (every output carries the weight as a factor) and a
        // harmless depth: no branch, no zero-initialised outputs
        T r[2], Jc[2][NCP], Jp[2][3];
        obs_resjac<T>(cp_safe, X0, X1, X2, m.x, m.y, valid ? ww : (T)0, r, Jc, Jp, valid);
        sq += robust_apply<T>(ps.loss(), r, Jc, Jp);
        // per-point blocks: V (6) and g_p (3), summed over the 16 cameras of the DPP row
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
        if (!fixedp) gmx = fmaxf(gmx, fmaxf(fabsf(g3[0]), fmaxf(fabsf(g3[1]), fabsf(g3[2]))));
        // point scaling: monotone max of the squared column norms (x_scale='jac', scipy trf.py:424,545)
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
        if (have_pt && c == 0) {
          D2p[3 * (size_t)p] = E0; D2p[3 * (size_t)p + 1] = E1; D2p[3 * (size_t)p + 2] = E2;
          gp[3 * (size_t)p] = (double)g3[0]; gp[3 * (size_t)p + 1] = (double)g3[1]; gp[3 * (size_t)p + 2] = (double)g3[2];
          float4* o4 = reinterpret_cast<float4*>(pf + (size_t)p * PF);
          o4[0] = make_float4(f[0], f[1], f[2], f[3]);
          o4[1] = make_float4(f[4], f[5], f[6], f[7]);
          o4[2] = make_float4(f[8], f[9], f[10], f[11]);
        }
        if (cam_ok) {
          schur_emit_block<T>(panel, q, c * NCP, 1, Jc, Jp, f);
          if (c == 0) { s_z[3 * q + 0] = f[6]; s_z[3 * q + 1] = f[7]; s_z[3 * q + 2] = f[8]; }
        }
        // camera blocks: U_c upper triangle + g_c in registers (this lane always serves camera c)
        static_for<0, NCP>([&](auto ac) {
          constexpr int a = decltype(ac)::value;
          static_for<a, NCP>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            constexpr int k = a * NCP - (a * (a - 1)) / 2 + (b - a);
            Uacc[k] = __builtin_fmaf(Jc[1][a], Jc[1][b], __builtin_fmaf(Jc[0][a], Jc[0][b], Uacc[k]));   // two FMAs, no add
          });
          Uacc[NCP * (NCP + 1) / 2 + a] = __builtin_fmaf(Jc[1][a], r[1], __builtin_fmaf(Jc[0][a], r[0], Uacc[NCP * (NCP + 1) / 2 + a]));
        });
      }
      if (dbg && blockIdx.x == 0 && threadIdx.x == 0 && it < 20) dbg[2 * it] = clock64();
      __syncthreads();
    }
    if (stamp_wg && threadIdx.x == 0) dbg[50] = clock64();       // main loop done (producer side)
    // hand the accumulators over
    static_for<0, UPK>([&](auto kc) { constexpr int k = decltype(kc)::value; s_U[threadIdx.x * UPK + k] = Uacc[k]; });
    const double cs = wave_sum((double)sq), gm = wave_max((double)gmx);
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
    __builtin_amdgcn_s_setprio(2);
    for (int it = 0; it <= nchunk; ++it) {
      if (it >= 1) {
        const T* panel = s_buf + ((it - 1) & 1) * BUF;
        const T* s_z = panel + K * GROUP_ROWS;
        if (ct < GROUP_ROWS) {
          T s0 = 0, s1 = 0;
#pragma unroll 8
          for (int k = 0; k < K; k += 2) { s0 += panel[k * GROUP_ROWS + ct] * s_z[k]; s1 += panel[(k + 1) * GROUP_ROWS + ct] * s_z[k + 1]; }
          bacc += (double)(s0 + s1);
        }
        schur_consume_v<Cfg, K>(cw, panel + lane_off, panel + lane_off, acc);
      }
      if (dbg && blockIdx.x == 0 && threadIdx.x == NPROD && it < 20) dbg[2 * it + 1] = clock64();
      __syncthreads();
    }
    __syncthreads();
    // wait for the camera blocks U_c of this workgroup (folded below by everybody), then take them out of the tiles:
    // the slab then holds this workgroup's share of  sum Ytilde Ytilde^T - U,  and k_build_exchange's plain sum of the
    // slabs is -S.  Only tiles on and next to the diagonal contain entries of an 11x11 camera block.
    fold_u();
    __syncthreads();
    constexpr int LO = 0;
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
    (void)LO;
    if (stamp_wg && threadIdx.x == NPROD) dbg[51] = clock64();   // U folded and taken out of the tiles
    T* slab = slabs + (size_t)blockIdx.x * (size_t)(GROUP_TILES * GROUP_TILES) * 256;
    schur_store_v<Cfg>(cw, slab, lane, acc);
    if (stamp_wg && threadIdx.x == NPROD) dbg[52] = clock64();   // slab stores issued
    if (ct < GROUP_ROWS) {
      const int c = ct / NCP, e = ct - c * NCP;
      const double gpart = (c < C) ? (double)s_Ured[c * UPK + NCP * (NCP + 1) / 2 + e] : 0.0;
      const double dpart = (c < C) ? (double)s_Ured[c * UPK + (e * NCP - (e * (e - 1)) / 2)] : 0.0;
      bpart[(size_t)blockIdx.x * GROUP_ROWS + ct] = bacc - gpart;           // rhs = sum (b - g_c) over the workgroups
      gdpart[((size_t)blockIdx.x * 2 + 0) * GROUP_ROWS + ct] = gpart;
      gdpart[((size_t)blockIdx.x * 2 + 1) * GROUP_ROWS + ct] = dpart;
    }
  }
  if (threadIdx.x == 0) {
    double cs = 0, gm = 0;
    for (int wv = 0; wv < NPROD / 64; ++wv) { cs += s_scr[0][wv]; gm = fmax(gm, s_scr[1][wv]); }
    cost_part[blockIdx.x] = 0.5 * cs;
    gmax_part[blockIdx.x] = gm;
    if (stamp_wg) dbg[53] = clock64();                           // kernel exit (thread 0)
  }
}


// ------------------------------------------------------------------ K3+K4 fused, Schur products on the bf16 matrix pipe
// Same kernel structure and the same producer mathematics as k_schur_fused; what changes is how panel^T panel is formed.
// On gfx950 the f32-input MFMA runs on the SIMD's f32 FMA lanes (tools/micro/mix_waves.hip: a VALU-only wave beside a
// saturating f32-MFMA wave makes no progress at all), so the 6.3k MFMA cycles of a 16-point chunk and the ~5k cycles of
// producer arithmetic simply add up.  The bf16 MFMA is a separate pipe.  Each f32 panel value y is therefore split EXACTLY
// into three bf16 pieces, y = h + m + l exactly (8 + 8 + 8 mantissa bits; h = bf16(y), m = bf16(y - h), l = (y - h) - m,
// round to nearest), and a product of two panel values is accumulated in f32 as the six partial
// products of combined order <= 2,
//     y y' ~= h h' + h m' + m h' + h l' + l h' + m m'        (dropped: m l', l m', l l' <= 2^-24 |y y'|),
// each of them exact in f32 (8 x 8 bits).  Error per product <= 3 * 2^-24 relative, the order of the f32 MFMA's own
// rounding -- this is f32 arithmetic carried by six 16x16x32 bf16 MFMAs (16 cycles each, K = 32 = 8 points x (3 + 1 pad))
// instead of eight 16x16x4 f32 MFMAs (32 cycles each): 2.7x fewer matrix cycles, and they overlap with the producers.
//   LDS: three bf16 planes per buffer, double-buffered: 2 x 67,584 B.  A plane is [11 tiles][2 halves][16 rows][8 slots] of
//   8-byte point slots (3 values + a zero pad): point q of a 16-point chunk goes to half (q >> 2) & 1, slot
//   sigma = (q & 3) | ((q >> 3) << 2), so that the two points a consumer lane needs for one k-step (q = 8 s + g and
//   8 s + g + 4, g = lane >> 4; the k order inside a k-step is arbitrary as long as both operands agree) are the same slot
//   of the two halves, 1024 bytes apart: one ds_read2st64_b64 returns the whole 16-byte MFMA operand (the halves of one tile
//   are neighbours so that the compiler's load merging pairs them and not two tiles).  Rows are PARAMETER-major, row = 16 e + c
//   (tile b of the MFMA = parameter b of all 16 cameras; 11 parameters = 11 tiles), and slot sigma of row (e, c) is stored
//   at sigma ^ (c >> 1):
//     * a producer ds_write_b64 is issued by 16 lanes with the same q and c = 0..15: rows are 64 bytes, so the row parity
//       picks the bank half and the 8 cameras of one parity hit 8 different slots -> all 32 banks, no conflict (the
//       camera-major [176][72] layout of the first version measured 44 % of its LDS cycles as bank conflicts,
//       SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE);
//     * one 32-lane pass of a consumer read covers rows i = 0..15 and g = 0, 1: i & 3 picks the 16-bank quarter and the 8
//       lanes that share it read 8 different slots -> no conflict.
//   k_build_exchange undoes the row order (flag emajor).
//   The right-hand side b = panel^T z moves into the producers (11 more register accumulators per lane), so the
//   consumers touch nothing but bf16 fragments.
struct SchurBf3Cfg {
  using elem = float;
  static constexpr bool diag = true;
  static constexpr int THREADS = SCHUR_THREADS, NPROD = 256, NCW = 4, TS = 1, NV = 4;
  static constexpr int NTILE = (GROUP_TILES * (GROUP_TILES + 1)) / 2;
  static constexpr int TPW = (NTILE + NV - 1) / NV;
  static constexpr int PTS = 16, KQ = 4, K = PTS * KQ;            // 64 panel columns per chunk = 2 MFMA k-steps of 32
  static constexpr int HALF_BYTES = 16 * 64;                       // one half of a tile: [16 rows][8 slots][4 bf16]
  static constexpr int PLANE = GROUP_ROWS * K;                     // bf16 elements per plane (two half-planes)
  static constexpr int BUF_BYTES = 3 * PLANE * 2;                  // h, m, l
  static constexpr int UPKB = UPK + NCP;                           // per-lane accumulators handed over at the end: U (66) + g (11) + b (11)
  static constexpr int UPKS = UPKB | 1;                            // their row stride in the hand-over area: odd, so that the 64 lanes of a
                                                                   // store hit all banks (a stride of 88 floats put them on 4 banks, 16-way)
  static constexpr size_t LDS_BYTES = 2 * (size_t)BUF_BYTES + (size_t)GROUP_CAMS * CAMPRE * sizeof(float);
  static_assert(2 * (size_t)BUF_BYTES >= (size_t)(NPROD + GROUP_CAMS) * UPKS * sizeof(float), "the accumulator hand-over reuses the panel buffers");
};

__global__ __launch_bounds__(SCHUR_THREADS) void k_schur_fused_bf3(
    const ParamSets<float> ps, const FusedDecide fd, int C,
    const float2* __restrict__ uv, const float* __restrict__ w, const int32_t* __restrict__ pt_start,
    const uint16_t* __restrict__ vis, int N, int ksplit, double* __restrict__ D2p, double* __restrict__ gp,
    float* __restrict__ pf, float* __restrict__ slabs, double* __restrict__ bpart, double* __restrict__ gdpart,
    double* __restrict__ cost_part, double* __restrict__ gmax_part, long long* __restrict__ dbg,
    int exp_arg /* timing experiments, only in a build with -DSBA_SCHUR_EXP_BUILD (make EXTRA=-DSBA_SCHUR_EXP_BUILD; SBA_SCHUR_EXP=1: no m / l planes,
                   2: no U_c / g_c accumulation; results are WRONG when set -- docs/EXPERIMENTS.md, profiles/r4_fused_bf3_sensitivity.txt) */) {
  extern __shared__ __align__(16) unsigned char smem[];
  using T = float;
  using Cfg = SchurBf3Cfg;
#ifdef SBA_SCHUR_EXP_BUILD
  const int exp_flags = exp_arg;
#else
  constexpr int exp_flags = 0;          // the production kernel carries none of the experiment's branches
  (void)exp_arg;
#endif
  constexpr int THREADS = Cfg::THREADS, NPROD = Cfg::NPROD, TPW = Cfg::TPW, PTS = Cfg::PTS, UPKB = Cfg::UPKB, UPKS = Cfg::UPKS;
  const bool stamp_wg = dbg && blockIdx.x == 0;
  if (stamp_wg && threadIdx.x == 0) dbg[48] = clock64();
  __shared__ LMState s_st;
  __shared__ LMLogRow s_row;
  __shared__ int s_have_row;
  __shared__ double s_scr[2][NPROD / 64];
  T* s_cam = reinterpret_cast<T*>(smem + 2 * Cfg::BUF_BYTES);            // [16][CAMPRE]
  T* s_U = reinterpret_cast<T*>(smem);                                   // [256][UPKS] once the panels are done with
  T* s_Ured = s_U + NPROD * UPKS;                                        // [C][UPKB]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool producer = threadIdx.x < NPROD;
  // the slice of this workgroup: a multiple of 8 points (one 32-deep k-step), NOT of the 16-point chunk -- at 50 000 points and 256
  // workgroups that is 200 points = 12 chunks + 8 points on 250 workgroups instead of 13 chunks on 240; the half chunk at the end
  // costs half (producer waves 0 and 1 work, the consumers take one k-step)
  int per = (N + ksplit - 1) / ksplit;
  per = ((per + 7) / 8) * 8;
  const int pbeg = min(N, (int)blockIdx.x * per), pend = min(N, pbeg + per);
  const int nchunk = (pend - pbeg + PTS - 1) / PTS;
  // ---- prologue.  (Requesting the camera table and the first chunk for BOTH parameter sets next to the record, so that the
  // whole prologue is one round trip to memory and the selection happens in registers, was measured: +1.5 us per iteration.)
  DecidePartials dp;
  {
    constexpr int NWORD = sizeof(LMState) / 4;
    if ((int)threadIdx.x < NWORD) reinterpret_cast<int*>(&s_st)[threadIdx.x] = reinterpret_cast<const int*>(fd.st_in)[threadIdx.x];
    if (fd.do_decide) decide_gather(dp, fd.scal_all, fd.trial_part, fd.gmax_in, fd.n_trial, fd.n_gmax);
    {   // zero both panel buffers meanwhile: the rows of cameras >= C are never written
      uint4* z4 = reinterpret_cast<uint4*>(smem);
      for (int i = threadIdx.x; i < 2 * Cfg::BUF_BYTES / 16; i += THREADS) z4[i] = make_uint4(0, 0, 0, 0);
    }
    // pin the partial sums in front of the barrier: left alone, LLVM sinks their loads into the conditional block that uses them
    // (behind the barrier), which put a whole memory round trip (4k cycles) back between the record's arrival and the decision
    asm volatile("" : "+v"(dp.a), "+v"(dp.b), "+v"(dp.c), "+v"(dp.d), "+v"(dp.g));
    if (stamp_wg && threadIdx.x == 0) dbg[54] = clock64();
    __syncthreads();
    if (stamp_wg && threadIdx.x == 0) dbg[55] = clock64();
    if (fd.do_decide) {
      const bool running = s_st.status < 0;                 // uniform; a finished solve only has its record carried over
      bool have_row = false;
      static_assert(THREADS == DECIDE_THREADS, "decide_fold is written for the thread count of this kernel");
      // scratch of the reductions: the first 20 KB of the (zeroed, still unused) panel buffers, cleared again afterwards
      if (running) have_row = decide_core(&s_st, dp, fd.scal_all, fd.n_ranks, &s_row, fd.log_cap, reinterpret_cast<double*>(smem),
                                          stamp_wg ? dbg + 58 : nullptr);
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
  if (stamp_wg && threadIdx.x == 0) dbg[57] = clock64();
  if (s_st.status >= 0) return;
  const LMState* st = &s_st;
  const int cur_ = ps_cur(ps, st);
  const T* __restrict__ ptsT = ps.ptsT[cur_];
  for (int i = threadIdx.x; i < C * CAMPRE; i += THREADS) s_cam[i] = ps.campre[cur_][i];
  if (stamp_wg && threadIdx.x == 0) dbg[56] = clock64();
  const T lam = (T)st->lam;
  auto fold_u = [&]() {
    for (int o = threadIdx.x; o < C * UPKB; o += THREADS) {
      const int c = o / UPKB, k = o - c * UPKB;
      T sum = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) sum += s_U[(q * 16 + c) * UPKS + k];
      s_Ured[o] = sum;
    }
  };
  __syncthreads();

  if (producer) {
    const int q = threadIdx.x >> 4, c = threadIdx.x & 15;
    const bool cam_ok = c < C;
    const T* cp_safe = s_cam + (cam_ok ? c : 0) * CAMPRE;
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
      if (i_pt && vis) { i_mask = vis[p]; i_start = pt_start[p]; }
    };
    auto request = [&](int chunk) {
      const int p = pbeg + chunk * PTS + q;
      n_pt = i_pt;
      n_valid = i_pt && cam_ok && ((i_mask >> c) & 1u);
      if (n_pt) {
        n_X[0] = ptsT[3 * (size_t)p]; n_X[1] = ptsT[3 * (size_t)p + 1]; n_X[2] = ptsT[3 * (size_t)p + 2];
        n_D[0] = D2p[3 * (size_t)p]; n_D[1] = D2p[3 * (size_t)p + 1]; n_D[2] = D2p[3 * (size_t)p + 2];
      }
      if (n_valid) {
        const size_t o = vis ? (size_t)i_start + __builtin_popcount(i_mask & ((1u << c) - 1u)) : (size_t)p * C + c;
        n_uv = uv[o];
        n_w = w ? w[o] : (T)1;
      }
      request_index(chunk + 1);
    };
    request_index(0);
    request(0);
    if (stamp_wg && threadIdx.x == 0) dbg[49] = clock64();
    // byte offset of this lane's 8-byte slot inside a plane: half (q >> 2) & 1, row 16 e + c, slot sigma(q) ^ (c >> 1)
    const int lane_slot = ((q >> 2) & 1) * Cfg::HALF_BYTES + c * 64 + ((((q & 3) | ((q >> 3) << 2)) ^ (c >> 1)) << 3);
    for (int it = 0; it <= nchunk; ++it) {
      if (it < nchunk && (wid < 2 || pbeg + it * PTS + 8 < pend)) {      // (points 8 .. 15 of a chunk = the second k-step: waves 2, 3)
        unsigned char* pbuf = smem + (it & 1) * Cfg::BUF_BYTES;
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
        if (have_pt && c == 0) {
          D2p[3 * (size_t)p] = E0; D2p[3 * (size_t)p + 1] = E1; D2p[3 * (size_t)p + 2] = E2;
          gp[3 * (size_t)p] = (double)g3[0]; gp[3 * (size_t)p + 1] = (double)g3[1]; gp[3 * (size_t)p + 2] = (double)g3[2];
          float4* o4 = reinterpret_cast<float4*>(pf + (size_t)p * PF);
          o4[0] = make_float4(f[0], f[1], f[2], f[3]);
          o4[1] = make_float4(f[4], f[5], f[6], f[7]);
          o4[2] = make_float4(f[8], f[9], f[10], f[11]);
        }
        // Ytilde = Jc^T (Jp L^-T) (11x3; all zero for a degenerate point because f is), its three bf16 pieces into the planes,
        // and the right-hand side  b_c += Ytilde z
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
            // (Jc[1][cx] = Jc[0][cy] = 0 structurally -- obs_resjac, and the robust scaling keeps zeros: those products are left out,
            //  here and in U_c / g_c below; the results are the same bit for bit)
            T y[3];
#pragma unroll
            for (int d = 0; d < 3; ++d)
              y[d] = e == CP_CX ? Jc[0][e] * Jt[0][d] : e == CP_CY ? Jc[1][e] * Jt[1][d] : Jc[0][e] * Jt[0][d] + Jc[1][e] * Jt[1][d];
            Uacc[UPK + e] = __builtin_fmaf(y[2], f[8], __builtin_fmaf(y[1], f[7], __builtin_fmaf(y[0], f[6], Uacc[UPK + e])));
            // three-way split by round-to-nearest (v_cvt_pk_bf16_f32 converts and packs two values in one instruction): the
            // remainders y - h and (y - h) - m are exact in f32 and the last one has at most 8 significant bits
            // (inline asm: written as __bf16 conversions the compiler converts the low halves a second time and masks the pad)
            auto pk = [](float lo, float hi) -> unsigned {
              unsigned v;
              asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v) : "v"(lo), "v"(hi));
              return v;
            };
            auto pk1 = [](float lo) -> unsigned {           // upper half = bf16(0) = the slot's zero pad
              unsigned v;
              asm("v_cvt_pk_bf16_f32 %0, %1, 0" : "=v"(v) : "v"(lo));
              return v;
            };
            auto lo_f = [](unsigned pkd) -> float { return __builtin_bit_cast(float, pkd << 16); };
            auto hi_f = [](unsigned pkd) -> float { return __builtin_bit_cast(float, pkd & 0xffff0000u); };
            const unsigned h01 = pk(y[0], y[1]), h2 = pk1(y[2]);
            unsigned char* dst = pbuf + lane_slot + e * (2 * Cfg::HALF_BYTES);
            *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h2);
            if (!(exp_flags & 1)) {
              const float r0 = y[0] - lo_f(h01), r1 = y[1] - hi_f(h01), r2 = y[2] - lo_f(h2);
              const unsigned m01 = pk(r0, r1), m2 = pk1(r2);
              const float s0 = r0 - lo_f(m01), s1 = r1 - hi_f(m01), s2 = r2 - lo_f(m2);
              const unsigned l01 = pk(s0, s1), l2 = pk1(s2);
              *reinterpret_cast<uint2*>(dst + Cfg::PLANE * 2) = make_uint2(m01, m2);
              *reinterpret_cast<uint2*>(dst + 2 * Cfg::PLANE * 2) = make_uint2(l01, l2);
            }
          });
        }
        if (!(exp_flags & 2))
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
    __syncthreads();                       // the consumers have read the last panel: the buffers become the hand-over area
    static_for<0, UPKB>([&](auto kc) { constexpr int k = decltype(kc)::value; s_U[threadIdx.x * UPKS + k] = Uacc[k]; });
    const double cs = wave_sum((double)sq), gm = wave_max((double)gmx);
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
    // fragment of row tile b, k-step s: slot (g | 4 s) ^ (i >> 1) of row 16 b + i in both half-planes (i = lane & 15, g = lane >> 4)
    int frag_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) frag_off[s] = (lane & 15) * 64 + ((((lane >> 4) | (4 * s)) ^ ((lane & 15) >> 1)) << 3);
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    for (int it = 0; it <= nchunk; ++it) {
      if (it >= 1) {
        const unsigned char* pbuf = smem + ((it - 1) & 1) * Cfg::BUF_BYTES;
        const int nstep = (pend - (pbeg + (it - 1) * PTS) > 8) ? 2 : 1;     // k-steps the producers have written (last chunk: maybe one)
        static_for<0, Cfg::NV>([&](auto vc) {
          constexpr int V = decltype(vc)::value;
          if (cw == V) {
            constexpr int LO = schur_lo(Cfg::NTILE, Cfg::NV, V), HI = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
            constexpr int RMIN = schur_tile_R(true, LO);
#pragma unroll
            for (int s = 0; s < Cfg::K / 32; ++s) {
              if (s >= nstep) break;
              // all three planes of this k-step are requested before the first MFMA (the scheduling barrier keeps the
              // compiler from sinking the reads next to their uses, which exposed one LDS round trip per few MFMAs)
              bf16x8_t fh[GROUP_TILES], fm[GROUP_TILES], fl[GROUP_TILES];
              auto load = [&](bf16x8_t (&dst)[GROUP_TILES], int plane) {
#pragma unroll
                for (int b = RMIN; b < GROUP_TILES; ++b) {
                  const unsigned char* rowp = pbuf + plane * (Cfg::PLANE * 2) + b * (2 * Cfg::HALF_BYTES) + frag_off[s];
                  const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(rowp);
                  const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(rowp + Cfg::HALF_BYTES);
                  dst[b] = __builtin_bit_cast(bf16x8_t, u32x4_t{lo[0], lo[1], hi[0], hi[1]});
                }
              };
              load(fh, 0);
              load(fm, 1);
              load(fl, 2);
              __builtin_amdgcn_sched_barrier(0);
              // h h'
              static_for<LO, HI>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int R = schur_tile_R(true, t), Tc = schur_tile_T(true, t);
                acc[t - LO] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[R], fh[Tc], acc[t - LO], 0, 0, 0);
              });
              // h m' + m h' + m m'
              static_for<LO, HI>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int R = schur_tile_R(true, t), Tc = schur_tile_T(true, t);
                acc[t - LO] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[R], fm[Tc], acc[t - LO], 0, 0, 0);
                acc[t - LO] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm[R], fh[Tc], acc[t - LO], 0, 0, 0);
                acc[t - LO] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm[R], fm[Tc], acc[t - LO], 0, 0, 0);
              });
              // h l' + l h'
              static_for<LO, HI>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int R = schur_tile_R(true, t), Tc = schur_tile_T(true, t);
                acc[t - LO] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[R], fl[Tc], acc[t - LO], 0, 0, 0);
                acc[t - LO] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[R], fh[Tc], acc[t - LO], 0, 0, 0);
              });
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        });
      }
      if (stamp_wg && threadIdx.x == NPROD && it < 20) dbg[2 * it + 1] = clock64();
      __syncthreads();
    }
    __syncthreads();                       // matches the producers' barrier in front of the hand-over
    __syncthreads();                       // accumulators are in s_U
    fold_u();
    __syncthreads();
    static_for<0, Cfg::NV>([&](auto vc) {
      constexpr int V = decltype(vc)::value;
      if (cw == V) {
        constexpr int T0 = schur_lo(Cfg::NTILE, Cfg::NV, V), T1 = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
        static_for<T0, T1>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          constexpr int R = schur_tile_R(true, t), Tc = schur_tile_T(true, t);
          // parameter-major rows: tile (R, Tc) holds parameters (R, Tc) of every camera pair, the camera's own block U sits
          // on the tile's diagonal (row camera == column camera): at most one of a lane's four registers
          const int cj_ = lane & 15, rgm = cj_ - 4 * (lane >> 4);
          const T u = (rgm >= 0 && rgm < 4 && cj_ < C) ? s_Ured[cj_ * UPKB + (R * NCP - (R * (R - 1)) / 2 + (Tc - R))] : (T)0;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) acc[t - T0][rg] -= (rg == rgm) ? u : (T)0;
        });
      }
    });
    if (stamp_wg && threadIdx.x == NPROD) dbg[51] = clock64();
    T* slab = slabs + (size_t)blockIdx.x * (size_t)(GROUP_TILES * GROUP_TILES) * 256;
    schur_store_v<Cfg>(cw, slab, lane, acc);
    if (stamp_wg && threadIdx.x == NPROD) dbg[52] = clock64();
    if (ct < GROUP_ROWS) {
      const int c = ct / NCP, e = ct - c * NCP;
      const double gpart = (c < C) ? (double)s_Ured[c * UPKB + NCP * (NCP + 1) / 2 + e] : 0.0;
      const double dpart = (c < C) ? (double)s_Ured[c * UPKB + (e * NCP - (e * (e - 1)) / 2)] : 0.0;
      const double bsum = (c < C) ? (double)s_Ured[c * UPKB + UPK + e] : 0.0;
      bpart[(size_t)blockIdx.x * GROUP_ROWS + ct] = bsum - gpart;             // rhs = sum (b - g_c) over the workgroups
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

#endif  // SBA_NCP == 11 (fused linearise + Schur kernels)

// ------------------------------------------------------------------ group pairs of a multi-group rig on the bf16 pipe (both camera models)
// LDS geometry of one 16-camera panel in parameter-major order (NCP tiles of 16 rows), shared with k_schur_fused_bf3
struct SchurPairCfg {
  using elem = float;
  static constexpr bool diag = true;
  static constexpr int THREADS = SCHUR_THREADS, NPROD = 256, NCW = 4, TS = 1, NV = 4;
  static constexpr int NTILE = (GROUP_TILES * (GROUP_TILES + 1)) / 2;
  static constexpr int TPW = (NTILE + NV - 1) / NV;
  static constexpr int PTS = 16, K = 64;                           // 16-point chunks = 2 MFMA k-steps of 32
  static constexpr int HALF_BYTES = 16 * 64;                       // one half of a tile: [16 rows][8 slots][4 bf16]
  static constexpr int PLANE = GROUP_ROWS * K;                     // bf16 elements per plane (two half-planes)
  static constexpr int BUF_BYTES = 3 * PLANE * 2;                  // h, m, l
  static constexpr size_t LDS_BYTES = 2 * (size_t)BUF_BYTES + (size_t)GROUP_CAMS * CAMPRE * sizeof(float);
  static_assert(LDS_BYTES <= 160 * 1024, "the double-buffered panel has to fit the 160 KB LDS (13 parameters: 161,472 B)");
};

// ------------------------------------------------------------------ diagonal group pair of a multi-group rig on the bf16 pipe
// The Schur block of ONE camera group with itself (16 cameras, 176 rows) when the rig has several groups: the producers and
// consumers of k_schur_fused_bf3 without the linearisation around them -- the point factors L^-1, z come from k_point_factor
// (they involve every camera that sees the point, not just this group), U and g_c from k_linearize_cams, the decision from
// k_decide.  Lane (q, c) = (point of the 16-point chunk, camera of the group) finds its observation through the k_group_index
// tables; a lane without one runs the same code with weight 0 and writes zeros.  Same LDS planes, same six bf16 MFMAs per
// tile and k-step, same parameter-major tile order (k_build_exchange: emajor = 2 undoes it for the diagonal pairs).
// grid = (ksplit, ngroups).  Replaces k_schur<float, true> there: 4 x fewer matrix cycles, and they overlap with the producers.
__global__ __launch_bounds__(SCHUR_THREADS) void k_schur_diag_bf3(
    const ParamSets<float> ps, const LMState* __restrict__ st, int C,
    const float2* __restrict__ uv, const float* __restrict__ w, const uint16_t* __restrict__ gmask, const int32_t* __restrict__ gstart,
    int N, const float* __restrict__ pf, const int32_t* __restrict__ pair_ga, int pair0, int ksplit,
    float* __restrict__ slabs, double* __restrict__ bpart,
    double* __restrict__ gdpart /* round 4: [group][ksplit][2][GROUP_ROWS] partial g_c / diag U rows.  The producers evaluate Jc anyway, so they also
                                    accumulate the group's camera blocks U_c = sum Jc^T Jc and g_c = sum Jc^T r (the fused kernel's 77 / 104 per-lane
                                    register accumulators); U_c is subtracted on the tiles' camera diagonals (the slab then holds Schur partials - U),
                                    bpart becomes b - g_c, and k_linearize_cams + k_reduce_cams are not launched.  NULL: the round-3 behaviour */) {
  extern __shared__ __align__(16) unsigned char smem[];
  using T = float;
  using Cfg = SchurPairCfg;
  constexpr int THREADS = Cfg::THREADS, NPROD = Cfg::NPROD, TPW = Cfg::TPW, PTS = Cfg::PTS;
  constexpr int UPKB = UPK + NCP, UPKS = UPKB | 1;                     // U (upper triangle) + g + b per lane; odd row stride in the hand-over area
  static_assert(2 * (size_t)Cfg::BUF_BYTES >= (size_t)(NPROD + GROUP_CAMS) * UPKS * sizeof(float), "the accumulator hand-over reuses the panel buffers");
  if (st->status >= 0) return;
  const int cur_ = ps_cur(ps, st);
  const T* __restrict__ ptsT = ps.ptsT[cur_];
  const int pair = pair0 + blockIdx.y;
  const int ga = pair_ga[pair];
  const int camA0 = ga * GROUP_CAMS, nA = min(GROUP_CAMS, C - camA0);
  const uint16_t* __restrict__ gm = gmask + (size_t)ga * N;
  const int32_t* __restrict__ gs = gstart + (size_t)ga * N;
  T* s_cam = reinterpret_cast<T*>(smem + 2 * Cfg::BUF_BYTES);            // [16][CAMPRE]
  T* s_B = reinterpret_cast<T*>(smem);                                   // [256][NCP] once the panels are done with (gdpart == NULL)
  T* s_U = reinterpret_cast<T*>(smem);                                   // [256][UPKS] once the panels are done with (gdpart != NULL)
  T* s_Ured = s_U + NPROD * UPKS;                                        // [16][UPKB]
  const bool fold_u = gdpart != nullptr;                                 // (a kernel argument: uniform)
  auto fold_lanes = [&]() {                                             // the 16 lanes (points of a chunk) that served camera c fold their accumulators
    for (int o = threadIdx.x; o < nA * UPKB; o += THREADS) {
      const int c = o / UPKB, kk = o - c * UPKB;
      T sum = 0;
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) sum += s_U[(qq * 16 + c) * UPKS + kk];
      s_Ured[o] = sum;
    }
  };
  {   // zero both panel buffers: the rows of cameras >= nA are never written
    uint4* z4 = reinterpret_cast<uint4*>(smem);
    for (int i = threadIdx.x; i < 2 * Cfg::BUF_BYTES / 16; i += THREADS) z4[i] = make_uint4(0, 0, 0, 0);
  }
  for (int i = threadIdx.x; i < nA * CAMPRE; i += THREADS) s_cam[i] = ps.campre[cur_][(size_t)camA0 * CAMPRE + i];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool producer = threadIdx.x < NPROD;
  int per = (N + ksplit - 1) / ksplit;
  per = ((per + PTS - 1) / PTS) * PTS;
  const int pbeg = min(N, (int)blockIdx.x * per), pend = min(N, pbeg + per);
  const int nchunk = (pend - pbeg + PTS - 1) / PTS;
  __syncthreads();

  if (producer) {
    const int q = threadIdx.x >> 4, c = threadIdx.x & 15;
    const bool cam_ok = c < nA;
    const T* cp_safe = s_cam + (cam_ok ? c : 0) * CAMPRE;
    T bacc[NCP];
#pragma unroll
    for (int e = 0; e < NCP; ++e) bacc[e] = (T)0;
    T Uacc[UPK];                                                         // U_c upper triangle + g_c (only touched when fold_u)
    static_for<0, UPK>([&](auto kc) { Uacc[decltype(kc)::value] = (T)0; });
    float2 n_uv = make_float2(0.f, 0.f);
    T n_w = 1, n_X[3] = {0, 0, 0}, n_f[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) n_f[k] = (T)0;
    bool n_valid = false;
    unsigned i_mask = 0; int i_start = 0; bool i_pt = false;
    auto request_index = [&](int chunk) {
      const int p = pbeg + chunk * PTS + q;
      i_pt = chunk < nchunk && p < pend;
      i_mask = 0; i_start = 0;
      if (i_pt) { i_mask = gm[p]; i_start = gs[p]; }
    };
    auto request = [&](int chunk) {
      const int p = pbeg + chunk * PTS + q;
      n_valid = i_pt && cam_ok && ((i_mask >> c) & 1u);
      if (i_pt) {
        n_X[0] = ptsT[3 * (size_t)p]; n_X[1] = ptsT[3 * (size_t)p + 1]; n_X[2] = ptsT[3 * (size_t)p + 2];
        const float4* f4 = reinterpret_cast<const float4*>(pf + (size_t)p * PF);
        const float4 a = f4[0], b = f4[1], d = f4[2];
        n_f[0] = a.x; n_f[1] = a.y; n_f[2] = a.z; n_f[3] = a.w; n_f[4] = b.x; n_f[5] = b.y; n_f[6] = b.z; n_f[7] = b.w;
        n_f[8] = d.x; n_f[9] = d.y; n_f[10] = d.z; n_f[11] = d.w;
      } else {
#pragma unroll
        for (int k = 0; k < PF; ++k) n_f[k] = (T)0;
      }
      if (n_valid) {
        const size_t o = (size_t)i_start + __builtin_popcount(i_mask & ((1u << c) - 1u));
        n_uv = uv[o];
        n_w = w ? w[o] : (T)1;
      }
      request_index(chunk + 1);
    };
    request_index(0);
    request(0);
    const int lane_slot = ((q >> 2) & 1) * Cfg::HALF_BYTES + c * 64 + ((((q & 3) | ((q >> 3) << 2)) ^ (c >> 1)) << 3);
    for (int it = 0; it <= nchunk; ++it) {
      if (it < nchunk) {
        unsigned char* pbuf = smem + (it & 1) * Cfg::BUF_BYTES;
        const bool valid = n_valid;
        const float2 m = n_uv;
        const T ww = n_w, X0 = n_X[0], X1 = n_X[1], X2 = n_X[2];
        T f[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) f[k] = n_f[k];
        request(it + 1);
        T r[2], Jc[2][NCP], Jp[2][3];
        obs_resjac<T>(cp_safe, X0, X1, X2, m.x, m.y, valid ? ww : (T)0, r, Jc, Jp, valid);
        (void)robust_apply<T>(ps.loss(), r, Jc, Jp);
        // Ytilde = Jc^T (Jp L^-T) (all zero for a degenerate or fixed point: its factor row is zero), its three bf16 pieces
        // into the planes, and the right-hand side  b_c += Ytilde z
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
            for (int d = 0; d < 3; ++d) y[d] = Jc[0][e] * Jt[0][d] + Jc[1][e] * Jt[1][d];
            bacc[e] = __builtin_fmaf(y[2], f[8], __builtin_fmaf(y[1], f[7], __builtin_fmaf(y[0], f[6], bacc[e])));
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
            unsigned char* dst = pbuf + lane_slot + e * (2 * Cfg::HALF_BYTES);
            *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h2);
            *reinterpret_cast<uint2*>(dst + Cfg::PLANE * 2) = make_uint2(m01, m2);
            *reinterpret_cast<uint2*>(dst + 2 * Cfg::PLANE * 2) = make_uint2(l01, l2);
          });
        }
        if (fold_u) {
          // the camera's own block and gradient, as in k_schur_fused_bf3 (a lane without an observation has Jc = 0, r = 0)
          static_for<0, NCP>([&](auto ac) {
            constexpr int a = decltype(ac)::value;
            static_for<a, NCP>([&](auto bc) {
              constexpr int b = decltype(bc)::value;
              constexpr int kk = a * NCP - (a * (a - 1)) / 2 + (b - a);
              Uacc[kk] = __builtin_fmaf(Jc[1][a], Jc[1][b], __builtin_fmaf(Jc[0][a], Jc[0][b], Uacc[kk]));
            });
            constexpr int kg = NCP * (NCP + 1) / 2 + a;
            Uacc[kg] = __builtin_fmaf(Jc[1][a], r[1], __builtin_fmaf(Jc[0][a], r[0], Uacc[kg]));
          });
        }
      }
      __syncthreads();
    }
    __syncthreads();                       // the consumers have read the last panel: the buffers become the hand-over area
    if (fold_u) {
      static_for<0, UPK>([&](auto kc) { constexpr int kk = decltype(kc)::value; s_U[threadIdx.x * UPKS + kk] = Uacc[kk]; });
#pragma unroll
      for (int e = 0; e < NCP; ++e) s_U[threadIdx.x * UPKS + UPK + e] = bacc[e];
      __syncthreads();
      fold_lanes();
      __syncthreads();
    } else {
#pragma unroll
      for (int e = 0; e < NCP; ++e) s_B[threadIdx.x * NCP + e] = bacc[e];
      __syncthreads();
      if ((int)threadIdx.x < GROUP_ROWS) {
        const int cc = threadIdx.x / NCP, e = threadIdx.x - cc * NCP;
        double sum = 0;
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) sum += (double)s_B[(qq * 16 + cc) * NCP + e];
        bpart[((size_t)ga * ksplit + blockIdx.x) * GROUP_ROWS + threadIdx.x] = (cc < nA) ? sum : 0.0;
      }
    }
  } else {
    const int cw = wid - NPROD / 64;
    typename Mfma<T>::acc_t acc[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) acc[s] = typename Mfma<T>::acc_t{0, 0, 0, 0};
    int frag_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) frag_off[s] = (lane & 15) * 64 + ((((lane >> 4) | (4 * s)) ^ ((lane & 15) >> 1)) << 3);
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    for (int it = 0; it <= nchunk; ++it) {
      if (it >= 1) {
        const unsigned char* pbuf = smem + ((it - 1) & 1) * Cfg::BUF_BYTES;
        static_for<0, Cfg::NV>([&](auto vc) {
          constexpr int V = decltype(vc)::value;
          if (cw == V) {
            constexpr int LO = schur_lo(Cfg::NTILE, Cfg::NV, V), HI = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
            constexpr int RMIN = schur_tile_R(true, LO);
#pragma unroll
            for (int s = 0; s < Cfg::K / 32; ++s) {
              auto load = [&](bf16x8_t (&dst)[GROUP_TILES], int plane) {
#pragma unroll
                for (int b = RMIN; b < GROUP_TILES; ++b) {
                  const unsigned char* rowp = pbuf + plane * (Cfg::PLANE * 2) + b * (2 * Cfg::HALF_BYTES) + frag_off[s];
                  const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(rowp);
                  const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(rowp + Cfg::HALF_BYTES);
                  dst[b] = __builtin_bit_cast(bf16x8_t, u32x4_t{lo[0], lo[1], hi[0], hi[1]});
                }
              };
              auto products = [&](const bf16x8_t (&fx)[GROUP_TILES], const bf16x8_t (&fy)[GROUP_TILES], bool both) {
                static_for<LO, HI>([&](auto tc) {
                  constexpr int t = decltype(tc)::value;
                  constexpr int R = schur_tile_R(true, t), Tc = schur_tile_T(true, t);
                  acc[t - LO] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[R], fy[Tc], acc[t - LO], 0, 0, 0);
                  if (both) acc[t - LO] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[R], fx[Tc], acc[t - LO], 0, 0, 0);
                });
              };
              if constexpr (NCP <= 11) {
                // 11 tiles: all three planes of the k-step in registers before the first MFMA (132 fragment + 68 accumulator VGPRs)
                bf16x8_t fh[GROUP_TILES], fm[GROUP_TILES], fl[GROUP_TILES];
                load(fh, 0);
                load(fm, 1);
                load(fl, 2);
                __builtin_amdgcn_sched_barrier(0);
                products(fh, fh, false);
                products(fh, fm, true);
                products(fm, fm, false);
                products(fh, fl, true);
                __builtin_amdgcn_sched_barrier(0);
              } else {
                // 13 tiles (91 of them, 92 accumulator VGPRs): two fragment sets, the l plane reuses m's registers
                bf16x8_t fh[GROUP_TILES], fo[GROUP_TILES];
                load(fh, 0);
                load(fo, 1);
                __builtin_amdgcn_sched_barrier(0);
                products(fh, fh, false);
                products(fh, fo, true);
                products(fo, fo, false);
                __builtin_amdgcn_sched_barrier(0);
                load(fo, 2);
                __builtin_amdgcn_sched_barrier(0);
                products(fh, fo, true);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
        });
      }
      __syncthreads();
    }
    __syncthreads();                       // matches the producers' barrier in front of the hand-over
    __syncthreads();                       // (and the one behind it)
    if (fold_u) {
      fold_lanes();
      __syncthreads();
      // parameter-major rows: tile (R, Tc) holds parameters (R, Tc) of every camera pair of the group; the camera's own block U_c sits on
      // the tile's diagonal (row camera == column camera): at most one of a lane's four registers (k_schur_fused_bf3's epilogue)
      static_for<0, Cfg::NV>([&](auto vc) {
        constexpr int V = decltype(vc)::value;
        if (cw == V) {
          constexpr int T0 = schur_lo(Cfg::NTILE, Cfg::NV, V), T1 = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
          static_for<T0, T1>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int R = schur_tile_R(true, t), Tc = schur_tile_T(true, t);
            const int cj_ = lane & 15, rgm = cj_ - 4 * (lane >> 4);
            const T u = (rgm >= 0 && rgm < 4 && cj_ < nA) ? s_Ured[cj_ * UPKB + (R * NCP - (R * (R - 1)) / 2 + (Tc - R))] : (T)0;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) acc[t - T0][rg] -= (rg == rgm) ? u : (T)0;
          });
        }
      });
      const int ct = threadIdx.x - NPROD;
      if (ct < GROUP_ROWS) {
        const int c = ct / NCP, e = ct - c * NCP;
        const double gpart = (c < nA) ? (double)s_Ured[c * UPKB + NCP * (NCP + 1) / 2 + e] : 0.0;
        const double dpart = (c < nA) ? (double)s_Ured[c * UPKB + (e * NCP - (e * (e - 1)) / 2)] : 0.0;
        const double bsum = (c < nA) ? (double)s_Ured[c * UPKB + UPK + e] : 0.0;
        const size_t slot = (size_t)ga * ksplit + blockIdx.x;
        bpart[slot * GROUP_ROWS + ct] = bsum - gpart;                         // rhs = sum (b - g_c) over the workgroups
        gdpart[(slot * 2 + 0) * GROUP_ROWS + ct] = gpart;
        gdpart[(slot * 2 + 1) * GROUP_ROWS + ct] = dpart;
      }
    }
    T* slab = slabs + ((size_t)pair * ksplit + blockIdx.x) * (size_t)(GROUP_TILES * GROUP_TILES) * 256;
    schur_store_v<Cfg>(cw, slab, lane, acc);
  }
}

// ------------------------------------------------------------------ off-diagonal group pair on the bf16 pipe
// S(ga, gb) = - sum_p Ytilde_a Ytilde_b^T for two different camera groups: two panels, 121 tiles (row tiles = parameters of group a,
// column tiles = parameters of group b).  Both panels of a 16-point chunk do not fit the LDS twice, so a chunk is 8 points = ONE
// 32-deep k-step: 256 producer lanes = 8 points x (16 + 16) cameras, waves 0-1 build panel A, waves 2-3 panel B.
//   LDS: per buffer 2 panels x 3 bf16 planes x [11 tiles][2 halves][16 rows][4 slots] of 8-byte point slots (32-byte rows):
//   point q -> half q >> 2, slot (q & 3) ^ ((c >> 2) & 3).  A producer ds_write_b64 (16 lanes: one point, 16 cameras) covers all 32
//   banks once; a consumer lane (i, g) reads points g and g + 4 of row i with one ds_read2st64_b64 (the halves are 512 B apart), the
//   four lanes that share a bank column (i, i + 8; g, g ^ 1) hit four different slots.  2 x 67,584 B + camera tables.
//   Consumers: 30-31 tiles per wave (124 accumulator VGPRs); the wave's <= 4 row fragments of all planes stay in registers for the
//   chunk, the column tiles stream through a double-buffered 3-plane fragment set (loads of column Tc + 1 under the MFMAs of Tc).
struct SchurBf3OffCfg {
  using elem = float;
  static constexpr bool diag = false;
  static constexpr int THREADS = SCHUR_THREADS, NPROD = 256, NCW = 4, TS = 1, NV = 4;
  static constexpr int NTILE = GROUP_TILES * GROUP_TILES;
  static constexpr int TPW = (NTILE + NV - 1) / NV;
  static constexpr int PTS = 8, K = 32;
  static constexpr int HALF_BYTES = 16 * 32, TILE_BYTES = 2 * HALF_BYTES, PLANE_BYTES = GROUP_TILES * TILE_BYTES;
  static constexpr int PANEL_BYTES = 3 * PLANE_BYTES, BUF_BYTES = 2 * PANEL_BYTES;
  static constexpr size_t LDS_BYTES = 2 * (size_t)BUF_BYTES + 2 * (size_t)GROUP_CAMS * CAMPRE * sizeof(float);
  static_assert(LDS_BYTES <= 160 * 1024, "two double-buffered panels have to fit the 160 KB LDS (13 parameters: 163,200 B)");
};
__global__ __launch_bounds__(SCHUR_THREADS) void k_schur_offdiag_bf3(
    const ParamSets<float> ps, const LMState* __restrict__ st, int C,
    const float2* __restrict__ uv, const float* __restrict__ w, const uint16_t* __restrict__ gmask, const int32_t* __restrict__ gstart,
    int N, const float* __restrict__ pf, const int32_t* __restrict__ pair_ga, const int32_t* __restrict__ pair_gb, int pair0, int ksplit,
    float* __restrict__ slabs) {
  extern __shared__ __align__(16) unsigned char smem[];
  using T = float;
  using Cfg = SchurBf3OffCfg;
  constexpr int THREADS = Cfg::THREADS, NPROD = Cfg::NPROD, TPW = Cfg::TPW, PTS = Cfg::PTS;
  if (st->status >= 0) return;
  const int cur_ = ps_cur(ps, st);
  const T* __restrict__ ptsT = ps.ptsT[cur_];
  const int pair = pair0 + blockIdx.y;
  const int ga = pair_ga[pair], gb = pair_gb[pair];
  T* s_cam = reinterpret_cast<T*>(smem + 2 * Cfg::BUF_BYTES);            // [2][16][CAMPRE]
  {
    uint4* z4 = reinterpret_cast<uint4*>(smem);
    for (int i = threadIdx.x; i < 2 * Cfg::BUF_BYTES / 16; i += THREADS) z4[i] = make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int cam0 = (h ? gb : ga) * GROUP_CAMS, nc = min(GROUP_CAMS, C - cam0);
    for (int i = threadIdx.x; i < nc * CAMPRE; i += THREADS) s_cam[h * GROUP_CAMS * CAMPRE + i] = ps.campre[cur_][(size_t)cam0 * CAMPRE + i];
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool producer = threadIdx.x < NPROD;
  int per = (N + ksplit - 1) / ksplit;
  per = ((per + PTS - 1) / PTS) * PTS;
  const int pbeg = min(N, (int)blockIdx.x * per), pend = min(N, pbeg + per);
  const int nchunk = (pend - pbeg + PTS - 1) / PTS;
  __syncthreads();

  if (producer) {
    const int h = threadIdx.x >> 7, q = (threadIdx.x >> 4) & 7, c = threadIdx.x & 15;
    const int g = h ? gb : ga;
    const int nc = min(GROUP_CAMS, C - g * GROUP_CAMS);
    const bool cam_ok = c < nc;
    const T* cp_safe = s_cam + (h * GROUP_CAMS + (cam_ok ? c : 0)) * CAMPRE;
    const uint16_t* __restrict__ gm = gmask + (size_t)g * N;
    const int32_t* __restrict__ gs = gstart + (size_t)g * N;
    float2 n_uv = make_float2(0.f, 0.f);
    T n_w = 1, n_X[3] = {0, 0, 0}, n_f[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) n_f[k] = (T)0;
    bool n_valid = false;
    unsigned i_mask = 0; int i_start = 0; bool i_pt = false;
    auto request_index = [&](int chunk) {
      const int p = pbeg + chunk * PTS + q;
      i_pt = chunk < nchunk && p < pend;
      i_mask = 0; i_start = 0;
      if (i_pt) { i_mask = gm[p]; i_start = gs[p]; }
    };
    auto request = [&](int chunk) {
      const int p = pbeg + chunk * PTS + q;
      n_valid = i_pt && cam_ok && ((i_mask >> c) & 1u);
      if (i_pt) {
        n_X[0] = ptsT[3 * (size_t)p]; n_X[1] = ptsT[3 * (size_t)p + 1]; n_X[2] = ptsT[3 * (size_t)p + 2];
        const float4* f4 = reinterpret_cast<const float4*>(pf + (size_t)p * PF);
        const float4 a = f4[0], b = f4[1];
        n_f[0] = a.x; n_f[1] = a.y; n_f[2] = a.z; n_f[3] = a.w; n_f[4] = b.x; n_f[5] = b.y;      // L^-1 only: no right-hand side here
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) n_f[k] = (T)0;
      }
      if (n_valid) {
        const size_t o = (size_t)i_start + __builtin_popcount(i_mask & ((1u << c) - 1u));
        n_uv = uv[o];
        n_w = w ? w[o] : (T)1;
      }
      request_index(chunk + 1);
    };
    request_index(0);
    request(0);
    const int lane_slot = h * Cfg::PANEL_BYTES + (q >> 2) * Cfg::HALF_BYTES + c * 32 + (((q & 3) ^ ((c >> 2) & 3)) << 3);
    for (int it = 0; it <= nchunk; ++it) {
      if (it < nchunk) {
        unsigned char* pbuf = smem + (it & 1) * Cfg::BUF_BYTES;
        const bool valid = n_valid;
        const float2 m = n_uv;
        const T ww = n_w, X0 = n_X[0], X1 = n_X[1], X2 = n_X[2];
        T f[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) f[k] = n_f[k];
        request(it + 1);
        T r[2], Jc[2][NCP], Jp[2][3];
        obs_resjac<T>(cp_safe, X0, X1, X2, m.x, m.y, valid ? ww : (T)0, r, Jc, Jp, valid);
        (void)robust_apply<T>(ps.loss(), r, Jc, Jp);
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
            for (int d = 0; d < 3; ++d) y[d] = Jc[0][e] * Jt[0][d] + Jc[1][e] * Jt[1][d];
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
            unsigned char* dst = pbuf + lane_slot + e * Cfg::TILE_BYTES;
            *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h2);
            *reinterpret_cast<uint2*>(dst + Cfg::PLANE_BYTES) = make_uint2(m01, m2);
            *reinterpret_cast<uint2*>(dst + 2 * Cfg::PLANE_BYTES) = make_uint2(l01, l2);
          });
        }
      }
      __syncthreads();
    }
  } else {
    const int cw = wid - NPROD / 64;
    typename Mfma<T>::acc_t acc[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) acc[s] = typename Mfma<T>::acc_t{0, 0, 0, 0};
    // fragment of a tile: points g and g + 4 of row i = lane & 15, slot g ^ ((i >> 2) & 3) of the two halves
    const int frag_off = (lane & 15) * 32 + (((lane >> 4) ^ (((lane & 15) >> 2) & 3)) << 3);
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    for (int it = 0; it <= nchunk; ++it) {
      if (it >= 1) {
        const unsigned char* pbuf = smem + ((it - 1) & 1) * Cfg::BUF_BYTES + frag_off;
        static_for<0, Cfg::NV>([&](auto vc) {
          constexpr int V = decltype(vc)::value;
          if (cw == V) {
            constexpr int LO = schur_lo(Cfg::NTILE, Cfg::NV, V), HI = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
            constexpr int RMIN = schur_tile_R(false, LO), RMAX = schur_tile_R(false, HI - 1);
            constexpr int NR = RMAX - RMIN + 1;
            auto frag = [&](int panel, int plane, int b) -> bf16x8_t {
              const unsigned char* rowp = pbuf + panel * Cfg::PANEL_BYTES + plane * Cfg::PLANE_BYTES + b * Cfg::TILE_BYTES;
              const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(rowp);
              const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(rowp + Cfg::HALF_BYTES);
              return __builtin_bit_cast(bf16x8_t, u32x4_t{lo[0], lo[1], hi[0], hi[1]});
            };
            if constexpr (NCP <= 11) {
              // registers: 124 accumulators + the wave's <= 4 row fragments of all three planes (stationary for the chunk) + the
              // three planes of ONE column tile, double-buffered: column Tc + 1 is in flight while the <= 24 MFMAs of column Tc issue
              bf16x8_t fa[3][NR], fb[2][3];
  #pragma unroll
              for (int pl = 0; pl < 3; ++pl)
  #pragma unroll
                for (int rr = 0; rr < NR; ++rr) fa[pl][rr] = frag(0, pl, RMIN + rr);
  #pragma unroll
              for (int pl = 0; pl < 3; ++pl) fb[0][pl] = frag(1, pl, 0);
              static_for<0, GROUP_TILES>([&](auto cc) {
                constexpr int Tc = decltype(cc)::value;
                constexpr int cur = Tc & 1;
                if constexpr (Tc + 1 < GROUP_TILES) {
  #pragma unroll
                  for (int pl = 0; pl < 3; ++pl) fb[cur ^ 1][pl] = frag(1, pl, Tc + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                // the six partial products, row tiles interleaved so that consecutive MFMAs never share an accumulator
                static_for<0, 6>([&](auto pc) {
                  constexpr int pr = decltype(pc)::value;
                  constexpr int pa = (pr == 0 || pr == 3 || pr == 5) ? 0 : (pr == 1 || pr == 4) ? 1 : 2;     // h h' | m h' | l h' | h m' | m m' | h l'
                  constexpr int pb = pr < 3 ? 0 : pr < 5 ? 1 : 2;
                  static_for<0, NR>([&](auto rc) {
                    constexpr int R = RMIN + decltype(rc)::value;
                    constexpr int t = R * GROUP_TILES + Tc;
                    if constexpr (t >= LO && t < HI)
                      acc[t - LO] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[pa][R - RMIN], fb[cur][pb], acc[t - LO], 0, 0, 0);
                  });
                });
                __builtin_amdgcn_sched_barrier(0);
              });
            } else {
              // 13 parameters: 169 tiles, 42-43 per wave = 172 accumulator VGPRs.  The wave's <= 5 row fragments of ONE plane are
              // stationary while the column tiles stream past (double-buffered, only the planes that pair with it):
              // h x (h', m', l'), then m x (h', m'), then l x h' -- the columns are read three times, the consumers have the slack
              static_for<0, 3>([&](auto pac) {
                constexpr int pa = decltype(pac)::value;
                constexpr int NB = 3 - pa;
                bf16x8_t fa[NR], fb[2][NB];
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) fa[rr] = frag(0, pa, RMIN + rr);
#pragma unroll
                for (int pl = 0; pl < NB; ++pl) fb[0][pl] = frag(1, pl, 0);
                static_for<0, GROUP_TILES>([&](auto cc) {
                  constexpr int Tc = decltype(cc)::value;
                  constexpr int cur = Tc & 1;
                  if constexpr (Tc + 1 < GROUP_TILES) {
#pragma unroll
                    for (int pl = 0; pl < NB; ++pl) fb[cur ^ 1][pl] = frag(1, pl, Tc + 1);
                  }
                  __builtin_amdgcn_sched_barrier(0);
                  static_for<0, NB>([&](auto pbc) {
                    constexpr int pb = decltype(pbc)::value;
                    static_for<0, NR>([&](auto rc) {
                      constexpr int R = RMIN + decltype(rc)::value;
                      constexpr int t = R * GROUP_TILES + Tc;
                      if constexpr (t >= LO && t < HI)
                        acc[t - LO] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[R - RMIN], fb[cur][pb], acc[t - LO], 0, 0, 0);
                    });
                  });
                  __builtin_amdgcn_sched_barrier(0);
                });
              });
            }
          }
        });
      }
      __syncthreads();
    }
    float4* slab4 = reinterpret_cast<float4*>(slabs + ((size_t)pair * ksplit + blockIdx.x) * (size_t)(GROUP_TILES * GROUP_TILES) * 256) + lane;
    static_for<0, Cfg::NV>([&](auto vc) {
      constexpr int V = decltype(vc)::value;
      if (cw == V) {
        constexpr int LO = schur_lo(Cfg::NTILE, Cfg::NV, V), HI = schur_lo(Cfg::NTILE, Cfg::NV, V + 1);
        // a running pointer the compiler cannot fold back into 30 separate 64-bit address constants (those took 60 VGPRs and
        // spilled around this epilogue)
        float4* pt = slab4 + LO * 64;
        static_for<LO, HI>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          *pt = make_float4(acc[t - LO][0], acc[t - LO][1], acc[t - LO][2], acc[t - LO][3]);     // [tile][lane][reg]
          pt += 64;
          asm volatile("" : "+v"(pt));
        });
      }
    });
  }
}

}  // namespace SBA_NS
