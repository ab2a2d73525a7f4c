// sba_model.hpp -- per-observation camera model and analytic Jacobian for gfx950 device code.
//
// Model being replaced: PySBA.rotate / PySBA.project / PySBA.fun
// (/root/reference/lasercalib/pySBA.py:61-101): Rodrigues rotate -> + t -> pinhole divide ->
// two-term radial distortion with ONE focal length -> + principal point; residual = w*(proj - uv).
// The reference differentiates this numerically (scipy 3-point, 28 evaluations per Jacobian);
// here the 2x11 camera block and 2x3 point block are closed-form.
//
// Per-camera quantities that do not depend on the observation (sin/cos, the Rodrigues series
// coefficients and the rotation matrix) are computed once per parameter update into a "CamPre"
// table that every workgroup stages in LDS, so the per-observation path has no transcendental.
#pragma once
#include <hip/hip_runtime.h>

// Camera model selection: the whole device code base is compiled once per model into its own namespace.
#ifndef SBA_NCP
#define SBA_NCP 11
#endif
#if SBA_NCP == 11
#define SBA_NS sba11
#elif SBA_NCP == 13
#define SBA_NS sba13
#else
#error "SBA_NCP must be 11 (radial) or 13 (radial + tangential)"
#endif

namespace SBA_NS {

constexpr int NCP = SBA_NCP;     // camera parameters per camera: 11 (pySBA.py:31-35) or 13 (+ tangential p1, p2 before cx, cy)
constexpr bool TANGENTIAL = (NCP == 13);
constexpr int CAMPRE = NCP + 14; // CamPre row length (odd stride => LDS reads of different cameras spread over banks)
// CamPre row layout
constexpr int CP_RHO = 0;        // rho[3]
constexpr int CP_T = 3;          // t[3]
constexpr int CP_F = 6, CP_K1 = 7, CP_K2 = 8;
constexpr int CP_P1 = 9, CP_P2 = 10;               // tangential model only
constexpr int CP_CX = NCP - 2, CP_CY = NCP - 1;
constexpr int CP_R = NCP;        // R[9] row-major
constexpr int CP_A = NCP + 9, CP_B = NCP + 10, CP_A2 = NCP + 11, CP_B2 = NCP + 12;   // sin t/t, (1-cos t)/t^2, (cos t - a)/t^2, (a-2b)/t^2
constexpr int CP_PAD = NCP + 13;

// Build one CamPre row from the NCP raw parameters (always evaluated in double, then narrowed).
template <typename T>
__device__ inline void campre_build(const double* __restrict__ cam, T* __restrict__ out) {
  const double r0 = cam[0], r1 = cam[1], r2 = cam[2];
  const double th2 = r0 * r0 + r1 * r1 + r2 * r2;
  double c, a, b, a2, b2;
  if (th2 < 1e-4) {   // series: theta = 0 must behave as the identity (pySBA.py:66-68)
    c = 1.0 - th2 * (0.5 - th2 * (1.0 / 24 - th2 * (1.0 / 720)));
    a = 1.0 - th2 * (1.0 / 6 - th2 * (1.0 / 120 - th2 * (1.0 / 5040)));
    b = 0.5 - th2 * (1.0 / 24 - th2 * (1.0 / 720 - th2 * (1.0 / 40320)));
    a2 = -1.0 / 3 + th2 * (1.0 / 30 - th2 * (1.0 / 840 - th2 * (1.0 / 45360)));
    b2 = -1.0 / 12 + th2 * (1.0 / 180 - th2 * (1.0 / 6720 - th2 * (1.0 / 453600)));
  } else {
    const double th = sqrt(th2);
    double s;
    sincos(th, &s, &c);
    a = s / th;
    b = (1.0 - c) / th2;
    a2 = (c - a) / th2;
    b2 = (a - 2.0 * b) / th2;
  }
#pragma unroll
  for (int i = 0; i < NCP; ++i) out[i] = (T)cam[i];
  // R = c I + a [rho]x + b rho rho^T
  out[CP_R + 0] = (T)(c + b * r0 * r0);
  out[CP_R + 1] = (T)(-a * r2 + b * r0 * r1);
  out[CP_R + 2] = (T)(a * r1 + b * r0 * r2);
  out[CP_R + 3] = (T)(a * r2 + b * r1 * r0);
  out[CP_R + 4] = (T)(c + b * r1 * r1);
  out[CP_R + 5] = (T)(-a * r0 + b * r1 * r2);
  out[CP_R + 6] = (T)(-a * r1 + b * r2 * r0);
  out[CP_R + 7] = (T)(a * r0 + b * r2 * r1);
  out[CP_R + 8] = (T)(c + b * r2 * r2);
  out[CP_A] = (T)a;
  out[CP_B] = (T)b;
  out[CP_A2] = (T)a2;
  out[CP_B2] = (T)b2;
  out[CP_PAD] = (T)0;
}

// Forward projection only.  cp points at a CamPre row (LDS or global).
template <typename T>
__device__ inline void obs_project(const T* __restrict__ cp, T X0, T X1, T X2, T& u, T& v) {
  const T p0 = cp[CP_R + 0] * X0 + cp[CP_R + 1] * X1 + cp[CP_R + 2] * X2 + cp[CP_T + 0];
  const T p1 = cp[CP_R + 3] * X0 + cp[CP_R + 4] * X1 + cp[CP_R + 5] * X2 + cp[CP_T + 1];
  const T p2 = cp[CP_R + 6] * X0 + cp[CP_R + 7] * X1 + cp[CP_R + 8] * X2 + cp[CP_T + 2];
  const T iz = (T)1 / p2;
  const T x = p0 * iz, y = p1 * iz;
  const T n = x * x + y * y;
  const T d = (T)1 + n * (cp[CP_K1] + cp[CP_K2] * n);
  if constexpr (TANGENTIAL) {      // OpenCV convention: x' = x d + 2 p1 x y + p2 (n + 2 x^2), y' = y d + p1 (n + 2 y^2) + 2 p2 x y
    const T tp1 = cp[CP_P1], tp2 = cp[CP_P2], xy2 = (T)2 * x * y;
    const T xd = d * x + tp1 * xy2 + tp2 * (n + (T)2 * x * x);
    const T yd = d * y + tp1 * (n + (T)2 * y * y) + tp2 * xy2;
    u = cp[CP_F] * xd + cp[CP_CX];
    v = cp[CP_F] * yd + cp[CP_CY];
  } else {
    const T fd = cp[CP_F] * d;
    u = fd * x + cp[CP_CX];
    v = fd * y + cp[CP_CY];
  }
}

// Residual + Jacobian blocks of one observation.
//   r[2]      = w * (project - uv)
//   Jc[2][NCP] = d r / d(cam params),  Jp[2][3] = d r / d X
template <typename T>
__device__ inline void obs_resjac(const T* __restrict__ cp, T X0, T X1, T X2, T uo, T vo, T w,
                                  T r[2], T Jc[2][NCP], T Jp[2][3], bool valid = true /* false: the caller passes w = 0 and only
                                  needs finite outputs, so the depth is replaced by 1 (no 1/0 -> inf * 0 = NaN) */) {
  const T R0 = cp[CP_R + 0], R1 = cp[CP_R + 1], R2 = cp[CP_R + 2];
  const T R3 = cp[CP_R + 3], R4 = cp[CP_R + 4], R5 = cp[CP_R + 5];
  const T R6 = cp[CP_R + 6], R7 = cp[CP_R + 7], R8 = cp[CP_R + 8];
  const T p0 = R0 * X0 + R1 * X1 + R2 * X2 + cp[CP_T + 0];
  const T p1 = R3 * X0 + R4 * X1 + R5 * X2 + cp[CP_T + 1];
  const T p2 = R6 * X0 + R7 * X1 + R8 * X2 + cp[CP_T + 2];
  const T iz = (T)1 / (valid ? p2 : (T)1);
  const T x = p0 * iz, y = p1 * iz;
  const T n = x * x + y * y;
  const T k1 = cp[CP_K1], k2 = cp[CP_K2], f = cp[CP_F];
  const T d = (T)1 + n * (k1 + k2 * n);
  const T dn = k1 + (T)2 * k2 * n;
  // A = w * d(u,v)/dp
  const T wf = w * f;
  T gxx = d + (T)2 * x * x * dn, gxy = (T)2 * x * y * dn, gyy = d + (T)2 * y * y * dn;     // d(xd,yd)/d(x,y), symmetric
  T xd = x, yd = y;                         // tangential model: the distorted normalised coordinates
  if constexpr (TANGENTIAL) {
    const T tp1 = cp[CP_P1], tp2 = cp[CP_P2], xy2 = (T)2 * x * y;
    xd = d * x + tp1 * xy2 + tp2 * (n + (T)2 * x * x);
    yd = d * y + tp1 * (n + (T)2 * y * y) + tp2 * xy2;
    gxx += (T)2 * tp1 * y + (T)6 * tp2 * x;
    gxy += (T)2 * (tp1 * x + tp2 * y);
    gyy += (T)6 * tp1 * y + (T)2 * tp2 * x;
    r[0] = w * (f * xd + cp[CP_CX] - uo);
    r[1] = w * (f * yd + cp[CP_CY] - vo);
  } else {                                  // the reference's model, operation order unchanged since round 1
    r[0] = w * (f * d * x + cp[CP_CX] - uo);
    r[1] = w * (f * d * y + cp[CP_CY] - vo);
  }
  const T ux = wf * gxx;
  const T uy = wf * gxy;
  const T vy = wf * gyy;
  const T A00 = ux * iz, A01 = uy * iz, A02 = -(ux * x + uy * y) * iz;
  const T A10 = uy * iz, A11 = vy * iz, A12 = -(uy * x + vy * y) * iz;

  // point block: A R
  Jp[0][0] = A00 * R0 + A01 * R3 + A02 * R6;
  Jp[0][1] = A00 * R1 + A01 * R4 + A02 * R7;
  Jp[0][2] = A00 * R2 + A01 * R5 + A02 * R8;
  Jp[1][0] = A10 * R0 + A11 * R3 + A12 * R6;
  Jp[1][1] = A10 * R1 + A11 * R4 + A12 * R7;
  Jp[1][2] = A10 * R2 + A11 * R5 + A12 * R8;
#ifdef SBA_F64_SCHED
  if constexpr (sizeof(T) == 8) __builtin_amdgcn_sched_barrier(0);
#endif

  // rotation block: A * dP/drho,  dP/drho = q rho^T - a [X]x + b rho X^T + b (rho.X) I
  const T h0 = cp[CP_RHO + 0], h1 = cp[CP_RHO + 1], h2 = cp[CP_RHO + 2];
  const T a = cp[CP_A], b = cp[CP_B], a2 = cp[CP_A2], b2 = cp[CP_B2];
  const T c0 = h1 * X2 - h2 * X1, c1 = h2 * X0 - h0 * X2, c2 = h0 * X1 - h1 * X0;   // rho x X
  const T hd = h0 * X0 + h1 * X1 + h2 * X2;
  const T q0 = -a * X0 + a2 * c0 + b2 * hd * h0;
  const T q1 = -a * X1 + a2 * c1 + b2 * hd * h1;
  const T q2 = -a * X2 + a2 * c2 + b2 * hd * h2;
  const T bhd = b * hd;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const T a0 = rr ? A10 : A00, a1 = rr ? A11 : A01, a2r = rr ? A12 : A02;
    const T Aq = a0 * q0 + a1 * q1 + a2r * q2;
    const T Ah = b * (a0 * h0 + a1 * h1 + a2r * h2);
    // row * [X]x  = (row x X) transposed sign: (row^T [X]x)_j = (X x row)_j ... use explicit form
    // [X]x = [[0,-X2,X1],[X2,0,-X0],[-X1,X0,0]]  => row*[X]x = (a1*X2 - a2r*X1, a2r*X0 - a0*X2, a0*X1 - a1*X0)
    const T x0 = a1 * X2 - a2r * X1, x1 = a2r * X0 - a0 * X2, x2 = a0 * X1 - a1 * X0;
    Jc[rr][0] = Aq * h0 - a * x0 + Ah * X0 + bhd * a0;
    Jc[rr][1] = Aq * h1 - a * x1 + Ah * X1 + bhd * a1;
    Jc[rr][2] = Aq * h2 - a * x2 + Ah * X2 + bhd * a2r;
    Jc[rr][3] = a0;
    Jc[rr][4] = a1;
    Jc[rr][5] = a2r;
  }
  const T wfn = wf * n;
  if constexpr (TANGENTIAL) { Jc[0][6] = w * xd; Jc[1][6] = w * yd; }
  else { const T wd = w * d; Jc[0][6] = wd * x; Jc[1][6] = wd * y; }
  Jc[0][7] = wfn * x;       Jc[1][7] = wfn * y;
  Jc[0][8] = wfn * n * x;   Jc[1][8] = wfn * n * y;
  if constexpr (TANGENTIAL) {
    const T xy2 = (T)2 * x * y;
    Jc[0][CP_P1] = wf * xy2;                     Jc[1][CP_P1] = wf * (n + (T)2 * y * y);
    Jc[0][CP_P2] = wf * (n + (T)2 * x * x);      Jc[1][CP_P2] = wf * xy2;
  }
  Jc[0][CP_CX] = w;         Jc[1][CP_CX] = (T)0;
  Jc[0][CP_CY] = (T)0;      Jc[1][CP_CY] = w;
}

// Point block + DIRECTIONAL derivative along a camera step: what the back substitution needs of an observation is
//   Jp (2x3)  and  s = Jc dc (2),  not the 2 x NCP block Jc itself.  With A = w d(u,v)/dP (2x3, as in obs_resjac)
//   s = A ( (dP/drho) drho + dt ) + (d r / d f) df + (d r / d k1) dk1 + (d r / d k2) dk2 [+ p1, p2] + w (dcx, dcy),
//   (dP/drho) drho = q (rho . drho) - a (X x drho) + b rho (X . drho) + b (rho . X) drho
// (the row form of obs_resjac's rotation block, a_row . v, since (a_row x X) . drho = a_row . (X x drho)): one 3-vector instead of a
// 2x3 block per row -- about 60 instructions less than building Jc and contracting it.  Same formulas, another association:
// agrees with obs_resjac's Jc dc to rounding.
template <typename T>
__device__ inline void obs_jp_jvp(const T* __restrict__ cp, T X0, T X1, T X2, T uo, T vo, T w, const T (&dc)[NCP],
                                  T r[2], T Jp[2][3], T s[2]) {
  const T R0 = cp[CP_R + 0], R1 = cp[CP_R + 1], R2 = cp[CP_R + 2];
  const T R3 = cp[CP_R + 3], R4 = cp[CP_R + 4], R5 = cp[CP_R + 5];
  const T R6 = cp[CP_R + 6], R7 = cp[CP_R + 7], R8 = cp[CP_R + 8];
  const T p0 = R0 * X0 + R1 * X1 + R2 * X2 + cp[CP_T + 0];
  const T p1 = R3 * X0 + R4 * X1 + R5 * X2 + cp[CP_T + 1];
  const T p2 = R6 * X0 + R7 * X1 + R8 * X2 + cp[CP_T + 2];
  const T iz = (T)1 / p2;
  const T x = p0 * iz, y = p1 * iz;
  const T n = x * x + y * y;
  const T k1 = cp[CP_K1], k2 = cp[CP_K2], f = cp[CP_F];
  const T d = (T)1 + n * (k1 + k2 * n);
  const T dn = k1 + (T)2 * k2 * n;
  const T wf = w * f;
  T gxx = d + (T)2 * x * x * dn, gxy = (T)2 * x * y * dn, gyy = d + (T)2 * y * y * dn;
  T xd = x, yd = y;
  if constexpr (TANGENTIAL) {
    const T tp1 = cp[CP_P1], tp2 = cp[CP_P2], xy2 = (T)2 * x * y;
    xd = d * x + tp1 * xy2 + tp2 * (n + (T)2 * x * x);
    yd = d * y + tp1 * (n + (T)2 * y * y) + tp2 * xy2;
    gxx += (T)2 * tp1 * y + (T)6 * tp2 * x;
    gxy += (T)2 * (tp1 * x + tp2 * y);
    gyy += (T)6 * tp1 * y + (T)2 * tp2 * x;
    r[0] = w * (f * xd + cp[CP_CX] - uo);
    r[1] = w * (f * yd + cp[CP_CY] - vo);
  } else {
    r[0] = w * (f * d * x + cp[CP_CX] - uo);
    r[1] = w * (f * d * y + cp[CP_CY] - vo);
  }
  const T ux = wf * gxx, uy = wf * gxy, vy = wf * gyy;
  const T A00 = ux * iz, A01 = uy * iz, A02 = -(ux * x + uy * y) * iz;
  const T A10 = uy * iz, A11 = vy * iz, A12 = -(uy * x + vy * y) * iz;
  Jp[0][0] = A00 * R0 + A01 * R3 + A02 * R6;
  Jp[0][1] = A00 * R1 + A01 * R4 + A02 * R7;
  Jp[0][2] = A00 * R2 + A01 * R5 + A02 * R8;
  Jp[1][0] = A10 * R0 + A11 * R3 + A12 * R6;
  Jp[1][1] = A10 * R1 + A11 * R4 + A12 * R7;
  Jp[1][2] = A10 * R2 + A11 * R5 + A12 * R8;
  // v = (dP/drho) drho + dt
  const T h0 = cp[CP_RHO + 0], h1 = cp[CP_RHO + 1], h2 = cp[CP_RHO + 2];
  const T a = cp[CP_A], b = cp[CP_B], a2 = cp[CP_A2], b2 = cp[CP_B2];
  const T c0 = h1 * X2 - h2 * X1, c1 = h2 * X0 - h0 * X2, c2 = h0 * X1 - h1 * X0;   // rho x X
  const T hd = h0 * X0 + h1 * X1 + h2 * X2;
  const T q0 = -a * X0 + a2 * c0 + b2 * hd * h0;
  const T q1 = -a * X1 + a2 * c1 + b2 * hd * h1;
  const T q2 = -a * X2 + a2 * c2 + b2 * hd * h2;
  const T e0 = dc[0], e1 = dc[1], e2 = dc[2];
  const T hde = h0 * e0 + h1 * e1 + h2 * e2;            // rho . drho
  const T xde = X0 * e0 + X1 * e1 + X2 * e2;            // X . drho
  const T bx = b * xde, bh = b * hd;
  const T v0 = q0 * hde - a * (X1 * e2 - X2 * e1) + bx * h0 + bh * e0 + dc[3];
  const T v1 = q1 * hde - a * (X2 * e0 - X0 * e2) + bx * h1 + bh * e1 + dc[4];
  const T v2 = q2 * hde - a * (X0 * e1 - X1 * e0) + bx * h2 + bh * e2 + dc[5];
  // focal length, k1, k2 (one scalar per component), principal point
  const T wfn = wf * n;
  T g;
  if constexpr (TANGENTIAL) {
    g = wfn * (dc[7] + n * dc[8]);
    const T xy2 = (T)2 * x * y;
    s[0] = A00 * v0 + A01 * v1 + A02 * v2 + w * xd * dc[6] + g * x + wf * (xy2 * dc[CP_P1] + (n + (T)2 * x * x) * dc[CP_P2]) + w * dc[CP_CX];
    s[1] = A10 * v0 + A11 * v1 + A12 * v2 + w * yd * dc[6] + g * y + wf * ((n + (T)2 * y * y) * dc[CP_P1] + xy2 * dc[CP_P2]) + w * dc[CP_CY];
  } else {
    g = w * d * dc[6] + wfn * (dc[7] + n * dc[8]);
    s[0] = A00 * v0 + A01 * v1 + A02 * v2 + g * x + w * dc[CP_CX];
    s[1] = A10 * v0 + A11 * v1 + A12 * v2 + g * y + w * dc[CP_CY];
  }
}

// ------------------------------------------------------------------ robust loss (opt-in; the objective of scipy least_squares(loss=..., f_scale=delta))
// The loss applies to every residual component f: z = (f/delta)^2, cost = 0.5 delta^2 sum rho(z) with scipy's rho
// (scipy/optimize/_lsq/least_squares.py:189-226):
//   huber    rho = z (z <= 1), 2 sqrt(z) - 1 beyond          rho' = 1, z^-1/2
//   soft_l1  rho = 2 (sqrt(1 + z) - 1)                       rho' = (1 + z)^-1/2
//   cauchy   rho = ln(1 + z)                                 rho' = 1 / (1 + z)
// Its gradient is J^T (rho' f).  For the Gauss-Newton model the rows of J and f are both scaled by
// sqrt(rho') -- iteratively re-weighted least squares: J_s^T f_s = J^T rho' f is the exact gradient and J_s^T J_s = J^T rho' J
// stays positive -- applied right after the residual/Jacobian blocks, so that every kernel downstream (normal-equation
// blocks, Schur complement, step) is robust without knowing about it; only the cost sums take rho instead of f^2.
// scipy scales rows by sqrt(max(rho' + 2 rho'' z, EPS)) instead (common.py, scale_for_robust_loss_function), which for Huber
// is sqrt(EPS) for EVERY row beyond f_scale (and for Cauchy for every row with z > 1): the model loses all curvature there and
// only the trust region bounds the step (12 719 evaluations on the 6 x 300 test rig); with multiplicative Levenberg-Marquardt
// damping that is unusable (a point whose observations all start beyond f_scale has a zero 3x3 block).  Same objective, same
// stationary points, different quadratic model -- the Ceres choice when rho' + 2 rho'' z <= 0.
// delta <= 0: linear loss, nothing happens (one uniform branch).
__device__ inline double log1p_t(double x) { return log1p(x); }
__device__ inline float log1p_t(float x) { return log1pf(x); }
constexpr int LOSS_HUBER = 1, LOSS_SOFT_L1 = 2, LOSS_CAUCHY = 3;          // = sba_loss (include/sba_hip.h)
template <typename T> struct RLoss { T delta = (T)0; int kind = 0; };
template <typename T>
__device__ __forceinline__ T robust_component(RLoss<T> L, T& f, T& jscale) {       // returns delta^2 rho(z) (= f^2 when nothing is scaled)
  const T delta = L.delta;
  const T z = (f / delta) * (f / delta);
  jscale = (T)1;
  if (L.kind == LOSS_SOFT_L1) {                        // (uniform branch: the kind is a kernel argument)
    const T s1 = sqrt((T)1 + z);
    jscale = sqrt((T)1 / s1);
    f = f * jscale;
    return delta * delta * (T)2 * (z / (s1 + (T)1));   // 2 (sqrt(1 + z) - 1) without the cancellation at small z
  }
  if (L.kind == LOSS_CAUCHY) {
    const T cost = delta * delta * log1p_t(z);
    jscale = sqrt((T)1 / ((T)1 + z));
    f = f * jscale;
    return cost;
  }
  if (z <= (T)1) return f * f;
  const T sz = sqrt(z);                                  // |f| / delta;  rho'(z) = 1 / sz
  jscale = sqrt((T)1 / sz);
  const T cost = delta * delta * ((T)2 * sz - (T)1);
  f = f * jscale;
  return cost;
}
// residual only (trial points, sba_residual): delta^2 (rho(z0) + rho(z1)), or r0^2 + r1^2 for the linear loss
template <typename T>
__device__ __forceinline__ T robust_cost(RLoss<T> L, T r0, T r1) {
  if (!(L.delta > (T)0)) return r0 * r0 + r1 * r1;
  T js;
  return robust_component<T>(L, r0, js) + robust_component<T>(L, r1, js);
}
// residual + Jacobian blocks: scale in place, return the cost term
template <typename T>
__device__ __forceinline__ T robust_apply(RLoss<T> L, T r[2], T Jc[2][NCP], T Jp[2][3]) {
  if (!(L.delta > (T)0)) return r[0] * r[0] + r[1] * r[1];
  T cost = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    T js;
    cost += robust_component<T>(L, r[k], js);
    if (js != (T)1) {
#pragma unroll
      for (int e = 0; e < NCP; ++e) Jc[k][e] *= js;
#pragma unroll
      for (int d = 0; d < 3; ++d) Jp[k][d] *= js;
    }
  }
  return cost;
}

// the same for (r, Jp, s = Jc dc): the rows of Jc are scaled like the residual, so s is
template <typename T>
__device__ __forceinline__ void robust_apply_jvp(RLoss<T> L, T r[2], T Jp[2][3], T s[2]) {
  if (!(L.delta > (T)0)) return;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    T js;
    (void)robust_component<T>(L, r[k], js);
    if (js != (T)1) {
      s[k] *= js;
#pragma unroll
      for (int d = 0; d < 3; ++d) Jp[k][d] *= js;
    }
  }
}

__device__ inline void sincos_t(double x, double* s, double* c) { sincos(x, s, c); }
__device__ inline void sincos_t(float x, float* s, float* c) { sincosf(x, s, c); }

// Plain Rodrigues rotation from a raw rotation vector (PySBA.rotate, pySBA.py:61-73), used by the
// stateless sba_rotate / sba_project entry points where every row carries its own camera.
template <typename T>
__device__ inline void rotate_raw(T h0, T h1, T h2, T X0, T X1, T X2, T& P0, T& P1, T& P2) {
  const T th2 = h0 * h0 + h1 * h1 + h2 * h2;
  T c, a, b;
  if (th2 < (T)1e-4) {
    c = (T)1 - th2 * ((T)0.5 - th2 * ((T)(1.0 / 24) - th2 * (T)(1.0 / 720)));
    a = (T)1 - th2 * ((T)(1.0 / 6) - th2 * ((T)(1.0 / 120) - th2 * (T)(1.0 / 5040)));
    b = (T)0.5 - th2 * ((T)(1.0 / 24) - th2 * ((T)(1.0 / 720) - th2 * (T)(1.0 / 40320)));
  } else {
    const T th = sqrt(th2);
    T s;
    sincos_t(th, &s, &c);
    a = s / th;
    b = ((T)1 - c) / th2;
  }
  const T hd = b * (h0 * X0 + h1 * X1 + h2 * X2);
  P0 = c * X0 + a * (h1 * X2 - h2 * X1) + hd * h0;
  P1 = c * X1 + a * (h2 * X0 - h0 * X2) + hd * h1;
  P2 = c * X2 + a * (h0 * X1 - h1 * X0) + hd * h2;
}

}  // namespace SBA_NS
