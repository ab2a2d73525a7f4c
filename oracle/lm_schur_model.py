"""CPU MODEL of the device algorithm -- test infrastructure, NOT product code.

The reference solves bundle adjustment with scipy's TRF + finite differences
(pySBA.py:141).  The product replaces that with analytic Jacobian blocks and a
Levenberg-Marquardt / Schur-complement loop on the GPU.  This file states that
replacement algorithm in plain numpy so the tests can check the HIP kernels'
intermediate quantities (Jacobian blocks, U/V/W, reduced camera system, step)
one by one, and so the multi-rank driver can be exercised on CPU with gloo.

It is validated two ways in tests/: (1) its analytic Jacobian against the
reference's scipy 3-point finite-difference Jacobian fixture (f3_jacobian.npz),
(2) its converged cost/RMS against the reference's converged solves
(f4_solves.npz).  Nothing under lasercalib_amd/ imports it.
"""
from __future__ import annotations

import numpy as np

NCP = 11


# ----------------------------------------------------------------------------- per-observation math
def rodrigues_coeffs(theta2):
    """a=sin t/t, b=(1-cos t)/t^2, a2=(cos t - a)/t^2, b2=(a-2b)/t^2 with series near 0."""
    theta2 = np.asarray(theta2, dtype=np.float64)
    small = theta2 < 1e-4
    t2s = np.where(small, theta2, 1.0)
    t2 = np.where(small, 1.0, theta2)
    th = np.sqrt(t2)
    s, c = np.sin(th), np.cos(th)
    a = s / th
    b = (1 - c) / t2
    a2 = (c - a) / t2
    b2 = (a - 2 * b) / t2
    a_s = 1 - t2s / 6 + t2s ** 2 / 120 - t2s ** 3 / 5040
    b_s = 0.5 - t2s / 24 + t2s ** 2 / 720 - t2s ** 3 / 40320
    a2_s = -1.0 / 3 + t2s / 30 - t2s ** 2 / 840 + t2s ** 3 / 45360
    b2_s = -1.0 / 12 + t2s / 180 - t2s ** 2 / 6720 + t2s ** 3 / 453600
    c_s = 1 - t2s / 2 + t2s ** 2 / 24 - t2s ** 3 / 720
    return (np.where(small, c_s, c), np.where(small, a_s, a), np.where(small, b_s, b),
            np.where(small, a2_s, a2), np.where(small, b2_s, b2))


def residual_jacobian(cams, pts, uv, ci, pi, w):
    """Analytic residual (M,2) and Jacobian blocks Jc (M,2,11), Jp (M,2,3) of w*(project-uv).

    Model: pySBA.py:61-89 (Rodrigues, pinhole, two radial terms, one focal length).
    """
    cam = cams[ci]
    X = pts[pi]
    rho = cam[:, 0:3]
    t = cam[:, 3:6]
    f, k1, k2 = cam[:, 6], cam[:, 7], cam[:, 8]
    th2 = np.sum(rho * rho, axis=1)
    c, a, b, a2, b2 = rodrigues_coeffs(th2)
    rxX = np.cross(rho, X)
    rdX = np.sum(rho * X, axis=1)
    P = c[:, None] * X + a[:, None] * rxX + (b * rdX)[:, None] * rho
    p = P + t
    iz = 1.0 / p[:, 2]
    x, y = p[:, 0] * iz, p[:, 1] * iz
    n = x * x + y * y
    d = 1 + k1 * n + k2 * n * n
    dn = k1 + 2 * k2 * n
    u = f * d * x + cam[:, 9]
    v = f * d * y + cam[:, 10]
    M = ci.shape[0]
    wv = np.asarray(w, dtype=np.float64).reshape(-1)
    if wv.size == 1:
        wv = np.full(M, wv[0])
    res = np.stack([u, v], axis=1) - uv
    res = res * wv[:, None]

    # A = d(u,v)/dp  (2x3)
    ux = f * (d + 2 * x * x * dn)
    uy = f * (2 * x * y * dn)
    vy = f * (d + 2 * y * y * dn)
    A = np.zeros((M, 2, 3))
    A[:, 0, 0] = ux * iz
    A[:, 0, 1] = uy * iz
    A[:, 0, 2] = -(ux * x + uy * y) * iz
    A[:, 1, 0] = uy * iz
    A[:, 1, 1] = vy * iz
    A[:, 1, 2] = -(uy * x + vy * y) * iz

    # R = c I + a [rho]x + b rho rho^T
    R = np.zeros((M, 3, 3))
    eye = np.eye(3)
    R += c[:, None, None] * eye
    R += b[:, None, None] * (rho[:, :, None] * rho[:, None, :])
    K = np.zeros((M, 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -rho[:, 2], rho[:, 1]
    K[:, 1, 0], K[:, 1, 2] = rho[:, 2], -rho[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -rho[:, 1], rho[:, 0]
    R += a[:, None, None] * K

    # dP/drho = q rho^T - a [X]x + b rho X^T + b (rho.X) I ,  q = -a X + a2 (rho x X) + b2 (rho.X) rho
    q = -a[:, None] * X + a2[:, None] * rxX + (b2 * rdX)[:, None] * rho
    Xx = np.zeros((M, 3, 3))
    Xx[:, 0, 1], Xx[:, 0, 2] = -X[:, 2], X[:, 1]
    Xx[:, 1, 0], Xx[:, 1, 2] = X[:, 2], -X[:, 0]
    Xx[:, 2, 0], Xx[:, 2, 1] = -X[:, 1], X[:, 0]
    dPdr = (q[:, :, None] * rho[:, None, :] - a[:, None, None] * Xx
            + b[:, None, None] * (rho[:, :, None] * X[:, None, :])
            + (b * rdX)[:, None, None] * eye)

    Jc = np.zeros((M, 2, NCP))
    Jc[:, :, 0:3] = A @ dPdr
    Jc[:, :, 3:6] = A
    Jc[:, 0, 6], Jc[:, 1, 6] = d * x, d * y
    Jc[:, 0, 7], Jc[:, 1, 7] = f * x * n, f * y * n
    Jc[:, 0, 8], Jc[:, 1, 8] = f * x * n * n, f * y * n * n
    Jc[:, 0, 9] = 1.0
    Jc[:, 1, 10] = 1.0
    Jp = A @ R
    Jc *= wv[:, None, None]
    Jp *= wv[:, None, None]
    return res, Jc, Jp


def jacobian_csr(Jc, Jp, ci, pi, n_cams, n_pts):
    """Assemble blocks into the (2M, 11C+3N) CSR layout scipy uses (pySBA.py:110-116)."""
    from scipy.sparse import csr_matrix
    M = ci.shape[0]
    cols_c = (ci[:, None] * NCP + np.arange(NCP)[None, :])
    cols_p = (n_cams * NCP + pi[:, None] * 3 + np.arange(3)[None, :])
    cols = np.concatenate([cols_c, cols_p], axis=1)                  # (M,14)
    cols = np.repeat(cols[:, None, :], 2, axis=1).reshape(2 * M, 14)
    data = np.concatenate([Jc, Jp], axis=2).reshape(2 * M, 14)
    indptr = np.arange(0, 2 * M * 14 + 1, 14)
    J = csr_matrix((data.ravel(), cols.ravel(), indptr), shape=(2 * M, n_cams * NCP + n_pts * 3))
    J.sort_indices()
    return J


def normal_blocks(res, Jc, Jp, ci, pi, n_cams, n_pts):
    """U (C,11,11), gc (C,11), V (N,3,3), gp (N,3), W (M,11,3) -- blocks of J^T J and J^T r."""
    U = np.zeros((n_cams, NCP, NCP))
    gc = np.zeros((n_cams, NCP))
    V = np.zeros((n_pts, 3, 3))
    gp = np.zeros((n_pts, 3))
    np.add.at(U, ci, np.einsum("mri,mrj->mij", Jc, Jc))
    np.add.at(gc, ci, np.einsum("mri,mr->mi", Jc, res))
    np.add.at(V, pi, np.einsum("mri,mrj->mij", Jp, Jp))
    np.add.at(gp, pi, np.einsum("mri,mr->mi", Jp, res))
    W = np.einsum("mri,mrj->mij", Jc, Jp)
    return U, gc, V, gp, W


def reduced_system(U, gc, V, gp, W, ci, pi, lam, D2p):
    """Undamped-camera Schur complement: S = U - sum W V'^-1 W^T, rhs = -(gc - sum W V'^-1 gp).

    V' = V + lam*diag(D2p).  The camera damping lam*D2c is added by the caller AFTER the
    cross-rank sum (it needs the global diag(U)).
    """
    C = U.shape[0]
    n = C * NCP
    Vd = V + lam * (D2p[:, :, None] * np.eye(3)[None])
    Vinv = np.linalg.inv(Vd)
    S = np.zeros((n, n))
    for c in range(C):
        S[c * NCP:(c + 1) * NCP, c * NCP:(c + 1) * NCP] = U[c]
    rhs = -gc.reshape(-1).copy()
    Y = np.einsum("mij,mjk->mik", W, Vinv[pi])                     # (M,11,3)
    rhs_add = np.einsum("mik,mk->mi", Y, gp[pi])
    np.add.at(rhs.reshape(C, NCP), ci, rhs_add)
    # pairwise blocks per point
    order = np.argsort(pi, kind="stable")
    start = np.searchsorted(pi[order], np.arange(V.shape[0] + 1))
    for p in range(V.shape[0]):
        idx = order[start[p]:start[p + 1]]
        if idx.size == 0:
            continue
        Yp = Y[idx].reshape(-1, 3)                                  # (11k,3)
        Wp = W[idx].reshape(-1, 3)
        blk = Yp @ Wp.T
        rows = (ci[idx][:, None] * NCP + np.arange(NCP)[None, :]).ravel()
        S[np.ix_(rows, rows)] -= blk
    return S, rhs, Vinv


# ----------------------------------------------------------------------------- LM control (shared constants)
LAMBDA0 = 1e-4
LAMBDA_MIN = 1e-12
LAMBDA_MAX = 1e12

TERMINATION_MESSAGES = {
    -1: "Improper input parameters status returned from `leastsq`",
    0: "The maximum number of function evaluations is exceeded.",
    1: "`gtol` termination condition is satisfied.",
    2: "`ftol` termination condition is satisfied.",
    3: "`xtol` termination condition is satisfied.",
    4: "Both `ftol` and `xtol` termination conditions are satisfied.",
}


class ModelEngine:
    """One rank's shard of the problem, phase by phase (mirrors the C-ABI phase entry points).

    Cameras are replicated, points [p0,p1) and their observations are local.
    Exchange buffer layout (float64), n = 11*C:
        [ S (n*n, row-major) | rhs (n) | diagU (n) | gc (n) | cost ]
    """

    def __init__(self, cams, pts_local, uv, ci, pi_local, w=None, shared_intrinsics=False):
        self.cams = np.array(cams, dtype=np.float64)
        self.shared = shared_intrinsics
        self.pts = np.array(pts_local, dtype=np.float64)
        self.uv, self.ci, self.pi = uv, ci, pi_local
        self.w = np.ones(ci.shape[0]) if w is None else np.asarray(w, dtype=np.float64).reshape(-1)
        self.C, self.N = self.cams.shape[0], self.pts.shape[0]
        self.n = self.C * NCP
        if self.shared:     # pySBA.py:252-325: [f,k1,k2 | 6 extrinsics x C | 2 centre x C]
            C = self.C
            self.ns = 3 + 8 * C
            self.T = np.zeros((self.n, self.ns))
            for c in range(C):
                for e in range(NCP):
                    col = 3 + 6 * c + e if e < 6 else e - 6 if e < 9 else 3 + 6 * C + 2 * c + (e - 9)
                    self.T[c * NCP + e, col] = 1.0
        self.D2c = np.zeros(self.ns if self.shared else self.n)
        self.D2p = np.zeros((self.N, 3))
        self.lam, self.nu = LAMBDA0, 2.0
        self.nfev = self.njev = 0
        self.fresh = False
        self.begin()

    def begin(self, ftol=1e-8, xtol=1e-8, gtol=1e-8, max_nfev=None):
        self.ftol, self.xtol, self.gtol = ftol, xtol, gtol
        self.max_nfev = max_nfev if max_nfev else 100 * (self.n + 3 * self.N)

    def exchange_size(self):
        return self.n * self.n + 3 * self.n + 1

    batch = 1

    def decide_async(self, scalars_all, n_ranks=None):
        self._status, _ = self.decide(scalars_all, n_ranks)
        self._iters = getattr(self, "_iters", 0) + 1
        self.need_lin = self.accepted

    def poll(self):
        return self._status, self._iters

    def linearize(self):
        if not getattr(self, "need_lin", True):
            return
        res, Jc, Jp = residual_jacobian(self.cams, self.pts, self.uv, self.ci, self.pi, self.w)
        self.res = res
        self.U, self.gc, self.V, self.gp, self.W = normal_blocks(res, Jc, Jp, self.ci, self.pi, self.C, self.N)
        self.cost_loc = 0.5 * float(np.sum(res * res))
        dV = np.einsum("nii->ni", self.V)
        self.D2p = np.maximum(self.D2p, dV)
        self.D2p_eff = np.where(self.D2p > 0, self.D2p, 1.0)
        self.gmax_p = float(np.max(np.abs(self.gp))) if self.N else 0.0
        self.fresh = True
        if self.njev == 0:
            self.nfev = 1
        self.njev += 1

    def form_reduced(self):
        S, rhs, self.Vinv = reduced_system(self.U, self.gc, self.V, self.gp, self.W, self.ci, self.pi,
                                           self.lam, self.D2p_eff)
        n = self.n
        E = np.empty(self.exchange_size())
        E[:n * n] = S.ravel()
        E[n * n:n * n + n] = rhs
        E[n * n + n:n * n + 2 * n] = np.einsum("cii->ci", self.U).ravel()
        E[n * n + 2 * n:n * n + 3 * n] = self.gc.ravel()
        E[-1] = self.cost_loc
        return E

    def solve_trial(self, E):
        """E is the cross-rank SUM of form_reduced() buffers.  Returns local scalars (8,)."""
        n = self.n
        S = E[:n * n].reshape(n, n).copy()
        rhs = E[n * n:n * n + n]
        diagU = E[n * n + n:n * n + 2 * n]
        self.gc_tot = E[n * n + 2 * n:n * n + 3 * n]
        self.cost = float(E[-1])
        if self.shared:
            T = self.T
            S, rhs, diagU, self.gc_tot = T.T @ S @ T, T.T @ rhs, T.T @ diagU, T.T @ self.gc_tot
            n = self.ns
        if self.fresh:
            self.D2c = np.maximum(self.D2c, diagU)
            self.fresh = False
        self.D2c_eff = np.where(self.D2c > 0, self.D2c, 1.0)
        S[np.diag_indices(n)] += self.lam * self.D2c_eff
        fail = 0.0
        try:
            L = np.linalg.cholesky(S)
            dc = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
        except np.linalg.LinAlgError:
            fail, dc = 1.0, np.zeros(n)
        self.ds = dc                                    # step in the system's own unknowns
        self.xs = (np.linalg.pinv(self.T) @ self.cams.ravel()) if self.shared else self.cams.ravel()
        if self.shared:
            dc = self.T @ dc
        self.dc = dc
        # back-substitution: dp = V'^-1 ( -gp - W^T dc )
        Wt_dc = np.zeros((self.N, 3))
        np.add.at(Wt_dc, self.pi, np.einsum("mij,mi->mj", self.W, dc.reshape(self.C, NCP)[self.ci]))
        self.dp = np.einsum("nij,nj->ni", self.Vinv, -self.gp - Wt_dc)
        self.cams_new = self.cams + dc.reshape(self.C, NCP)
        self.pts_new = self.pts + self.dp
        res_new, _, _ = residual_jacobian(self.cams_new, self.pts_new, self.uv, self.ci, self.pi, self.w)
        self.res_new = res_new
        cost_new = 0.5 * float(np.sum(res_new * res_new))
        self.nfev += 1
        pred_p = 0.5 * float(np.sum(self.dp * (self.lam * self.D2p_eff * self.dp - self.gp)))
        return np.array([cost_new, pred_p, float(np.sum(self.dp ** 2)), float(np.sum(self.pts ** 2)),
                         self.gmax_p, fail, 0.0, 0.0])

    def decide(self, scalars_all, n_ranks=None):
        """scalars_all: (R,8) rows from every rank in rank order.  Returns (status or None, accepted)."""
        status = self._decide(np.asarray(scalars_all).reshape(-1, 8), self.ftol, self.xtol, self.gtol, self.max_nfev)
        return status, self.accepted

    def _decide(self, scalars_all, ftol, xtol, gtol, max_nfev):
        cost_new = float(np.sum(scalars_all[:, 0]))
        pred = float(np.sum(scalars_all[:, 1]))
        dx2 = float(np.sum(scalars_all[:, 2]))
        x2 = float(np.sum(scalars_all[:, 3]))
        gmax = float(np.max(scalars_all[:, 4]))
        fail = float(np.max(scalars_all[:, 5])) > 0
        dc = self.ds
        pred += 0.5 * float(np.sum(dc * (self.lam * self.D2c_eff * dc - self.gc_tot)))
        dx2 += float(np.sum(dc ** 2))
        x2 += float(np.sum(self.xs ** 2))
        gmax = max(gmax, float(np.max(np.abs(self.gc_tot))))
        self.optimality = gmax
        self.step_norm = np.sqrt(dx2)
        self.cost_trial = cost_new
        if gmax < gtol:                 # scipy tests this before taking a step (trf.py:452): the trial does not count
            self.accepted = False
            self.actual = self.rho = 0.0
            self.nfev -= 1
            return 1
        ok = (not fail) and np.isfinite(cost_new) and pred > 0
        actual = self.cost - cost_new if ok else -1.0
        rho = actual / pred if ok else -1.0
        self.actual, self.rho = actual, rho
        status = None
        if ok:
            f_ok = actual < ftol * self.cost and rho > 0.25
            x_ok = np.sqrt(dx2) < xtol * (xtol + np.sqrt(x2))
            status = 4 if (f_ok and x_ok) else 2 if f_ok else 3 if x_ok else None
        if actual > 0:
            self.cams, self.pts = self.cams_new, self.pts_new
            self.res = self.res_new
            self.cost_accepted = cost_new
            self.lam = min(max(self.lam * max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3), LAMBDA_MIN), LAMBDA_MAX)
            self.nu = 2.0
            self.accepted = True
        else:
            self.lam = min(self.lam * self.nu, LAMBDA_MAX)
            self.nu *= 2.0
            self.accepted = False
        if status is None and self.nfev >= max_nfev:
            status = 0
        return status


def run_lm_single(engine, ftol=1e-4, xtol=1e-8, gtol=1e-8, max_nfev=None, verbose=0, max_iter=10000):
    """Single-rank driver over a ModelEngine (the multi-rank one lives in lasercalib_amd/dist.py)."""
    engine.begin(ftol, xtol, gtol, max_nfev)
    status, it = None, 0
    need_lin = True
    while status is None and it < max_iter:
        if need_lin:
            engine.linearize()
        E = engine.form_reduced()
        sc = engine.solve_trial(E)
        status, need_lin = engine.decide(sc[None, :], 1)
        it += 1
        if verbose:
            print(f"{it:4d} nfev {engine.nfev:4d} cost {engine.cost:.6e} -> {engine.cost_trial:.6e} "
                  f"rho {engine.rho:8.4f} lam {engine.lam:.2e} |dx| {engine.step_norm:.3e} opt {engine.optimality:.3e}"
                  f" {'acc' if engine.accepted else 'REJ'}")
    final_cost = 0.5 * float(np.sum(engine.res ** 2))
    return dict(status=status, cams=engine.cams, pts=engine.pts, cost=final_cost, nfev=engine.nfev,
                njev=engine.njev, iterations=it, optimality=engine.optimality)
