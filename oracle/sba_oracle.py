"""CPU ORACLE -- test infrastructure, NOT product code.

A numpy/scipy restatement of the reference's bundle-adjustment path
(/root/reference/lasercalib/pySBA.py).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; nothing under
lasercalib_amd/ does.  The arithmetic that is not in the reference itself lives
in the un-vendored third-party scipy.optimize.least_squares (reference call
sites pySBA.py:141,169,194,246,315; version used for the golden fixtures:
scipy 1.15.3 / numpy 2.2.6), so the restatement calls that same public function
with the reference's exact keyword set.

Parity pin: tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which oracle/make_golden.py produced by importing the
reference itself in the build container (the reference ships no tests or
golden vectors of its own -- SURVEY.md section 4).

Operation order inside rotate()/project() follows the reference exactly so
that results are bit-identical, which the golden tests assert.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import least_squares
from scipy.sparse import lil_matrix

N_CAM_PARAMS = 11  # [rotvec(3), t(3), f, k1, k2, cx, cy]   (pySBA.py:31-35)


# --------------------------------------------------------------------------- model
def rotate(points, rot_vecs):
    """Rodrigues rotation of (M,3) points by (M,3) rotation vectors (pySBA.py:61-73).

    theta == 0 maps to the identity because 0/0 -> nan -> 0 for the axis (pySBA.py:66-68).
    """
    angle = np.linalg.norm(rot_vecs, axis=1)[:, np.newaxis]
    with np.errstate(invalid="ignore"):
        axis = rot_vecs / angle
        axis = np.nan_to_num(axis)
    along = np.sum(points * axis, axis=1)[:, np.newaxis]
    ca = np.cos(angle)
    sa = np.sin(angle)
    return ca * points + sa * np.cross(axis, points) + along * (1 - ca) * axis


def project(points, camera_rows):
    """(M,3) world points x (M,11) gathered camera rows -> (M,2) pixels (pySBA.py:76-89)."""
    q = rotate(points, camera_rows[:, :3])
    q += camera_rows[:, 3:6]
    q = q[:, :2] / q[:, 2, np.newaxis]
    focal = camera_rows[:, 6]
    k1 = camera_rows[:, 7]
    k2 = camera_rows[:, 8]
    rad2 = np.sum(q ** 2, axis=1)
    distort = 1 + k1 * rad2 + k2 * rad2 ** 2
    q *= (distort * focal)[:, np.newaxis]
    q += camera_rows[:, 9:]
    return q


def default_weights(point_ind):
    """Integer ones, shape (M,1) -- the reference's default (pySBA.py:56-58)."""
    return np.full_like(point_ind, 1).reshape((-1, 1))


def fun(x, n_cams, n_pts, cam_ind, pt_ind, uv, weights):
    """Weighted reprojection residual, interleaved [u0,v0,u1,v1,...] (pySBA.py:92-101)."""
    cams = x[: n_cams * N_CAM_PARAMS].reshape((n_cams, N_CAM_PARAMS))
    pts = x[n_cams * N_CAM_PARAMS:].reshape((n_pts, 3))
    proj = project(pts[pt_ind], cams[cam_ind])
    return (weights * (proj - uv)).ravel()


def sparsity(n_cams, n_pts, cam_ind, pt_ind):
    """Jacobian pattern, 28 ones per observation (pySBA.py:103-118)."""
    m = cam_ind.size * 2
    n = n_cams * N_CAM_PARAMS + n_pts * 3
    A = lil_matrix((m, n), dtype=int)
    row = np.arange(cam_ind.size)
    for s in range(N_CAM_PARAMS):
        A[2 * row, cam_ind * N_CAM_PARAMS + s] = 1
        A[2 * row + 1, cam_ind * N_CAM_PARAMS + s] = 1
    for s in range(3):
        A[2 * row, n_cams * N_CAM_PARAMS + pt_ind * 3 + s] = 1
        A[2 * row + 1, n_cams * N_CAM_PARAMS + pt_ind * 3 + s] = 1
    return A


def split_params(x, n_cams, n_pts):
    """x -> (C,11), (N,3) views (pySBA.py:121-129)."""
    return (x[: n_cams * N_CAM_PARAMS].reshape((n_cams, N_CAM_PARAMS)),
            x[n_cams * N_CAM_PARAMS:].reshape((n_pts, 3)))


# --------------------------------------------------------------------------- solvers
def bundle_adjust(cams, pts, uv, cam_ind, pt_ind, weights=None, ftol=1e-4, verbose=0,
                  max_nfev=None, **lsq_kw):
    """Full BA exactly as pySBA.py:132-147 drives scipy.  Returns (res, cams_opt, pts_opt).
    lsq_kw: extra least_squares keywords the reference leaves at their defaults (xtol, gtol: the LM-control tests set them)."""
    if weights is None:
        weights = default_weights(pt_ind)
    C, N = cams.shape[0], pts.shape[0]
    x0 = np.hstack((cams.ravel(), pts.ravel()))
    A = sparsity(C, N, cam_ind, pt_ind)
    res = least_squares(fun, x0, jac_sparsity=A, verbose=verbose, x_scale="jac", ftol=ftol,
                        method="trf", jac="3-point", max_nfev=max_nfev,
                        args=(C, N, cam_ind, pt_ind, uv, weights), **lsq_kw)
    c_opt, p_opt = split_params(res.x, C, N)
    return res, c_opt, p_opt


def tight_optimum(cams, pts, uv, cam_ind, pt_ind, weights=None, max_nfev=(150, 60), model=None):
    """The minimum of the reference's `fun` NEAR (cams, pts), by optimisers that share nothing with the device algorithm or with
    oracle/lm_schur_model.py: scipy's TRF with the EXACT (SVD) trust-region subproblem on a DENSE 3-point finite-difference
    Jacobian of `fun` (x_scale='jac', every tolerance 1e-14), polished by MINPACK's lmder (`method='lm'`) on the same Jacobian.
    The reference's own call (TRF + LSMR on the sparse Jacobian, pySBA.py:141) stalls on ftol above this minimum, which is why
    comparisons with it at a tight tolerance can only be one-sided; tests that need a two-sided pin on a rig that is not in
    tests/golden/f9_tight.npz (which holds the same thing, computed from the reference's own `fun` from the initial guess) call
    this from the device's solution.  Dense: keep to a few thousand parameters.  model: a module with fun / sparsity /
    N_CAM_PARAMS (oracle.sba_oracle_tangential for 13-parameter rows).  Returns (cost, x)."""
    import sys
    from scipy.optimize._numdiff import approx_derivative
    m = model if model is not None else sys.modules[__name__]
    if weights is None:
        weights = default_weights(pt_ind)
    C, N = cams.shape[0], pts.shape[0]
    args = (C, N, cam_ind, pt_ind, uv, weights)
    A = m.sparsity(C, N, cam_ind, pt_ind)

    def f(x):
        return m.fun(x, *args)

    def jac(x):
        return approx_derivative(m.fun, x, method="3-point", sparsity=A, args=args).toarray()

    x0 = np.hstack((np.asarray(cams, dtype=np.float64).ravel(), np.asarray(pts, dtype=np.float64).ravel()))
    ra = least_squares(f, x0, jac=jac, method="trf", tr_solver="exact", x_scale="jac", ftol=1e-14, xtol=1e-14, gtol=1e-14,
                       max_nfev=max_nfev[0])
    rb = least_squares(f, ra.x, jac=jac, method="lm", ftol=1e-14, xtol=1e-14, gtol=1e-14, max_nfev=max_nfev[1])
    best = ra if ra.cost <= rb.cost else rb
    return float(best.cost), best.x


def bundle_adjust_ext(cams, pts, uv, cam_ind, pt_ind, fixed_mask=None, loss="linear", f_scale=1.0, weights=None,
                      ftol=1e-4, verbose=0, max_nfev=None, **lsq_kw):
    """EXTENSION oracle -- not a restatement of reference behaviour: the reference stores ``points3Dfixed``
    without using it (pySBA.py:28,55) and calls least_squares with the default linear loss (pySBA.py:141).  This is the same
    least_squares call with (i) the fixed points removed from the parameter vector (their coordinates enter ``fun`` as
    constants) and (ii) scipy's own ``loss=`` / ``f_scale=`` keywords.  With no fixed point and the linear loss it IS
    ``bundle_adjust`` (pinned by f4_solves.npz).  Returns (res, cams_opt, pts_opt)."""
    if weights is None:
        weights = default_weights(pt_ind)
    C, N = cams.shape[0], pts.shape[0]
    free = np.ones(N, dtype=bool) if fixed_mask is None else ~np.asarray(fixed_mask, dtype=bool)
    free_idx = np.nonzero(free)[0]
    col_of_pt = np.full(N, -1, dtype=np.int64)
    col_of_pt[free_idx] = np.arange(free_idx.size)
    nc = C * N_CAM_PARAMS

    def fun_ext(x):
        full = pts.copy()
        full[free_idx] = x[nc:].reshape((-1, 3))
        return (weights * (project(full[pt_ind], x[:nc].reshape((C, N_CAM_PARAMS))[cam_ind]) - uv)).ravel()

    A = lil_matrix((cam_ind.size * 2, nc + 3 * free_idx.size), dtype=int)
    row = np.arange(cam_ind.size)
    for s in range(N_CAM_PARAMS):
        A[2 * row, cam_ind * N_CAM_PARAMS + s] = 1
        A[2 * row + 1, cam_ind * N_CAM_PARAMS + s] = 1
    sel = free[pt_ind]
    for s in range(3):
        A[2 * row[sel], nc + col_of_pt[pt_ind[sel]] * 3 + s] = 1
        A[2 * row[sel] + 1, nc + col_of_pt[pt_ind[sel]] * 3 + s] = 1
    x0 = np.hstack((cams.ravel(), pts[free_idx].ravel()))
    res = least_squares(fun_ext, x0, jac_sparsity=A, verbose=verbose, x_scale="jac", ftol=ftol, method="trf", jac="3-point",
                        max_nfev=max_nfev, loss=loss, f_scale=f_scale, **lsq_kw)
    p_opt = pts.copy()
    p_opt[free_idx] = res.x[nc:].reshape((-1, 3))
    return res, res.x[:nc].reshape((C, N_CAM_PARAMS)), p_opt


def sparsity_nocam(n_pts, pt_ind):
    """pySBA.py:216-226."""
    A = lil_matrix((pt_ind.size * 2, n_pts * 3), dtype=int)
    row = np.arange(pt_ind.size)
    for s in range(3):
        A[2 * row, pt_ind * 3 + s] = 1
        A[2 * row + 1, pt_ind * 3 + s] = 1
    return A


def fun_nocam(x, cams, n_pts, cam_ind, pt_ind, uv, weights):
    """pySBA.py:228-235."""
    pts = x.reshape((n_pts, 3))
    return (weights * (project(pts[pt_ind], cams[cam_ind]) - uv)).ravel()


def bundle_adjust_nocam(cams, pts, uv, cam_ind, pt_ind, weights=None, ftol=1e-7, verbose=0):
    """Points-only BA (pySBA.py:237-250).  Returns (res, pts_opt)."""
    if weights is None:
        weights = default_weights(pt_ind)
    N = pts.shape[0]
    A = sparsity_nocam(N, pt_ind)
    res = least_squares(fun_nocam, pts.ravel(), jac_sparsity=A, verbose=verbose, x_scale="jac",
                        ftol=ftol, method="trf", jac="3-point",
                        args=(cams, N, cam_ind, pt_ind, uv, weights))
    return res, res.x.reshape((N, 3))


def fun_camonly(x, n_cams, n_pts, cam_ind, pt_ind, uv, weights, pts):
    """Cameras-only residual; note the SQUARED pixel error (pySBA.py:151-156)."""
    cams = x.reshape(n_cams, N_CAM_PARAMS)
    return (weights * (project(pts[pt_ind], cams[cam_ind]) - uv) ** 2).ravel()


def bundle_adjust_camonly(cams, pts, uv, cam_ind, pt_ind, weights=None, ftol=1e-4, verbose=0):
    """pySBA.py:160-173: dense default '2-point' Jacobian, no x_scale."""
    if weights is None:
        weights = default_weights(pt_ind)
    C, N = cams.shape[0], pts.shape[0]
    res = least_squares(fun_camonly, cams.ravel(), verbose=verbose, ftol=ftol, method="trf",
                        args=(C, N, cam_ind, pt_ind, uv, weights, pts))
    return res, res.x.reshape(C, N_CAM_PARAMS)


def fun_transform_points_3d(x, n_cams, n_pts, cams, cam_ind, pt_ind, uv, weights, pts):
    """3x4 affine applied to the points; squared pixel error (pySBA.py:176-187)."""
    T = np.vstack((x.reshape(3, 4), [0, 0, 0, 1]))
    homog = np.vstack((pts.transpose(), np.ones(shape=(1, n_pts))))
    moved = np.dot(T, homog).transpose()[:, :3]
    return (weights * (project(moved[pt_ind], cams[cam_ind]) - uv) ** 2).ravel()


def bundle_adjust_transform_points_3d(cams, pts, uv, cam_ind, pt_ind, weights=None, ftol=1e-3,
                                      verbose=0):
    """pySBA.py:190-205.  Returns (res, transformed points)."""
    if weights is None:
        weights = default_weights(pt_ind)
    C, N = cams.shape[0], pts.shape[0]
    x0 = np.hstack((np.eye(3), np.zeros((3, 1)))).ravel()
    res = least_squares(fun_transform_points_3d, x0, verbose=verbose, ftol=ftol, method="trf",
                        args=(C, N, cams, cam_ind, pt_ind, uv, weights, pts))
    T = np.vstack((res.x.reshape(3, 4), [0, 0, 0, 1]))
    homog = np.vstack((pts.transpose(), np.ones(shape=(1, N))))
    return res, np.dot(T, homog).transpose()[:, :3]


N_SHARED_INTR = 3   # f, k1, k2 shared by all cameras  (pySBA.py:254,281)
N_EXTR = 6
N_CENTRE = 2


def sparsity_sharedcam(n_cams, n_pts, cam_ind, pt_ind):
    """pySBA.py:252-275."""
    n_cam_params = n_cams * N_EXTR + n_cams * N_CENTRE + N_SHARED_INTR
    A = lil_matrix((cam_ind.size * 2, n_cam_params + n_pts * 3), dtype=int)
    row = np.arange(cam_ind.size)
    A[2 * row, 0:N_SHARED_INTR] = 1
    A[2 * row + 1, 0:N_SHARED_INTR] = 1
    for s in range(N_EXTR):
        A[2 * row, N_SHARED_INTR + cam_ind * N_EXTR + s] = 1
        A[2 * row + 1, N_SHARED_INTR + cam_ind * N_EXTR + s] = 1
    for s in range(N_CENTRE):
        A[2 * row, N_SHARED_INTR + n_cams * N_EXTR + cam_ind * N_CENTRE + s] = 1
        A[2 * row + 1, N_SHARED_INTR + n_cams * N_EXTR + cam_ind * N_CENTRE + s] = 1
    for s in range(3):
        A[2 * row, n_cam_params + pt_ind * 3 + s] = 1
        A[2 * row + 1, n_cam_params + pt_ind * 3 + s] = 1
    return A


def unpack_sharedcam(x, n_cams):
    """x -> (C,11) camera rows with tiled shared intrinsics, and the offset of the points."""
    n_cam_params = n_cams * (N_EXTR + N_CENTRE) + N_SHARED_INTR
    shared = x[:N_SHARED_INTR]
    extr = x[N_SHARED_INTR:N_SHARED_INTR + n_cams * N_EXTR].reshape((n_cams, N_EXTR))
    centre = x[N_SHARED_INTR + n_cams * N_EXTR: n_cam_params].reshape((n_cams, N_CENTRE))
    cams = np.concatenate((extr, np.tile(shared, (n_cams, 1)), centre), axis=1)
    return cams, n_cam_params


def fun_sharedcam(x, n_cams, n_pts, cam_ind, pt_ind, uv, weights):
    """pySBA.py:277-295."""
    cams, off = unpack_sharedcam(x, n_cams)
    pts = x[off:].reshape((n_pts, 3))
    return (weights * (project(pts[pt_ind], cams[cam_ind]) - uv)).ravel()


def bundle_adjust_sharedcam(cams, pts, uv, cam_ind, pt_ind, weights=None, ftol=1e-6, verbose=0):
    """pySBA.py:297-325.  Returns (res, cams_opt, pts_opt)."""
    if weights is None:
        weights = default_weights(pt_ind)
    C, N = cams.shape[0], pts.shape[0]
    shared0 = np.mean(cams[:, 6:9], axis=0).ravel()
    x0 = np.hstack((shared0, cams[:, :6].ravel(), cams[:, 9:].ravel(), pts.ravel()))
    A = sparsity_sharedcam(C, N, cam_ind, pt_ind)
    res = least_squares(fun_sharedcam, x0, jac_sparsity=A, verbose=verbose, x_scale="jac", ftol=ftol,
                        method="trf", jac="3-point", args=(C, N, cam_ind, pt_ind, uv, weights))
    c_opt, off = unpack_sharedcam(res.x, C)
    return res, c_opt, res.x[off:].reshape((N, 3))


# --------------------------------------------------------------------------- summaries
def rms_reprojection(cams, pts, uv, cam_ind, pt_ind):
    """RMS pixel distance over observations (the quantity sba_print.py:17-19 histograms)."""
    d = project(pts[pt_ind], cams[cam_ind]) - uv
    return float(np.sqrt(np.mean(np.sum(d * d, axis=1))))


def gauge_invariants(cams):
    """Similarity-gauge-free summaries of a calibration.

    Returns intrinsics (C,5) = [f,k1,k2,cx,cy] and the matrix of pairwise camera-centre
    distances divided by their mean.
    """
    C = cams.shape[0]
    centres = -rotate(cams[:, 3:6], -cams[:, 0:3])        # c = -R^T t
    d = np.linalg.norm(centres[:, None, :] - centres[None, :, :], axis=2)
    iu = np.triu_indices(C, 1)
    ratios = d[iu] / np.mean(d[iu]) if C > 1 else np.zeros(0)
    return cams[:, 6:11].copy(), ratios
