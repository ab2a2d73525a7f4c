"""CPU ORACLE for the 13-parameter (radial + tangential) camera model -- test infrastructure, NOT product code.

PARITY UNPINNED for p1, p2 != 0: the reference has no tangential distortion term (its model is pySBA.py:76-89, radial
only; its exporter writes zeros for p1, p2, lasercalib/convert_params.py:110), so there is no reference output this
extension could be checked against.  What pins it:
  * with p1 = p2 = 0 it must reproduce oracle.sba_oracle.project bit for bit (tests/test_oracle_golden.py), i.e. it is
    the reference's model plus one additive term;
  * the term itself is OpenCV's (the convention of the calibration YAMLs the reference reads and writes,
    convert_params.py:66-87,105-123):  x' = x d + 2 p1 x y + p2 (r2 + 2 x^2),  y' = y d + p1 (r2 + 2 y^2) + 2 p2 x y.
The solver call is the reference's (pySBA.py:141): scipy least_squares(trf, 3-point FD, x_scale='jac', jac_sparsity).

Camera row: [rvec(3), t(3), f, k1, k2, p1, p2, cx, cy].
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import least_squares
from scipy.sparse import lil_matrix

from . import sba_oracle as base

N_CAM_PARAMS = 13


def project(points, camera_rows):
    """(M,3) world points x (M,13) gathered camera rows -> (M,2) pixels.  Operation order follows base.project
    (pySBA.py:76-89) so that p1 = p2 = 0 gives the same bits."""
    q = base.rotate(points, camera_rows[:, :3])
    q += camera_rows[:, 3:6]
    q = q[:, :2] / q[:, 2, np.newaxis]
    focal = camera_rows[:, 6]
    k1 = camera_rows[:, 7]
    k2 = camera_rows[:, 8]
    p1 = camera_rows[:, 9]
    p2 = camera_rows[:, 10]
    rad2 = np.sum(q ** 2, axis=1)
    distort = 1 + k1 * rad2 + k2 * rad2 ** 2
    x, y = q[:, 0].copy(), q[:, 1].copy()
    tang = np.stack([2 * p1 * x * y + p2 * (rad2 + 2 * x * x), p1 * (rad2 + 2 * y * y) + 2 * p2 * x * y], axis=1)
    q *= (distort * focal)[:, np.newaxis]
    q += tang * focal[:, np.newaxis]
    q += camera_rows[:, 11:13]
    return q


def fun(x, n_cams, n_pts, cam_ind, pt_ind, uv, weights):
    cams = x[: n_cams * N_CAM_PARAMS].reshape((n_cams, N_CAM_PARAMS))
    pts = x[n_cams * N_CAM_PARAMS:].reshape((n_pts, 3))
    return (weights * (project(pts[pt_ind], cams[cam_ind]) - uv)).ravel()


def sparsity(n_cams, n_pts, cam_ind, pt_ind):
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


def bundle_adjust(cams, pts, uv, cam_ind, pt_ind, weights=None, ftol=1e-4, verbose=0, max_nfev=None, **lsq_kw):
    """The reference's least_squares call (pySBA.py:141) on the 13-parameter model.  Returns (res, cams_opt, pts_opt)."""
    if weights is None:
        weights = base.default_weights(pt_ind)
    C, N = cams.shape[0], pts.shape[0]
    x0 = np.hstack((cams.ravel(), pts.ravel()))
    A = sparsity(C, N, cam_ind, pt_ind)
    res = least_squares(fun, x0, jac_sparsity=A, verbose=verbose, x_scale="jac", ftol=ftol, method="trf", jac="3-point",
                        max_nfev=max_nfev, args=(C, N, cam_ind, pt_ind, uv, weights), **lsq_kw)
    return res, res.x[: C * N_CAM_PARAMS].reshape((C, N_CAM_PARAMS)), res.x[C * N_CAM_PARAMS:].reshape((N, 3))


def fd_jacobian(x, n_cams, n_pts, cam_ind, pt_ind, uv, weights):
    """scipy's sparse 3-point finite-difference Jacobian of `fun` (what least_squares builds internally)."""
    from scipy.optimize._numdiff import approx_derivative, group_columns
    A = sparsity(n_cams, n_pts, cam_ind, pt_ind)
    groups = group_columns(A)
    return approx_derivative(fun, x, method="3-point", sparsity=(A, groups), args=(n_cams, n_pts, cam_ind, pt_ind, uv, weights))


def rms_reprojection(cams, pts, uv, cam_ind, pt_ind):
    d = project(pts[pt_ind], cams[cam_ind]) - uv
    return float(np.sqrt(np.mean(np.sum(d * d, axis=1))))
