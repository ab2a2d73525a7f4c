"""Seed-fixed synthetic laser-calibration rigs (measurement + test input only).

Shaped after the reference's example rig (example/config.json:8-11,22-23 and
example/calib_init_2024_05_02/*.yaml): cameras on a ring looking at a laser
pointer swept over two z-planes, image 3208x2200, f ~ 2400 px.  The recipe is
the one SURVEY.md section 8(d) fixes so that every round measures the same
problem.  Observations are emitted point-major / camera-minor, the order
scripts/get_points3d.py:78-86 produces.

This module is pure numpy and holds no solver code.
"""
from __future__ import annotations

import numpy as np


def _look_at_rotation(centre):
    """Rotation matrix whose rows are the camera axes; z looks at the origin."""
    z = -centre / np.linalg.norm(centre)
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z])


def _rotvec_from_matrix(Rm):
    """Log map of SO(3) (no scipy dependency; angles here stay far from pi... but handle it)."""
    cos_t = np.clip((np.trace(Rm) - 1.0) * 0.5, -1.0, 1.0)
    theta = np.arccos(cos_t)
    if theta < 1e-12:
        return np.zeros(3)
    if np.pi - theta < 1e-6:
        # axis from the symmetric part
        A = (Rm + np.eye(3)) * 0.5
        k = int(np.argmax(np.diag(A)))
        v = A[k] / np.sqrt(A[k, k])
        return v * theta
    w = np.array([Rm[2, 1] - Rm[1, 2], Rm[0, 2] - Rm[2, 0], Rm[1, 0] - Rm[0, 1]])
    return w * (theta / (2.0 * np.sin(theta)))


def _project_np(points, cams):
    """Plain pinhole + 2-term radial model used only to synthesise measurements."""
    rv = cams[:, :3]
    theta = np.linalg.norm(rv, axis=1)[:, None]
    with np.errstate(invalid="ignore", divide="ignore"):
        v = np.nan_to_num(rv / theta)
    dot = np.sum(points * v, axis=1)[:, None]
    c, s = np.cos(theta), np.sin(theta)
    p = c * points + s * np.cross(v, points) + dot * (1 - c) * v
    p = p + cams[:, 3:6]
    xy = p[:, :2] / p[:, 2:3]
    n = np.sum(xy * xy, axis=1)
    r = 1 + cams[:, 7] * n + cams[:, 8] * n * n
    if cams.shape[1] == 13:      # radial + tangential rows [.., k1, k2, p1, p2, cx, cy] (OpenCV convention)
        p1, p2, x, y = cams[:, 9], cams[:, 10], xy[:, 0], xy[:, 1]
        xd = x * r + 2 * p1 * x * y + p2 * (n + 2 * x * x)
        yd = y * r + p1 * (n + 2 * y * y) + 2 * p2 * x * y
        return np.stack([xd, yd], axis=1) * cams[:, 6][:, None] + cams[:, 11:13]
    return xy * (r * cams[:, 6])[:, None] + cams[:, 9:11]


def make_rig(n_cams, n_points, seed=0, visibility=1.0, noise_px=0.3,
             min_cams_per_point=2, perturb=True, tangential=False):
    """Return a dict with truth, initial guess and the observation list.

    tangential=True: 13-parameter camera rows [rvec, t, f, k1, k2, p1, p2, cx, cy] with p1, p2 ~ N(0, 1e-3)
    (BASELINE config 5 / SURVEY 8d; the initial guess starts them at zero, as a calibration without that term would).

    keys: cams_true, pts_true, cams0, pts0 (float64), points_2d (M,2) float64,
          camera_ind (M,) int64, point_ind (M,) int64 (non-decreasing).
    """
    rng = np.random.default_rng(seed)
    C, N = int(n_cams), int(n_points)
    phi = 2.0 * np.pi * np.arange(C) / C
    cams = np.zeros((C, 11))
    for i in range(C):
        centre = np.array([1500.0 * np.cos(phi[i]), 1500.0 * np.sin(phi[i]), 1200.0])
        Rm = _look_at_rotation(centre)
        cams[i, 0:3] = _rotvec_from_matrix(Rm)
        cams[i, 3:6] = -Rm @ centre
    cams[:, 6] = rng.normal(2400.0, 20.0, C)
    cams[:, 7] = rng.normal(0.0, 1e-3, C)
    cams[:, 8] = rng.normal(0.0, 1e-2, C)
    cams[:, 9] = rng.normal(1604.0, 10.0, C)
    cams[:, 10] = rng.normal(1100.0, 10.0, C)
    if tangential:               # drawn after everything the 11-parameter recipe draws at this point: same rig otherwise
        tp = np.random.default_rng(seed + 7919).normal(0.0, 1e-3, (C, 2))
        cams = np.hstack([cams[:, :9], tp, cams[:, 9:11]])

    pts = np.empty((N, 3))
    pts[:, 0:2] = rng.uniform(-700.0, 700.0, (N, 2))
    pts[:, 2] = np.where(rng.random(N) < 0.5, 0.0, 106.0)

    if visibility >= 1.0:
        vis = np.ones((N, C), dtype=bool)
    else:
        vis = rng.random((N, C)) < visibility
        # every point must keep >= min_cams_per_point cameras (get_points3d.py:52-56 filters likewise)
        short = vis.sum(axis=1) < min_cams_per_point
        for p in np.nonzero(short)[0]:
            need = rng.choice(C, size=min(min_cams_per_point, C), replace=False)
            vis[p, need] = True
    point_ind, camera_ind = np.nonzero(vis)          # row-major => point-major, camera-minor
    point_ind = point_ind.astype(np.int64)
    camera_ind = camera_ind.astype(np.int64)

    # (projected in blocks of observations: the gathered (M, 3) / (M, 13) operands of one call would be 16 GB at config 5's
    #  128 M observations; the values and the random stream are those of the single call)
    uv = np.empty((point_ind.size, 2))
    blk = 1 << 22
    for o in range(0, point_ind.size, blk):
        uv[o:o + blk] = _project_np(pts[point_ind[o:o + blk]], cams[camera_ind[o:o + blk]])
    uv += rng.normal(0.0, noise_px, uv.shape)

    cams0 = cams.copy()
    pts0 = pts.copy()
    if tangential:
        cams0[:, 9:11] = 0.0
    if perturb:
        cams0[:, 0:3] += rng.normal(0.0, 5e-3, (C, 3))
        cams0[:, 3:6] += rng.normal(0.0, 5.0, (C, 3))
        cams0[:, 6] += rng.normal(0.0, 20.0, C)
        pts0 += rng.normal(0.0, 5.0, (N, 3))
    return dict(cams_true=cams, pts_true=pts, cams0=cams0, pts0=pts0,
                points_2d=uv, camera_ind=camera_ind, point_ind=point_ind,
                n_cams=C, n_points=N)
