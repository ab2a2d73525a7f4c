"""Reporting pass after a solve: the numeric part of the reference's ``lasercalib/sba_print.py``.

``sba_print`` (sba_print.py:8-46) prints the camera table, histograms the per-observation reprojection error and draws
the rig.  The plotting stays with the caller; what it computes is here, with the 800k-observation reprojection pass
going through the device ``project`` kernel (``sba_project``) instead of numpy temporaries:

* :func:`reprojection_errors`  -- sba_print.py:17-19  (``||project(points3D[pi], cameraArray[ci]) - points2D||`` per observation)
* :func:`camera_table`         -- sba_print.py:12-15  (one row per camera; plain text, no ``prettytable``)
* :func:`camera_extrinsics`    -- sba_print.py:33-42  (4x4 pose handed to the pyramid drawer)
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation as R

__all__ = ["reprojection_errors", "reprojection_summary", "camera_table", "camera_extrinsics"]


def reprojection_errors(sba) -> np.ndarray:
    """(M,) Euclidean pixel error of every observation at the instance's current parameters (sba_print.py:17-19)."""
    r = sba.project(sba.points3D[sba.point2DIndices], sba.cameraArray[sba.cameraIndices]) - sba.points2D
    return np.sqrt(np.sum(r ** 2, axis=1))


def reprojection_summary(sba) -> dict:
    """Mean / RMS / median / 99th percentile of :func:`reprojection_errors` (the histogram's range, sba_print.py:21)."""
    e = reprojection_errors(sba)
    return {"n_obs": int(e.size), "mean": float(e.mean()), "rms": float(np.sqrt(np.mean(e ** 2))),
            "median": float(np.median(e)), "p99": float(np.percentile(e, 99)), "max": float(e.max())}


_COLS = ("rx", "ry", "rz", "tx", "ty", "tz", "f", "k1", "k2", "cx", "cy")


def camera_table(sba) -> str:
    """The camera rows ``sba_print`` feeds to PrettyTable (sba_print.py:12-15), as fixed-width text."""
    cams = np.asarray(sba.cameraArray, dtype=np.float64)
    head = " cam " + " ".join(f"{c:>12s}" for c in _COLS[: cams.shape[1]])
    rows = [f"{i:4d} " + " ".join(f"{v:12.6g}" for v in row) for i, row in enumerate(cams)]
    return "\n".join([head] + rows)


def camera_extrinsics(sba) -> np.ndarray:
    """(C,4,4) matrices ``ex`` of sba_print.py:33-42: rotation block = R(-rotvec), translation = -R(-rotvec) t."""
    cams = np.asarray(sba.cameraArray, dtype=np.float64)
    out = np.tile(np.eye(4), (cams.shape[0], 1, 1))
    for i, row in enumerate(cams):
        r_f = R.from_rotvec(-row[0:3]).as_matrix()
        out[i, :3, :3] = r_f              # (r_inv).T with r_inv = r_f.T
        out[i, :3, 3] = -r_f @ row[3:6]
    return out
