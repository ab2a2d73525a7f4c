"""Observation-list builder and dataset concatenation: the step right before ``PySBA``.

The reference builds the observation list with Python loops inside ``scripts/get_points3d.py`` and glues the per-laser
datasets together at the top of ``scripts/calibrate_camera.py``; neither is importable (both are scripts with
top-level I/O), so this module restates the array semantics, vectorised:

* :func:`filter_points`        -- get_points3d.py:52-58  (>= ``min_num_cam_per_point`` views and seen by the 3-D init camera)
* :func:`observation_list`     -- get_points3d.py:73-86  (point-major / camera-minor, NaN = not seen)
* :func:`concatenate_datasets` -- calibrate_camera.py:32-44 (incl. the NON-cumulative point offset, see below)
* :func:`is_point_major`       -- the ordering guarantee the device upload relies on to skip its counting sort

All arrays are numpy float64 / int64, exactly what ``PySBA.__init__`` (pySBA.py:28-59) takes.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

__all__ = ["filter_points", "observation_list", "make_dataset", "concatenate_datasets", "is_point_major"]


def filter_points(centroids: np.ndarray, min_num_cam_per_point: int, cam_idx_3dpts: int) -> np.ndarray:
    """Boolean keep-mask over the frames of one laser dataset (get_points3d.py:52-58).

    ``centroids`` is (n_pts, 2, n_cams) with NaN where a camera did not see the laser spot.  A frame is kept when at
    least ``min_num_cam_per_point`` cameras saw it AND the camera used for the 3-D initialisation saw it.  Only the
    first image coordinate is inspected, like the reference (``v = centroids[i, 0, :]``).
    """
    centroids = np.asarray(centroids)
    seen = ~np.isnan(centroids[:, 0, :])
    return (seen.sum(axis=1) >= int(min_num_cam_per_point)) & seen[:, int(cam_idx_3dpts)]


def observation_list(in_pts: np.ndarray):
    """(camera_ind, point_ind, points_2d) of one dataset, point-major / camera-minor (get_points3d.py:73-86).

    ``in_pts`` is (n_pts, 2, n_cams); an observation exists where ``in_pts[i, 0, j]`` is not NaN.  The reference fills
    the three arrays with a double Python loop (i outer, j inner); ``np.nonzero`` on the (n_pts, n_cams) mask walks the
    same order, so the result is element-for-element identical.  dtypes: int64, int64, float64.
    """
    in_pts = np.asarray(in_pts, dtype=np.float64)
    if in_pts.ndim != 3 or in_pts.shape[1] != 2:
        raise ValueError("in_pts must have shape (n_pts, 2, n_cams)")
    seen = ~np.isnan(in_pts[:, 0, :])
    point_ind, camera_ind = np.nonzero(seen)
    points_2d = in_pts[point_ind, :, camera_ind].astype(np.float64, copy=True)
    return camera_ind.astype(np.int64), point_ind.astype(np.int64), points_2d


def make_dataset(in_pts: np.ndarray, points_3d: np.ndarray) -> Dict[str, object]:
    """One entry of ``points_dataset.pkl`` (get_points3d.py:100-108): same keys, same dtypes."""
    camera_ind, point_ind, points_2d = observation_list(in_pts)
    return {
        "n_cams": int(np.asarray(in_pts).shape[2]),
        "n_pts": int(np.asarray(in_pts).shape[0]),
        "points_2d": points_2d,
        "points_3d": np.asarray(points_3d, dtype=np.float64),
        "camera_ind": camera_ind,
        "point_ind": point_ind,
    }


def concatenate_datasets(points_dataset: Sequence[Dict[str, object]], cumulative_offsets: bool = False):
    """Stack the per-laser datasets into the five ``PySBA`` inputs (calibrate_camera.py:32-44).

    Returns ``(n_cams, points_3d, points_2d, camera_ind, point_ind)``.

    Reference quirk, kept by default: the point-index offset of dataset ``i`` is ``n_pts`` of dataset ``i-1`` alone
    (calibrate_camera.py:41-43 appends ``points_dataset[i]['n_pts']``, it never accumulates), which is only right for
    one or two datasets -- the shipped example has two.  ``cumulative_offsets=True`` uses the running sum instead.
    """
    if len(points_dataset) == 0:
        raise ValueError("points_dataset is empty")
    n_cams = points_dataset[0]["n_cams"]
    points_3d = np.vstack([d["points_3d"] for d in points_dataset])
    points_2d = np.vstack([d["points_2d"] for d in points_dataset])
    camera_ind = np.hstack([d["camera_ind"] for d in points_dataset])
    offsets: List[int] = [0]
    for i in range(len(points_dataset) - 1):
        prev = int(points_dataset[i]["n_pts"])
        offsets.append(offsets[-1] + prev if cumulative_offsets else prev)
    point_ind = np.hstack([np.asarray(d["point_ind"]) + offsets[i] for i, d in enumerate(points_dataset)])
    return n_cams, points_3d, points_2d, camera_ind, point_ind


def is_point_major(point_ind: np.ndarray) -> bool:
    """True when observations are grouped by non-decreasing point index (what get_points3d.py:78-86 emits).

    The device upload (``sba_upload``) keeps the caller's order in that case and only falls back to a counting sort
    otherwise; either way results are reported in the caller's observation order.
    """
    point_ind = np.asarray(point_ind)
    return bool(point_ind.size == 0 or np.all(point_ind[1:] >= point_ind[:-1]))
