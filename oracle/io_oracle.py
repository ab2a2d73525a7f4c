"""CPU restatement of the array code on either side of the bundle adjustment -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(lasercalib_amd/dataset.py, convert_params.py, report.py) never does.

Pinning: the dataset-builder functions restate loops that live inside two reference SCRIPTS with top-level file and GUI
I/O (scripts/get_points3d.py, scripts/calibrate_camera.py) which cannot be imported; they are **pinned by
tests/golden/f7_dataset.npz**, which oracle/make_golden.py f7 recorded by exec'ing the reference's own line ranges
(get_points3d.py:48-61,73-86; calibrate_camera.py:35-44) on synthetic centroids.  The conversion functions
are pinned by tests/golden/f6_convert.npz, produced by the reference's own ``sba_to_readable_format`` /
``readable_to_red_format`` (oracle/make_golden.py f6).
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation


def filter_points_loop(centroids, min_num_cam_per_point, cam_idx_3dpts):
    """scripts/get_points3d.py:52-58, loop for loop."""
    n_pts = centroids.shape[0]
    keep = np.zeros(shape=(n_pts,), dtype=bool)
    for i in range(n_pts):
        v = centroids[i, 0, :]
        if (np.sum(~np.isnan(v)) >= min_num_cam_per_point) and (~np.isnan(v[cam_idx_3dpts])):
            keep[i] = True
    return keep


def observation_list_loop(in_pts):
    """scripts/get_points3d.py:61,73-86: count the observations, then fill the three arrays point by point."""
    n_cams = in_pts.shape[2]
    n_obs = np.sum(~np.isnan(in_pts[:, 0, :].ravel()))
    camera_ind = np.zeros(shape=(n_obs,), dtype=int)
    point_ind = np.zeros(shape=(n_obs,), dtype=int)
    points_2d = np.zeros(shape=(n_obs, 2), dtype=float)
    ind = 0
    for i in range(in_pts.shape[0]):
        for j in range(n_cams):
            if np.isnan(in_pts[i, 0, j]):
                continue
            camera_ind[ind] = j
            point_ind[ind] = i
            points_2d[ind, :] = in_pts[i, :, j]
            ind += 1
    return camera_ind, point_ind, points_2d


def concatenate_loop(points_dataset):
    """scripts/calibrate_camera.py:32-44, including the offset list that holds n_pts of the PREVIOUS dataset only."""
    n = len(points_dataset)
    n_cams = points_dataset[0]["n_cams"]
    points_3d = np.vstack([points_dataset[i]["points_3d"] for i in range(n)])
    points_2d = np.vstack([points_dataset[i]["points_2d"] for i in range(n)])
    camera_ind = np.hstack([points_dataset[i]["camera_ind"] for i in range(n)])
    points_ind_offset = [0]
    for i in range(n - 1):
        points_ind_offset.append(points_dataset[i]["n_pts"])
    point_ind = np.hstack([points_dataset[i]["point_ind"] + points_ind_offset[i] for i in range(n)])
    return n_cams, points_3d, points_2d, camera_ind, point_ind


def camera_row_from_calibration(K, dist, Rm, T):
    """lasercalib/convert_params.py:79-86 for one camera, from the four matrices cv2.FileStorage would return."""
    row = np.zeros(11)
    row[0:3] = Rotation.from_matrix(Rm).as_rotvec()
    row[3:6] = np.asarray(T).reshape(-1)[:3]
    row[6:9] = [K[0, 0], np.asarray(dist).reshape(-1)[0], np.asarray(dist).reshape(-1)[1]]
    row[9:11] = [K[0, 2], K[1, 2]]
    return row


def readable_from_row(row):
    """lasercalib/convert_params.py:18-27 written out element by element."""
    K = np.zeros((3, 3))
    K[0, 0] = row[6]
    K[1, 1] = row[6]
    K[2, 2] = 1.0
    K[2, 0] = row[9]
    K[2, 1] = row[10]
    return {"K": K, "R": Rotation.from_rotvec(-np.asarray(row[:3])).as_matrix(), "t": np.asarray(row[3:6]), "d": np.asarray(row[7:9])}


def red_row(p):
    """lasercalib/convert_params.py:11-15 for one camera."""
    return np.hstack((np.transpose(p["K"]).ravel(), np.transpose(p["R"]).ravel(), p["t"], p["d"], [0.0, 0.0]))


def reprojection_errors(project, points3D, cameraArray, points2D, cameraIndices, point2DIndices):
    """lasercalib/sba_print.py:17-19 with ``project`` = the oracle's project."""
    r = project(points3D[point2DIndices], cameraArray[cameraIndices]) - points2D
    return np.sqrt(np.sum(r ** 2, axis=1))
