"""GPU test of the calibrate_camera.py flow around the hot path, on the 17-camera rig of the
reference's shipped example calibration: centroids -> filter -> observation list -> dataset concatenation -> PySBA on
the device -> reprojection report (device ``project``) -> readable / red / YAML export."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from lasercalib_amd import _native, convert_params as cp, dataset as ds, report  # noqa: E402
from lasercalib_amd.pySBA import PySBA  # noqa: E402
from oracle import io_oracle, sba_oracle as orc  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _example_problem(seed=0, n_frames=(900, 700), noise_px=0.3):
    f6 = np.load(os.path.join(GOLD, "f6_convert.npz"))
    cams_true = f6["example_cameraArray"]                       # 17 real cameras (example/calib_init_2024_05_02)
    C = cams_true.shape[0]
    rng = np.random.default_rng(seed)
    sets, truth_pts = [], []
    for n, z in zip(n_frames, (0.0, 106.0)):                    # the two laser planes of example/config.json
        X = np.column_stack([rng.uniform(-700, 700, n), rng.uniform(-700, 700, n), np.full(n, z)])
        cent = np.full((n, 2, C), np.nan)
        for j in range(C):
            uv = orc.project(X, np.repeat(cams_true[j][None], n, 0)) + rng.normal(0, noise_px, (n, 2))
            seen = (uv[:, 0] > 0) & (uv[:, 0] < 3208) & (uv[:, 1] > 0) & (uv[:, 1] < 2200) & (rng.random(n) < 0.8)
            cent[seen, :, j] = uv[seen]
        keep = ds.filter_points(cent, 4, C - 1)                 # min_num_cam_per_point = 4, init camera = Cam710038 (last)
        np.testing.assert_array_equal(keep, io_oracle.filter_points_loop(cent, 4, C - 1))
        sets.append(ds.make_dataset(cent[keep], X[keep] + rng.normal(0, 5.0, (int(keep.sum()), 3))))
        truth_pts.append(X[keep])
    n_cams, p3, p2, ci, pi = ds.concatenate_datasets(sets)
    cams0 = cams_true.copy()
    cams0[:, :3] += rng.normal(0, 3e-3, (C, 3))
    cams0[:, 3:6] += rng.normal(0, 3.0, (C, 3))
    cams0[:, 6] += rng.normal(0, 10.0, C)
    return dict(n_cams=n_cams, cams0=cams0, pts0=p3, uv=p2, ci=ci, pi=pi, cams_true=cams_true,
                pts_true=np.vstack(truth_pts), noise=noise_px, sets=sets)


def test_example_rig_flow_end_to_end(tmp_path):
    assert _native.device_count() > 0
    P = _example_problem()
    assert P["n_cams"] == 17 and ds.is_point_major(P["pi"])
    sba = PySBA(P["cams0"], P["pts0"], P["uv"], P["ci"], P["pi"])
    e0 = report.reprojection_errors(sba)                        # sba_print.py:17-19 through the device project kernel
    e0_ref = io_oracle.reprojection_errors(orc.project, sba.points3D, sba.cameraArray, sba.points2D, sba.cameraIndices, sba.point2DIndices)
    assert np.max(np.abs(e0 - e0_ref)) <= 1e-7
    res = sba.bundleAdjust(1e-4)
    assert res.status in (1, 2, 3, 4) and res.success
    s = report.reprojection_summary(sba)
    # converged to the measurement noise: per-axis sigma 0.3 px => RMS radial error ~ 0.3*sqrt(2) = 0.42 px
    assert s["rms"] < 1.5 * P["noise"] * np.sqrt(2) and s["rms"] < 0.05 * np.sqrt(np.mean(e0 ** 2))
    e1_ref = io_oracle.reprojection_errors(orc.project, sba.points3D, sba.cameraArray, sba.points2D, sba.cameraIndices, sba.point2DIndices)
    assert np.max(np.abs(report.reprojection_errors(sba) - e1_ref)) <= 1e-7
    # intrinsics are gauge-free: recovered to well under the 10 px perturbation (a camera with few views stays looser)
    df = np.abs(sba.cameraArray[:, 6] - P["cams_true"][:, 6])
    assert np.median(df) < 1.0 and df.max() < 8.0
    # export chain of calibrate_camera.py:75-83
    camList = cp.camera_array_to_readable(sba.cameraArray)
    red = cp.readable_to_red_format(camList)
    assert red.shape == (17, 25) and np.isfinite(red).all()
    np.testing.assert_array_equal(red[:, 0], sba.cameraArray[:, 6])
    names = [f"Cam{i}" for i in range(17)]
    cp.readable_format_to_aruco_format(str(tmp_path) + "/", 17, camList, names)
    back = cp.initialize_from_checkerboard(str(tmp_path), 17, names)
    np.testing.assert_allclose(back, sba.cameraArray, rtol=0, atol=1e-8)
    assert "tx" in report.camera_table(sba)
