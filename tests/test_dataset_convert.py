"""CPU tests for the rows either side of the hot path: observation-list builder,
dataset concatenation, parameter conversion and YAML I/O -- product (lasercalib_amd) vs oracle restatement and vs the
golden vectors recorded from the reference's own conversion functions (tests/golden/f6_convert.npz)."""
import os

import numpy as np
import pytest

from lasercalib_amd import convert_params as cp
from lasercalib_amd import dataset as ds
from oracle import io_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _centroids(rng, n_pts, n_cams, p_seen):
    c = rng.uniform(0, 3000, size=(n_pts, 2, n_cams))
    hide = rng.random((n_pts, n_cams)) > p_seen
    c[np.broadcast_to(hide[:, None, :], c.shape)] = np.nan
    return c


@pytest.mark.parametrize("n_pts,n_cams,p", [(0, 4, 0.5), (1, 1, 1.0), (50, 3, 0.0), (200, 17, 0.6), (333, 5, 1.0)])
def test_observation_list_matches_reference_loops(n_pts, n_cams, p):
    rng = np.random.default_rng(n_pts + n_cams)
    c = _centroids(rng, n_pts, n_cams, p)
    ci, pi, uv = ds.observation_list(c)
    ci0, pi0, uv0 = orc.observation_list_loop(c)
    assert ci.dtype == np.int64 and pi.dtype == np.int64 and uv.dtype == np.float64
    np.testing.assert_array_equal(ci, ci0)
    np.testing.assert_array_equal(pi, pi0)
    np.testing.assert_array_equal(uv, uv0)            # bit-exact: values are copied, not computed
    assert ds.is_point_major(pi)


def test_observation_list_rejects_bad_shape():
    with pytest.raises(ValueError):
        ds.observation_list(np.zeros((4, 3, 2)))


@pytest.mark.parametrize("min_cams", [1, 4, 17])
def test_filter_points_matches_reference_loop(min_cams):
    rng = np.random.default_rng(3)
    c = _centroids(rng, 400, 17, 0.4)
    for cam3d in (0, 16):
        np.testing.assert_array_equal(ds.filter_points(c, min_cams, cam3d), orc.filter_points_loop(c, min_cams, cam3d))


def _datasets(k):
    rng = np.random.default_rng(10 + k)
    out = []
    for d in range(k):
        n = 20 + 7 * d
        c = _centroids(rng, n, 4, 0.7)
        c[:, :, 0] = rng.uniform(0, 100, (n, 2))          # camera 0 sees everything
        out.append(ds.make_dataset(c, rng.normal(size=(n, 3))))
    return out


@pytest.mark.parametrize("k", [1, 2, 3])
def test_concatenate_keeps_the_reference_offset_quirk(k):
    sets = _datasets(k)
    got = ds.concatenate_datasets(sets)
    ref = orc.concatenate_loop(sets)
    assert got[0] == ref[0]
    for a, b in zip(got[1:], ref[1:]):
        np.testing.assert_array_equal(a, b)
    cum = ds.concatenate_datasets(sets, cumulative_offsets=True)
    if k <= 2:
        np.testing.assert_array_equal(cum[4], got[4])    # the quirk only bites from the third dataset on
    else:
        assert cum[4].max() == sum(s["n_pts"] for s in sets) - 1
        assert got[4].max() < cum[4].max()
        assert not ds.is_point_major(got[4]) and ds.is_point_major(cum[4])


def test_dataset_dict_layout():
    d = _datasets(1)[0]
    assert sorted(d) == ["camera_ind", "n_cams", "n_pts", "point_ind", "points_2d", "points_3d"]
    assert d["points_2d"].shape == (d["camera_ind"].size, 2)


# ----------------------------------------------------------------------------- conversion (pinned by the reference)

@pytest.fixture(scope="module")
def f6():
    return np.load(os.path.join(GOLD, "f6_convert.npz"))


def test_sba_to_readable_and_red_match_reference_golden(f6):
    cams = f6["cameraArray"]
    camList = cp.camera_array_to_readable(cams)
    for i, c in enumerate(camList):
        np.testing.assert_array_equal(c["K"], f6["readable_K"][i])
        np.testing.assert_array_equal(c["R"], f6["readable_R"][i])
        np.testing.assert_array_equal(c["t"], f6["readable_t"][i])
        np.testing.assert_array_equal(c["d"], f6["readable_d"][i])
        o = orc.readable_from_row(cams[i])
        np.testing.assert_array_equal(o["K"], c["K"])
        np.testing.assert_array_equal(o["R"], c["R"])
    red = cp.readable_to_red_format(camList)
    assert red.shape == (cams.shape[0], 25) and not np.isnan(red).any()
    np.testing.assert_array_equal(red, f6["red"])
    np.testing.assert_array_equal(red[0], orc.red_row(camList[0]))
    assert np.shares_memory(camList[0]["t"], cams)          # t and d are views into the row, like upstream


def test_initialize_from_checkerboard_on_example_calibration(f6, tmp_path):
    names = [str(n) for n in f6["names"]]
    for i, n in enumerate(names):
        cp.write_opencv_yaml(str(tmp_path / f"{n}.yaml"), {
            "camera_matrix": f6["K"][i], "distortion_coefficients": f6["dist"][i], "rc_ext": f6["R"][i], "tc_ext": f6["T"][i]})
    cams = cp.initialize_from_checkerboard(str(tmp_path), len(names), names)
    np.testing.assert_array_equal(cams, f6["example_cameraArray"])     # %.16e round-trips f64 exactly
    assert cams.shape == (17, 11)
    assert abs(cams[0, 6] - f6["K"][0, 0, 0]) == 0


def test_opencv_yaml_round_trip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    nodes = {"image_width": 3208, "camera_matrix": rng.normal(size=(3, 3)) * 1e3, "tc_ext": rng.normal(size=3),
             "distortion_coefficients": np.array([1e-4, -2e-2, 0, 0, 0])}
    p = str(tmp_path / "c.yaml")
    cp.write_opencv_yaml(p, nodes)
    text = open(p).read()
    assert text.startswith("%YAML:1.0\n---\n") and "!!opencv-matrix" in text and "dt: d" in text
    back = cp.read_opencv_yaml(p)
    assert back["image_width"] == 3208
    np.testing.assert_array_equal(back["camera_matrix"], nodes["camera_matrix"])
    np.testing.assert_array_equal(back["tc_ext"], nodes["tc_ext"].reshape(3, 1))
    assert back["distortion_coefficients"].shape == (5, 1)
    with open(p, "a") as f:
        f.write("bad: !!opencv-matrix\n   rows: 2\n   cols: 2\n   dt: d\n   data: [ 1., 2., 3. ]\n")
    with pytest.raises(ValueError):
        cp.read_opencv_yaml(p)


def test_aruco_export_round_trips_through_initialize(f6, tmp_path):
    cams = f6["example_cameraArray"]
    names = [f"Cam{i}" for i in range(cams.shape[0])]
    cp.readable_format_to_aruco_format(str(tmp_path) + "/", cams.shape[0], cp.camera_array_to_readable(cams), names)
    back = cp.initialize_from_checkerboard(str(tmp_path), cams.shape[0], names)
    np.testing.assert_allclose(back, cams, rtol=0, atol=1e-9)


def test_shim_module_resolves_to_product(monkeypatch):
    import lasercalib.convert_params as shim
    assert shim.sba_to_readable_format is cp.sba_to_readable_format
    monkeypatch.delenv("LASERCALIB_UPSTREAM", raising=False)
    with pytest.raises(AttributeError):
        shim.load_from_blender


def test_report_helpers_cpu():
    from lasercalib_amd import report

    class Stub:
        cameraArray = np.array([[0.1, -0.2, 0.3, 10.0, 20.0, 30.0, 1000.0, 0.0, 0.0, 500.0, 400.0]])
    ex = report.camera_extrinsics(Stub())
    from scipy.spatial.transform import Rotation as R
    r_f = R.from_rotvec(-Stub.cameraArray[0, :3]).as_matrix()
    np.testing.assert_allclose(ex[0, :3, :3], r_f)
    np.testing.assert_allclose(ex[0, :3, 3], -r_f @ Stub.cameraArray[0, 3:6])
    assert "cx" in report.camera_table(Stub()).splitlines()[0]


def test_tangential_rows_export_p1_p2(tmp_path):
    """13-parameter rows (extension): p1, p2 travel through the readable dict, the 25-column table and the YAML, in OpenCV's
    [k1, k2, p1, p2, k3] order; upstream writes zeros there (convert_params.py:110) and 11-parameter rows still do."""
    from lasercalib_amd import convert_params as cp
    row11 = np.array([0.1, -0.2, 0.3, 10.0, 20.0, 1500.0, 2400.0, 1e-3, -2e-2, 1604.0, 1100.0])
    row13 = np.concatenate([row11[:9], [4e-4, -7e-4], row11[9:]])
    a, b = cp.sba_to_readable_format(row11), cp.sba_to_readable_format(row13)
    assert "p" not in a and np.array_equal(b["p"], [4e-4, -7e-4])
    assert np.array_equal(a["K"], b["K"]) and np.array_equal(a["R"], b["R"]) and np.array_equal(a["d"], b["d"])
    red = cp.readable_to_red_format([a, b])
    assert np.array_equal(red[0, 21:25], [1e-3, -2e-2, 0, 0]) and np.array_equal(red[1, 21:25], [1e-3, -2e-2, 4e-4, -7e-4])
    cp.readable_format_to_aruco_format(str(tmp_path) + "/", 2, [a, b], ["A", "B"])
    da = cp.read_opencv_yaml(str(tmp_path / "A.yaml"))["distortion_coefficients"].ravel()
    db = cp.read_opencv_yaml(str(tmp_path / "B.yaml"))["distortion_coefficients"].ravel()
    assert np.allclose(da, [1e-3, -2e-2, 0, 0, 0]) and np.allclose(db, [1e-3, -2e-2, 4e-4, -7e-4, 0])
    back = cp.initialize_from_checkerboard(str(tmp_path), 1, ["B"], tangential=True)
    assert back.shape == (1, 13) and np.allclose(back[0, 6:], row13[6:], rtol=1e-12) and np.allclose(back[0, 3:6], row13[3:6])
    assert cp.initialize_from_checkerboard(str(tmp_path), 1, ["B"]).shape == (1, 11)


# ----------------------------------------------------------------------------- pinned by the reference's own loops (F7)
def test_dataset_builder_pinned_by_reference_loops(golden):
    """tests/golden/f7_dataset.npz was recorded by exec'ing the reference's own line ranges -- scripts/get_points3d.py:48-61,
    73-86 and scripts/calibrate_camera.py:35-44 -- on synthetic centroids (oracle/make_golden.py f7): the vectorised product
    code and the restated oracle must both reproduce those arrays exactly, including the xy flip, the keep rule, the
    point-major / camera-minor order, dtypes, and the NON-cumulative point offset with three datasets."""
    g = golden("f7_dataset.npz")
    datasets = []
    for d in range(3):
        cen = np.flip(g[f"d{d}_centroids"], axis=1)                    # get_points3d.py:48
        keep = ds.filter_points(cen, 3, 2)
        assert np.array_equal(keep, g[f"d{d}_keep"]) and np.array_equal(keep, orc.filter_points_loop(cen, 3, 2))
        in_pts = cen[keep]
        assert np.array_equal(in_pts, g[f"d{d}_in_pts"], equal_nan=True)
        ci, pi, uv = ds.observation_list(in_pts)
        for got, key in ((ci, "camera_ind"), (pi, "point_ind"), (uv, "points_2d")):
            assert got.dtype == g[f"d{d}_{key}"].dtype and np.array_equal(got, g[f"d{d}_{key}"])
        ci2, pi2, uv2 = orc.observation_list_loop(in_pts)
        assert np.array_equal(ci2, ci) and np.array_equal(pi2, pi) and np.array_equal(uv2, uv)
        assert ds.is_point_major(pi)
        datasets.append(ds.make_dataset(in_pts, g[f"d{d}_points_3d"]))
    for tag, sel in (("two", datasets[:2]), ("three", datasets)):
        n_cams, p3, p2, ci, pi = ds.concatenate_datasets(sel)
        assert n_cams == int(g[f"cat_{tag}_n_cams"])
        for got, key in ((p3, "points_3d"), (p2, "points_2d"), (ci, "camera_ind"), (pi, "point_ind")):
            assert np.array_equal(got, g[f"cat_{tag}_{key}"])
        o = orc.concatenate_loop(sel)
        assert np.array_equal(o[4], pi) and np.array_equal(o[3], ci)
    # the quirk is real in the recorded data: with three datasets the last offset is n_pts of dataset 1 alone
    assert g["cat_three_point_ind"].max() < sum(int(d["n_pts"]) for d in datasets) - 1
    assert ds.concatenate_datasets(datasets, cumulative_offsets=True)[4].max() == sum(int(d["n_pts"]) for d in datasets) - 1
