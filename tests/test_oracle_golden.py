"""CPU: the oracle (oracle/sba_oracle.py) is pinned against fixtures produced by the reference itself
(oracle/make_golden.py imports /root/reference/lasercalib/pySBA.py), and the numpy model of the device
algorithm (oracle/lm_schur_model.py) is pinned against the same fixtures."""
import os

import numpy as np
import pytest
import scipy
from scipy.sparse import csr_matrix

from oracle import lm_schur_model as model
from oracle import sba_oracle as orc


def test_rotate_project_bit_identical(golden):
    g = golden("f1_project.npz")
    assert np.array_equal(orc.rotate(g["points"], g["cam_rows"][:, :3]), g["rotated"])
    assert np.array_equal(orc.project(g["points"], g["cam_rows"]), g["projected"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_fun_bit_identical(golden, tag):
    g = golden("f2_fun.npz")
    C, N = g[f"{tag}_shape"]
    x0, uv, ci, pi = g[f"{tag}_x0"], g[f"{tag}_uv"], g[f"{tag}_ci"], g[f"{tag}_pi"]
    assert np.array_equal(orc.fun(x0, C, N, ci, pi, uv, orc.default_weights(pi)), g[f"{tag}_unit_res"])
    assert np.array_equal(orc.fun(x0, C, N, ci, pi, uv, g[f"{tag}_w"].reshape(-1, 1)), g[f"{tag}_w_res"])


def test_default_weights_are_integer_ones(golden):
    g = golden("f5_variants.npz")
    w = orc.default_weights(g["pi"])
    assert str(w.dtype) == str(g["default_weight_dtype"]) and tuple(w.shape) == tuple(g["default_weight_shape"])
    assert int(g["getResiduals_raises"]) == 1


@pytest.mark.parametrize("tag", ["a", "b"])
def test_sparsity_and_fd_jacobian(golden, tag):
    from scipy.optimize._numdiff import approx_derivative
    g = golden("f3_jacobian.npz")
    C, N = g[f"{tag}_shape"]
    x0, uv, ci, pi = g[f"{tag}_x0"], g[f"{tag}_uv"], g[f"{tag}_ci"], g[f"{tag}_pi"]
    A = orc.sparsity(C, N, ci, pi).tocsr()
    A.sort_indices()
    assert np.array_equal(A.indices, g[f"{tag}_A_indices"]) and np.array_equal(A.indptr, g[f"{tag}_A_indptr"])
    assert A.nnz == 28 * ci.size and np.all(A.data == 1)
    if str(g["scipy_version"]) == scipy.__version__:
        J = approx_derivative(orc.fun, x0, method="3-point", sparsity=orc.sparsity(C, N, ci, pi),
                              args=(C, N, ci, pi, uv, orc.default_weights(pi))).tocsr()
        J.sort_indices()
        assert np.array_equal(J.data, g[f"{tag}_J_data"])
    # analytic Jacobian of the device-algorithm model vs the reference's finite differences
    res, Jc, Jp = model.residual_jacobian(x0[:11 * C].reshape(C, 11), x0[11 * C:].reshape(N, 3), uv, ci, pi, 1.0)
    Ja = model.jacobian_csr(Jc, Jp, ci, pi, C, N)
    Jref = csr_matrix((g[f"{tag}_J_data"], g[f"{tag}_J_indices"], g[f"{tag}_J_indptr"]), shape=Ja.shape)
    assert abs(Ja - Jref).max() <= 1e-6 * abs(Jref).max()
    assert np.max(np.abs(res.ravel() - orc.fun(x0, C, N, ci, pi, uv, 1.0))) <= 1e-9


@pytest.mark.parametrize("tag", ["cfg1", "sparse"])
def test_bundle_adjust_reproduces_reference(golden, tag):
    g = golden("f4_solves.npz")
    res, cams, pts = orc.bundle_adjust(g[f"{tag}_cams0"], g[f"{tag}_pts0"], g[f"{tag}_uv"], g[f"{tag}_ci"], g[f"{tag}_pi"],
                                       ftol=1e-4)
    assert res.status == int(g[f"{tag}_loose_status"])
    if str(g["scipy_version"]) == scipy.__version__:
        assert res.nfev == int(g[f"{tag}_loose_nfev"])
        assert np.allclose(res.x, g[f"{tag}_loose_x"], rtol=0, atol=1e-9)
        assert abs(res.cost - float(g[f"{tag}_loose_cost"])) <= 1e-12 * res.cost
    else:
        assert abs(res.cost - float(g[f"{tag}_loose_cost"])) <= 1e-5 * res.cost
    rms = orc.rms_reprojection(cams, pts, g[f"{tag}_uv"], g[f"{tag}_ci"], g[f"{tag}_pi"])
    assert abs(rms - float(g[f"{tag}_loose_rms"])) <= 1e-6
    intr, ratios = orc.gauge_invariants(cams)
    assert np.allclose(intr, g[f"{tag}_loose_intr"], atol=1e-6) and np.allclose(ratios, g[f"{tag}_loose_centre_ratios"], atol=1e-9)


def test_variants_reproduce_reference(golden):
    g = golden("f5_variants.npz")
    a = (g["cams0"], g["pts0"], g["uv"], g["ci"], g["pi"])
    same = str(g["scipy_version"]) == scipy.__version__
    tol = 1e-9 if same else 1e-4
    res, pts = orc.bundle_adjust_nocam(*a)
    assert abs(res.cost - float(g["nocam_cost"])) <= tol * res.cost and np.allclose(pts, g["nocam_pts"], atol=1e-6 if same else 1e-2)
    res, cams = orc.bundle_adjust_camonly(*a)
    assert abs(res.cost - float(g["camonly_cost"])) <= max(tol, 1e-6) * res.cost
    res, cams, pts = orc.bundle_adjust_sharedcam(*a)
    assert abs(res.cost - float(g["sharedcam_cost"])) <= tol * res.cost
    assert np.all(cams[:, 6:9] == cams[0, 6:9])          # shared f, k1, k2
    res, pts = orc.bundle_adjust_transform_points_3d(*a)
    assert abs(res.cost - float(g["transform_cost"])) <= max(tol, 1e-6) * res.cost


@pytest.mark.parametrize("tag", ["cfg1", "sparse"])
def test_device_algorithm_model_converges_to_reference(golden, tag):
    """The LM/Schur algorithm the GPU runs (here in numpy) reaches the reference's optimum at ftol=1e-4
    (one-sided: never worse; gap bounded by 1e-5 relative on these rigs)."""
    g = golden("f4_solves.npz")
    eng = model.ModelEngine(g[f"{tag}_cams0"], g[f"{tag}_pts0"], g[f"{tag}_uv"], g[f"{tag}_ci"], g[f"{tag}_pi"])
    out = model.run_lm_single(eng, ftol=1e-4)
    ref = float(g[f"{tag}_loose_cost"])
    assert out["status"] == 2 and out["cost"] <= ref * (1 + 1e-9) and (ref - out["cost"]) <= 1e-5 * ref
    # normal-equation blocks are consistent with the sparse Jacobian: J^T J and J^T r
    res, Jc, Jp = model.residual_jacobian(eng.cams, eng.pts, eng.uv, eng.ci, eng.pi, eng.w)
    C, N = eng.C, eng.N
    J = model.jacobian_csr(Jc, Jp, eng.ci, eng.pi, C, N)
    U, gc, V, gp, W = model.normal_blocks(res, Jc, Jp, eng.ci, eng.pi, C, N)
    JTJ = (J.T @ J).toarray()
    for c in range(C):
        assert np.allclose(JTJ[11 * c:11 * c + 11, 11 * c:11 * c + 11], U[c], rtol=1e-10, atol=1e-6)
    g_full = J.T @ res.ravel()
    assert np.allclose(g_full[:11 * C], gc.ravel(), rtol=1e-10, atol=1e-6)
    assert np.allclose(g_full[11 * C:], gp.ravel(), rtol=1e-10, atol=1e-6)
    # Schur complement vs dense elimination
    lam = 1e-3
    D2p = np.where(np.einsum("nii->ni", V) > 0, np.einsum("nii->ni", V), 1.0)
    S, rhs, _ = model.reduced_system(U, gc, V, gp, W, eng.ci, eng.pi, lam, D2p)
    H = JTJ.copy()
    H[11 * C:, 11 * C:] += lam * np.diag(D2p.ravel())
    B, E_, Cb = H[:11 * C, :11 * C], H[:11 * C, 11 * C:], H[11 * C:, 11 * C:]
    S_ref = B - E_ @ np.linalg.solve(Cb, E_.T)
    assert np.allclose(S, S_ref, rtol=1e-8, atol=1e-5 * abs(S_ref).max())
    rhs_ref = -(g_full[:11 * C] - E_ @ np.linalg.solve(Cb, g_full[11 * C:]))
    assert np.allclose(rhs, rhs_ref, rtol=1e-8, atol=1e-8 * abs(rhs_ref).max())


def test_tangential_oracle_is_the_reference_model_plus_one_term(golden):
    """oracle/sba_oracle_tangential.py (13-parameter rows, NOT pinnable to the reference for p != 0): with p1 = p2 = 0 it
    must reproduce the reference-generated F1 projections bit for bit, and the extra term is OpenCV's."""
    from oracle import sba_oracle_tangential as t13
    g = golden("f1_project.npz")
    rows = g["cam_rows"]
    rows13 = np.hstack([rows[:, :9], np.zeros((rows.shape[0], 2)), rows[:, 9:11]])
    assert np.array_equal(t13.project(g["points"], rows13), g["projected"])
    rows13[:, 9], rows13[:, 10] = 1e-3, -2e-3
    far = np.r_[0:128, 160:rows.shape[0]]
    q = orc.rotate(g["points"], rows[:, :3]) + rows[:, 3:6]
    x, y = q[:, 0] / q[:, 2], q[:, 1] / q[:, 2]
    r2 = x * x + y * y
    dx = 2 * 1e-3 * x * y - 2e-3 * (r2 + 2 * x * x)
    dy = 1e-3 * (r2 + 2 * y * y) - 2 * 2e-3 * x * y
    extra = np.stack([dx, dy], axis=1) * rows[:, 6:7]
    assert np.max(np.abs(t13.project(g["points"], rows13)[far] - (g["projected"] + extra)[far])) <= 1e-9


def test_tangential_oracle_against_exact_opencv_formula_vectors():
    """F8: the only pin the 13-parameter extension can have -- the published OpenCV formula evaluated in exact rational
    arithmetic (oracle/make_opencv_vectors.py), poses with exact rotations.  The last row has p1 = p2 = 0 and must also be the
    11-parameter reference model's value."""
    import json
    from oracle import sba_oracle_tangential as orc13
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "f8_opencv_tangential.json")))
    P = np.array([r["point"] for r in d["rows"]])
    cam = np.array([r["camera"] for r in d["rows"]])
    uv = np.array([r["uv"] for r in d["rows"]])
    assert np.max(np.abs(orc13.project(P, cam) - uv)) <= 1e-9
    last = cam[-1]
    assert last[9] == 0 and last[10] == 0
    row11 = np.hstack([last[:9], last[11:13]])
    assert np.max(np.abs(orc.project(P[-1:], row11[None]) - uv[-1:])) <= 1e-9
