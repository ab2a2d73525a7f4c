"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

Usage:  python oracle/make_golden.py            (needs /root/reference; never runs on the GPU box)

The reference ships no tests or golden vectors, so the pins for this
path are produced here by importing /root/reference/lasercalib/pySBA.py and recording its
inputs and outputs.  Only data (arrays) is written -- no reference source travels.
"tight" solves use ftol=1e-8 (scipy's own default): below that the reference's TRF/LSMR step crawls
(cost reductions ~4e-7 per iteration on the 2x500 rig) and does not terminate in practical time.
Fixture families follow SURVEY.md section 8(c): F1 project/rotate, F2 fun, F3 Jacobian +
sparsity pattern, F4 converged solves, F5 variant solvers, F6 gauge-invariant summaries;
F9 (round 4) the tight optimum of the reference's `fun` by independent exact optimisers.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys

import numpy as np
import scipy

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

from lasercalib.pySBA import PySBA  # noqa: E402  (the reference, NOT the repo's package)
from scipy.optimize._numdiff import approx_derivative  # noqa: E402

from lasercalib_amd.synth import make_rig  # noqa: E402
from oracle import sba_oracle as orc  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"
VERS = dict(scipy_version=scipy.__version__, numpy_version=np.__version__)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def ref_instance(rig, weights=None):
    return PySBA(rig["cams0"].copy(), rig["pts0"].copy(), rig["points_2d"], rig["camera_ind"],
                 rig["point_ind"], pointWeights=weights)


def f1_project():
    rng = np.random.default_rng(101)
    M = 4096
    rig = make_rig(8, 64, seed=5)
    cams = rig["cams_true"][rng.integers(0, 8, M)].copy()
    pts = rng.uniform(-700, 700, (M, 3))
    pts[:, 2] = rng.uniform(-50, 150, M)
    # special rotations: theta = 0, ~1e-9, ~1e-4, ~pi ; near-zero depth rows
    cams[0:32, 0:3] = 0.0
    cams[32:64, 0:3] = rng.normal(0, 1e-9, (32, 3))
    cams[64:96, 0:3] = rng.normal(0, 1e-4, (32, 3))
    ax = rng.normal(0, 1, (32, 3))
    ax /= np.linalg.norm(ax, axis=1)[:, None]
    cams[96:128, 0:3] = ax * (np.pi - rng.uniform(0, 1e-3, (32, 1)))
    cams[128:160, 3:6] = rng.normal(0, 50, (32, 3))        # points close to the camera (small z)
    sba = PySBA(cams, pts, None, None, np.zeros(1, dtype=np.int64))
    rot = sba.rotate(pts, cams[:, :3])
    uv = sba.project(pts, cams)
    np.savez_compressed(os.path.join(OUT, "f1_project.npz"), points=pts, cam_rows=cams,
                        rotated=rot, projected=uv, **VERS)


def f2_fun():
    out = {}
    for tag, (C, N, vis) in dict(a=(2, 500, 1.0), b=(5, 200, 0.6)).items():
        rig = make_rig(C, N, seed=11, visibility=vis)
        M = rig["point_ind"].size
        rng = np.random.default_rng(7)
        w = rng.uniform(0.5, 2.0, M)
        x0 = np.hstack((rig["cams0"].ravel(), rig["pts0"].ravel()))
        for wtag, weights in (("unit", None), ("w", w)):
            sba = ref_instance(rig, weights)
            r = sba.fun(x0, C, N, sba.cameraIndices, sba.point2DIndices, sba.points2D, sba.pointWeights)
            out[f"{tag}_{wtag}_res"] = r
        out[f"{tag}_x0"] = x0
        out[f"{tag}_uv"] = rig["points_2d"]
        out[f"{tag}_ci"] = rig["camera_ind"]
        out[f"{tag}_pi"] = rig["point_ind"]
        out[f"{tag}_w"] = w
        out[f"{tag}_shape"] = np.array([C, N])
    np.savez_compressed(os.path.join(OUT, "f2_fun.npz"), **out, **VERS)


def f3_jacobian():
    out = {}
    for tag, (C, N, vis) in dict(a=(3, 40, 0.8), b=(2, 500, 1.0)).items():
        rig = make_rig(C, N, seed=21, visibility=vis)
        sba = ref_instance(rig)
        x0 = np.hstack((rig["cams0"].ravel(), rig["pts0"].ravel()))
        A = sba.bundle_adjustment_sparsity(C, N, sba.cameraIndices, sba.point2DIndices)
        J = approx_derivative(sba.fun, x0, method="3-point", sparsity=A,
                              args=(C, N, sba.cameraIndices, sba.point2DIndices, sba.points2D,
                                    sba.pointWeights)).tocsr()
        J.sort_indices()
        Ac = A.tocsr()
        Ac.sort_indices()
        out.update({f"{tag}_x0": x0, f"{tag}_uv": rig["points_2d"], f"{tag}_ci": rig["camera_ind"],
                    f"{tag}_pi": rig["point_ind"], f"{tag}_shape": np.array([C, N]),
                    f"{tag}_J_data": J.data, f"{tag}_J_indices": J.indices, f"{tag}_J_indptr": J.indptr,
                    f"{tag}_A_indices": Ac.indices, f"{tag}_A_indptr": Ac.indptr})
    np.savez_compressed(os.path.join(OUT, "f3_jacobian.npz"), **out, **VERS)


def f4_f6_solves():
    out = {}
    for tag, (C, N, vis) in dict(cfg1=(2, 500, 1.0), mid=(8, 2000, 1.0), sparse=(6, 600, 0.6)).items():
        rig = make_rig(C, N, seed=0, visibility=vis)
        out.update({f"{tag}_cams0": rig["cams0"], f"{tag}_pts0": rig["pts0"], f"{tag}_uv": rig["points_2d"],
                    f"{tag}_ci": rig["camera_ind"], f"{tag}_pi": rig["point_ind"]})
        for ftag, ftol in (("loose", 1e-4), ("tight", 1e-8)):
            sba = ref_instance(rig)
            res = quiet(sba.bundleAdjust, ftol)
            rms = orc.rms_reprojection(sba.cameraArray, sba.points3D, rig["points_2d"],
                                       rig["camera_ind"], rig["point_ind"])
            intr, ratios = orc.gauge_invariants(sba.cameraArray)
            out.update({f"{tag}_{ftag}_x": res.x, f"{tag}_{ftag}_cost": res.cost,
                        f"{tag}_{ftag}_nfev": res.nfev, f"{tag}_{ftag}_njev": res.njev,
                        f"{tag}_{ftag}_status": res.status, f"{tag}_{ftag}_optimality": res.optimality,
                        f"{tag}_{ftag}_rms": rms, f"{tag}_{ftag}_intr": intr,
                        f"{tag}_{ftag}_centre_ratios": ratios})
            print(tag, ftag, "cost", res.cost, "nfev", res.nfev, "status", res.status, "rms", rms, flush=True)
    np.savez_compressed(os.path.join(OUT, "f4_solves.npz"), **out, **VERS)


def f5_variants():
    C, N = 4, 300
    rig = make_rig(C, N, seed=3, visibility=0.9)
    out = {"cams0": rig["cams0"], "pts0": rig["pts0"], "uv": rig["points_2d"],
           "ci": rig["camera_ind"], "pi": rig["point_ind"]}

    sba = ref_instance(rig)
    res = quiet(sba.bundleAdjust_nocam)
    out.update(nocam_x=res.x, nocam_cost=res.cost, nocam_status=res.status, nocam_pts=sba.points3D)

    sba = ref_instance(rig)
    res = quiet(sba.bundle_adjustment_camonly)
    out.update(camonly_x=res.x, camonly_cost=res.cost, camonly_status=res.status,
               camonly_cams=sba.cameraArray)

    sba = ref_instance(rig)
    res = quiet(sba.bundleAdjust_sharedcam)
    out.update(sharedcam_x=res.x, sharedcam_cost=res.cost, sharedcam_status=res.status,
               sharedcam_cams=sba.cameraArray, sharedcam_pts=sba.points3D)

    sba = ref_instance(rig)
    res = quiet(sba.bundleAdjust_transform_points_3d)
    out.update(transform_x=res.x, transform_cost=res.cost, transform_status=res.status,
               transform_pts=sba.points3D)

    # quirk pins: getResiduals raises for M != 2 ; default weights are int ones (M,1)
    sba = ref_instance(rig)
    try:
        sba.getResiduals()
        raised = 0
    except ValueError:
        raised = 1
    out.update(getResiduals_raises=raised, default_weight_dtype=str(sba.pointWeights.dtype),
               default_weight_shape=np.array(sba.pointWeights.shape))
    np.savez_compressed(os.path.join(OUT, "f5_variants.npz"), **out, **VERS)



def _reference_lines(rel_path, first, last, dedent=0):
    """Source lines [first, last] (1-based, inclusive) of a reference script, de-indented -- executed here, never stored."""
    with open(os.path.join(REF, rel_path)) as f:
        lines = f.read().split("\n")[first - 1:last]
    return "\n".join(l[dedent:] if l.strip() else "" for l in lines)


def f7_dataset():
    """Observation-list builder and dataset concatenation, recorded FROM THE REFERENCE'S OWN LOOPS.

    The loops live inside two scripts with top-level I/O (not importable), so their line ranges are read from
    /root/reference and exec'd on synthetic centroids: scripts/get_points3d.py:48-61 (flip, filter) and :73-86 (the
    camera_ind / point_ind / points_2d double loop), scripts/calibrate_camera.py:35-44 (stacking with the non-cumulative
    point offset).  Only the input and output ARRAYS are stored (tests/golden/f7_dataset.npz)."""
    rng = np.random.default_rng(77)
    out = {}
    datasets = []
    n_cams = 5
    cam_names = [f"Cam{i}" for i in range(n_cams)]
    for d, n_pts in enumerate((60, 45, 30)):
        centroids = rng.uniform(0.0, 3000.0, (n_pts, 2, n_cams))
        unseen = rng.random((n_pts, n_cams)) < 0.35                   # NaN pairs: camera did not see the spot
        centroids[np.repeat(unseen[:, None, :], 2, axis=1)] = np.nan
        out[f"d{d}_centroids"] = centroids.copy()
        ns = dict(np=np, centroids=centroids.copy(), cam_names=cam_names, cam_name_for_3d_init="Cam2", n_pts=n_pts,
                  min_num_cam_per_point=3, n_cams=n_cams, print=lambda *a, **k: None)
        exec(_reference_lines("scripts/get_points3d.py", 48, 61, dedent=4), ns)
        exec(_reference_lines("scripts/get_points3d.py", 73, 86, dedent=4), ns)
        out[f"d{d}_keep"] = ns["keep"]
        out[f"d{d}_in_pts"] = ns["in_pts"]
        out[f"d{d}_camera_ind"] = ns["camera_ind"]
        out[f"d{d}_point_ind"] = ns["point_ind"]
        out[f"d{d}_points_2d"] = ns["points_2d"]
        pts3 = rng.normal(0.0, 300.0, (ns["n_in_pts"], 3))
        out[f"d{d}_points_3d"] = pts3
        datasets.append({"n_cams": n_cams, "n_pts": ns["n_in_pts"], "points_2d": ns["points_2d"], "points_3d": pts3,
                         "camera_ind": ns["camera_ind"], "point_ind": ns["point_ind"]})
    for tag, sel in (("two", datasets[:2]), ("three", datasets)):
        ns = dict(np=np, points_dataset=sel)
        exec(_reference_lines("scripts/calibrate_camera.py", 35, 44), ns)
        out[f"cat_{tag}_n_cams"] = np.array(ns["n_cams"])
        for k in ("points_3d", "points_2d", "camera_ind", "point_ind"):
            out[f"cat_{tag}_{k}"] = ns[k]
    np.savez_compressed(os.path.join(OUT, "f7_dataset.npz"), **out, **VERS)


def f6_convert():
    """Conversion functions of the reference (lasercalib/convert_params.py:7-27) on the shipped example calibration.

    convert_params imports cv2 at module level (absent here) although the two pure functions recorded below never touch
    it, and spells NaN as ``np.NaN`` (gone in numpy 2): an empty ``cv2`` module object and the alias are put in place
    for the import only -- none of the reference's cv2-using functions is called.  The example YAML numbers are stored as
    plain arrays (data shipped by the reference under example/calib_init_2024_05_02).
    """
    import glob
    import types
    from lasercalib_amd.convert_params import read_opencv_yaml
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    if not hasattr(np, "NaN"):
        np.NaN = np.nan
    from lasercalib import convert_params as ref_cp
    files = sorted(glob.glob("/root/reference/example/calib_init_2024_05_02/*.yaml"))
    names = [os.path.splitext(os.path.basename(f))[0] for f in files]
    K = np.stack([read_opencv_yaml(f)["camera_matrix"] for f in files])
    D = np.stack([read_opencv_yaml(f)["distortion_coefficients"] for f in files])
    Rm = np.stack([read_opencv_yaml(f)["rc_ext"] for f in files])
    T = np.stack([read_opencv_yaml(f)["tc_ext"] for f in files])
    from oracle import io_oracle
    cams = np.stack([io_oracle.camera_row_from_calibration(K[i], D[i], Rm[i], T[i]) for i in range(len(files))])
    rng = np.random.default_rng(6)
    extra = make_rig(5, 10, seed=6)["cams0"] + rng.normal(0, 1e-3, (5, 11))
    extra[0, :3] = 0.0                       # zero rotation
    extra[1, :3] = [np.pi, 0.0, 0.0]         # half turn
    allc = np.vstack([cams, extra])
    camList = [ref_cp.sba_to_readable_format(allc[i, :]) for i in range(allc.shape[0])]
    red = ref_cp.readable_to_red_format(camList)
    np.savez_compressed(os.path.join(OUT, "f6_convert.npz"), names=np.array(names), K=K, dist=D, R=Rm, T=T,
                        example_cameraArray=cams, cameraArray=allc,
                        readable_K=np.stack([c["K"] for c in camList]), readable_R=np.stack([c["R"] for c in camList]),
                        readable_t=np.stack([c["t"] for c in camList]), readable_d=np.stack([c["d"] for c in camList]),
                        red=red, **VERS)
    print("f6_convert: %d example cameras + %d synthetic" % (len(files), extra.shape[0]), flush=True)


def f9_tight_optimum():
    """The minimum of the REFERENCE'S OWN `fun`, found by optimisers that share nothing with the device algorithm or with
    oracle/lm_schur_model.py: (A) scipy's trust-region-reflective method with the EXACT (SVD) trust-region subproblem on a dense
    3-point finite-difference Jacobian of PySBA.fun, `x_scale='jac'`, all three tolerances at 1e-14; (B) MINPACK's lmder
    (`method='lm'`) on the same dense Jacobian, started from A's result; (C) only where neither converges within its cap, a textbook dense
    Levenberg-Marquardt from their best point.  Every entry carries `optimality` = ||J^T r||_inf of the reference's fun at the stored point:
    the stationarity certificate does not depend on who found the point.  The reference's own call (TRF + LSMR, sparse) stalls
    on ftol above this minimum (23.7608 vs 23.4586 on the 2 x 500 rig), which is why the tight-tolerance parity tests used to be
    one-sided; this family gives them a two-sided pin: cost, RMS reprojection, gauge-free summaries.  Arrays only."""
    from scipy.optimize import least_squares
    out = {}
    path = os.path.join(OUT, "f9_tight.npz")
    rigs = dict(cfg1=make_rig(2, 500, seed=0), sparse=make_rig(6, 600, seed=0, visibility=0.6), var=make_rig(4, 300, seed=3, visibility=0.9))
    # F9_RIGS=cfg1 regenerates one rig and keeps the others of an existing file; F9_NFEV = evaluation caps "trf,minpack" (the
    # two-camera rig crawls along its weak directions under the trust region: 120 evaluations are not enough there)
    only = [t for t in os.environ.get("F9_RIGS", "").split(",") if t]
    caps = [int(v) for v in os.environ.get("F9_NFEV", "120,40").split(",")]
    if only and os.path.exists(path):
        with np.load(path) as z:
            out.update({k: z[k] for k in z.files if k.split("_")[0] not in only and k not in VERS})
        rigs = {t: rigs[t] for t in only}
    for tag, rig in rigs.items():
        C, N = rig["n_cams"], rig["n_points"]
        sba = ref_instance(rig)
        args = (C, N, sba.cameraIndices, sba.point2DIndices, sba.points2D, sba.pointWeights)
        A = sba.bundle_adjustment_sparsity(C, N, sba.cameraIndices, sba.point2DIndices)

        def fun(x):
            return sba.fun(x, *args)

        def jac(x):
            return approx_derivative(sba.fun, x, method="3-point", sparsity=A, args=args).toarray()

        x0 = np.hstack((rig["cams0"].ravel(), rig["pts0"].ravel()))
        ra = least_squares(fun, x0, jac=jac, method="trf", tr_solver="exact", x_scale="jac", ftol=1e-14, xtol=1e-14, gtol=1e-14,
                           max_nfev=caps[0])
        rb = least_squares(fun, ra.x, jac=jac, method="lm", ftol=1e-14, xtol=1e-14, gtol=1e-14, max_nfev=caps[1])
        best = ra if ra.cost <= rb.cost else rb
        # (C) where neither of the two has converged within its cap (the two-camera rig: its trust region crawls along weakly determined
        # directions), a textbook dense Levenberg-Marquardt on the same finite-difference Jacobian takes over from their best point:
        # (J^T J + mu D^2) d = -J^T r with D = running maximum of the column norms, Nielsen's mu update.  Whoever finds the point, the
        # certificate is solver-independent: ||J^T r||_inf of the reference's fun, stored as `optimality`.
        rc_cost, rc_nfev = np.nan, 0
        if ra.status <= 0 and rb.status <= 0:
            x = best.x.copy()
            r = fun(x); J = jac(x); cost = 0.5 * float(r @ r)
            D2 = np.maximum(np.sum(J * J, axis=0), 1e-30)
            mu, nu = 1e-6, 2.0
            for rc_nfev in range(1, 401):
                g = J.T @ r
                if np.max(np.abs(g)) < 1e-9:
                    break
                H = J.T @ J
                H[np.diag_indices_from(H)] += mu * D2
                d = -np.linalg.solve(H, g)
                xn = x + d
                rn = fun(xn); cn = 0.5 * float(rn @ rn)
                pred = 0.5 * float(d @ (mu * D2 * d - g))
                rho = (cost - cn) / pred if pred > 0 else -1.0
                if cn < cost:
                    small = cost - cn < 1e-15 * cost
                    x, r, cost = xn, rn, cn
                    J = jac(x)
                    D2 = np.maximum(D2, np.sum(J * J, axis=0))
                    mu *= max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3); nu = 2.0
                    if small:
                        break
                else:
                    mu *= nu; nu *= 2.0
            rc_cost = cost
            if cost < best.cost:
                class _R:          # noqa: N801  (same fields as an OptimizeResult, as far as they are used below)
                    pass
                best = _R(); best.x, best.cost, best.optimality = x, cost, float(np.max(np.abs(J.T @ r)))
            print(tag, "dense LM from there:", repr(cost), rc_nfev, "optimality", float(np.max(np.abs(J.T @ r))), flush=True)
        cams, pts = best.x[:C * 11].reshape(C, 11), best.x[C * 11:].reshape(N, 3)
        rms = orc.rms_reprojection(cams, pts, rig["points_2d"], rig["camera_ind"], rig["point_ind"])
        intr, ratios = orc.gauge_invariants(cams)
        out.update({f"{tag}_cams0": rig["cams0"], f"{tag}_pts0": rig["pts0"], f"{tag}_uv": rig["points_2d"],
                    f"{tag}_ci": rig["camera_ind"], f"{tag}_pi": rig["point_ind"],
                    f"{tag}_cost_trf_exact": ra.cost, f"{tag}_cost_minpack": rb.cost, f"{tag}_cost": best.cost,
                    f"{tag}_nfev_trf_exact": ra.nfev, f"{tag}_nfev_minpack": rb.nfev, f"{tag}_optimality": best.optimality,
                    f"{tag}_status_trf_exact": ra.status, f"{tag}_status_minpack": rb.status, f"{tag}_cost_dense_lm": rc_cost,
                    f"{tag}_nfev_dense_lm": rc_nfev,
                    f"{tag}_x": best.x, f"{tag}_rms": rms, f"{tag}_intr": intr, f"{tag}_centre_ratios": ratios})
        print(tag, "trf-exact", repr(ra.cost), ra.nfev, ra.status, "| minpack", repr(rb.cost), rb.nfev, rb.status, "| optimality", best.optimality, "rms", rms, flush=True)
    np.savez_compressed(path, **out, **VERS)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    stages = dict(f1=f1_project, f2=f2_fun, f3=f3_jacobian, f4=f4_f6_solves, f5=f5_variants, f6=f6_convert, f7=f7_dataset, f9=f9_tight_optimum)
    for name in (sys.argv[1:] or list(stages)):
        stages[name]()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
