"""GPU tests of the Levenberg-Marquardt control path (k_decide, csrc/sba_lm_kernels.hpp): scipy's status codes and the
"non-finite trial -> shrink and retry" rule.

scipy semantics being mirrored (scipy/optimize/_lsq/): status 0 = max_nfev reached (trf.py:437-438, least_squares.py:963),
1 = gtol, tested BEFORE a step is taken (trf.py:452), 2 = ftol, 3 = xtol, 4 = both (common.py:705-717); a trial point with
non-finite residuals is not an error: the trust region shrinks and the step is retried (trf.py:504-506).
Each status is provoked through the tolerances alone and compared with the numpy model of the device algorithm, which runs
the same tests in the same order (oracle/lm_schur_model.py).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from lasercalib_amd import _native  # noqa: E402
from lasercalib_amd.synth import make_rig  # noqa: E402
from oracle import lm_schur_model as model  # noqa: E402
from oracle import sba_oracle as orc  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert _native.device_count() > 0, "no HIP device visible: GPU tests must run on the MI355X box"


def _p(golden, tag="sparse"):
    g = golden("f4_solves.npz")
    return {k: g[f"{tag}_{k}"] for k in ("cams0", "pts0", "uv", "ci", "pi")}


def _dev(p, dtype="f64", **kw):
    with _native.Problem(p["cams0"], p["pts0"], p["uv"], p["ci"], p["pi"], dtype=dtype) as prob:
        return prob.solve_lm(prob.make_opts(**kw))


@pytest.mark.parametrize("name,kw,status", [
    ("max_nfev", dict(ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=3), 0),
    ("gtol", dict(ftol=1e-8, xtol=1e-8, gtol=1e12), 1),
    ("ftol", dict(ftol=1e-4, xtol=1e-15, gtol=1e-15), 2),
    ("xtol", dict(ftol=1e-15, xtol=1e-4, gtol=1e-15), 3),
    ("ftol+xtol", dict(ftol=1.0, xtol=1.0, gtol=1e-15), 4),
])
def test_scipy_status_codes(golden, name, kw, status):
    p = _p(golden)
    cams, pts, rep, log = _dev(p, **kw)
    eng = model.ModelEngine(p["cams0"], p["pts0"], p["uv"], p["ci"], p["pi"])
    out = model.run_lm_single(eng, **kw)
    assert rep.status == status == out["status"]
    assert rep.iterations == out["iterations"] and rep.nfev == out["nfev"]
    assert abs(rep.cost - out["cost"]) <= 1e-9 * out["cost"]
    if status == 0:
        assert rep.nfev == kw["max_nfev"]                               # stops exactly at the budget
    if status == 1:                                                      # no step was taken: x = x0, nfev = 1 like scipy
        assert rep.nfev == 1 and np.array_equal(cams, p["cams0"]) and np.array_equal(pts, p["pts0"])
        ref, _, _ = orc.bundle_adjust(p["cams0"], p["pts0"], p["uv"], p["ci"], p["pi"], ftol=1e-8, gtol=1e12)
        assert ref.status == 1 and ref.nfev == 1 and abs(ref.cost - rep.cost) <= 1e-12 * ref.cost


def test_default_max_nfev_is_100_n_like_scipy(golden):
    """max_nfev <= 0 means 100 * n parameters (trf.py:437-438): a tiny problem with unreachable tolerances must stop at it."""
    rig = make_rig(2, 3, seed=1, visibility=1.0, min_cams_per_point=2)
    p = dict(cams0=rig["cams0"], pts0=rig["pts0"], uv=rig["points_2d"], ci=rig["camera_ind"], pi=rig["point_ind"])
    cams, pts, rep, _ = _dev(p, ftol=0.0, xtol=0.0, gtol=0.0)
    assert rep.status == 0 and rep.nfev == 100 * (2 * 11 + 3 * 3)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_nonfinite_trial_is_rejected_and_retried(golden, dtype):
    """Phase API with the caller owning the scalar exchange (the multi-rank entry points): the trial cost of the second
    iteration is replaced by NaN / +inf before sba_lm_decide.  The step must be rejected (not an error, not accepted),
    the damping must grow, and the solve must then reach the same optimum as an undisturbed one."""
    import torch
    p = _p(golden)
    torch.cuda.set_device(0)
    for poison in (float("nan"), float("inf")):
        prob = _native.Problem(p["cams0"], p["pts0"], p["uv"], p["ci"], p["pi"], dtype=dtype,
                               stream=torch.cuda.current_stream().cuda_stream)
        prob.lm_begin(prob.make_opts(ftol=1e-6))
        E = torch.zeros(prob.exchange_size(), dtype=torch.float64, device="cuda")
        sc = torch.zeros(8, dtype=torch.float64, device="cuda")
        rows, status = [], -1
        while status < 0 and len(rows) < 200:
            prob.lm_linearize()
            prob.lm_form_reduced(E.data_ptr())
            prob.lm_solve_trial(E.data_ptr(), sc.data_ptr())
            if len(rows) == 1:
                sc[0] = poison
            status, accepted, row = prob.lm_decide(sc.data_ptr(), 1)
            rows.append((accepted, row.lambda_, row.cost, row.nfev))
        cams, pts, rep = prob.lm_finish()
        prob.close()
        assert rows[1][0] is False and rows[1][1] > rows[0][1]          # rejected, damping raised
        # cost unchanged (row 0 reports the trial kernel's sum, row 1 the linearisation kernel's sum at the same point: equal
        # to rounding of the summation order), the evaluation counted
        assert abs(rows[1][2] - rows[0][2]) <= (1e-12 if dtype == "f64" else 1e-6) * rows[0][2] and rows[1][3] == rows[0][3] + 1
        assert rows[2][0] is True                                         # the retry with the larger damping goes through
        assert status in (2, 3, 4) and np.all(np.isfinite(cams)) and np.all(np.isfinite(pts))
        c2, p2, clean, _ = _dev(p, dtype=dtype, ftol=1e-6)
        # two damping histories stopped by the same ftol = 1e-6: in fp64 they end within a few ftol of each other (observed
        # 3.7e-6).  In fp32 a cost difference of 1e-6 relative is below the rounding noise of the cost itself, so where such a
        # crawling solve (50+ iterations on this sparse 5 x 200 rig) stops is not reproducible between two trajectories:
        # observed 5e-4 apart, both within 1e-3 of the fp64 optimum
        assert abs(rep.cost - clean.cost) <= (1e-5 if dtype == "f64" else 1e-3) * clean.cost


def test_far_start_with_outliers_ends_in_a_stationary_point():
    """tools/robustness_sweep.py, scale 12, seed 0 (profiles/r1_robustness_sweep.txt): started 12x further from the truth
    than the usual initial guess, with 1 % gross (50 px) outliers, the device LM ends 0.25 % ABOVE scipy's TRF.  That is
    a different local minimum, not an early stop: tightening ftol from 1e-6 to 1e-10 moves the device cost by 4e-5 only,
    and scipy restarted from the device's point stays there.  Asserted: same stationary point at both tolerances, the
    reference cannot improve on it, and it lies within 0.5 % of the reference's own minimum."""
    rig = make_rig(6, 250, seed=100, visibility=0.8)
    rng = np.random.default_rng(0)
    cams0 = rig["cams_true"] + (rig["cams0"] - rig["cams_true"]) * 12.0
    pts0 = rig["pts_true"] + (rig["pts0"] - rig["pts_true"]) * 12.0
    uv = rig["points_2d"].copy()
    bad = rng.random(uv.shape[0]) < 0.01
    uv[bad] += rng.normal(0, 50.0, (int(bad.sum()), 2))
    p = dict(cams0=cams0, pts0=pts0, uv=uv, ci=rig["camera_ind"], pi=rig["point_ind"])
    ref, _, _ = orc.bundle_adjust(cams0, pts0, uv, p["ci"], p["pi"], ftol=1e-6)
    cams, pts, rep, _ = _dev(p, ftol=1e-6, max_nfev=5000)
    _, _, tight, _ = _dev(p, ftol=1e-10, max_nfev=5000)
    assert rep.status == 2 and tight.status in (2, 3, 4)
    assert 0 <= rep.cost - tight.cost <= 1e-4 * tight.cost
    again, _, _ = orc.bundle_adjust(cams, pts, uv, p["ci"], p["pi"], ftol=1e-6, max_nfev=50)
    assert again.cost >= rep.cost * (1 - 1e-5)
    assert abs(rep.cost - ref.cost) <= 5e-3 * ref.cost


@pytest.mark.parametrize("kw", [
    dict(ftol=1e-4),                                                                   # terminates by ftol inside a batch
    dict(ftol=0.0, xtol=0.0, gtol=0.0, max_iter=7, always_relinearize=True),           # stops on the iteration budget
    dict(ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=5, lambda0=1e-12),              # rejected steps in the mix
])
def test_decision_in_the_fused_prologue_equals_the_separate_kernel(kw, monkeypatch):
    """fp32, <= 16 cameras: k_schur_fused_bf3 takes the accept/reject decision of the previous trial in its prologue (every
    workgroup redundantly, workgroup 0 publishing the record; csrc/sba_kernels.hpp FusedDecide).  SBA_DECIDE_KERNEL=1 keeps
    the separate k_decide launch: same decision function on the same inputs -> identical logs and identical parameters."""
    rig = make_rig(12, 3000, seed=11, visibility=0.85)
    args = (rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"])

    def run():
        with _native.Problem(*args, dtype="f32") as prob:
            return prob.solve_lm(prob.make_opts(**kw))

    cams_a, pts_a, rep_a, log_a = run()
    monkeypatch.setenv("SBA_DECIDE_KERNEL", "1")
    cams_b, pts_b, rep_b, log_b = run()
    assert rep_a.status == rep_b.status and rep_a.iterations == rep_b.iterations and rep_a.nfev == rep_b.nfev
    assert rep_a.iterations >= 2
    assert len(log_a) == len(log_b) == rep_a.iterations
    for ra, rb in zip(log_a, log_b):
        for f in ("iteration", "accepted", "nfev", "cost", "cost_reduction", "step_norm", "optimality", "lambda_", "rho"):
            if hasattr(ra, f):
                assert getattr(ra, f) == getattr(rb, f), f
    assert rep_a.cost == rep_b.cost and rep_a.optimality == rep_b.optimality
    assert np.array_equal(cams_a, cams_b) and np.array_equal(pts_a, pts_b)


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_lm_run_is_the_loop_of_solve_lm(dtype):
    """sba_solve_lm = sba_lm_begin + sba_lm_run + sba_lm_finish (include/sba_hip.h): the three calls give the log, the report and
    the parameters of the one call, bit for bit.  bench.py times sba_lm_run alone."""
    rig = make_rig(16, 2500, seed=5, visibility=0.9)
    args = (rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"])
    kw = dict(ftol=1e-6)
    with _native.Problem(*args, dtype=dtype) as prob:
        cams_a, pts_a, rep_a, log_a = prob.solve_lm(prob.make_opts(**kw))
    with _native.Problem(*args, dtype=dtype) as prob:
        prob.lm_begin(prob.make_opts(**kw))
        status, iters = prob.lm_run()
        log_b = prob.iteration_log()
        cams_b, pts_b, rep_b = prob.lm_finish()
    assert status == rep_a.status and iters == rep_a.iterations == rep_b.iterations
    assert [r.cost for r in log_a] == [r.cost for r in log_b] and [r.accepted for r in log_a] == [r.accepted for r in log_b]
    assert rep_a.cost == rep_b.cost and rep_a.optimality == rep_b.optimality and rep_a.nfev == rep_b.nfev
    assert np.array_equal(cams_a, cams_b) and np.array_equal(pts_a, pts_b)
    with _native.Problem(*args, dtype=dtype) as prob:                # without a solve in progress the call is refused
        with pytest.raises(RuntimeError):
            prob.lm_run()
