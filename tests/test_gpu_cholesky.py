"""GPU: every factorisation kernel of the reduced camera system on its own, against numpy.

The phase API takes the reduced system from a device buffer the caller owns (the exchange buffer
``[S n*n | rhs n | diagU n | g_c n | cost 1]``, include/sba_hip.h), so a test can hand ``sba_lm_solve_trial`` ANY symmetric
positive definite S and read the camera step back with ``sba_lm_get_step``: the damped solve
``(S + lam diag(max(D, diagU)))^-1 rhs`` must agree with numpy whatever kernel the size selects --

    n <= 176 (<= 16 cameras):        k_cholesky_blocked (right-looking, all in LDS)
    176 < n <= 256 (17..23 cameras): k_cholesky_ll      (left-looking, factor on chip, sba_chol_ll.hpp)
    SBA_CHOL=ll:                     k_cholesky_ll for every n <= 256
    SBA_CHOL=blocked, n <= 209:      k_cholesky_stream instead of k_cholesky_ll
    larger, up to 22 block rows of 64 (24 .. 127 cameras): k_chol_big_dag (one launch: walker + tiles; fp32 engine: on f32
                                     lanes, the f64 instance behind it on a refusal) + k_chol_big_back_all
    SBA_CHOL_BIG=launches, or more:  k_chol_big_prepare / k_chol_big_step per block column

including sizes that are not multiples of the 16-wide blocks, an ill-conditioned matrix, and an indefinite one (the step
must come back as zero with the failure flag set, which the LM control turns into a rejected trial).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from lasercalib_amd import _native  # noqa: E402
from lasercalib_amd.synth import make_rig  # noqa: E402


def _spd(n, rng, cond=1e3):
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.geomspace(1.0, cond, n)
    A = (Q * ev) @ Q.T
    return 0.5 * (A + A.T)


def _solve_on_device(C, S, rhs, dU, dtype, lam=1e-3, want_retries=False):
    rig = make_rig(C, 40, seed=C)
    n = 11 * C
    with _native.Problem(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], dtype=dtype) as prob:
        prob.lm_begin(prob.make_opts(ftol=0, xtol=0, gtol=0, lambda0=lam))
        assert prob.exchange_size() == n * n + 3 * n + 1
        E = torch.zeros(prob.exchange_size(), dtype=torch.float64, device="cuda")
        E[: n * n] = torch.from_numpy(S.ravel())
        E[n * n: n * n + n] = torch.from_numpy(rhs)
        E[n * n + n: n * n + 2 * n] = torch.from_numpy(dU)
        sc = torch.zeros(8, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        prob.lm_solve_trial(E.data_ptr(), sc.data_ptr())
        step = prob.lm_get_step().ravel()
        _, _, rep = prob.lm_finish()
    return (step, int(rep.reserved)) if want_retries else step


CASES = [(2, "f64"), (5, "f64"), (11, "f64"), (16, "f64"), (16, "f32"), (17, "f64"), (17, "f32"), (19, "f64"), (20, "f32"),
         (23, "f64"), (23, "f32"), (24, "f64"), (32, "f32")]


@pytest.mark.parametrize("mode", ["default", "blocked", "ll"])
@pytest.mark.parametrize("C,dtype", CASES)
def test_reduced_system_solve_matches_numpy(monkeypatch, C, dtype, mode):
    if mode != "default":
        monkeypatch.setenv("SBA_CHOL", mode)
    else:
        monkeypatch.delenv("SBA_CHOL", raising=False)
    rng = np.random.default_rng(100 + C)
    n = 11 * C
    S = _spd(n, rng)
    rhs = rng.standard_normal(n)
    dU = np.abs(rng.standard_normal(n)) + 0.5
    lam = 1e-3
    step = _solve_on_device(C, S, rhs, dU, dtype, lam)
    A = S + lam * np.diag(dU)            # first linearisation: D = max(0, diagU)
    ref = np.linalg.solve(A, rhs)
    # f64 engine: pivots refined to 4e-15; f32 engine, f64 factorisation (more than 176 unknowns, or SBA_CHOL=ll / SBA_CHOL_F32=0): the
    # 5e-8 hardware estimate of 1/sqrt is used as it is, which perturbs the factored matrix by 1e-7 relative =>
    # solution error <= cond * 1e-7; f32 engine up to 256 unknowns (round 4): the factorisation runs on f32 lanes, cond * 1e-6
    f32_lanes = dtype == "f32" and ((n <= 176 and mode != "ll") or (176 < n <= 256 and mode == "default"))
    tol = 1e-9 if dtype == "f64" else 1e3 * (1e-6 if f32_lanes else 2e-7)
    assert np.max(np.abs(step - ref)) <= tol * np.max(np.abs(ref)), np.max(np.abs(step - ref)) / np.max(np.abs(ref))


@pytest.mark.parametrize("mode", ["default", "ll"])
@pytest.mark.parametrize("C", [16, 17, 21, 23])
def test_ill_conditioned_and_indefinite(monkeypatch, C, mode):
    if mode != "default":
        monkeypatch.setenv("SBA_CHOL", mode)
    else:
        monkeypatch.delenv("SBA_CHOL", raising=False)
    rng = np.random.default_rng(7 + C)
    n = 11 * C
    S = _spd(n, rng, cond=1e9)
    rhs = rng.standard_normal(n)
    dU = np.ones(n)
    step = _solve_on_device(C, S, rhs, dU, "f64", lam=1e-6)
    ref = np.linalg.solve(S + 1e-6 * np.eye(n), rhs)
    assert np.max(np.abs(step - ref)) <= 1e-5 * np.max(np.abs(ref))
    # an indefinite system: the factorisation must notice and return a zero step
    S2 = S.copy()
    k = n - 5
    S2[k, k] = -1.0
    step = _solve_on_device(C, S2, rhs, dU, "f64", lam=1e-6)
    assert np.all(step == 0.0)


@pytest.mark.parametrize("C", [5, 11, 16])
def test_f32_lane_factorisation_of_the_fp32_engine(monkeypatch, C):
    """fp32 engine, up to 176 unknowns: k_cholesky_blocked factors on f32 lanes (f32 pivot chain, v_mfma_f32_16x16x4) and repeats
    the factorisation in f64 only when the f32 one is refused.  Bars: solution error <= cond * 1e-6 for cond <= 1e5 with NO
    repeat; at cond 1e9 (lam 1e-6) the pivots sink below 2^-23 of their diagonal entries, the repeat is taken (counted in the
    report) and the answer has the f64 kernel's accuracy; SBA_CHOL_F32=0 never takes the f32 path."""
    monkeypatch.delenv("SBA_CHOL", raising=False)
    monkeypatch.delenv("SBA_CHOL_F32", raising=False)
    rng = np.random.default_rng(300 + C)
    n = 11 * C
    rhs = rng.standard_normal(n)
    dU = np.ones(n)
    for cond in (1e2, 1e4, 1e5):
        S = _spd(n, rng, cond=cond)
        step, retries = _solve_on_device(C, S, rhs, dU, "f32", lam=1e-6, want_retries=True)
        ref = np.linalg.solve(S + 1e-6 * np.eye(n), rhs)
        err = np.max(np.abs(step - ref)) / np.max(np.abs(ref))
        assert retries == 0 and err <= cond * 1e-6, (cond, retries, err)
    S = _spd(n, rng, cond=1e9)
    ref = np.linalg.solve(S + 1e-6 * np.eye(n), rhs)
    step, retries = _solve_on_device(C, S, rhs, dU, "f32", lam=1e-6, want_retries=True)
    assert retries == 1 and np.max(np.abs(step - ref)) <= 1e-5 * np.max(np.abs(ref)), (retries, np.max(np.abs(step - ref)) / np.max(np.abs(ref)))
    S2 = S.copy()
    S2[n - 5, n - 5] = -1.0                      # indefinite: refused in f32, refused again in f64, zero step
    step, retries = _solve_on_device(C, S2, rhs, dU, "f32", lam=1e-6, want_retries=True)
    assert retries == 1 and np.all(step == 0.0)
    monkeypatch.setenv("SBA_CHOL_F32", "0")
    S = _spd(n, rng, cond=1e4)
    step, retries = _solve_on_device(C, S, rhs, dU, "f32", lam=1e-6, want_retries=True)
    ref = np.linalg.solve(S + 1e-6 * np.eye(n), rhs)
    assert retries == 0 and np.max(np.abs(step - ref)) <= 1e4 * 2e-7 * np.max(np.abs(ref))


@pytest.mark.parametrize("C", [17, 20, 23])
def test_f32_lane_factorisation_between_177_and_256_unknowns(monkeypatch, C):
    """fp32 engine, 17 .. 23 cameras: the same right-looking kernel built for 16 block rows (only the f32 triangle fits the LDS: 20-float
    rows up to 14 block rows, 17-float rows beyond) runs in front of k_cholesky_ll, which then only runs when the f32 factorisation
    refused the system (LMState::chol_retry).  Bars: error <= cond * 1e-6 for cond <= 1e5 without a repeat; at cond 1e9 and for an
    indefinite system the repeat is taken (the latter still ends as a zero step)."""
    monkeypatch.delenv("SBA_CHOL", raising=False)
    monkeypatch.delenv("SBA_CHOL_F32", raising=False)
    rng = np.random.default_rng(400 + C)
    n = 11 * C
    rhs = rng.standard_normal(n)
    dU = np.ones(n)
    for cond in (1e2, 1e4, 1e5):
        S = _spd(n, rng, cond=cond)
        step, retries = _solve_on_device(C, S, rhs, dU, "f32", lam=1e-6, want_retries=True)
        ref = np.linalg.solve(S + 1e-6 * np.eye(n), rhs)
        err = np.max(np.abs(step - ref)) / np.max(np.abs(ref))
        assert retries == 0 and err <= cond * 1e-6, (cond, retries, err)
    S = _spd(n, rng, cond=1e9)
    step, retries = _solve_on_device(C, S, rhs, dU, "f32", lam=1e-6, want_retries=True)
    assert retries == 1 and np.all(np.isfinite(step))
    S[n - 5, n - 5] = -1.0
    step, retries = _solve_on_device(C, S, rhs, dU, "f32", lam=1e-6, want_retries=True)
    assert retries == 1 and np.all(step == 0.0)
    monkeypatch.setenv("SBA_CHOL_F32", "0")
    S = _spd(n, rng, cond=1e3)
    step, retries = _solve_on_device(C, S, rhs, dU, "f32", lam=1e-6, want_retries=True)
    ref = np.linalg.solve(S + 1e-6 * np.eye(n), rhs)
    assert retries == 0 and np.max(np.abs(step - ref)) <= 1e3 * 2e-7 * np.max(np.abs(ref))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_shared_intrinsics_system_through_the_left_looking_kernel(monkeypatch, dtype):
    """bundleAdjust_sharedcam ties f, k1, k2 of all cameras: 24 cameras give 3 + 8 * 24 = 195 tied unknowns, which the left-looking
    kernel factors (with its tie / first maps in the epilogue); SBA_CHOL=blocked sends the same system through the streamed kernel."""
    rig = make_rig(24, 220, seed=12, visibility=0.6)
    args = (rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"])

    def run(mode):
        if mode:
            monkeypatch.setenv("SBA_CHOL", mode)
        else:
            monkeypatch.delenv("SBA_CHOL", raising=False)
        with _native.Problem(*args, dtype=dtype) as prob:
            return prob.solve_lm(prob.make_opts(ftol=1e-6, mode=_native.MODE_SHARED_INTR, max_iter=8))

    ca, pa, ra, la = run(None)
    cb, pb, rb, lb = run("blocked")
    assert ra.iterations == rb.iterations == 8 or ra.status == rb.status
    tol = 1e-9 if dtype == "f64" else 1e-4
    assert abs(ra.cost - rb.cost) <= tol * rb.cost
    # the tied parameters received the same step in every camera, in both runs
    d = ca[:, 6:9] - rig["cams0"][:, 6:9]
    assert np.max(np.abs(d - d[0])) <= 1e-9 * max(1.0, np.max(np.abs(d)))
    assert np.max(np.abs(ca - cb)) <= (1e-6 if dtype == "f64" else 1e-2) * max(1.0, np.max(np.abs(cb)))


@pytest.mark.parametrize("C,dtype", [(24, "f32"), (47, "f64"), (47, "f32"), (64, "f64"), (64, "f32"), (100, "f64"), (127, "f32"), (128, "f64")])
@pytest.mark.parametrize("mode", ["dag", "launches"])
def test_large_systems_one_launch_and_per_column_launches(monkeypatch, C, dtype, mode):
    """24 cameras and more: the one-launch factorisation (k_chol_big_dag: a walker workgroup on the diagonal, one workgroup per
    tile behind flags, the system built from the exchange buffer inside the launch, back substitution and LM epilogue in
    k_chol_big_back_all) and the per-column launches of rounds 1-3 (SBA_CHOL_BIG=launches; 128 cameras = 23 block rows take them
    anyway) against numpy, at sizes whose last block row holds only the right-hand-side row (64 cameras: n = 704) and sizes that
    are no multiple of anything (47, 127)."""
    monkeypatch.delenv("SBA_CHOL", raising=False)
    monkeypatch.delenv("SBA_CHOL_F32", raising=False)
    if mode == "launches":
        monkeypatch.setenv("SBA_CHOL_BIG", "launches")
    else:
        monkeypatch.delenv("SBA_CHOL_BIG", raising=False)
    rng = np.random.default_rng(500 + C)
    n = 11 * C
    S = _spd(n, rng)
    rhs = rng.standard_normal(n)
    dU = np.abs(rng.standard_normal(n)) + 0.5
    step, retries = _solve_on_device(C, S, rhs, dU, dtype, 1e-3, want_retries=True)
    ref = np.linalg.solve(S + 1e-3 * np.diag(dU), rhs)
    f32_lanes = dtype == "f32" and mode == "dag" and C <= 127
    tol = 1e-9 if dtype == "f64" else 1e3 * (1e-6 if f32_lanes else 2e-7)
    assert retries == 0
    assert np.max(np.abs(step - ref)) <= tol * np.max(np.abs(ref)), np.max(np.abs(step - ref)) / np.max(np.abs(ref))


@pytest.mark.parametrize("C", [24, 32, 64])
def test_f32_lane_factorisation_of_large_systems(monkeypatch, C):
    """fp32 engine, 24 cameras and more: k_chol_big_dag<float> (f32 pivots with the pivot-growth test, f32 MFMAs, f32 tiles and
    images) with k_chol_big_dag<double> launched behind it, which only runs when the f32 factorisation refused the system.  Bars as
    for the smaller systems: error <= cond * 1e-6 for cond <= 1e5 without a repeat; at cond 1e9 the repeat is taken and the answer
    has f64 accuracy; an indefinite system is refused twice and ends as a zero step; SBA_CHOL_F32=0 never takes the f32 path."""
    monkeypatch.delenv("SBA_CHOL", raising=False)
    monkeypatch.delenv("SBA_CHOL_F32", raising=False)
    monkeypatch.delenv("SBA_CHOL_BIG", raising=False)
    rng = np.random.default_rng(600 + C)
    n = 11 * C
    rhs = rng.standard_normal(n)
    dU = np.ones(n)
    for cond in (1e2, 1e4, 1e5):
        S = _spd(n, rng, cond=cond)
        step, retries = _solve_on_device(C, S, rhs, dU, "f32", lam=1e-6, want_retries=True)
        ref = np.linalg.solve(S + 1e-6 * np.eye(n), rhs)
        err = np.max(np.abs(step - ref)) / np.max(np.abs(ref))
        assert retries == 0 and err <= cond * 1e-6, (cond, retries, err)
    S = _spd(n, rng, cond=1e9)
    ref = np.linalg.solve(S + 1e-6 * np.eye(n), rhs)
    step, retries = _solve_on_device(C, S, rhs, dU, "f32", lam=1e-6, want_retries=True)
    assert retries == 1 and np.max(np.abs(step - ref)) <= 1e-5 * np.max(np.abs(ref)), (retries, np.max(np.abs(step - ref)) / np.max(np.abs(ref)))
    S2 = S.copy()
    S2[n - 5, n - 5] = -1.0
    step, retries = _solve_on_device(C, S2, rhs, dU, "f32", lam=1e-6, want_retries=True)
    assert retries == 1 and np.all(step == 0.0)
    monkeypatch.setenv("SBA_CHOL_F32", "0")
    S = _spd(n, rng, cond=1e4)
    step, retries = _solve_on_device(C, S, rhs, dU, "f32", lam=1e-6, want_retries=True)
    ref = np.linalg.solve(S + 1e-6 * np.eye(n), rhs)
    assert retries == 0 and np.max(np.abs(step - ref)) <= 1e4 * 2e-7 * np.max(np.abs(ref))
