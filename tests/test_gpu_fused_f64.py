"""GPU: the fp64 fused linearise + Schur kernel of one-group rigs (k_schur_fused_f64, csrc/sba_schur_f64.hpp) -- the path an
unmodified calibrate_camera.py takes (PySBA.bundleAdjust, /root/reference/lasercalib/pySBA.py:194-212, runs in float64).

* the reduced camera system [S | rhs | diagU | g_c | cost], the trial scalars and the camera step of the fused kernel against the
  three-launch path it replaces (SBA_NO_FUSED64=1: k_linearize_cams + k_reduce_cams + k_schur_sym<LIN>) on the same shuffled,
  weighted problem: agreement at fp64 rounding; dense and masked visibility, camera counts below 16 (zero panel rows), slices
  whose last chunk is partial (1 .. 3 producer waves), fewer points than workgroups;
* the LM log of a whole solve, fused against three-launch: same accept/reject sequence, costs to 1e-12;
* converged solves against the oracle (scipy's least_squares as the reference calls it): never worse, and scipy restarted from
  the device solution cannot lower the cost;
* Huber loss, fixed points and shared intrinsics go through the same kernel;
* two runs of the same solve are bit-identical (the LDS accumulators are per producer wave: no order depends on timing).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from lasercalib_amd import _native  # noqa: E402
from lasercalib_amd.synth import make_rig  # noqa: E402
from oracle import sba_oracle as orc  # noqa: E402


def _system(rig, uv, ci, pi, wts, monkeypatch, fused, lam=1e-4, **kw):
    if fused:
        monkeypatch.delenv("SBA_NO_FUSED64", raising=False)
    else:
        monkeypatch.setenv("SBA_NO_FUSED64", "1")
    torch.cuda.set_device(0)
    with _native.Problem(rig["cams0"], rig["pts0"], uv, ci, pi, weights=wts, dtype="f64",
                         stream=torch.cuda.current_stream().cuda_stream, **kw) as prob:
        prob.lm_begin(prob.make_opts(ftol=1e-8, lambda0=lam))
        prob.lm_linearize()
        E = torch.zeros(prob.exchange_size(), dtype=torch.float64, device="cuda")
        prob.lm_form_reduced(E.data_ptr())
        sc = torch.zeros(8, dtype=torch.float64, device="cuda")
        prob.lm_solve_trial(E.data_ptr(), sc.data_ptr())
        torch.cuda.synchronize()
        step = prob.lm_get_step().ravel().copy()
        Eh, sch = E.cpu().numpy().copy(), sc.cpu().numpy().copy()
        prob.lm_finish()
    return Eh, sch, step


def _compare(Ea, sa, da, Eb, sb, db, n, tol=1e-11):
    S_a, S_b = Ea[: n * n].reshape(n, n), Eb[: n * n].reshape(n, n)
    assert np.array_equal(S_a, S_a.T)
    scale = np.sqrt(np.outer(np.abs(np.diag(S_b)), np.abs(np.diag(S_b))))
    assert np.max(np.abs(S_a - S_b) / scale) <= tol               # fp64 sums of ~N terms in two different orders
    for k, name in enumerate(("rhs", "diagU", "g_c")):
        a, b = Ea[n * n + k * n: n * n + (k + 1) * n], Eb[n * n + k * n: n * n + (k + 1) * n]
        assert np.max(np.abs(a - b)) <= tol * np.max(np.abs(b)), name
    assert abs(Ea[-1] - Eb[-1]) <= 1e-13 * Eb[-1]                 # cost
    assert np.max(np.abs(sa[:4] - sb[:4]) / (np.abs(sb[:4]) + 1e-30)) <= 1e-8
    assert np.max(np.abs(da - db)) <= 1e-7 * np.max(np.abs(db))   # camera step: the damped solve amplifies the rounding of S


# (cameras, points, visibility): 16 x 4099 = 256 workgroups x 16 points + 3 (a last chunk with one producer wave), 16 x 8200 -> 36
# points per workgroup (2 chunks + 4 points), 16 x 100 (fewer chunks than workgroups), camera counts that leave panel rows empty
@pytest.mark.parametrize("C,N,vis", [(16, 333, 1.0), (16, 4099, 1.0), (16, 8200, 1.0), (16, 100, 1.0), (16, 700, 0.5), (12, 500, 1.0),
                                     (7, 410, 0.7), (2, 300, 1.0), (13, 2500, 0.45)])
def test_fused_f64_builds_the_three_launch_system(monkeypatch, C, N, vis):
    rig = make_rig(C, N, seed=11 + C, visibility=vis, min_cams_per_point=2)
    rng = np.random.default_rng(5)
    perm = rng.permutation(rig["camera_ind"].size)
    uv, ci, pi = rig["points_2d"][perm], rig["camera_ind"][perm], rig["point_ind"][perm]
    wts = rng.uniform(0.5, 1.5, ci.size)
    Ea, sa, da = _system(rig, uv, ci, pi, wts, monkeypatch, fused=True)
    Eb, sb, db = _system(rig, uv, ci, pi, wts, monkeypatch, fused=False)
    _compare(Ea, sa, da, Eb, sb, db, 11 * C)


@pytest.mark.parametrize("C,N,vis", [(16, 900, 1.0), (10, 700, 0.6)])
def test_fused_f64_solve_follows_the_three_launch_solve(monkeypatch, C, N, vis):
    rig = make_rig(C, N, seed=3 + C, visibility=vis, min_cams_per_point=3)
    args = (rig["points_2d"], rig["camera_ind"], rig["point_ind"])
    logs = []
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("SBA_NO_FUSED64", raising=False)
        else:
            monkeypatch.setenv("SBA_NO_FUSED64", "1")
        with _native.Problem(rig["cams0"], rig["pts0"], *args, dtype="f64") as prob:
            cams, pts, rep, log = prob.solve_lm(prob.make_opts(ftol=1e-10))
        logs.append((cams, pts, rep, log))
    (ca, pa, ra, la), (cb, pb, rb, lb) = logs
    assert ra.status == rb.status and len(la) == len(lb)
    assert [r.accepted for r in la] == [r.accepted for r in lb]
    assert np.max(np.abs(np.array([r.cost for r in la]) / np.array([r.cost for r in lb]) - 1)) <= 1e-10
    assert abs(ra.cost - rb.cost) <= 1e-12 * rb.cost
    assert np.max(np.abs(ca - cb)) <= 1e-7 * np.max(np.abs(cb))


@pytest.mark.parametrize("C,N,vis", [(16, 400, 1.0), (8, 600, 0.6)])
def test_fused_f64_converges_to_the_reference_solution(C, N, vis):
    rig = make_rig(C, N, seed=9 + C, visibility=vis, min_cams_per_point=3)
    args = (rig["points_2d"], rig["camera_ind"], rig["point_ind"])
    with _native.Problem(rig["cams0"], rig["pts0"], *args, dtype="f64") as prob:
        cams, pts, rep, log = prob.solve_lm(prob.make_opts(ftol=1e-10))
    ref, _, _ = orc.bundle_adjust(rig["cams0"], rig["pts0"], *args, ftol=1e-10)
    assert rep.status in (2, 3, 4)
    # one-sided at tight tolerance, like tests/test_gpu_parity.py (scipy's TRF stops on xtol a little above the optimum): never
    # worse than the reference, same basin, and scipy restarted from the device solution cannot lower it
    assert rep.cost <= ref.cost * (1 + 1e-9) and rep.cost >= 0.99 * ref.cost
    again, _, _ = orc.bundle_adjust(cams, pts, *args, ftol=1e-10, max_nfev=20)
    assert again.cost >= rep.cost * (1 - 1e-6)
    cost64 = 0.5 * np.sum(orc.fun(np.hstack((cams.ravel(), pts.ravel())), C, N, rig["camera_ind"], rig["point_ind"], rig["points_2d"], 1.0) ** 2)
    assert abs(cost64 - rep.cost) <= 1e-10 * cost64               # the reported cost is the oracle's cost at the returned point


def test_fused_f64_with_huber_loss_fixed_points_and_shared_intrinsics(monkeypatch):
    rig = make_rig(16, 450, seed=21, visibility=0.8, min_cams_per_point=3)
    rng = np.random.default_rng(8)
    uv = rig["points_2d"].copy()
    bad = rng.choice(uv.shape[0], uv.shape[0] // 25, replace=False)
    uv[bad] += rng.normal(0, 40.0, (bad.size, 2))                 # gross outliers: the Huber branch is taken
    fixed = np.zeros(450, dtype=np.uint8)
    fixed[rng.choice(450, 30, replace=False)] = 1
    args = (uv, rig["camera_ind"], rig["point_ind"])
    out = []
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("SBA_NO_FUSED64", raising=False)
        else:
            monkeypatch.setenv("SBA_NO_FUSED64", "1")
        with _native.Problem(rig["cams0"], rig["pts0"], *args, dtype="f64") as prob:
            prob.set_fixed_points(fixed)
            prob.set_robust_loss("huber", 2.0)
            cams, pts, rep, log = prob.solve_lm(prob.make_opts(ftol=1e-9, mode=_native.MODE_SHARED_INTR))
        out.append((cams, pts, rep, log))
    (ca, pa, ra, la), (cb, pb, rb, lb) = out
    fx = fixed.astype(bool)
    assert np.array_equal(pa[fx], rig["pts0"][fx])
    assert [r.accepted for r in la] == [r.accepted for r in lb]
    assert abs(ra.cost - rb.cost) <= 1e-10 * rb.cost
    assert np.max(np.abs(ca - cb)) <= 1e-6 * np.max(np.abs(cb))
    d = ca[:, 6:9] - rig["cams0"][:, 6:9]                         # f, k1, k2 are tied: every camera gets the same step
    assert np.max(np.abs(d - d[0])) <= 1e-9 * max(1.0, np.max(np.abs(d)))


def test_fused_f64_is_reproducible_bit_for_bit():
    rig = make_rig(16, 3000, seed=4)
    args = (rig["points_2d"], rig["camera_ind"], rig["point_ind"])
    runs = []
    for _ in range(2):
        with _native.Problem(rig["cams0"], rig["pts0"], *args, dtype="f64") as prob:
            cams, pts, rep, log = prob.solve_lm(prob.make_opts(ftol=1e-9))
        runs.append((cams, pts, [r.cost for r in log]))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]) and runs[0][2] == runs[1][2]


def test_fused_f64_full_size_16x50k_properties():
    """The shape the fp64 timing quotes: 16 cameras x 50,000 points, dense (196 points per workgroup: 12 chunks + 4 points)."""
    rig = make_rig(16, 50000, seed=0)
    args = (rig["points_2d"], rig["camera_ind"], rig["point_ind"])
    with _native.Problem(rig["cams0"], rig["pts0"], *args, dtype="f64") as prob:
        r0, c0 = prob.residual()
        cams, pts, rep, log = prob.solve_lm(prob.make_opts(ftol=1e-6))
        r1, c1 = prob.residual(np.hstack((cams.ravel(), pts.ravel())))
    assert rep.status == 2 and rep.cost < 1e-3 * c0 and abs(c1 - rep.cost) <= 1e-12 * c1
    costs = [row.cost for row in log if row.accepted]
    assert all(b <= a for a, b in zip(costs, costs[1:]))
    rms = np.sqrt(np.mean(np.sum(r1.reshape(-1, 2) ** 2, axis=1)))
    assert 0.35 < rms < 0.45                                      # 0.3 px noise per axis


# ---------------------------------------------------------------- 17 .. 23 cameras: k_schur_fused_wide_f64 (compact rows; 19+: two
# workgroups per slice share the tiles)
@pytest.mark.parametrize("C,N,vis", [(17, 333, 1.0), (17, 4099, 1.0), (17, 500, 0.5), (18, 260, 1.0), (18, 700, 0.6), (17, 50, 1.0),
                                     (19, 203, 0.8), (20, 150, 1.0), (20, 2100, 1.0), (21, 97, 0.7), (22, 120, 1.0), (23, 260, 0.6)])
@pytest.mark.parametrize("pw2", [False, True])
def test_wide_f64_builds_the_pair_kernel_system(monkeypatch, C, N, vis, pw2):
    """The reference's own rig shape (17 cameras, example/config.json:24-42) in PySBA's own dtype: the one-launch kernel against
    the path it replaces (SBA_NO_FUSED64=1: k_linearize_* + k_point_factor + the k_schur_sym group pairs); three points per
    producer wave (the default) and two (SBA_WIDE_PW2=1)."""
    if pw2:
        monkeypatch.setenv("SBA_WIDE_PW2", "1")
    rig = make_rig(C, N, seed=41 + C, visibility=vis, min_cams_per_point=2)
    rng = np.random.default_rng(7)
    perm = rng.permutation(rig["camera_ind"].size)
    uv, ci, pi = rig["points_2d"][perm], rig["camera_ind"][perm], rig["point_ind"][perm]
    wts = rng.uniform(0.5, 1.5, ci.size)
    Ea, sa, da = _system(rig, uv, ci, pi, wts, monkeypatch, fused=True)
    Eb, sb, db = _system(rig, uv, ci, pi, wts, monkeypatch, fused=False)
    _compare(Ea, sa, da, Eb, sb, db, 11 * C)


@pytest.mark.parametrize("C,N,vis", [(17, 600, 1.0), (18, 500, 0.55), (20, 400, 0.7), (23, 300, 1.0)])
def test_wide_f64_solve_follows_the_pair_kernel_solve_and_the_oracle(monkeypatch, C, N, vis):
    rig = make_rig(C, N, seed=5 + C, visibility=vis, min_cams_per_point=4)
    args = (rig["points_2d"], rig["camera_ind"], rig["point_ind"])
    logs = []
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("SBA_NO_FUSED64", raising=False)
        else:
            monkeypatch.setenv("SBA_NO_FUSED64", "1")
        with _native.Problem(rig["cams0"], rig["pts0"], *args, dtype="f64") as prob:
            logs.append(prob.solve_lm(prob.make_opts(ftol=1e-10)))
    (ca, pa, ra, la), (cb, pb, rb, lb) = logs
    assert ra.status == rb.status and [r.accepted for r in la] == [r.accepted for r in lb]
    assert abs(ra.cost - rb.cost) <= 1e-11 * rb.cost and np.max(np.abs(ca - cb)) <= 1e-6 * np.max(np.abs(cb))
    ref, _, _ = orc.bundle_adjust(rig["cams0"], rig["pts0"], *args, ftol=1e-10)
    assert ra.cost <= ref.cost * (1 + 1e-9) and ra.cost >= 0.99 * ref.cost
    again, _, _ = orc.bundle_adjust(ca, pa, *args, ftol=1e-10, max_nfev=20)
    assert again.cost >= ra.cost * (1 - 1e-6)


def test_wide_f64_with_huber_loss_and_fixed_points(monkeypatch):
    rig = make_rig(17, 300, seed=4, visibility=0.8)
    uv = rig["points_2d"].copy()
    rng = np.random.default_rng(1)
    bad = rng.choice(uv.shape[0], 30, replace=False)
    uv[bad] += rng.normal(0, 40.0, (30, 2))
    fixed = np.zeros(300, dtype=np.uint8)
    fixed[:6] = 1
    out = []
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("SBA_NO_FUSED64", raising=False)
        else:
            monkeypatch.setenv("SBA_NO_FUSED64", "1")
        with _native.Problem(rig["cams0"], rig["pts0"], uv, rig["camera_ind"], rig["point_ind"], dtype="f64") as prob:
            prob.set_fixed_points(fixed)
            prob.set_robust_loss("huber", 1.0)
            out.append(prob.solve_lm(prob.make_opts(ftol=1e-9)))
    (ca, pa, ra, la), (cb, pb, rb, lb) = out
    assert np.array_equal(pa[:6], rig["pts0"][:6]) and np.array_equal(pb[:6], rig["pts0"][:6])
    assert [r.accepted for r in la] == [r.accepted for r in lb]
    assert abs(ra.cost - rb.cost) <= 1e-10 * rb.cost


@pytest.mark.parametrize("C,N", [(16, 3), (16, 17), (5, 1), (17, 2), (18, 13)])
def test_fused_f64_kernels_on_tiny_rigs(monkeypatch, C, N):
    """Fewer points than one chunk, slices of a single point, most workgroups without work."""
    rig = make_rig(C, N, seed=100 + C + N)
    rng = np.random.default_rng(2)
    wts = rng.uniform(0.5, 1.5, rig["camera_ind"].size)
    args = (rig, rig["points_2d"], rig["camera_ind"], rig["point_ind"], wts, monkeypatch)
    Ea, sa, da = _system(*args, fused=True, lam=1e-2)
    Eb, sb, db = _system(*args, fused=False, lam=1e-2)
    _compare(Ea, sa, da, Eb, sb, db, 11 * C, tol=1e-10)


@pytest.mark.parametrize("seed", range(16))
def test_fused_f64_kernels_on_random_shapes(monkeypatch, seed):
    """Shapes nobody picked by hand: 2 .. 23 cameras, 1 .. 3000 points, visibility 0.4 .. 1, with and without weights."""
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.integers(2, 24))
    N = int(rng.integers(1, 3001))
    vis = float(rng.choice([1.0, rng.uniform(0.4, 1.0)]))
    rig = make_rig(C, N, seed=2000 + seed, visibility=vis, min_cams_per_point=min(2, C))
    ci = rig["camera_ind"]
    wts = rng.uniform(0.5, 1.5, ci.size) if seed % 2 else None
    args = (rig, rig["points_2d"], ci, rig["point_ind"], wts, monkeypatch)
    Ea, sa, da = _system(*args, fused=True, lam=1e-3)
    Eb, sb, db = _system(*args, fused=False, lam=1e-3)
    _compare(Ea, sa, da, Eb, sb, db, 11 * C, tol=1e-10)
