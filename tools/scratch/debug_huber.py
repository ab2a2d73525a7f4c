import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lasercalib_amd import _native
from lasercalib_amd.synth import make_rig
from oracle import sba_oracle as orc
C, N, vis = 6, 300, 0.8
rig = make_rig(C, N, seed=41, visibility=vis)
rng = np.random.default_rng(2)
uv = rig["points_2d"].copy()
bad = rng.random(uv.shape[0]) < 0.03
uv[bad] += rng.normal(0, 40.0, (int(bad.sum()), 2))
args = (uv, rig["camera_ind"], rig["point_ind"])
ref, cams_ref, pts_ref = orc.bundle_adjust_ext(rig["cams0"], rig["pts0"], *args, loss="huber", f_scale=1.0, ftol=1e-8, verbose=1)
print("scipy", ref.cost, ref.nfev, ref.status)
for dtype in ("f64",):
    with _native.Problem(rig["cams0"], rig["pts0"], *args, dtype=dtype) as prob:
        prob.set_robust_loss("huber", 1.0)
        cams, pts, rep, log = prob.solve_lm(prob.make_opts(ftol=1e-8, max_nfev=60))
    print(dtype, rep.cost, rep.status, rep.iterations)
    for r in log[:60]:
        print(f"  it {r.iteration:3d} acc {r.accepted} cost {r.cost:.8f} red {r.cost_reduction:+.3e} rho {r.rho:+.3f} lam {r.lambda_:.2e} step {r.step_norm:.3e} opt {r.optimality:.3e}")
