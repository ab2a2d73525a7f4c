"""Per-iteration costs of a multi-group fp32 solve: diagonal pairs on the bf16 pipe vs f32-input MFMAs vs the fp64 engine."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lasercalib_amd import _native
from lasercalib_amd.synth import make_rig
C, N, vis = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
tang = len(sys.argv) > 4 and sys.argv[4] == 'tangential'
rig = make_rig(C, N, seed=33, visibility=vis, tangential=tang)
a = (rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"])
kw = dict(ftol=0.0, xtol=0.0, gtol=0.0, max_iter=8, always_relinearize=True)
def run(dtype):
    with _native.Problem(*a, dtype=dtype) as prob:
        return prob.solve_lm(prob.make_opts(**kw))[3]
lb = run("f32")
os.environ["SBA_NO_BF3_PAIRS"] = "1"
lf = run("f32")
del os.environ["SBA_NO_BF3_PAIRS"]
ld = run("f64")
for rb, rf, rd in zip(lb, lf, ld):
    print(f"it {rb.iteration}: bf3 acc {rb.accepted} cost {rb.cost:.6f} rho {rb.rho:+.3f} lam {rb.lambda_:.2e} | f32 acc {rf.accepted} cost {rf.cost:.6f} rho {rf.rho:+.3f} lam {rf.lambda_:.2e} | f64 acc {rd.accepted} cost {rd.cost:.6f} rho {rd.rho:+.3f} lam {rd.lambda_:.2e}")
