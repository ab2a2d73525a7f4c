"""Per-iteration costs of the fused bf16x3 kernel vs the f32-MFMA fused kernel vs fp64 on a small one-group rig."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lasercalib_amd import _native
from lasercalib_amd.synth import make_rig
C, N = int(sys.argv[1]), int(sys.argv[2])
rig = make_rig(C, N, seed=0)
a = (rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"])
kw = dict(ftol=0.0, xtol=0.0, gtol=0.0, max_iter=5, always_relinearize=True)
def run(dtype):
    with _native.Problem(*a, dtype=dtype) as prob:
        return prob.solve_lm(prob.make_opts(**kw))[3]
lb = run("f32")
os.environ["SBA_FUSED_MFMA"] = "f32"
lf = run("f32")
del os.environ["SBA_FUSED_MFMA"]
ld = run("f64")
for rb, rf, rd in zip(lb, lf, ld):
    print(f"it {rb.iteration}: bf3 acc {rb.accepted} cost {rb.cost:.6f} step {rb.step_norm:.4e} | f32mfma acc {rf.accepted} cost {rf.cost:.6f} step {rf.step_norm:.4e} | f64 acc {rd.accepted} cost {rd.cost:.6f} step {rd.step_norm:.4e}")
