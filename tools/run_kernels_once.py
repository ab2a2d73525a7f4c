"""Launch every streaming kernel a few times at 16x50k (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lasercalib_amd import _native
from lasercalib_amd.synth import make_rig
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
rig = make_rig(16, 50000, seed=0)
prob = _native.Problem(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], dtype=dtype)
for k in ("residual", "resjac"):
    prob.time_kernel(k, 3)
prob.solve_lm(prob.make_opts(ftol=0, xtol=0, gtol=0, max_iter=6, always_relinearize=True))
prob.close()
