"""One-rank RCCL loop vs plain loop, iteration by iteration (multi-group fp32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lasercalib_amd import _native
from lasercalib_amd.synth import make_rig
C, N, vis = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
rig = make_rig(C, N, seed=13, visibility=vis)
a = (rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"])
res = []
for with_comm in (False, True):
    with _native.Problem(*a, dtype="f32") as prob:
        if with_comm:
            prob.comm_init(_native.comm_unique_id(), 0, 1)
        res.append(prob.solve_lm(prob.make_opts(ftol=0.0, xtol=0.0, gtol=0.0, max_iter=10, always_relinearize=True)))
(c0, p0, r0, l0), (c1, p1, r1, l1) = res
for x, y in zip(l0, l1):
    print(f"it {x.iteration}: plain acc {x.accepted} cost {x.cost:.9f} red {x.cost_reduction:+.3e} step {x.step_norm:.6e} | rccl acc {y.accepted} cost {y.cost:.9f} red {y.cost_reduction:+.3e} step {y.step_norm:.6e}")
print("max |dcams|", np.max(np.abs(c0 - c1)), "max |dpts|", np.max(np.abs(p0 - p1)))
