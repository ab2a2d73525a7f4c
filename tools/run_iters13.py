"""Run K forced LM iterations on a 13-parameter (radial + tangential) rig."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lasercalib_amd import _native
from lasercalib_amd.synth import make_rig
C, N, K, dtype = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
rig = make_rig(C, N, seed=0, tangential=True)
prob = _native.Problem(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], dtype=dtype)
cams, pts, rep, log = prob.solve_lm(prob.make_opts(ftol=0, xtol=0, gtol=0, max_iter=K, always_relinearize=True))
print(f"{dtype} 13-param C={C} N={N}: {K} iterations, device {rep.seconds_device*1e3:.2f} ms -> {rep.seconds_device/K*1e6:.1f} us/iter, cost {log[0].cost:.3f} -> {log[-1].cost:.3f}")
prob.close()
