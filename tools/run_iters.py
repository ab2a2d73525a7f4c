"""Run K forced LM iterations at a given size (for rocprofv3).

usage: run_iters.py CAMS POINTS ITERS DTYPE [VISIBILITY [MIN_VIEWS [tangential]]]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lasercalib_amd import _native
if os.environ.get('SBA_LIB_AB'):      # A/B runs: another build of the library
    _native.LIB_PATH = os.environ['SBA_LIB_AB']
from lasercalib_amd.synth import make_rig
C, N, K, dtype = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
vis = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
minv = int(sys.argv[6]) if len(sys.argv) > 6 else 2
tang = len(sys.argv) > 7 and sys.argv[7] == "tangential"
rig = make_rig(C, N, seed=0, visibility=vis, min_cams_per_point=minv, tangential=tang)
prob = _native.Problem(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], dtype=dtype)
cams, pts, rep, log = prob.solve_lm(prob.make_opts(ftol=0, xtol=0, gtol=0, max_iter=K, always_relinearize=True))
M = len(rig["camera_ind"])
print(f"{dtype} C={C} N={N} M={M} vis={vis}{' 13p' if tang else ''}: {K} iterations, device {rep.seconds_device*1e3:.2f} ms -> {rep.seconds_device/K*1e6:.1f} us/iter  cost {rep.cost:.6g}", flush=True)
prob.close()
