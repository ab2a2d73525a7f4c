"""Where the wall time of one drop-in solve goes (warm process): handle creation + upload, solve, residual read-back."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lasercalib_amd import _native
from lasercalib_amd.synth import make_rig
C, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 50000)
rig = make_rig(C, N, seed=0)
for dtype in ("f64", "f32", "f32"):
    t0 = time.perf_counter()
    prob = _native.Problem(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], dtype=dtype)
    t1 = time.perf_counter()
    cams, pts, rep, log = prob.solve_lm(prob.make_opts(ftol=1e-4))
    t2 = time.perf_counter()
    f, _ = prob.residual()
    t3 = time.perf_counter()
    prob.close()
    t4 = time.perf_counter()
    print(f"{dtype}: create+upload {1e3*(t1-t0):6.2f} ms  solve {1e3*(t2-t1):6.2f} ms (device {rep.seconds_device*1e3:.2f})  residual {1e3*(t3-t2):6.2f} ms  close {1e3*(t4-t3):6.2f} ms")
