"""Quick per-kernel timing at BASELINE config 3 (16 cams x 50k points)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lasercalib_amd import _native
from lasercalib_amd.synth import make_rig
C, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 50000)
VIS = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rig = make_rig(C, N, seed=0, visibility=VIS)
for dtype in ("f64", "f32"):
    t = time.time()
    prob = _native.Problem(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], dtype=dtype)
    t_up = time.time() - t
    M = rig["camera_ind"].size
    print(f"== {dtype}: C={C} N={N} M={M} upload+layout {t_up*1e3:.1f} ms")
    for k in ("residual", "resjac", "linearize_points", "linearize_cams", "schur", "backsub"):
        us = prob.time_kernel(k, 20)
        print(f"   {k:18s} {us:9.1f} us   {M/us:8.1f} Mobs/s")
    t = time.time()
    cams, pts, rep, log = prob.solve_lm(prob.make_opts(ftol=1e-4))
    wall = time.time() - t
    print(f"   solve ftol=1e-4: status {rep.status} iters {rep.iterations} nfev {rep.nfev} cost {rep.cost:.6f} device {rep.seconds_device*1e3:.2f} ms wall {wall*1e3:.2f} ms"
          f" -> {rep.iterations/rep.seconds_device:.1f} LM it/s")
    cams, pts, rep, log = prob.solve_lm(prob.make_opts(ftol=0, xtol=0, gtol=0, max_iter=30, always_relinearize=True))
    print(f"   30 forced iterations: device {rep.seconds_device*1e3:.2f} ms -> {rep.seconds_device/30*1e6:.1f} us/iter, {30/rep.seconds_device:.1f} LM it/s")
    prob.close()
