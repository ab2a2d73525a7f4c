import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lasercalib_amd import _native
from lasercalib_amd.synth import make_rig
for C, N in ((16, 500), (16, 512), (16, 1024), (16, 4096), (16, 8192)):
    rig = make_rig(C, N, seed=0)
    a = (rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"])
    out = []
    for env in ({}, {"SBA_DECIDE_KERNEL": "1"}, {"SBA_FUSED_MFMA": "f32"}):
        os.environ.update(env)
        with _native.Problem(*a, dtype="f32") as prob:
            cams, pts, rep, log = prob.solve_lm(prob.make_opts(ftol=0.0, xtol=0.0, gtol=0.0, max_iter=3, always_relinearize=True))
            out.append((rep.cost, rep.accepted, [(r.accepted, round(r.cost, 3), round(r.rho, 3)) for r in log]))
        for k in env: del os.environ[k]
    print(C, N, out)
