"""Exploration: large camera counts vs numpy model and scipy oracle (prints numbers used to set test tolerances)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lasercalib_amd import _native
from lasercalib_amd.synth import make_rig
from oracle import lm_schur_model as model
from oracle import sba_oracle as orc
cases = [(47, 60, 0.3), (64, 80, 0.25), (128, 100, 0.15)]
if len(sys.argv) > 1:
    cases = [tuple(float(x) if '.' in x else int(x) for x in a.split(',')) for a in sys.argv[1:]]
for C, N, vis in cases:
    rig = make_rig(C, N, seed=8, visibility=vis)
    M = rig["camera_ind"].size
    print(f"== C={C} N={N} vis={vis} M={M}", flush=True)
    for dtype in ("f64", "f32"):
        t = time.time()
        with _native.Problem(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], dtype=dtype) as prob:
            cams, pts, rep, log = prob.solve_lm(prob.make_opts(ftol=1e-6))
        print(f"  {dtype} ftol 1e-6: status {rep.status} iters {rep.iterations} nfev {rep.nfev} cost {rep.cost:.9f} wall {time.time()-t:.2f}s", flush=True)
        print("     costs:", [f"{r.cost:.6f}{'' if r.accepted else '*'}" for r in log][:12])
    t = time.time()
    eng = model.ModelEngine(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"])
    out = model.run_lm_single(eng, ftol=1e-6)
    print(f"  model: iters {out['iterations']} nfev {out['nfev']} status {out['status']} cost {out['cost']:.9f}  ({time.time()-t:.1f}s)", flush=True)
    for ftol in (1e-4,):
        t = time.time()
        ref, _, _ = orc.bundle_adjust(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], ftol=ftol)
        for dtype in ("f64", "f32"):
            with _native.Problem(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], dtype=dtype) as prob:
                cams, pts, rep, log = prob.solve_lm(prob.make_opts(ftol=ftol))
            rms = orc.rms_reprojection(cams, pts, rig["points_2d"], rig["camera_ind"], rig["point_ind"])
            print(f"  ftol {ftol}: scipy cost {ref.cost:.9f} nfev {ref.nfev} ({time.time()-t:.1f}s) | {dtype} cost {rep.cost:.9f} rel {(rep.cost-ref.cost)/ref.cost:+.2e} rms {rms:.6f} status {rep.status} iters {rep.iterations}", flush=True)
