import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lasercalib_amd import _native
from oracle import sba_oracle as orc
g = np.load("tests/golden/f9_tight.npz")
for tag in ("var", "sparse", "cfg1"):
    a = (g[tag + "_cams0"], g[tag + "_pts0"], g[tag + "_uv"], g[tag + "_ci"], g[tag + "_pi"])
    best = float(g[tag + "_cost"])
    for ftol in (1e-4, 1e-6, 1e-9):
        with _native.Problem(*a, dtype="f32") as prob:
            cams, pts, rep, log = prob.solve_lm(prob.make_opts(ftol=ftol))
        c = 0.5 * np.sum(orc.fun(np.hstack((cams.ravel(), pts.ravel())), cams.shape[0], pts.shape[0], a[3], a[4], a[2], 1.0) ** 2)
        mode = os.environ.get("SBA_CHOL_F32", "1")
        lam = rep.lambda_
        print(f"SBA_CHOL_F32={mode} {tag} ftol {ftol:g}: cost {c:.6f} rel to minimum {(c - best) / best:+.2e} status {rep.status} iterations {rep.iterations} accepted {rep.accepted} retries {rep.reserved} lambda {lam:.1e}")
