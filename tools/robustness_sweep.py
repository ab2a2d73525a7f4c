"""Device LM vs the scipy oracle from increasingly bad starts (seeds x perturbation scale, with outliers):
final cost ratio, status, iteration counts.  Run on the GPU box: python tools/robustness_sweep.py [CAMS [POINTS [VISIBILITY]]]
(default 6 x 250 at 0.8: the one-group kernels; 17 x 300 at 0.6: the wide kernel + the left-looking Cholesky)"""
import sys, os, io, contextlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lasercalib_amd import _native
from lasercalib_amd.synth import make_rig
from oracle import sba_oracle as orc

CAMS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
POINTS = int(sys.argv[2]) if len(sys.argv) > 2 else 250
VIS = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
rows = []
for scale in (1.0, 4.0, 12.0):
    for seed in range(4):
        rig = make_rig(CAMS, POINTS, seed=100 + seed, visibility=VIS, min_cams_per_point=min(4, CAMS))
        rng = np.random.default_rng(seed)
        cams0 = rig["cams_true"] + (rig["cams0"] - rig["cams_true"]) * scale
        pts0 = rig["pts_true"] + (rig["pts0"] - rig["pts_true"]) * scale
        uv = rig["points_2d"].copy()
        bad = rng.random(uv.shape[0]) < 0.01            # 1 % gross outliers (50 px)
        uv[bad] += rng.normal(0, 50.0, (int(bad.sum()), 2))
        for dtype in ("f64", "f32"):
            prob = _native.Problem(cams0, pts0, uv, rig["camera_ind"], rig["point_ind"], dtype=dtype)
            _, _, rep, _ = prob.solve_lm(prob.make_opts(ftol=1e-6, max_nfev=2000))
            prob.close()
            rows.append((scale, seed, dtype, rep.cost, rep.status, rep.iterations))
        t = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            ref, _, _ = orc.bundle_adjust(cams0, pts0, uv, rig["camera_ind"], rig["point_ind"], ftol=1e-6)
        rows.append((scale, seed, "scipy", ref.cost, ref.status, ref.nfev))
        print(f"scale {scale:4.1f} seed {seed}: scipy cost {ref.cost:12.4f} (status {ref.status}, nfev {ref.nfev}, {time.time()-t:.1f}s)  "
              f"f64 {rows[-3][3]:12.4f} (st {rows[-3][4]}, it {rows[-3][5]})  f32 {rows[-2][3]:12.4f} (st {rows[-2][4]}, it {rows[-2][5]})", flush=True)
