"""Exploration for test tolerances (prints measured values)."""
import sys, os, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lasercalib_amd import _native
from lasercalib_amd.synth import make_rig
from oracle import sba_oracle as orc
from oracle import lm_schur_model as model
G = np.load("tests/golden/f4_solves.npz")
def quiet(f, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return f(*a, **k)
# 1. f32 17-camera test
rig = make_rig(17, 300, seed=9, visibility=0.7)
with _native.Problem(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], dtype="f32") as prob:
    cams, pts, rep, _ = prob.solve_lm(prob.make_opts(ftol=1e-4))
with _native.Problem(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], dtype="f64") as prob:
    cams64, pts64, rep64, _ = prob.solve_lm(prob.make_opts(ftol=1e-4))
ref, _, _ = quiet(orc.bundle_adjust, rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], ftol=1e-4)
print(f"17x300 f32: cost {rep.cost:.6f} f64 {rep64.cost:.6f} scipy {ref.cost:.6f}  rel(f32-scipy) {(rep.cost-ref.cost)/ref.cost:+.2e} rel(f32-f64) {(rep.cost-rep64.cost)/rep64.cost:+.2e} iters {rep.iterations}/{rep64.iterations}")
# 2. smoke f32 leg
rig = make_rig(6, 400, seed=2)
with _native.Problem(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], dtype="f32") as prob:
    _, _, rep, _ = prob.solve_lm(prob.make_opts(ftol=1e-4))
ref, _, _ = quiet(orc.bundle_adjust, rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], ftol=1e-4)
print(f"smoke 6x400 f32: cost {rep.cost:.6f} scipy {ref.cost:.6f} rel {(rep.cost-ref.cost)/ref.cost:+.2e}")
# 3. f32 project golden: which rows exceed
g1 = np.load("tests/golden/f1_project.npz")
print("f1 keys", list(g1.keys()))
# 4. status semantics vs model
p = {k: G[f"sparse_{k}"] for k in ("cams0", "pts0", "uv", "ci", "pi")}
def dev(dtype="f64", **kw):
    with _native.Problem(p["cams0"], p["pts0"], p["uv"], p["ci"], p["pi"], dtype=dtype) as prob:
        return prob.solve_lm(prob.make_opts(**kw))
def mod(**kw):
    eng = model.ModelEngine(p["cams0"], p["pts0"], p["uv"], p["ci"], p["pi"])
    return model.run_lm_single(eng, **kw)
for name, kw in (("max_nfev=3", dict(ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=3)), ("gtol=1e12", dict(ftol=1e-8, xtol=1e-8, gtol=1e12)),
                 ("xtol=1e-4", dict(ftol=1e-15, xtol=1e-4, gtol=1e-15)), ("ftol=1e-2,xtol=1e-2", dict(ftol=1e-2, xtol=1e-2, gtol=1e-15)),
                 ("xtol=1e-6", dict(ftol=1e-15, xtol=1e-6, gtol=1e-15))):
    c, pp, rep, log = dev(**kw)
    out = mod(**kw)
    print(f"{name:22s} device status {rep.status} iters {rep.iterations} nfev {rep.nfev} cost {rep.cost:.9f} | model status {out['status']} iters {out['iterations']} nfev {out['nfev']} cost {out['cost']:.9f} | x==x0 {np.array_equal(c, p['cams0'])}")
# 5. robustness scale 12 seed 0
rig = make_rig(6, 250, seed=100, visibility=0.8)
rng = np.random.default_rng(0)
cams0 = rig["cams_true"] + (rig["cams0"] - rig["cams_true"]) * 12.0
pts0 = rig["pts_true"] + (rig["pts0"] - rig["pts_true"]) * 12.0
uv = rig["points_2d"].copy()
bad = rng.random(uv.shape[0]) < 0.01
uv[bad] += rng.normal(0, 50.0, (int(bad.sum()), 2))
ref, _, _ = quiet(orc.bundle_adjust, cams0, pts0, uv, rig["camera_ind"], rig["point_ind"], ftol=1e-6)
print(f"scale12 seed0: scipy ftol 1e-6 cost {ref.cost:.4f} nfev {ref.nfev}")
for ftol in (1e-6, 1e-8, 1e-10):
    for dtype in ("f64", "f32"):
        with _native.Problem(cams0, pts0, uv, rig["camera_ind"], rig["point_ind"], dtype=dtype) as prob:
            c, pp, rep, log = prob.solve_lm(prob.make_opts(ftol=ftol, max_nfev=5000))
        print(f"   {dtype} ftol {ftol:g}: cost {rep.cost:.4f} status {rep.status} iters {rep.iterations} rel {(rep.cost-ref.cost)/ref.cost:+.2e}")
        if ftol == 1e-6 and dtype == "f64":
            again, _, _ = quiet(orc.bundle_adjust, c, pp, uv, rig["camera_ind"], rig["point_ind"], ftol=1e-6, max_nfev=50)
            print(f"      scipy restarted from the device's ftol=1e-6 point: cost {again.cost:.4f} nfev {again.nfev}")
            acc = [r for r in log if r.accepted]
            print("      last accepted reductions:", [f"{r.cost_reduction:.3e}" for r in acc[-5:]], "rho", [f"{r.rho:.2f}" for r in acc[-5:]])
# 6. config2 parameter difference
from scipy.optimize import least_squares
from lasercalib_amd.pySBA import assemble_jacobian
rig = make_rig(8, 5000, seed=0)
ci, pi, uvv = rig["camera_ind"], rig["point_ind"], rig["points_2d"]
x0 = np.hstack((rig["cams0"].ravel(), rig["pts0"].ravel()))
with _native.Problem(rig["cams0"], rig["pts0"], uvv, ci, pi) as prob:
    fun = lambda x: prob.residual(x)[0]
    def jac(x):
        _, Jc, Jp = prob.residual_jacobian(x)
        return assemble_jacobian(Jc, Jp, ci, pi, 8, 5000)
    res = least_squares(fun, x0, jac=jac, x_scale="jac", ftol=1e-4, method="trf")
ref, _, _ = quiet(orc.bundle_adjust, rig["cams0"], rig["pts0"], uvv, ci, pi, ftol=1e-4)
d = np.abs(res.x - ref.x)
print(f"config2: cost rel {(res.cost-ref.cost)/ref.cost:+.2e} max|dx| cams {d[:88].max():.4e} (by column {d[:88].reshape(8,11).max(axis=0)}) points {d[88:].max():.4e}")
