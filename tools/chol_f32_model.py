"""CPU experiment behind the fp32-engine Cholesky decision (round 4): what does an f32 factorisation of the reduced camera system
do to the LM loop?  The numpy model of the device algorithm (oracle/lm_schur_model.py) is run with three reduced-system solvers:
  f64      np.linalg.cholesky in float64 on the exact S                       (the fp64 engine)
  f64n     the same on S with 1e-7 relative noise per entry                   (the fp32 engine today: S from f32 products)
  f32      S (with that noise) rounded to float32, LAPACK spotrf / spotrs     (an f32-lane Cholesky kernel)
  f32fb    f32, falling back to f64n when spotrf fails or min_i L_ii^2 / A_ii < TAU
Prints iterations, rejected steps, final cost relative to the f64 run and the lambda range visited.
usage: python tools/chol_f32_model.py [ftol ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg as sl
from lasercalib_amd.synth import make_rig
from oracle import lm_schur_model as lm

TAU = float(os.environ.get("TAU", "1e-4"))


class Eng(lm.ModelEngine):
    solver = "f64"
    stats = None

    def solve_trial(self, E):
        n = self.n
        if self.solver != "f64":
            rng = np.random.default_rng(12345 + self.nfev)
            S = E[:n * n].reshape(n, n)
            noise = 1e-7 * rng.standard_normal((n, n))
            noise = (noise + noise.T) / 2
            E = E.copy()
            E[:n * n] = (S * (1 + noise)).ravel()
        if self.solver in ("f32", "f32fb"):
            orig = np.linalg.cholesky
            osolve = np.linalg.solve
            st = self.stats
            fb = self.solver == "f32fb"

            def chol32(A):
                A32 = A.astype(np.float32)
                try:
                    L = sl.cholesky(A32, lower=True, check_finite=False)
                    ok = np.all(np.isfinite(L))
                except sl.LinAlgError:
                    ok = False
                if ok:
                    ratio = float(np.min(np.diag(L).astype(np.float64) ** 2 / np.diag(A)))
                    st["minratio"].append(ratio)
                    if fb and ratio < TAU:
                        ok = False
                if not ok:
                    st["fail32"] += 1
                    if fb:
                        return orig(A)
                    raise np.linalg.LinAlgError("f32")
                st["ok32"] += 1
                return L          # float32: the two triangular solves below run in float32 as well

            def solve32(L, b):
                if L.dtype == np.float32:
                    return sl.solve_triangular(L, b.astype(np.float32), lower=(np.abs(np.triu(L, 1)).sum() == 0), check_finite=False)
                return osolve(L, b)
            np.linalg.cholesky, np.linalg.solve = chol32, solve32
            try:
                out = super().solve_trial(E)
            finally:
                np.linalg.cholesky, np.linalg.solve = orig, osolve
            return out
        return super().solve_trial(E)


def run(rig, solver, ftol, max_iter=200):
    e = Eng(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"])
    e.solver = solver
    e.stats = dict(fail32=0, ok32=0, minratio=[])
    lams, rej = [], 0
    e.begin(ftol, 1e-8, 1e-8, None)
    status, it, need = None, 0, True
    while status is None and it < max_iter:
        if need:
            e.linearize()
        E = e.form_reduced()
        lams.append(e.lam)
        sc = e.solve_trial(E)
        status, need = e.decide(sc[None, :], 1)
        rej += 0 if e.accepted else 1
        it += 1
    cost = 0.5 * float(np.sum(e.res ** 2))
    return dict(it=it, rej=rej, cost=cost, status=status, lam_min=min(lams), lam_max=max(lams), **{k: v for k, v in e.stats.items() if k != "minratio"},
                ratio_min=(min(e.stats["minratio"]) if e.stats["minratio"] else None))


if __name__ == "__main__":
    ftols = [float(a) for a in sys.argv[1:]] or [1e-4, 1e-8]
    rigs = {"16x2000 dense": make_rig(16, 2000, seed=0), "8x2000 dense": make_rig(8, 2000, seed=0),
            "6x600 vis .6": make_rig(6, 600, seed=0, visibility=0.6), "16x1500 vis .5": make_rig(16, 1500, seed=2, visibility=0.5, min_cams_per_point=4)}
    for name, rig in rigs.items():
        for ftol in ftols:
            base = None
            for solver in ("f64", "f64n", "f32", "f32fb"):
                r = run(rig, solver, ftol)
                if base is None:
                    base = r["cost"]
                print(f"{name:16s} ftol {ftol:7.0e} {solver:6s} it {r['it']:3d} rej {r['rej']:3d} status {r['status']} cost/f64-1 {r['cost'] / base - 1:+.2e} "
                      f"lam [{r['lam_min']:.1e}, {r['lam_max']:.1e}] f32 ok/fail {r['ok32']}/{r['fail32']} min L_ii^2/A_ii {r['ratio_min']}", flush=True)
