"""Debug probe: compare the device exchange buffer [S|rhs|diagU|gc|cost] with the numpy model."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lasercalib_amd import _native
from lasercalib_amd.synth import make_rig
from oracle import lm_schur_model as model

def run(C, N, vis, dtype="f64"):
    rig = make_rig(C, N, seed=8, visibility=vis)
    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream().cuda_stream
    prob = _native.Problem(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], dtype=dtype, stream=stream)
    opts = prob.make_opts(ftol=1e-6)
    prob.lm_begin(opts)
    prob.lm_linearize()
    E = torch.zeros(prob.exchange_size(), dtype=torch.float64, device="cuda")
    prob.lm_form_reduced(E.data_ptr())
    torch.cuda.synchronize()
    Ed = E.cpu().numpy().copy()
    eng = model.ModelEngine(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"])
    eng.linearize()
    Em = eng.form_reduced()
    n = 11 * C
    Sd, Sm = Ed[:n*n].reshape(n, n), Em[:n*n].reshape(n, n)
    print(f"C={C} N={N} vis={vis} {dtype}: |S| max {abs(Sm).max():.3e}  S err {abs(Sd-Sm).max():.3e}  asym {abs(Sd-Sd.T).max():.3e}")
    err = abs(Sd - Sm)
    blk = err.reshape(C, 11, C, 11).max(axis=(1, 3))
    if blk.max() > 1e-6 * abs(Sm).max():
        np.set_printoptions(linewidth=250, precision=1)
        print("block error map (cams x cams):\n", (blk / abs(Sm).max()))
    for name, a, b in (("rhs", n*n, n*n+n), ("diagU", n*n+n, n*n+2*n), ("gc", n*n+2*n, n*n+3*n), ("cost", n*n+3*n, n*n+3*n+1)):
        print(f"   {name}: max {abs(Em[a:b]).max():.3e} err {abs(Ed[a:b]-Em[a:b]).max():.3e}")
    sc = torch.zeros(8, dtype=torch.float64, device="cuda")
    prob.lm_solve_trial(E.data_ptr(), sc.data_ptr())
    torch.cuda.synchronize()
    scm = eng.solve_trial(Em)
    print("   scalars dev", sc.cpu().numpy()[:6])
    print("   scalars mod", scm[:6])
    prob.close()

if __name__ == "__main__":
    run(6, 100, 0.7)
    run(16, 200, 1.0)
    run(20, 200, 0.7)
    run(40, 100, 0.5)
    run(20, 200, 0.7, "f32")
    run(16, 200, 1.0, "f32")     # dense, one group: fused linearise + Schur kernel
    run(6, 101, 1.0, "f32")
    run(16, 37, 1.0, "f32")
    run(16, 300, 0.5, "f32")     # sparse, one group: fused kernel through the visibility mask
    run(12, 150, 0.6, "f32")
    run(7, 90, 0.4, "f32")
