"""Where does an LM iteration's time go at a large camera count?  Phase API with a stream sync (lm_poll) after each phase."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lasercalib_amd import _native
from lasercalib_amd.synth import make_rig
C, N, K, dtype = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
t = time.time(); rig = make_rig(C, N, seed=0); print(f"rig {time.time()-t:.2f}s", flush=True)
t = time.time()
prob = _native.Problem(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], dtype=dtype)
print(f"upload {time.time()-t:.2f}s", flush=True)
import ctypes
lib = _native.load()
hip = ctypes.CDLL(None)   # the HIP runtime _native preloaded (RTLD_GLOBAL)
nE = prob.exchange_size()
E = ctypes.c_void_p(); sc = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(E), nE * 8) == 0 and hip.hipMalloc(ctypes.byref(sc), 64) == 0
opts = prob.make_opts(ftol=0, xtol=0, gtol=0, max_iter=K, always_relinearize=True)
t = time.time(); prob.lm_begin(opts); print(f"lm_begin {time.time()-t:.3f}s", flush=True)
def lap(name, f):
    t = time.time(); f(); prob.lm_poll(); dt = time.time() - t
    print(f"   {name:14s} {dt*1e3:10.3f} ms", flush=True)
for it in range(K):
    print(f"iteration {it}", flush=True)
    lap("linearize", prob.lm_linearize)
    lap("form_reduced", lambda: prob.lm_form_reduced(E.value))
    lap("solve_trial", lambda: prob.lm_solve_trial(E.value, sc.value))
    lap("decide", lambda: prob.lm_decide_async(None, 1))
cams, pts, rep = prob.lm_finish()
print("cost", rep.cost, "iters", rep.iterations, "status", rep.status)
print([ (r.iteration, r.accepted, r.cost) for r in prob.iteration_log()])
prob.close()
