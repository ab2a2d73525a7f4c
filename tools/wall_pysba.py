"""Wall-clock of the drop-in call itself: PySBA(...).bundleAdjust(1e-4) at BASELINE config 3, second call (warm process)."""
import sys, os, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lasercalib_amd.pySBA import PySBA
from lasercalib_amd.synth import make_rig
C, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 50000)
rig = make_rig(C, N, seed=0)
for dtype in ("f64", "f32"):
    os.environ["LASERCALIB_SBA_DTYPE"] = dtype
    for rep in range(3):
        sba = PySBA(rig["cams0"].copy(), rig["pts0"].copy(), rig["points_2d"], rig["camera_ind"], rig["point_ind"])
        t = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            res = sba.bundleAdjust(1e-4)
        dt = time.perf_counter() - t
        print(f"{dtype} call {rep}: bundleAdjust wall {dt*1e3:8.1f} ms  status {res.status} nfev {res.nfev} cost {res.cost:.4f}")
