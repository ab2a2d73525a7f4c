import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lasercalib_amd import _native
from oracle import sba_oracle as orc
g = np.load("tests/golden/f1_project.npz")
uv = _native.project_rows(g["points"], g["cam_rows"], dtype="f32")
rot = _native.rotate_rows(g["points"], g["cam_rows"][:, :3], dtype="f32")
pc = g["rotated"] + g["cam_rows"][:, 3:6]
cond = np.linalg.norm(pc, axis=1) / np.abs(pc[:, 2])
err = np.max(np.abs(uv - g["projected"]), axis=1)
eps = np.finfo(np.float32).eps
bound = eps * (np.max(np.abs(g["projected"]), axis=1) + 2400.0 * cond * cond)
for name, sl in (("theta=0", slice(0, 32)), ("1e-9", slice(32, 64)), ("1e-4", slice(64, 96)), ("pi", slice(96, 128)), ("near", slice(128, 160)), ("ordinary", slice(160, 4096))):
    r = err[sl] / bound[sl]
    print(f"{name:10s} max err {err[sl].max():.3e} px  max cond {cond[sl].max():.2e}  max err/bound {r.max():.2f}  max|uv| {np.abs(g['projected'][sl]).max():.3e}")
print("rot err", np.max(np.abs(rot - g["rotated"])))
