import sys, os
sys.path.insert(0, '.')
import numpy as np
from lasercalib_amd.pySBA import PySBA
from oracle import sba_oracle as orc
g = np.load('tests/golden/f5_variants.npz')
a = (g["cams0"], g["pts0"], g["uv"], g["ci"], g["pi"])
sba = PySBA(g["cams0"].copy(), g["pts0"].copy(), g["uv"], g["ci"], g["pi"])
res = sba.bundle_adjustment_camonly()
print("camonly: cost", res.cost, "ref", float(g["camonly_cost"]), "status", res.status, "nfev", res.nfev)
f = orc.fun_camonly(res.x, 4, 300, g["ci"], g["pi"], g["uv"], 1.0, g["pts0"])
print("  fun check", np.max(np.abs(f - res.fun)), 0.5 * f @ f)
sba = PySBA(g["cams0"].copy(), g["pts0"].copy(), g["uv"], g["ci"], g["pi"])
res = sba.bundleAdjust_transform_points_3d()
print("transform: cost", res.cost, "ref", float(g["transform_cost"]), "status", res.status, "nfev", res.nfev)
print("  theta", res.x.reshape(3, 4)); print("  ref theta", g["transform_x"].reshape(3, 4))
f = orc.fun_transform_points_3d(res.x, 4, 300, g["cams0"], g["ci"], g["pi"], g["uv"], 1.0, g["pts0"])
print("  fun check", np.max(np.abs(f - res.fun)), 0.5 * f @ f, "pts diff vs ref", np.max(np.abs(sba.points3D - g["transform_pts"])))
