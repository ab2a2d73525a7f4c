"""Known-answer vectors for the 13-parameter camera model -- test infrastructure.

The reference has no tangential term, so nothing it produces can pin p1, p2 != 0 (oracle/sba_oracle_tangential.py).  What
CAN be pinned is the published formula itself, OpenCV's pinhole + radial + tangential model (the convention of the
calibration YAMLs the reference reads and writes, lasercalib/convert_params.py:66-87,105-123):

    x = X/Z, y = Y/Z, r2 = x^2 + y^2
    x' = x (1 + k1 r2 + k2 r2^2) + 2 p1 x y + p2 (r2 + 2 x^2)
    y' = y (1 + k1 r2 + k2 r2^2) + p1 (r2 + 2 y^2) + 2 p2 x y
    u = f x' + cx,  v = f y' + cy

This script evaluates it in EXACT rational arithmetic (fractions.Fraction; no code shared with the oracle or the product) for
camera poses whose rotation is exact -- the identity and quarter turns about the axes, given as rotation vectors -- and writes
tests/golden/f8_opencv_tangential.json.  Each row: point, 13-parameter camera row [rvec, t, f, k1, k2, p1, p2, cx, cy], expected (u, v).
"""
import json
import math
import os
from fractions import Fraction as F

ROT = {                       # rotation vector -> exact rotation matrix (rows)
    "identity": ((0.0, 0.0, 0.0), ((1, 0, 0), (0, 1, 0), (0, 0, 1))),
    "quarter_z": ((0.0, 0.0, math.pi / 2), ((0, -1, 0), (1, 0, 0), (0, 0, 1))),
    "quarter_x": ((math.pi / 2, 0.0, 0.0), ((1, 0, 0), (0, 0, -1), (0, 1, 0))),
    "half_y": ((0.0, math.pi, 0.0), ((-1, 0, 0), (0, 1, 0), (0, 0, -1))),
}
CASES = [
    # rotation, point (X, Y, Z), t, f, k1, k2, p1, p2, cx, cy   (decimal strings: exact rationals)
    ("identity", ("100", "-50", "0"), ("0", "0", "1000"), "2400", "-0.05", "0.01", "0.001", "-0.002", "1604", "1100"),
    ("identity", ("300", "200", "106"), ("-20", "35", "1500"), "2380.5", "0.02", "-0.015", "-0.0015", "0.0025", "1600.25", "1098.75"),
    ("identity", ("-640", "512", "0"), ("0", "0", "1280"), "2000", "0.1", "0.05", "0.003", "0.004", "1500", "1000"),
    ("quarter_z", ("120", "80", "10"), ("5", "-5", "900"), "2410", "-0.03", "0.002", "0.0007", "-0.0011", "1610", "1090"),
    ("quarter_x", ("50", "-700", "40"), ("10", "20", "300"), "2395", "0.015", "0.001", "-0.0009", "0.0013", "1599", "1101"),
    ("half_y", ("-75", "60", "-1200"), ("12", "-8", "100"), "2420", "-0.01", "0.004", "0.002", "0.001", "1604", "1100"),
    ("identity", ("10", "20", "0"), ("0", "0", "500"), "2400", "0.2", "-0.1", "0", "0", "1604", "1100"),       # p1 = p2 = 0: the reference's own model
]


def main():
    rows = []
    for rot, X, t, f, k1, k2, p1, p2, cx, cy in CASES:
        rvec, R = ROT[rot]
        X = [F(v) for v in X]
        t = [F(v) for v in t]
        f, k1, k2, p1, p2, cx, cy = (F(v) for v in (f, k1, k2, p1, p2, cx, cy))
        P = [sum(F(R[i][j]) * X[j] for j in range(3)) + t[i] for i in range(3)]
        x, y = P[0] / P[2], P[1] / P[2]
        r2 = x * x + y * y
        d = 1 + k1 * r2 + k2 * r2 * r2
        xd = x * d + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * d + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        u, v = f * xd + cx, f * yd + cy
        rows.append({"rotation": rot, "point": [float(a) for a in X],
                     "camera": list(rvec) + [float(a) for a in t] + [float(f), float(k1), float(k2), float(p1), float(p2), float(cx), float(cy)],
                     "uv": [float(u), float(v)], "uv_exact": [str(u), str(v)]})
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "f8_opencv_tangential.json")
    with open(out, "w") as fh:
        json.dump({"formula": "OpenCV pinhole + 2 radial + 2 tangential terms, exact rational arithmetic", "rows": rows}, fh, indent=1)
    print(f"wrote {len(rows)} rows to {out}")


if __name__ == "__main__":
    main()
