"""Camera-parameter conversion on either side of the bundle adjustment.

Mirrors the functions of the reference's ``lasercalib/convert_params.py`` that ``scripts/calibrate_camera.py`` uses
(lines 58, 77, 82, 95, 98), with the same names, argument order and array conventions:

* :func:`initialize_from_checkerboard`   -- convert_params.py:66-87   (OpenCV YAML -> (C,11) ``cameraArray``)
* :func:`sba_to_readable_format`         -- convert_params.py:18-27   (``cameraArray`` row -> {'K','R','t','d'})
* :func:`readable_to_red_format`         -- convert_params.py:7-16    ((C,25) table for "red")
* :func:`readable_format_to_aruco_format`, :func:`save_aruco_format` -- convert_params.py:105-123 (-> OpenCV YAML)

Differences, all deliberate:
  * the reference reads/writes the YAML through ``cv2.FileStorage``; OpenCV is not a dependency here -- the small
    ``!!opencv-matrix`` subset those files use is parsed/emitted by :func:`read_opencv_yaml` / :func:`write_opencv_yaml`;
  * ``np.NaN`` (removed in numpy 2, convert_params.py:8,41) is spelled ``np.nan``.
Conventions kept exactly, odd as they are: ``K`` of the readable format is the TRANSPOSE of the usual intrinsic matrix
(principal point in the last ROW, convert_params.py:23), ``R`` is built from the NEGATED rotation vector (:24), and the
"red" row is [K^T (9) | R^T (9) | t (3) | k1 k2 0 0].
"""
from __future__ import annotations

import os
import re
from typing import Dict, List, Sequence

import numpy as np
from scipy.spatial.transform import Rotation as R

__all__ = [
    "read_opencv_yaml", "write_opencv_yaml", "initialize_from_checkerboard", "sba_to_readable_format",
    "readable_to_red_format", "readable_format_to_aruco_format", "save_aruco_format", "camera_array_to_readable",
]

_MAT_RE = re.compile(
    r"^(?P<name>[A-Za-z_][A-Za-z0-9_]*):\s*!!opencv-matrix\s*\n"
    r"\s*rows:\s*(?P<rows>\d+)\s*\n\s*cols:\s*(?P<cols>\d+)\s*\n\s*dt:\s*(?P<dt>\w+)\s*\n"
    r"\s*data:\s*\[(?P<data>[^\]]*)\]",
    re.MULTILINE,
)
_SCALAR_RE = re.compile(r"^(?P<name>[A-Za-z_][A-Za-z0-9_]*):\s*(?P<val>[-+0-9.eE]+)\s*$", re.MULTILINE)


def read_opencv_yaml(path: str) -> Dict[str, object]:
    """Parse the ``%YAML:1.0`` files OpenCV's FileStorage writes: ``!!opencv-matrix`` nodes and plain numeric scalars.

    Matrices come back as float64 arrays of shape (rows, cols) -- what ``fs.getNode(name).mat()`` returns
    (convert_params.py:71-76).
    """
    with open(path, "r") as f:
        text = f.read()
    out: Dict[str, object] = {}
    for m in _MAT_RE.finditer(text):
        vals = [float(v) for v in m.group("data").replace("\n", " ").split(",") if v.strip()]
        rows, cols = int(m.group("rows")), int(m.group("cols"))
        if len(vals) != rows * cols:
            raise ValueError(f"{path}: matrix {m.group('name')} has {len(vals)} values, expected {rows}x{cols}")
        out[m.group("name")] = np.asarray(vals, dtype=np.float64).reshape(rows, cols)
    for m in _SCALAR_RE.finditer(text):
        if m.group("name") not in out and m.group("name") not in ("rows", "cols"):
            v = float(m.group("val"))
            out[m.group("name")] = int(v) if v.is_integer() and "." not in m.group("val") else v
    return out


def _fmt(v: float) -> str:
    v = float(v)
    if v == int(v) and abs(v) < 1e15:
        return f"{int(v)}."
    return f"{v:.16e}"


def write_opencv_yaml(path: str, nodes: Dict[str, object]) -> None:
    """Emit ``nodes`` (name -> scalar or array) in the FileStorage layout that :func:`read_opencv_yaml` and OpenCV read."""
    lines = ["%YAML:1.0", "---"]
    for name, val in nodes.items():
        if np.isscalar(val):
            lines.append(f"{name}: {val}")
            continue
        a = np.asarray(val, dtype=np.float64)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        lines.append(f"{name}: !!opencv-matrix")
        lines.append(f"   rows: {a.shape[0]}")
        lines.append(f"   cols: {a.shape[1]}")
        lines.append("   dt: d")
        body, cur = [], "   data: [ "
        flat = [_fmt(v) for v in a.ravel()]
        for i, s in enumerate(flat):
            piece = s + (", " if i + 1 < len(flat) else " ]")
            if len(cur) + len(piece) > 76 and cur.strip():
                body.append(cur.rstrip())
                cur = "       "
            cur += piece
        body.append(cur)
        lines.extend(body)
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def initialize_from_checkerboard(filedir: str, nCams: int, cam_names: Sequence[str], tangential: bool = False) -> np.ndarray:
    """(nCams, 11) initial ``cameraArray`` from per-camera OpenCV YAML files (convert_params.py:66-87).

    Row = [rotvec(rc_ext) (3), tc_ext (3), K[0,0], dist[0], dist[1], K[0,2], K[1,2]].
    tangential=True (extension, the reference drops these two): (nCams, 13) rows that also carry OpenCV's p1, p2 =
    dist[2], dist[3] before the principal point -- the camera model PySBA switches to for 13-column arrays.
    """
    rows = []
    for name in list(cam_names)[:nCams]:
        node = read_opencv_yaml(os.path.join(filedir, f"{name}.yaml"))
        K, dist = node["camera_matrix"], node["distortion_coefficients"].ravel()
        tang = [dist[2] if dist.size > 2 else 0.0, dist[3] if dist.size > 3 else 0.0] if tangential else []
        rows.append(np.concatenate([
            R.from_matrix(node["rc_ext"]).as_rotvec(), node["tc_ext"].ravel()[:3],
            [K[0, 0], dist[0], dist[1]], tang, [K[0, 2], K[1, 2]],
        ]))
    return np.asarray(rows, dtype=np.float64).reshape(nCams, 13 if tangential else 11)


def sba_to_readable_format(camParamVec: np.ndarray) -> Dict[str, np.ndarray]:
    """One ``cameraArray`` row -> {'K','R','t','d'} (convert_params.py:18-27); ``t`` and ``d`` are views, like upstream.
    A 13-parameter row (tangential model) also yields 'p' = (p1, p2); an 11-parameter row gives exactly upstream's dict."""
    tang = len(camParamVec) == 13
    f, cx, cy = camParamVec[6], camParamVec[-2], camParamVec[-1]
    K = np.array([[f, 0.0, 0.0], [0.0, f, 0.0], [cx, cy, 1.0]])          # transposed intrinsic matrix (see module doc)
    out = {"K": K, "R": R.from_rotvec(-camParamVec[:3]).as_matrix(), "t": camParamVec[3:6], "d": camParamVec[7:9]}
    if tang:
        out["p"] = camParamVec[9:11]
    return out


def camera_array_to_readable(cameraArray: np.ndarray) -> List[Dict[str, np.ndarray]]:
    """The ``camList`` loop of calibrate_camera.py:75-77 / 93-95."""
    return [sba_to_readable_format(cameraArray[i, :]) for i in range(cameraArray.shape[0])]


def readable_to_red_format(camList: Sequence[Dict[str, np.ndarray]]) -> np.ndarray:
    """(len(camList), 25) table [K^T (9) | R^T (9) | t (3) | d (2), 0, 0] (convert_params.py:7-16); the two zeros are
    the tangential slots and receive p1, p2 when the camera dict has them."""
    out = np.full((len(camList), 25), np.nan)
    for row, p in zip(out, camList):
        row[0:9] = np.asarray(p["K"]).T.ravel()
        row[9:18] = np.asarray(p["R"]).T.ravel()
        row[18:21] = p["t"]
        row[21:23] = p["d"]
        row[23:25] = p.get("p", (0.0, 0.0))
    return out


def readable_format_to_aruco_format(save_root: str, nCams: int, camList, cam_names: Sequence[str]) -> None:
    """Write one OpenCV YAML per camera (convert_params.py:105-113): camera_matrix = K^T, rc_ext = R^T, 5 distortion terms
    [k1, k2, p1, p2, k3 = 0] in OpenCV's order; upstream writes zeros for p1, p2 (its model has none, line 110), here they
    are exported when the camera came from a 13-parameter row."""
    for i in range(nCams):
        write_opencv_yaml(save_root + "{}.yaml".format(cam_names[i]), {
            "camera_matrix": camList[i]["K"].T,
            "distortion_coefficients": np.asarray([camList[i]["d"][0], camList[i]["d"][1], *camList[i].get("p", (0, 0)), 0]),
            "rc_ext": camList[i]["R"].T,
            "tc_ext": camList[i]["t"],
        })


def save_aruco_format(save_root: str, nCams: int, aruco_cam_list, cam_names: Sequence[str]) -> None:
    """convert_params.py:115-123."""
    for i in range(nCams):
        write_opencv_yaml(save_root + "{}.yaml".format(cam_names[i]), {
            "camera_matrix": aruco_cam_list[i]["camera_matrix"],
            "distortion_coefficients": aruco_cam_list[i]["distortion_coefficients"],
            "rc_ext": aruco_cam_list[i]["rc_ext"],
            "tc_ext": aruco_cam_list[i]["tc_ext"],
        })
