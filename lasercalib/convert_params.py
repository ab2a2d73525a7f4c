"""``lasercalib.convert_params`` -> the OpenCV-free, numpy-2-safe conversions of ``lasercalib_amd.convert_params``.

Names this module does not define (``getCameraArray``, ``load_from_blender``, ``red_to_aruco`` ... -- reference helpers
outside the calibration path) are looked up in the upstream file when LASERCALIB_UPSTREAM points at it.
"""
import importlib.util as _ilu
import os as _os

from lasercalib_amd.convert_params import (  # noqa: F401
    camera_array_to_readable, initialize_from_checkerboard, read_opencv_yaml, readable_format_to_aruco_format,
    readable_to_red_format, save_aruco_format, sba_to_readable_format, write_opencv_yaml,
)

_upstream = None


def __getattr__(name):
    global _upstream
    up = _os.environ.get("LASERCALIB_UPSTREAM")
    path = _os.path.join(up, "convert_params.py") if up else None
    if path and _os.path.isfile(path):
        if _upstream is None:
            spec = _ilu.spec_from_file_location("lasercalib._upstream_convert_params", path)
            _upstream = _ilu.module_from_spec(spec)
            spec.loader.exec_module(_upstream)
        if hasattr(_upstream, name):
            return getattr(_upstream, name)
    raise AttributeError(f"module 'lasercalib.convert_params' has no attribute {name!r}")
