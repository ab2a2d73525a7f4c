"""CPU: the C-ABI library builds, loads and exports every symbol include/sba_hip.h declares; without a GPU the
compute entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from lasercalib_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _native.load()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "sba_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sba_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    names = _declared_functions()
    assert len(names) >= 20
    raw = ctypes.CDLL(_native.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/sba_hip.h but not exported by libsba_hip.so"
    assert set(names) == set(_native.EXPORTED_SYMBOLS)
    assert lib.sba_abi_version() == 2


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_native.ProblemDesc) == 48
    assert ctypes.sizeof(_native.LmOpts) == 72
    assert ctypes.sizeof(_native.LmReport) == 88
    assert ctypes.sizeof(_native.LmIterLog) == 64


def test_no_gpu_means_loud_failure(lib):
    if lib.sba_device_count() > 0:
        pytest.skip("a GPU is visible; the no-device path is exercised on the CPU-only container")
    from lasercalib_amd.pySBA import PySBA
    from lasercalib_amd.synth import make_rig
    rig = make_rig(2, 20)
    sba = PySBA(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"])
    with pytest.raises(_native.SbaError, match="no HIP device"):
        sba.bundleAdjust(1e-4)
    with pytest.raises(_native.SbaError, match="no HIP device"):
        sba.project(rig["pts0"][rig["point_ind"]], rig["cams0"][rig["camera_ind"]])
    with pytest.raises(_native.SbaError):
        sba.bundleAdjust_nocam()
