"""ctypes binding of libsba_hip.so (C ABI declared in include/sba_hip.h).

There is NO CPU fallback: if the shared library is missing, or no gfx950 GPU is
visible, every compute entry point raises.  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C lasercalib_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsba_hip.so")

SBA_F64, SBA_F32 = 0, 1
CAM_RADIAL, CAM_RADIAL_TANGENTIAL = 0, 1        # sba_cam_model: 11 / 13 parameters per camera
LOSS_LINEAR, LOSS_HUBER, LOSS_SOFT_L1, LOSS_CAUCHY = 0, 1, 2, 3      # sba_loss
MODE_FULL, MODE_POINTS_ONLY, MODE_SHARED_INTR, MODE_CAMS_ONLY_SQ, MODE_TRANSFORM_SQ = 0, 1, 2, 3, 4
NSCALARS = 8

# every symbol include/sba_hip.h declares (tests/test_cabi_symbols.py checks the .so exports all of them)
EXPORTED_SYMBOLS = (
    "sba_abi_version", "sba_device_count", "sba_last_error", "sba_rotate", "sba_project", "sba_project_model",
    "sba_create", "sba_upload", "sba_set_params", "sba_get_params", "sba_destroy",
    "sba_get_gradient", "sba_get_transform", "sba_lm_get_step", "sba_ipc_export", "sba_ipc_attach", "sba_residual", "sba_residual_jacobian", "sba_solve_lm",
    "sba_lm_exchange_size", "sba_lm_begin", "sba_lm_linearize", "sba_lm_form_reduced",
    "sba_lm_solve_trial", "sba_lm_decide", "sba_lm_decide_async", "sba_lm_poll", "sba_lm_run", "sba_lm_finish", "sba_lm_get_log", "sba_time_kernel", "sba_get_kernel_profile",
    "sba_comm_get_unique_id", "sba_comm_init", "sba_set_fixed_points", "sba_set_robust_loss",
)


class SbaError(RuntimeError):
    """A libsba_hip call returned a negative status."""


class ProblemDesc(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int64),
                ("dtype", C.c_int32), ("device", C.c_int32), ("stream", C.c_void_p),
                ("use_stream", C.c_int32), ("cam_model", C.c_int32), ("reserved", C.c_int32 * 2)]


class LmOpts(C.Structure):
    _fields_ = [("ftol", C.c_double), ("xtol", C.c_double), ("gtol", C.c_double),
                ("max_nfev", C.c_int64), ("mode", C.c_int32), ("verbose", C.c_int32),
                ("max_iter", C.c_int32), ("always_relinearize", C.c_int32),
                ("lambda0", C.c_double), ("reserved", C.c_int32 * 4)]


class LmReport(C.Structure):
    _fields_ = [("cost", C.c_double), ("initial_cost", C.c_double), ("optimality", C.c_double),
                ("step_norm", C.c_double), ("lambda_", C.c_double), ("nfev", C.c_int64),
                ("njev", C.c_int64), ("iterations", C.c_int32), ("accepted", C.c_int32),
                ("status", C.c_int32), ("reserved", C.c_int32), ("seconds_total", C.c_double),
                ("seconds_device", C.c_double)]


class LmIterLog(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("accepted", C.c_int32), ("nfev", C.c_int64),
                ("cost", C.c_double), ("cost_reduction", C.c_double), ("step_norm", C.c_double),
                ("optimality", C.c_double), ("lambda_", C.c_double), ("rho", C.c_double)]


_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's);
    if this library pulled in the system copy first, a later ``import torch`` would bring up a second runtime that
    finds no GPU ("No HIP GPUs are available").  When torch is installed but not imported yet, its copy is loaded
    first so that both resolve to the same runtime; torch itself is not imported."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.isfile(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libsba_hip.so once; raise (never fall back) when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SbaError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C lasercalib_amd/csrc`. lasercalib_amd has no CPU fallback.")
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    H = C.c_void_p
    sig = {
        "sba_abi_version": (C.c_int, []),
        "sba_device_count": (C.c_int, []),
        "sba_last_error": (C.c_char_p, [H]),
        "sba_rotate": (C.c_int, [C.c_int, C.c_int, C.c_int64, dp, dp, dp]),
        "sba_project": (C.c_int, [C.c_int, C.c_int, C.c_int64, dp, dp, dp]),
        "sba_project_model": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, dp, dp, dp]),
        "sba_create": (C.c_int, [C.POINTER(ProblemDesc), C.POINTER(H)]),
        "sba_upload": (C.c_int, [H, dp, dp, dp, ip, ip, dp]),
        "sba_set_params": (C.c_int, [H, dp]),
        "sba_get_params": (C.c_int, [H, dp, dp]),
        "sba_destroy": (C.c_int, [H]),
        "sba_get_gradient": (C.c_int, [H, dp, dp]),
        "sba_get_transform": (C.c_int, [H, dp]),
        "sba_lm_get_step": (C.c_int, [H, dp]),
        "sba_residual": (C.c_int, [H, dp, dp, dp]),
        "sba_residual_jacobian": (C.c_int, [H, dp, dp, dp, dp]),
        "sba_solve_lm": (C.c_int, [H, C.POINTER(LmOpts), dp, dp, C.POINTER(LmReport),
                                   C.POINTER(LmIterLog), C.c_int32, C.POINTER(C.c_int32)]),
        "sba_lm_exchange_size": (C.c_int64, [H]),
        "sba_lm_begin": (C.c_int, [H, C.POINTER(LmOpts)]),
        "sba_lm_linearize": (C.c_int, [H]),
        "sba_lm_form_reduced": (C.c_int, [H, C.c_void_p]),
        "sba_lm_solve_trial": (C.c_int, [H, C.c_void_p, C.c_void_p]),
        "sba_lm_decide": (C.c_int, [H, C.c_void_p, C.c_int32, C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32), C.POINTER(LmIterLog)]),
        "sba_lm_decide_async": (C.c_int, [H, C.c_void_p, C.c_int32]),
        "sba_lm_poll": (C.c_int, [H, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "sba_lm_run": (C.c_int, [H, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "sba_lm_finish": (C.c_int, [H, dp, dp, C.POINTER(LmReport)]),
        "sba_lm_get_log": (C.c_int, [H, C.POINTER(LmIterLog), C.c_int32, C.POINTER(C.c_int32)]),
        "sba_time_kernel": (C.c_int, [H, C.c_char_p, C.c_int32, dp]),
        "sba_get_kernel_profile": (C.c_int, [H, dp, ip]),
        "sba_comm_get_unique_id": (C.c_int, [C.c_char_p]),
        "sba_comm_init": (C.c_int, [H, C.c_char_p, C.c_int32, C.c_int32]),
        "sba_ipc_export": (C.c_int, [H, C.c_int32, C.c_char_p]),
        "sba_ipc_attach": (C.c_int, [H, C.c_int32, C.c_int32, C.c_char_p]),
        "sba_set_fixed_points": (C.c_int, [H, C.c_void_p]),
        "sba_set_robust_loss": (C.c_int, [H, C.c_int32, C.c_double]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.sba_abi_version() != 2:
        raise SbaError("libsba_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _dptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _check(rc, handle=None):
    if rc == 0:
        return
    msg = load().sba_last_error(handle)
    text = msg.decode() if msg else ""
    if rc == -4:
        # scipy raises ValueError here (scipy/optimize/_lsq/least_squares.py:844-845); keep the type.
        raise ValueError(text or "Residuals are not finite in the initial point.")
    raise SbaError(f"libsba_hip status {rc}: {text}")


def dtype_code(dtype):
    if dtype in (SBA_F64, "f64", "float64", np.float64):
        return SBA_F64
    if dtype in (SBA_F32, "f32", "float32", np.float32):
        return SBA_F32
    raise ValueError(f"unknown dtype {dtype!r}")


def device_count():
    return int(load().sba_device_count())


COMM_ID_BYTES = 128
IPC_HANDLE_BYTES = 64


def comm_unique_id():
    """Rank 0: a fresh 128-byte ncclUniqueId (bytes) to hand to every rank's Problem.comm_init."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load().sba_comm_get_unique_id(buf))
    return buf.raw


def cam_model_of(n_cam_params):
    """11 columns = the reference's radial camera row (pySBA.py:31-35), 13 = radial + tangential (p1, p2 before cx, cy)."""
    if n_cam_params == 11:
        return CAM_RADIAL
    if n_cam_params == 13:
        return CAM_RADIAL_TANGENTIAL
    raise ValueError("camera rows must have 11 columns (radial model) or 13 (radial + tangential)")


def project_rows(points, cam_rows, dtype=SBA_F64, device=0):
    """PySBA.project on gathered rows (pySBA.py:76-89), computed on the GPU."""
    lib = load()
    p, c = _f64(points), _f64(cam_rows)
    if p.ndim != 2 or p.shape[1] != 3 or c.ndim != 2 or c.shape[1] not in (11, 13) or p.shape[0] != c.shape[0]:
        raise ValueError("project expects (M,3) points and (M,11) camera rows ((M,13) with tangential distortion)")
    out = np.empty((p.shape[0], 2))
    _check(lib.sba_project_model(device, dtype_code(dtype), cam_model_of(c.shape[1]), p.shape[0], _dptr(p), _dptr(c), _dptr(out)))
    return out


def rotate_rows(points, rot_vecs, dtype=SBA_F64, device=0):
    """PySBA.rotate on gathered rows (pySBA.py:61-73), computed on the GPU."""
    lib = load()
    p, r = _f64(points), _f64(rot_vecs)
    if p.ndim != 2 or p.shape[1] != 3 or r.shape != p.shape:
        raise ValueError("rotate expects (M,3) points and (M,3) rotation vectors")
    out = np.empty_like(p)
    _check(lib.sba_rotate(device, dtype_code(dtype), p.shape[0], _dptr(p), _dptr(r), _dptr(out)))
    return out


class Problem:
    """One device-resident bundle-adjustment problem (wraps an sba_handle)."""

    def __init__(self, cams, pts, uv, cam_idx, pt_idx, weights=None, dtype=SBA_F64, device=0, stream=None):
        lib = load()
        self._lib = lib
        self.cams0, self.pts0 = _f64(cams), _f64(pts)
        uv = _f64(uv)
        ci = np.ascontiguousarray(cam_idx, dtype=np.int64).reshape(-1)
        pi = np.ascontiguousarray(pt_idx, dtype=np.int64).reshape(-1)
        self.C, self.N, self.M = self.cams0.shape[0], self.pts0.shape[0], ci.shape[0]
        if self.cams0.ndim != 2 or self.cams0.shape[1] not in (11, 13):
            raise ValueError("cameraArray must have shape (n_cameras, 11) (or (n_cameras, 13) with tangential distortion)")
        self.P = self.cams0.shape[1]
        if self.pts0.ndim != 2 or self.pts0.shape[1] != 3:
            raise ValueError("points3D must have shape (n_points, 3)")
        if uv.shape != (self.M, 2) or pi.shape[0] != self.M:
            raise ValueError("points2D must be (n_observations, 2) and index arrays (n_observations,)")
        w = None
        if weights is not None:
            w = _f64(weights).reshape(-1)
            if w.shape[0] != self.M:
                raise ValueError("pointWeights must have one entry per observation")
        self.dtype = dtype_code(dtype)
        # stream=None: private stream.  stream=<int handle>: run on it (0 = the legacy default stream).
        desc = ProblemDesc(self.C, self.N, self.M, self.dtype, device,
                           C.c_void_p(stream) if stream else None, 0 if stream is None else 1, cam_model_of(self.P),
                           (C.c_int32 * 2)())
        h = C.c_void_p()
        _check(lib.sba_create(C.byref(desc), C.byref(h)))
        self._h = h
        try:
            _check(lib.sba_upload(h, _dptr(self.cams0), _dptr(self.pts0), _dptr(uv), _iptr(ci), _iptr(pi), _dptr(w)), h)
        except Exception:
            self.close()
            raise

    # -- multi-GPU inside the library (RCCL): after this, solve_lm runs the sharded loop on all ranks together
    def comm_init(self, unique_id, rank, n_ranks):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes comm_unique_id() returned on rank 0")
        _check(self._lib.sba_comm_init(self._h, bytes(unique_id), int(rank), int(n_ranks)), self._h)
        self.comm_rank, self.comm_n = int(rank), int(n_ranks)

    # -- the same sharded loop over peer-mapped buffers instead of RCCL (sba_ipc_export / sba_ipc_attach, include/sba_hip.h)
    def ipc_export(self, n_ranks):
        """This rank's exchange area for a job of n_ranks: returns the 64-byte handle every peer needs."""
        buf = C.create_string_buffer(IPC_HANDLE_BYTES)
        _check(self._lib.sba_ipc_export(self._h, int(n_ranks), buf), self._h)
        return bytes(buf.raw)

    def ipc_attach(self, rank, handles):
        """handles: the n_ranks handles in rank order (this rank's own included); after this, solve_lm runs the sharded loop."""
        if any(len(h) != IPC_HANDLE_BYTES for h in handles):
            raise ValueError("every handle must be the 64 bytes ipc_export() returned on its rank")
        _check(self._lib.sba_ipc_attach(self._h, int(rank), len(handles), b"".join(bytes(h) for h in handles)), self._h)
        self.comm_rank, self.comm_n = int(rank), len(handles)

    # -- opt-in extensions (off by default: the reference ignores points3Dfixed and uses the linear loss)
    def set_fixed_points(self, mask):
        """mask: (N,) booleans / 0-1, True = the point is held at its uploaded coordinates (gauge anchor); None clears."""
        if mask is None:
            _check(self._lib.sba_set_fixed_points(self._h, None), self._h)
            return
        m = np.ascontiguousarray(np.asarray(mask).reshape(-1) != 0, dtype=np.uint8)
        if m.shape[0] != self.N:
            raise ValueError("fixed-point mask must have one entry per 3-D point")
        _check(self._lib.sba_set_fixed_points(self._h, m.ctypes.data_as(C.c_void_p)), self._h)

    def set_robust_loss(self, loss="huber", f_scale=1.0):
        """scipy.optimize.least_squares(loss=..., f_scale=...) semantics; loss in ('linear', 'huber', 'soft_l1', 'cauchy')."""
        code = {"linear": LOSS_LINEAR, "huber": LOSS_HUBER, "soft_l1": LOSS_SOFT_L1, "cauchy": LOSS_CAUCHY}.get(loss)
        if code is None:
            raise ValueError("loss must be 'linear', 'huber', 'soft_l1' or 'cauchy'")
        _check(self._lib.sba_set_robust_loss(self._h, code, float(f_scale)), self._h)

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._lib.sba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def n_params(self):
        return self.P * self.C + 3 * self.N

    # -- model evaluation
    def residual(self, x=None, want_r=True):
        r = np.empty(2 * self.M) if want_r else None
        cost = C.c_double()
        xx = None if x is None else _f64(x)
        _check(self._lib.sba_residual(self._h, _dptr(xx), _dptr(r), C.byref(cost)), self._h)
        return r, cost.value

    def residual_jacobian(self, x=None):
        r = np.empty(2 * self.M)
        Jc = np.empty((self.M, 2, self.P))
        Jp = np.empty((self.M, 2, 3))
        xx = None if x is None else _f64(x)
        _check(self._lib.sba_residual_jacobian(self._h, _dptr(xx), _dptr(r), _dptr(Jc), _dptr(Jp)), self._h)
        return r, Jc, Jp

    def set_params(self, x):
        _check(self._lib.sba_set_params(self._h, _dptr(_f64(x))), self._h)

    def get_params(self):
        cams = np.empty((self.C, self.P))
        pts = np.empty((self.N, 3))
        _check(self._lib.sba_get_params(self._h, _dptr(cams), _dptr(pts)), self._h)
        return cams, pts

    def get_gradient(self):
        gc = np.empty((self.C, self.P))
        gp = np.empty((self.N, 3))
        _check(self._lib.sba_get_gradient(self._h, _dptr(gc), _dptr(gp)), self._h)
        return gc, gp

    def get_transform(self):
        th = np.empty(12)
        _check(self._lib.sba_get_transform(self._h, _dptr(th)), self._h)
        return th

    # -- solver
    @staticmethod
    def make_opts(ftol=1e-8, xtol=1e-8, gtol=1e-8, max_nfev=0, mode=MODE_FULL, verbose=0, max_iter=0,
                  always_relinearize=False, lambda0=0.0, profile=False):
        return LmOpts(ftol, xtol, gtol, int(max_nfev or 0), mode, verbose, int(max_iter or 0),
                      1 if always_relinearize else 0, float(lambda0), (C.c_int32 * 4)(1 if profile else 0, 0, 0, 0))

    def solve_lm(self, opts, log_capacity=4096):
        cams = np.empty((self.C, self.P))
        pts = np.empty((self.N, 3))
        rep = LmReport()
        log = (LmIterLog * log_capacity)()
        rows = C.c_int32(0)
        _check(self._lib.sba_solve_lm(self._h, C.byref(opts), _dptr(cams), _dptr(pts), C.byref(rep), log,
                                      log_capacity, C.byref(rows)), self._h)
        return cams, pts, rep, [log[i] for i in range(min(rows.value, log_capacity))]

    # -- phase-level API (multi-GPU driver)
    def exchange_size(self):
        return int(self._lib.sba_lm_exchange_size(self._h))

    def lm_begin(self, opts):
        _check(self._lib.sba_lm_begin(self._h, C.byref(opts)), self._h)

    def lm_linearize(self):
        _check(self._lib.sba_lm_linearize(self._h), self._h)

    def lm_form_reduced(self, exchange_ptr):
        _check(self._lib.sba_lm_form_reduced(self._h, C.c_void_p(exchange_ptr)), self._h)

    def lm_solve_trial(self, exchange_ptr, scalars_ptr):
        _check(self._lib.sba_lm_solve_trial(self._h, C.c_void_p(exchange_ptr), C.c_void_p(scalars_ptr)), self._h)

    def lm_get_step(self):
        """delta_c of the last lm_solve_trial (test hook for the reduced-system factorisations)."""
        d = np.empty((self.C, self.P))
        _check(self._lib.sba_lm_get_step(self._h, _dptr(d)), self._h)
        return d

    def lm_decide(self, scalars_all_ptr, n_ranks):
        status, acc = C.c_int32(), C.c_int32()
        row = LmIterLog()
        _check(self._lib.sba_lm_decide(self._h, C.c_void_p(scalars_all_ptr), n_ranks, C.byref(status),
                                       C.byref(acc), C.byref(row)), self._h)
        return status.value, bool(acc.value), row

    def lm_decide_async(self, scalars_all_ptr, n_ranks):
        _check(self._lib.sba_lm_decide_async(self._h, C.c_void_p(scalars_all_ptr), n_ranks), self._h)

    def lm_poll(self):
        status, iters = C.c_int32(), C.c_int32()
        _check(self._lib.sba_lm_poll(self._h, C.byref(status), C.byref(iters)), self._h)
        return status.value, iters.value

    def lm_run(self):
        """The iteration loop of solve_lm alone (between lm_begin and lm_finish); returns (status, iterations)."""
        status, iters = C.c_int32(), C.c_int32()
        _check(self._lib.sba_lm_run(self._h, C.byref(status), C.byref(iters)), self._h)
        return status.value, iters.value

    def lm_finish(self):
        cams = np.empty((self.C, self.P))
        pts = np.empty((self.N, 3))
        rep = LmReport()
        _check(self._lib.sba_lm_finish(self._h, _dptr(cams), _dptr(pts), C.byref(rep)), self._h)
        return cams, pts, rep

    def iteration_log(self, capacity=4096):
        log = (LmIterLog * capacity)()
        rows = C.c_int32(0)
        _check(self._lib.sba_lm_get_log(self._h, log, capacity, C.byref(rows)), self._h)
        return [log[i] for i in range(min(rows.value, capacity))]

    # -- measurement
    PROFILE_SLOTS = ("linearize_points", "linearize_cams", "schur", "schur_reduce", "cholesky_solve", "backsub")

    def kernel_profile(self):
        """Mean in-loop duration (us) per kernel class of the last solve run with profile=True."""
        tot = np.zeros(len(self.PROFILE_SLOTS))
        cnt = np.zeros(len(self.PROFILE_SLOTS), dtype=np.int64)
        _check(self._lib.sba_get_kernel_profile(self._h, _dptr(tot), _iptr(cnt)), self._h)
        return {k: (tot[i] / cnt[i] if cnt[i] else 0.0) for i, k in enumerate(self.PROFILE_SLOTS)}

    def time_kernel(self, name, reps=20):
        us = C.c_double()
        _check(self._lib.sba_time_kernel(self._h, name.encode(), reps, C.byref(us)), self._h)
        return us.value
