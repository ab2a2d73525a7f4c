"""Drop-in mirror of the reference's ``lasercalib.pySBA.PySBA`` over the MI355X engine.

Same class name, constructor, attributes, method names, positional order and defaults as
/root/reference/lasercalib/pySBA.py:25-325, so ``scripts/calibrate_camera.py`` (which does
``from lasercalib.pySBA import PySBA`` -> ``PySBA(...)`` -> ``sba.bundleAdjust(1e-4)`` ->
reads ``sba.cameraArray`` -> pickles ``sba``) runs unchanged.  Only the arithmetic moved:
projection, analytic Jacobian blocks and the whole Levenberg-Marquardt / Schur-complement
loop execute in libsba_hip.so on the GPU (see include/sba_hip.h).  There is no CPU path in
this module: without the shared library or without a gfx950 device the methods raise.

Extension beyond the reference: a cameraArray with 13 columns, [rvec(3), t(3), f, k1, k2, p1, p2, cx, cy], selects the
radial + tangential camera model (BASELINE config 5; the reference's model is radial only, pySBA.py:82-88).  Every method
below then works on 13-parameter rows; with the reference's 11 columns nothing changes.

Environment knobs (the script itself stays unchanged):
  LASERCALIB_SBA_DTYPE   f64 (default) | f32   arithmetic type of the per-observation math
  LASERCALIB_SBA_DEVICE  HIP device ordinal (default: LOCAL_RANK or 0)
  LASERCALIB_SBA_USE_FIXED 1: honour ``points3Dfixed`` (a boolean mask over the 3-D points, or an array of point indices) as gauge
                         anchors: those points keep their coordinates and drop out of the unknowns.  Default 0: stored and
                         ignored, exactly like the reference (pySBA.py:28,55).
  LASERCALIB_SBA_LOSS    linear (default, the reference's) | huber | soft_l1 | cauchy, with LASERCALIB_SBA_F_SCALE (default 1.0 px): scipy's
                         ``least_squares(loss=, f_scale=)`` semantics for bundleAdjust / _nocam / _sharedcam.
  LASERCALIB_SBA_SHARD   1: with a torch.distributed process group up, bundleAdjust / _nocam / _sharedcam shard the points
                         over the ranks (every rank calls with the same full problem).  Default 0: never implicit.
                         The squared-error variants (bundle_adjustment_camonly, bundleAdjust_transform_points_3d) always
                         run on one GPU per process.
"""
from __future__ import annotations

import os

import numpy as np
from scipy.optimize import OptimizeResult
from scipy.sparse import coo_matrix, csr_matrix, lil_matrix

from . import _native

N_CAM_PARAMS = 11

TERMINATION_MESSAGES = {
    -1: "Improper input parameters status returned from `leastsq`",
    0: "The maximum number of function evaluations is exceeded.",
    1: "`gtol` termination condition is satisfied.",
    2: "`ftol` termination condition is satisfied.",
    3: "`xtol` termination condition is satisfied.",
    4: "Both `ftol` and `xtol` termination conditions are satisfied.",
}


def _env_dtype():
    name = os.environ.get("LASERCALIB_SBA_DTYPE", "f64")
    return _native.dtype_code("f64" if name == "mixed" else name)


def _env_mixed():
    """LASERCALIB_SBA_DTYPE=mixed: bundleAdjust / _nocam / _sharedcam iterate on the fp32 engine (the fused bf16-pipe kernels) until
    the caller's tolerances stop it, then continue on the fp64 engine from that point with the same tolerances.  The returned
    point is the fp64 engine's -- it satisfies the same termination tests as a pure fp64 solve -- while most iterations ran at
    the fp32 rate (17 cameras x 10k points at the example's visibility: 109 us per iteration instead of 140; 232 before the fp64 engine
    got its own one-launch kernels in round 3)."""
    return os.environ.get("LASERCALIB_SBA_DTYPE", "f64") == "mixed"


def _env_device():
    return int(os.environ.get("LASERCALIB_SBA_DEVICE", os.environ.get("LOCAL_RANK", "0")))


class SBAResult(OptimizeResult):
    """scipy ``OptimizeResult`` whose ``fun`` and ``jac`` / ``grad`` are computed on first access.

    scipy returns the final residual vector and the final Jacobian -- a (2M x n) CSR matrix (least_squares.py:950-961), 22.4M
    non-zeros at 800k observations -- with every result; the reference's caller never reads either (scripts/calibrate_camera.py:71
    discards the result).  The residual read-back alone was 0.6 ms of the 2.7 ms a ``bundleAdjust`` call takes at 16 x 50k
    (tools/wall_pysba.py), so both are materialised by the device kernels when somebody asks.  The closures that do it live
    OUTSIDE the mapping (instance attributes), so the dict protocol stays consistent: ``'fun' in res``, ``'jac' in res`` and
    ``'grad' in res`` are true while they can be made, ``res.get(k)`` / ``res[k]`` / ``res.k`` make them, ``len`` / ``keys`` / ``items`` /
    ``copy`` / ``repr`` / pickling see a complete result (fun; jac and grad only once somebody has asked for them -- they are
    large) and never the closures.  Should the device be gone by then (a worker process without the GPU unpickles nothing lazy:
    pickling materialises ``fun`` first), the key is simply absent: KeyError / AttributeError, as for any missing key.
    """
    _LAZY = ("fun", "jac", "grad")

    def _maker(self, name):
        return self.__dict__.get(name)

    def set_makers(self, fun_maker=None, jac_maker=None):
        object.__setattr__(self, "_fun_maker", fun_maker)
        object.__setattr__(self, "_jac_maker", jac_maker)

    def __missing__(self, key):
        try:
            if key == "fun" and self._maker("_fun_maker") is not None:
                dict.__setitem__(self, "fun", self._maker("_fun_maker")())
                return dict.__getitem__(self, "fun")
            if key in ("jac", "grad") and self._maker("_jac_maker") is not None:
                J = self._maker("_jac_maker")()
                fvec = self["fun"]
                dict.__setitem__(self, "jac", J)
                dict.__setitem__(self, "grad", J.T.dot(fvec))
                return dict.__getitem__(self, key)
        except (_native.SbaError, OSError) as e:      # no device (any more): the lazy key does not exist
            raise KeyError(key) from e
        raise KeyError(key)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def _can_make(self, key):
        if key == "fun":
            return self._maker("_fun_maker") is not None
        if key in ("jac", "grad"):
            return self._maker("_jac_maker") is not None and (dict.__contains__(self, "fun") or self._maker("_fun_maker") is not None)
        return False

    def __contains__(self, key):
        return dict.__contains__(self, key) or self._can_make(key)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def _materialise(self):
        if not dict.__contains__(self, "fun") and self._maker("_fun_maker") is not None:
            try:
                self["fun"]
            except KeyError:
                pass

    def keys(self):
        self._materialise()
        return list(dict.keys(self))

    def items(self):
        return [(k, dict.__getitem__(self, k)) for k in self.keys()]

    def values(self):
        return [dict.__getitem__(self, k) for k in self.keys()]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def copy(self):
        """A plain, complete OptimizeResult (no closures)."""
        return OptimizeResult(self.items())

    def __repr__(self):
        return repr(OptimizeResult(self.items()))

    def __reduce__(self):   # the closures never travel; the residual vector goes along
        return (OptimizeResult, (dict(self.items()),))


def _print_table(log, initial_cost, report, message, verbose):
    """Iteration table in the format scipy prints for verbose=2 (scipy/optimize/_lsq/common.py:545-563)."""
    if verbose >= 2:
        print("{:^15}{:^15}{:^15}{:^15}{:^15}{:^15}".format(
            "Iteration", "Total nfev", "Cost", "Cost reduction", "Step norm", "Optimality"))
        first_opt = log[0].optimality if log else report.optimality
        print("{:^15}{:^15}{:^15.4e}{:^15}{:^15}{:^15.2e}".format(0, 1, initial_cost, "", "", first_opt))
        it = 0
        for k, row in enumerate(log):
            if not row.accepted:
                continue
            it += 1
            nxt = log[k + 1].optimality if k + 1 < len(log) else report.optimality
            print("{:^15}{:^15}{:^15.4e}{:^15.2e}{:^15.2e}{:^15.2e}".format(
                it, row.nfev, row.cost, row.cost_reduction, row.step_norm, nxt))
    if verbose >= 1:
        print(message)
        print("Function evaluations {0}, initial cost {1:.4e}, final cost {2:.4e}, "
              "first-order optimality {3:.2e}.".format(report.nfev, initial_cost, report.cost, report.optimality))


class PySBA:
    """Python class for Simple Bundle Adjustment (surface of pySBA.py:25-325, GPU engine underneath)."""

    def __init__(self, cameraArray, points3D, points2D, cameraIndices, point2DIndices, points3Dfixed=None,
                 pointWeights=None):
        # references are stored, not copies, like pySBA.py:50-59
        self.cameraArray = cameraArray
        self.points3D = points3D
        self.points2D = points2D
        self.cameraIndices = cameraIndices
        self.point2DIndices = point2DIndices
        self.points3Dfixed = points3Dfixed
        if pointWeights is None:
            pointWeights = np.full_like(point2DIndices, 1)     # integer ones (pySBA.py:57)
        self.pointWeights = pointWeights.reshape((-1, 1))
        self.points3Dfixed_labeled = None

    @property
    def _P(self):
        """Camera parameters per camera: 11 (the reference's row, pySBA.py:31-35) or 13 (with tangential p1, p2)."""
        return int(np.shape(self.cameraArray)[1]) if np.ndim(self.cameraArray) == 2 else N_CAM_PARAMS

    # ------------------------------------------------------------------ model (GPU kernels, numpy in/out)
    def rotate(self, points, rot_vecs):
        """Rodrigues rotation of (M,3) points by (M,3) rotation vectors (pySBA.py:61-73)."""
        return _native.rotate_rows(points, rot_vecs, dtype=_env_dtype(), device=_env_device())

    def project(self, points, cameraArray):
        """(M,3) points x (M,11) gathered camera rows -> (M,2) pixels (pySBA.py:76-89)."""
        return _native.project_rows(points, cameraArray, dtype=_env_dtype(), device=_env_device())

    def fun(self, params, n_cameras, n_points, camera_indices, point_indices, points_2d, pointWeights):
        """Weighted residual vector, interleaved (u,v) per observation (pySBA.py:92-101)."""
        nCamParams = self._P
        camera_params = params[:n_cameras * nCamParams].reshape((n_cameras, nCamParams))
        points_3d = params[n_cameras * nCamParams:].reshape((n_points, 3))
        points_proj = self.project(points_3d[point_indices], camera_params[camera_indices])
        return (pointWeights * (points_proj - points_2d)).ravel()

    def bundle_adjustment_sparsity(self, numCameras, numPoints, cameraIndices, pointIndices):
        """Jacobian pattern of `fun`: 28 ones per observation (pySBA.py:103-118), as a lil_matrix of int."""
        P = self._P
        m = cameraIndices.size * 2
        n = numCameras * P + numPoints * 3
        obs = np.arange(cameraIndices.size)
        cols = np.concatenate([cameraIndices[:, None] * P + np.arange(P)[None, :],
                               numCameras * P + pointIndices[:, None] * 3 + np.arange(3)[None, :]], axis=1)
        rows = np.repeat(2 * obs, cols.shape[1])
        cols = cols.ravel()
        rows = np.concatenate([rows, rows + 1])
        cols = np.concatenate([cols, cols])
        A = coo_matrix((np.ones(rows.size, dtype=int), (rows, cols)), shape=(m, n))
        A.sum_duplicates()
        A.data[:] = 1
        return lil_matrix(A)

    def optimizedParams(self, params, n_cameras, n_points):
        """Split x into (n_cameras,11) and (n_points,3) views (pySBA.py:121-129)."""
        P = self._P
        camera_params = params[:n_cameras * P].reshape((n_cameras, P))
        points_3d = params[n_cameras * P:].reshape((n_points, 3))
        return camera_params, points_3d

    # ------------------------------------------------------------------ solvers
    def _weights_or_none(self):
        w = np.asarray(self.pointWeights).reshape(-1)
        if w.size and np.all(w == 1):
            return None          # unit weights: the kernels skip the multiply and the 8 B/obs read
        return w.astype(np.float64)

    def _fixed_mask(self, n_points):
        """points3Dfixed -> boolean mask, only when LASERCALIB_SBA_USE_FIXED=1 (the reference never reads the attribute)."""
        if os.environ.get("LASERCALIB_SBA_USE_FIXED", "0") in ("", "0") or self.points3Dfixed is None:
            return None
        f = np.asarray(self.points3Dfixed)
        if f.dtype == bool:
            if f.shape != (n_points,):
                raise ValueError("a boolean points3Dfixed must have one entry per 3-D point")
            return f
        mask = np.zeros(n_points, dtype=bool)
        mask[f.astype(np.int64).reshape(-1)] = True
        return mask

    @staticmethod
    def _loss():
        loss = os.environ.get("LASERCALIB_SBA_LOSS", "linear")
        return (loss, float(os.environ.get("LASERCALIB_SBA_F_SCALE", "1.0"))) if loss != "linear" else None

    def _apply_extensions(self, prob, n_points, mask=None):
        mask = self._fixed_mask(n_points) if mask is None else mask
        if mask is not None:
            prob.set_fixed_points(mask)
        loss = self._loss()
        if loss is not None:
            prob.set_robust_loss(*loss)

    def _solve(self, mode, ftol, verbose=2, xtol=1e-8, gtol=1e-8, max_nfev=None):
        cams = np.ascontiguousarray(self.cameraArray, dtype=np.float64)
        pts = np.ascontiguousarray(self.points3D, dtype=np.float64)
        from . import dist
        if dist.sharding_requested():
            return dist.solve_sharded(self, mode, ftol, xtol, gtol, max_nfev, verbose, _env_dtype(), _env_device())
        first = None
        if _env_mixed():
            with _native.Problem(cams, pts, self.points2D, self.cameraIndices, self.point2DIndices,
                                 weights=self._weights_or_none(), dtype=_native.dtype_code("f32"), device=_env_device()) as prob:
                self._apply_extensions(prob, pts.shape[0])
                opts = prob.make_opts(ftol=ftol, xtol=xtol, gtol=gtol, max_nfev=max_nfev or 0, mode=mode, verbose=0)
                cams, pts, rep1, log1 = prob.solve_lm(opts)
            first = (rep1, log1)
            if max_nfev:
                max_nfev = max(1, int(max_nfev) - int(rep1.nfev))
        with _native.Problem(cams, pts, self.points2D, self.cameraIndices, self.point2DIndices,
                             weights=self._weights_or_none(), dtype=_env_dtype(), device=_env_device()) as prob:
            self._apply_extensions(prob, pts.shape[0])
            opts = prob.make_opts(ftol=ftol, xtol=xtol, gtol=gtol, max_nfev=max_nfev or 0, mode=mode, verbose=verbose)
            cams_opt, pts_opt, rep, log = prob.solve_lm(opts)
        fvec = None                    # res.fun is evaluated on first access (SBAResult)
        if first is not None:          # one report over both stages: counts add up, the table runs on
            rep1, log1 = first
            for row in log:
                row.iteration += rep1.iterations       # (not len(log1): the device log is capped at LOG_CAP rows)
                row.nfev += rep1.nfev - 1
            rep.initial_cost = rep1.initial_cost
            rep.nfev += rep1.nfev - 1          # (the fp64 stage's first evaluation is the fp32 stage's last point)
            rep.njev += rep1.njev - 1
            rep.iterations += rep1.iterations
            log = list(log1) + list(log)
        return self._package(mode, cams_opt, pts_opt, rep, log, fvec, verbose)

    def _package(self, mode, cams_opt, pts_opt, rep, log, fvec, verbose):
        C_, N_ = cams_opt.shape[0], pts_opt.shape[0]
        if mode == _native.MODE_POINTS_ONLY:
            x = pts_opt.ravel().copy()
        elif mode == _native.MODE_SHARED_INTR:      # parameter order of pySBA.py:313
            x = np.hstack((cams_opt[0, 6:9], cams_opt[:, :6].ravel(), cams_opt[:, 9:].ravel(), pts_opt.ravel()))   # 13-parameter rows: the tail is [p1, p2, cx, cy]
        else:
            x = np.hstack((cams_opt.ravel(), pts_opt.ravel()))
        message = TERMINATION_MESSAGES[rep.status]
        _print_table(log, rep.initial_cost, rep, message, verbose)
        ci, pi = np.asarray(self.cameraIndices), np.asarray(self.point2DIndices)
        uv, w = self.points2D, self._weights_or_none()
        dt, dev = _env_dtype(), _env_device()

        def make_jac():
            with _native.Problem(cams_opt, pts_opt, uv, ci, pi, weights=w, dtype=dt, device=dev) as prob:
                _, Jc, Jp = prob.residual_jacobian()
            return assemble_jacobian(Jc, Jp, ci, pi, C_, N_, points_only=(mode == _native.MODE_POINTS_ONLY),
                                     shared_intrinsics=(mode == _native.MODE_SHARED_INTR))

        def make_fun():
            with _native.Problem(cams_opt, pts_opt, uv, ci, pi, weights=w, dtype=dt, device=dev) as prob:
                self._apply_extensions(prob, N_)
                return prob.residual()[0]

        res = SBAResult(x=x, cost=rep.cost, optimality=rep.optimality,
                        active_mask=np.zeros_like(x), nfev=int(rep.nfev), njev=int(rep.njev),
                        status=int(rep.status), message=message, success=rep.status > 0)
        if fvec is not None:
            dict.__setitem__(res, "fun", fvec)
        res.set_makers(None if fvec is not None else make_fun, make_jac)
        return res, cams_opt, pts_opt

    def bundleAdjust(self, ftol=1e-4):
        """Bundle-adjust all cameras and all 3-D points (pySBA.py:132-147).

        Rebinds self.cameraArray / self.points3D to new arrays and returns the result object.
        """
        res, cams, pts = self._solve(_native.MODE_FULL, ftol)
        self.cameraArray = cams
        self.points3D = pts
        return res

    def bundleAdjust_nocam(self, ftol=1e-7):
        """Optimise the 3-D points with the cameras held fixed (pySBA.py:237-250)."""
        res, _cams, pts = self._solve(_native.MODE_POINTS_ONLY, ftol)
        self.points3D = pts
        return res

    def fun_nocam(self, params, camera_params, n_points, camera_indices, point_indices, points_2d, pointWeights):
        """pySBA.py:228-235."""
        points_3d = params.reshape((n_points, 3))
        points_proj = self.project(points_3d[point_indices], camera_params[camera_indices])
        return (pointWeights * (points_proj - points_2d)).ravel()

    def bundle_adjustment_sparsity_nocam(self, numPoints, pointIndices):
        """pySBA.py:216-226."""
        m = pointIndices.size * 2
        obs = np.arange(pointIndices.size)
        cols = (pointIndices[:, None] * 3 + np.arange(3)[None, :])
        rows = np.repeat(2 * obs, 3)
        cols = cols.ravel()
        A = coo_matrix((np.ones(2 * rows.size, dtype=int), (np.concatenate([rows, rows + 1]), np.concatenate([cols, cols]))),
                       shape=(m, numPoints * 3))
        A.sum_duplicates()
        A.data[:] = 1
        return lil_matrix(A)

    def getResiduals(self):
        """Residuals at the current parameters (pySBA.py:207-213).

        Like the reference, the weight vector passed here has shape (M,), which numpy cannot
        broadcast against the (M,2) residual unless M is 1 or 2 -- the same ValueError results.
        """
        numCameras = self.cameraArray.shape[0]
        numPoints = self.points3D.shape[0]
        x0 = np.hstack((self.cameraArray.ravel(), self.points3D.ravel()))
        return self.fun(x0, numCameras, numPoints, self.cameraIndices, self.point2DIndices, self.points2D,
                        np.full_like(self.point2DIndices, 1))

    # -- variants whose residual is defined on other parameterisations --------------------------
    def fun_camonly(self, params, n_cameras, n_points, camera_indices, point_indices, points_2d, pointWeights, points_3d):
        """Cameras-only residual; squares the pixel error like the reference (pySBA.py:151-156)."""
        camera_params = params.reshape(n_cameras, self._P)
        points_proj = self.project(points_3d[point_indices], camera_params[camera_indices])
        return (pointWeights * (points_proj - points_2d) ** 2).ravel()

    def fun_transform_points_3d(self, params, numCameras, n_points, camera_params, camera_indices, point_indices,
                                points_2d, pointWeights, points_3d):
        """3x4 affine on the points, squared pixel error (pySBA.py:176-187)."""
        T = np.vstack((params.reshape(3, 4), [0, 0, 0, 1]))
        homog = np.vstack((points_3d.transpose(), np.ones(shape=(1, n_points))))
        moved = np.dot(T, homog).transpose()[:, :3]
        points_proj = self.project(moved[point_indices], camera_params[camera_indices])
        return (pointWeights * (points_proj - points_2d) ** 2).ravel()

    def fun_sharedcam(self, params, n_cameras, n_points, camera_indices, point_indices, points_2d, pointWeights):
        """Shared (f,k1,k2) parameterisation (pySBA.py:277-295)."""
        nI, nE, nC = 3, 6, self._P - 9
        nCamParams = n_cameras * (nE + nC) + nI
        shared = params[:nI]
        extr = params[nI:nI + n_cameras * nE].reshape((n_cameras, nE))
        centre = params[nI + n_cameras * nE:nCamParams].reshape((n_cameras, nC))
        camera_params = np.concatenate((extr, np.tile(shared, (n_cameras, 1)), centre), axis=1)
        points_3d = params[nCamParams:].reshape((n_points, 3))
        points_proj = self.project(points_3d[point_indices], camera_params[camera_indices])
        return (pointWeights * (points_proj - points_2d)).ravel()

    def bundle_adjustment_sparsity_sharedcam(self, numCameras, numPoints, cameraIndices, pointIndices):
        """pySBA.py:252-275."""
        nI, nE, nC = 3, 6, self._P - 9
        nCamParams = numCameras * (nE + nC) + nI
        obs = np.arange(cameraIndices.size)
        cols = np.concatenate([
            np.tile(np.arange(nI), (obs.size, 1)),
            nI + cameraIndices[:, None] * nE + np.arange(nE)[None, :],
            nI + numCameras * nE + cameraIndices[:, None] * nC + np.arange(nC)[None, :],
            nCamParams + pointIndices[:, None] * 3 + np.arange(3)[None, :]], axis=1)
        rows = np.repeat(2 * obs, cols.shape[1])
        cols = cols.ravel()
        A = coo_matrix((np.ones(2 * rows.size, dtype=int), (np.concatenate([rows, rows + 1]), np.concatenate([cols, cols]))),
                       shape=(cameraIndices.size * 2, nCamParams + numPoints * 3))
        A.sum_duplicates()
        A.data[:] = 1
        return lil_matrix(A)

    def _solve_sq(self, mode, ftol, verbose=2):
        """Squared-pixel-error variants (pySBA.py:151-205): scipy defaults x_scale=1, xtol=gtol=1e-8."""
        cams = np.ascontiguousarray(self.cameraArray, dtype=np.float64)
        pts = np.ascontiguousarray(self.points3D, dtype=np.float64)
        w = self._weights_or_none()
        with _native.Problem(cams, pts, self.points2D, self.cameraIndices, self.point2DIndices, weights=w,
                             dtype=_env_dtype(), device=_env_device()) as prob:
            opts = prob.make_opts(ftol=ftol, xtol=1e-8, gtol=1e-8, mode=mode, verbose=verbose)
            cams_opt, pts_opt, rep, log = prob.solve_lm(opts)
            theta = prob.get_transform() if mode == _native.MODE_TRANSFORM_SQ else None
            # residual of the reference's function: w * (pixel error)^2, from the device's w * (pixel error)
            r, _ = prob.residual(np.hstack((cams_opt.ravel(), pts_opt.ravel())))
        wv = np.ones(r.size // 2) if w is None else w
        with np.errstate(divide="ignore", invalid="ignore"):
            fvec = np.where(np.repeat(wv, 2) != 0, r * r / np.repeat(wv, 2), 0.0)
        x = cams_opt.ravel().copy() if mode == _native.MODE_CAMS_ONLY_SQ else theta
        message = TERMINATION_MESSAGES[rep.status]
        _print_table(log, rep.initial_cost, rep, message, verbose)
        ci, pi = np.asarray(self.cameraIndices), np.asarray(self.point2DIndices)
        uv, dt, dev = self.points2D, _env_dtype(), _env_device()

        def make_jac():
            """Jacobian of the reference's squared residual rho = w * Delta^2 at the solution (scipy returns it dense for
            these variants, least_squares.py:950-961):  d rho = 2 (r / w) * (w dDelta/dtheta), with r and the blocks
            w dDelta/d(cam), w dDelta/dX from the device Jacobian kernel."""
            with _native.Problem(cams_opt, pts_opt, uv, ci, pi, weights=w, dtype=dt, device=dev) as prob:
                rr, Jc, Jp = prob.residual_jacobian()
            with np.errstate(divide="ignore", invalid="ignore"):
                two_delta = np.where(np.repeat(wv, 2) != 0, 2.0 * rr / np.repeat(wv, 2), 0.0).reshape(-1, 2)
            M_, P_ = ci.size, cams_opt.shape[1]
            if mode == _native.MODE_CAMS_ONLY_SQ:
                J = np.zeros((2 * M_, P_ * cams_opt.shape[0]))
                rows = np.arange(M_)
                for comp in range(2):
                    blk = two_delta[:, comp, None] * Jc[:, comp, :]
                    J[(2 * rows + comp)[:, None], ci[:, None] * P_ + np.arange(P_)[None, :]] = blk
                return J
            Xh = np.hstack((pts[pi], np.ones((M_, 1))))                       # d(A X + b)/d theta[k, :] = [X, 1]
            J = np.einsum("mc,mck,mj->mckj", two_delta, Jp, Xh).reshape(2 * M_, 12)
            return J

        res = SBAResult(x=x, cost=rep.cost, fun=fvec, optimality=rep.optimality, active_mask=np.zeros_like(x),
                        nfev=int(rep.nfev), njev=int(rep.njev), status=int(rep.status), message=message,
                        success=rep.status > 0)
        res.set_makers(None, make_jac)
        return res, cams_opt, pts_opt

    def bundle_adjustment_camonly(self, ftol=1e-4):
        """Optimise the cameras with the points held fixed, on the SQUARED pixel error (pySBA.py:151-173)."""
        res, cams, _pts = self._solve_sq(_native.MODE_CAMS_ONLY_SQ, ftol)
        self.cameraArray = cams
        return res

    def bundleAdjust_transform_points_3d(self, ftol=1e-3):
        """Fit one 3x4 affine applied to all 3-D points, cameras fixed, squared pixel error (pySBA.py:176-205).

        res.x holds the 12 parameters (row-major 3x4, starting from the identity); self.points3D is replaced by the
        transformed points like in the reference (pySBA.py:197-204).
        """
        res, _cams, pts = self._solve_sq(_native.MODE_TRANSFORM_SQ, ftol)
        self.points3D = pts
        return res

    def bundleAdjust_sharedcam(self, ftol=1e-6):
        """Bundle adjustment with one (f, k1, k2) shared by all cameras (pySBA.py:297-325).

        The shared values start from the mean over cameras (pySBA.py:309); on the device the tied parameters
        collapse the reduced camera system to 3 + 8*n_cameras unknowns (csrc/sba_lm_kernels.hpp: k_tie_system).
        """
        cams0 = np.array(self.cameraArray, dtype=np.float64, copy=True)
        cams0[:, 6:9] = np.mean(cams0[:, 6:9], axis=0)
        keep = self.cameraArray
        self.cameraArray = cams0
        try:
            res, cams, pts = self._solve(_native.MODE_SHARED_INTR, ftol)
        except Exception:
            self.cameraArray = keep
            raise
        self.cameraArray = cams
        self.points3D = pts
        return res


# pickles written by scripts/calibrate_camera.py:86-88 must load wherever `lasercalib.pySBA` resolves
PySBA.__module__ = "lasercalib.pySBA"


def assemble_jacobian(Jc, Jp, cam_idx, pt_idx, n_cams, n_pts, points_only=False, shared_intrinsics=False):
    """Blocks (M,2,11)/(M,2,3) -> the CSR matrix scipy would return as ``res.jac`` (pySBA.py:110-116 layout;
    pySBA.py:252-275 column layout for the shared-intrinsics variant)."""
    M = cam_idx.shape[0]
    P = Jc.shape[2]                      # 11, or 13 with tangential distortion
    if shared_intrinsics:
        ncp = 3 + (P - 3) * n_cams
        e = np.arange(P)
        ccol = np.where(e[None, :] < 6, 3 + 6 * cam_idx[:, None] + e[None, :],
                        np.where(e[None, :] < 9, e[None, :] - 6, 3 + 6 * n_cams + (P - 9) * cam_idx[:, None] + (e[None, :] - 9)))
        cols = np.concatenate([ccol, ncp + pt_idx[:, None] * 3 + np.arange(3)[None, :]], axis=1)
        data = np.concatenate([Jc, Jp], axis=2).reshape(2 * M, P + 3)
        width, ncol = P + 3, ncp + n_pts * 3
    elif points_only:
        cols = (pt_idx[:, None] * 3 + np.arange(3)[None, :])
        data = Jp.reshape(2 * M, 3)
        width, ncol = 3, n_pts * 3
    else:
        cols = np.concatenate([cam_idx[:, None] * P + np.arange(P)[None, :],
                               n_cams * P + pt_idx[:, None] * 3 + np.arange(3)[None, :]], axis=1)
        data = np.concatenate([Jc, Jp], axis=2).reshape(2 * M, P + 3)
        width, ncol = P + 3, n_cams * P + n_pts * 3
    cols = np.repeat(cols[:, None, :], 2, axis=1).reshape(2 * M, width)
    indptr = np.arange(0, 2 * M * width + 1, width)
    return csr_matrix((data.ravel(), cols.ravel(), indptr), shape=(2 * M, ncol))
