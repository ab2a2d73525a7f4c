"""Multi-GPU bundle adjustment: one process per GPU, points sharded, cameras replicated.

Every observation touches one camera block and one point block (pySBA.py:110-116), so the
point list is cut into contiguous ranges (balanced by observation count, at point boundaries)
and each rank owns its points, their observations, their 3x3 solves and back-substitution.
The only data-path exchange per LM trial is an all-reduce (sum) of the reduced camera system
``[S | rhs | diag U | g_c | cost]`` ((11C)^2 + 33C + 1 doubles) plus an all-gather of 8 scalars
per rank; every rank then solves the same camera system redundantly, so no broadcast is needed
and all ranks take bit-identical accept/reject decisions.

Product path (one GPU per rank): ``solve_sharded`` hands libsba_hip.so an RCCL communicator (``Problem.comm_init``)
and calls ``sba_solve_lm`` once -- the library runs the whole sharded loop itself, with its two collectives per trial
(``ncclAllReduce`` of the packed upper triangle, ``ncclAllGather`` of the scalars) enqueued on its own stream.
torch.distributed only carries the 128-byte ncclUniqueId before and the result shards after the solve.

``run_lm`` below is the same loop written against an abstract engine + communicator: the CPU tests drive it with a numpy
model over gloo (the logic check of the N > 1 path), and ``LASERCALIB_SBA_COMM=torch`` drives the HIP phase API with it
(what the one-card rehearsal in tests/test_gpu_dist.py uses, because RCCL refuses two ranks on one device).

Sharding is opt-in: ``PySBA`` methods shard only when ``LASERCALIB_SBA_SHARD=1`` is set AND a process group is up; every
rank must then call them with the same full problem (checked with a hash of the index arrays).
"""
from __future__ import annotations

import os
import sys

import numpy as np

NSCALARS = 8


# ----------------------------------------------------------------------------- process group helpers
def _td():
    """torch.distributed if torch is already imported and a process group is up, else None."""
    torch = sys.modules.get("torch")
    if torch is None:
        return None
    d = torch.distributed
    if d.is_available() and d.is_initialized():
        return d
    return None


def world_size():
    d = _td()
    return d.get_world_size() if d else 1


def sharding_requested():
    """PySBA shards a solve only on explicit request: a script that merely runs under torchrun, or calls the solver on one
    rank only, must not fall into collectives by accident."""
    return os.environ.get("LASERCALIB_SBA_SHARD", "0") not in ("", "0") and world_size() > 1


def rank():
    d = _td()
    return d.get_rank() if d else 0


# ----------------------------------------------------------------------------- sharding
def shard_bounds(pt_start, n_ranks):
    """Cut points [0,N) into n_ranks contiguous ranges with ~equal observation counts.

    pt_start: (N+1,) CSR offsets of the point-major observation list.  Returns (n_ranks+1,) point
    boundaries; cuts fall on point boundaries so no point straddles two ranks.
    """
    pt_start = np.asarray(pt_start, dtype=np.int64)
    N = pt_start.shape[0] - 1
    M = int(pt_start[-1])
    targets = (np.arange(1, n_ranks, dtype=np.float64) * M / n_ranks)
    cuts = np.searchsorted(pt_start, targets, side="left")
    # choose the nearer of the two neighbouring point boundaries
    lo = np.clip(cuts - 1, 0, N)
    cuts = np.where(np.abs(pt_start[np.clip(cuts, 0, N)] - targets) <= np.abs(pt_start[lo] - targets), cuts, lo)
    b = np.concatenate([[0], np.clip(cuts, 0, N), [N]]).astype(np.int64)
    return np.maximum.accumulate(b)


def make_shard(points3D, points2D, cam_idx, pt_idx, weights, n_ranks, r):
    """Slice rank r's share out of the full problem.

    Returns dict(pts, uv, ci, pi_local, w, p0, p1, obs_index) where obs_index are the positions of
    the shard's observations in the caller's observation order.
    """
    pt_idx = np.asarray(pt_idx, dtype=np.int64)
    N = points3D.shape[0]
    order = np.argsort(pt_idx, kind="stable") if np.any(np.diff(pt_idx) < 0) else np.arange(pt_idx.size)
    counts = np.bincount(pt_idx, minlength=N)
    pt_start = np.concatenate([[0], np.cumsum(counts)])
    b = shard_bounds(pt_start, n_ranks)
    p0, p1 = int(b[r]), int(b[r + 1])
    sel = order[pt_start[p0]:pt_start[p1]]
    w = None if weights is None else np.asarray(weights, dtype=np.float64).reshape(-1)[sel]
    return dict(pts=np.ascontiguousarray(points3D[p0:p1], dtype=np.float64),
                uv=np.ascontiguousarray(np.asarray(points2D)[sel], dtype=np.float64),
                ci=np.ascontiguousarray(np.asarray(cam_idx)[sel], dtype=np.int64),
                pi_local=pt_idx[sel] - p0, w=w, p0=p0, p1=p1, obs_index=sel, bounds=b)


# ----------------------------------------------------------------------------- collectives
class TorchComm:
    """Sum-all-reduce and row all-gather over torch.distributed (nccl == RCCL on ROCm, or gloo)."""

    def __init__(self):
        import torch
        self.torch = torch
        self.d = torch.distributed
        self.n = self.d.get_world_size()
        self.r = self.d.get_rank()

    def _t(self, x):
        return x if isinstance(x, self.torch.Tensor) else self.torch.from_numpy(x)

    def all_reduce_sum(self, x):
        self.d.all_reduce(self._t(x), op=self.d.ReduceOp.SUM)
        return x

    def all_gather_rows(self, x):
        t = self._t(x).reshape(-1)
        out = self.torch.empty(self.n * t.numel(), dtype=t.dtype, device=t.device)
        self.d.all_gather_into_tensor(out, t)
        return out if isinstance(x, self.torch.Tensor) else out.numpy().reshape(self.n, -1)

    def all_gather_var(self, a):
        """All-gather numpy arrays whose leading dimension differs per rank (host side, small)."""
        objs = [None] * self.n
        self.d.all_gather_object(objs, a)
        return objs


class SoloComm:
    n, r = 1, 0

    def all_reduce_sum(self, x):
        return x

    def all_gather_rows(self, x):
        return x if not isinstance(x, np.ndarray) else x.reshape(1, -1)

    def all_gather_var(self, a):
        return [a]


# ----------------------------------------------------------------------------- the loop
def run_lm(engine, comm, max_iter=0):
    """Engine-agnostic sharded LM loop.  Returns (status, iterations).

    ``engine.batch`` iterations are enqueued between two polls of the engine's state: the HIP engine decides
    accept / reject / terminate on the device (later launches of a finished solve are no-ops there, and the
    collectives of such a tail still match on every rank because all ranks take identical decisions), so the
    host does not synchronise with the GPU per iteration.
    """
    status, it = None, 0
    batch = max(1, int(getattr(engine, "batch", 1)))
    while status is None:
        todo = batch if not max_iter else max(1, min(batch, max_iter - it))
        for _ in range(todo):
            engine.linearize()                       # skipped inside the engine after a rejected step
            E = engine.form_reduced()
            comm.all_reduce_sum(E)
            sc = engine.solve_trial(E)
            sc_all = comm.all_gather_rows(sc)
            engine.decide_async(sc_all, comm.n)
        status, it = engine.poll()
        if status is None and max_iter and it >= max_iter:
            status = 0
    return status, it


# ----------------------------------------------------------------------------- product engine
class HipEngine:
    """libsba_hip.so phase calls on this rank's shard; exchange buffers are torch CUDA tensors."""

    def __init__(self, cams, shard, dtype, device, opts_kwargs, before_begin=None):
        import torch
        from . import _native
        self.torch = torch
        torch.cuda.set_device(device)
        stream = torch.cuda.current_stream().cuda_stream
        self.prob = _native.Problem(cams, shard["pts"], shard["uv"], shard["ci"], shard["pi_local"],
                                    weights=shard["w"], dtype=dtype, device=device, stream=stream)
        self.opts = self.prob.make_opts(**opts_kwargs)
        self.always_relinearize = bool(opts_kwargs.get("always_relinearize", False))
        dev = torch.device("cuda", device)
        self.E = torch.empty(self.prob.exchange_size(), dtype=torch.float64, device=dev)
        self.sc = torch.empty(NSCALARS, dtype=torch.float64, device=dev)
        self.batch = 1 if opts_kwargs.get("profile") else 4
        self.begin_error = None
        if before_begin is not None:
            before_begin(self.prob)
        try:
            self.prob.lm_begin(self.opts)
        except ValueError as e:          # "Residuals are not finite in the initial point." on THIS shard
            self.begin_error = str(e)

    def linearize(self):
        self.prob.lm_linearize()

    def form_reduced(self):
        self.prob.lm_form_reduced(self.E.data_ptr())
        return self.E

    def solve_trial(self, E):
        self.prob.lm_solve_trial(E.data_ptr(), self.sc.data_ptr())
        return self.sc

    def decide_async(self, sc_all, n_ranks):
        self.prob.lm_decide_async(sc_all.data_ptr(), n_ranks)

    def poll(self):
        status, iters = self.prob.lm_poll()
        return (None if status < 0 else status), iters

    def finish(self):
        return self.prob.lm_finish()

    def close(self):
        self.prob.close()


def _bounded_barrier(d, seconds=60.0):
    """Barrier that gives up: gloo has monitored_barrier(timeout); other backends get a plain barrier.  Never raises."""
    import datetime
    try:
        if d.get_backend() == "gloo":
            d.monitored_barrier(timeout=datetime.timedelta(seconds=seconds))
        else:
            d.barrier()
    except Exception:      # noqa: BLE001  (a peer is gone: nothing left to protect)
        pass


def raise_everywhere(comm, local_error):
    """A failure on ANY rank (e.g. a non-finite initial cost in one shard) becomes the same ValueError on ALL ranks, so
    nobody walks into a collective its peers will never join.  local_error: message or None."""
    flags = comm.all_gather_var(local_error)
    bad = [(r, f) for r, f in enumerate(flags) if f is not None]
    if bad:
        raise ValueError(f"{bad[0][1]} (rank {bad[0][0]})")


def _n_params(mode, P, C, N_total):
    """Number of free parameters of the whole job (scipy's default max_nfev is 100 x this, trf.py:437-438)."""
    from . import _native
    if mode == _native.MODE_POINTS_ONLY:
        return 3 * N_total
    if mode == _native.MODE_SHARED_INTR:
        return 3 + (P - 3) * C + 3 * N_total
    return P * C + 3 * N_total


def _same_problem_everywhere(comm, sba, cams, pts):
    """Cheap guard against per-rank data: every rank must hold the same full problem."""
    import hashlib
    h = hashlib.sha1()
    for a in (np.asarray(sba.cameraIndices), np.asarray(sba.point2DIndices)):
        h.update(np.ascontiguousarray(a, dtype=np.int64).tobytes())
    h.update(np.asarray([cams.shape[0], cams.shape[1], pts.shape[0]], dtype=np.int64).tobytes())
    digests = comm.all_gather_var(h.hexdigest())
    if len(set(digests)) != 1:
        raise ValueError("sharded solve: the ranks do not hold the same problem (index arrays / sizes differ); "
                         "every rank must call the solver with the full observation list")


def solve_sharded(sba, mode, ftol, xtol, gtol, max_nfev, verbose, dtype, device, max_iter=0,
                  always_relinearize=False):
    """PySBA._solve for world_size > 1.  Every rank calls this with the same full problem and gets
    the same full result back."""
    from . import _native
    comm = TorchComm()
    cams = np.ascontiguousarray(sba.cameraArray, dtype=np.float64)
    pts = np.ascontiguousarray(sba.points3D, dtype=np.float64)
    _same_problem_everywhere(comm, sba, cams, pts)
    w = sba._weights_or_none()
    shard = make_shard(pts, sba.points2D, sba.cameraIndices, sba.point2DIndices, w, comm.n, comm.r)
    # the evaluation budget is a property of the whole job, not of a shard: ranks own different numbers of points
    max_nfev = int(max_nfev) if max_nfev else 100 * _n_params(mode, cams.shape[1], cams.shape[0], pts.shape[0])
    kw = dict(ftol=ftol, xtol=xtol, gtol=gtol, max_nfev=max_nfev, mode=mode, verbose=0, max_iter=max_iter,
              always_relinearize=always_relinearize)
    # who carries the per-trial exchanges: "rccl" (the library binds RCCL: default on the nccl backend), "ipc" (the library's
    # one-shot exchange through peer-mapped buffers, csrc/sba_ipc.hpp: no RCCL launch per trial; also works with several ranks
    # on ONE device), "torch" (the phase C ABI with torch.distributed collectives: default elsewhere)
    mode_comm = os.environ.get("LASERCALIB_SBA_COMM", "rccl" if comm.d.get_backend() == "nccl" else "torch")
    use_rccl = mode_comm in ("rccl", "ipc")
    if use_rccl:
        # the library owns the exchanges: hand it a communicator (or the peers' areas) and make ONE call
        ids = [None]
        if mode_comm == "rccl":
            if comm.r == 0:
                try:
                    ids = [_native.comm_unique_id()]
                except Exception as e:      # noqa: BLE001  (RCCL missing / wrong version: the peers are waiting for the broadcast)
                    ids = [("error", f"{type(e).__name__}: {e}")]
            comm.d.broadcast_object_list(ids, src=0)
            if isinstance(ids[0], tuple):
                raise ValueError(f"sba_comm_get_unique_id failed on rank 0: {ids[0][1]}")
        # Everything that can fail on ONE rank (allocation, upload, a bad mask) happens before the first collective the
        # library enters (ncclCommInitRank), and the outcome is agreed over the torch group first: a rank that raised would
        # otherwise leave its peers blocked inside RCCL.  What this cannot cover is a failure inside ncclCommInitRank itself
        # (the peers are already in it) -- RCCL's own timeout is the only way out of that one.
        prob, local_err = None, None
        try:
            prob = _native.Problem(cams, shard["pts"], shard["uv"], shard["ci"], shard["pi_local"], weights=shard["w"],
                                   dtype=dtype, device=device)
            fm = sba._fixed_mask(pts.shape[0])
            sba._apply_extensions(prob, shard["pts"].shape[0], None if fm is None else fm[shard["p0"]:shard["p1"]])
        except Exception as e:      # noqa: BLE001  (whatever it was, every rank has to hear about it)
            local_err = f"{type(e).__name__}: {e}"
        try:
            raise_everywhere(comm, local_err)
        except ValueError:
            if prob is not None:
                prob.close()
            raise
        try:
            if mode_comm == "rccl":
                prob.comm_init(ids[0], comm.r, comm.n)
            else:
                # every rank exports its area; a failure anywhere is agreed before anybody maps anything
                handle, ipc_err = None, None
                try:
                    handle = prob.ipc_export(comm.n)
                except Exception as e:      # noqa: BLE001
                    ipc_err = f"{type(e).__name__}: {e}"
                raise_everywhere(comm, ipc_err)
                handles = comm.all_gather_var(handle)
                try:
                    prob.ipc_attach(comm.r, handles)
                except Exception as e:      # noqa: BLE001
                    ipc_err = f"{type(e).__name__}: {e}"
                raise_everywhere(comm, ipc_err)
            bad = None
            try:
                cams_opt, pts_loc, rep, log = prob.solve_lm(prob.make_opts(**kw))
            except ValueError as e:       # non-finite start: the library all-reduces the initial cost, so every rank lands here
                bad = e
            if bad is not None:
                raise bad
            fvec_loc, _ = prob.residual()
        finally:
            if mode_comm == "ipc":
                # nobody unmaps or frees an area a peer's last kernels may still be reading -- on the failure paths too (a rank
                # that raised would otherwise free memory a slower peer is still spinning on in k_ipc_gate); best effort and
                # bounded, because a peer that died will never arrive
                _bounded_barrier(comm.d)
            prob.close()
        status, cost, opt, cost0 = rep.status, rep.cost, rep.optimality, rep.initial_cost     # whole-job figures already
        parts = comm.all_gather_var((shard["p0"], pts_loc, shard["obs_index"], fvec_loc))
    else:
        eng = HipEngine(cams, shard, dtype, device, kw, before_begin=lambda prob: sba._apply_extensions(
            prob, shard["pts"].shape[0], None if sba._fixed_mask(pts.shape[0]) is None else sba._fixed_mask(pts.shape[0])[shard["p0"]:shard["p1"]]))
        try:
            # a non-finite initial cost on ANY rank must fail the solve on ALL ranks before the first collective of the loop
            raise_everywhere(comm, eng.begin_error)
            status, _ = run_lm(eng, comm, max_iter=max_iter)
            cams_opt, pts_loc, rep = eng.finish()
            gc_loc, gp_loc = eng.prob.get_gradient()
            fvec_loc, cost_loc = eng.prob.residual()
            log = eng.prob.iteration_log()
        finally:
            eng.close()
        gp_max = float(np.max(np.abs(gp_loc))) if gp_loc.size else 0.0
        extra = comm.all_gather_var((cost_loc, gc_loc, gp_max, rep.initial_cost))
        cost = sum(e[0] for e in extra)
        cost0 = sum(e[3] for e in extra)
        gc = sum(e[1] for e in extra)
        if mode == _native.MODE_SHARED_INTR:      # gradient in the tied unknowns, like the single-rank report (f, k1, k2 summed)
            gc_t = np.concatenate([gc[:, 6:9].sum(axis=0), gc[:, :6].ravel(), gc[:, 9:].ravel()])
        elif mode == _native.MODE_POINTS_ONLY:
            gc_t = np.zeros(1)
        else:
            gc_t = gc.ravel()
        opt = max(max(e[2] for e in extra), float(np.max(np.abs(gc_t))))
        parts = comm.all_gather_var((shard["p0"], pts_loc, shard["obs_index"], fvec_loc))
    pts_opt = np.empty_like(pts)
    fvec = np.empty(2 * np.asarray(sba.point2DIndices).size)
    for p0, pl, oi, fv in parts:
        pts_opt[p0:p0 + pl.shape[0]] = pl
        fvec.reshape(-1, 2)[oi] = fv.reshape(-1, 2)
    rep.cost, rep.optimality, rep.status, rep.initial_cost = cost, opt, status, cost0
    res, c, p = sba._package(mode, cams_opt, pts_opt, rep, log, fvec, verbose if comm.r == 0 else 0)
    return res, c, p
