"""GPU: BASELINE configs 4 and 5 as SHARDED jobs -- 64 cameras x 11 parameters and 128 cameras x 13 parameters over 4 and 8 ranks.

The GPU box has one card, RCCL refuses two ranks on one device and the box allows at most 6 processes on the card, so the ranks
are handles of the C ABI spread over 4 worker processes (one or two ranks per process, each rank solving on its own thread and
stream) and the per-trial exchanges go through the library's one-shot exchange (sba_ipc_export / sba_ipc_attach,
csrc/sba_ipc.hpp): peers in another process are mapped with hipIpcOpenMemHandle, peers in the same process are found in the
library's table of local areas.  What runs is exactly what `sba_solve_lm` runs on 8 GPUs -- the packed n(n+1)/2 + 3n + 1 exchange
(2 MB at n = 704, 11 MB at n = 1664), k_ipc_sum_system over 8 copies, the multi-workgroup Cholesky of sba_chol_big.hpp on every
rank, the 8 x 8 trial scalars -- except that the copies travel through one card's memory instead of xGMI.

Asserted: every rank returns the same bits (cameras, cost, nfev, status); the sharded solve follows the single-rank solve
(fp64: nfev equal, cost 1e-9 relative; fp32: the SURVEY 8(d) bar, 1e-4); on the smallest rig the reference's optimum at ftol 1e-4.
The parent process only carries the 64-byte handles between the workers (any channel may: include/sba_hip.h).
"""
import os
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RIGS = {
    # tag: (cameras, points, tangential, visibility, min views, seed)
    "c64_dense": (64, 600, False, 1.0, 2, 11),
    "c64_sparse": (64, 900, False, 0.4, 4, 12),
    "c128t_dense": (128, 400, True, 1.0, 2, 13),
    "c128t_sparse": (128, 640, True, 0.3, 4, 14),
}


def _rig(tag):
    sys.path.insert(0, ROOT)
    from lasercalib_amd.synth import make_rig
    C, N, tang, vis, mv, seed = RIGS[tag]
    rig = make_rig(C, N, seed=seed, visibility=vis, min_cams_per_point=mv, tangential=tang)
    return rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"]


def _worker(proc, n_procs, per_proc, tag, dtype, ftol, q_out, q_in):
    """Ranks proc * per_proc ... of a world of n_procs * per_proc; q_out -> parent, q_in <- parent."""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")       # every rank's stream on its own hardware queue (a gate must never sit in front of a peer's publish)
    try:
        from lasercalib_amd import _native, dist
        world = n_procs * per_proc
        cams0, pts0, uv, ci, pi = _rig(tag)
        ranks = [proc * per_proc + k for k in range(per_proc)]
        shards = [dist.make_shard(pts0, uv, ci, pi, None, world, r) for r in ranks]
        probs = [_native.Problem(cams0, s["pts"], s["uv"], s["ci"], s["pi_local"], dtype=dtype) for s in shards]
        q_out.put(("handles", proc, [p.ipc_export(world) for p in probs]))
        handles = q_in.get(timeout=300)
        for p, r in zip(probs, ranks):
            p.ipc_attach(r, handles)
        out = [None] * per_proc

        def run(k):
            try:
                cams, pts, rep, log = probs[k].solve_lm(probs[k].make_opts(ftol=ftol))
                out[k] = (ranks[k], rep.status, rep.cost, int(rep.nfev), cams, pts, shards[k]["p0"], rep.optimality, int(rep.iterations))
            except BaseException as e:      # noqa: BLE001
                out[k] = (ranks[k], repr(e))
        threads = [threading.Thread(target=run, args=(k,)) for k in range(per_proc)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        q_out.put(("done", proc, out))
        q_in.get(timeout=300)              # every rank everywhere has finished: nobody reads anybody's area any more
        for p in probs:
            p.close()
    except BaseException as e:              # never leave the parent waiting
        q_out.put(("error", proc, repr(e)))
        raise


def _run(tag, dtype, n_procs, per_proc, ftol=1e-4):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    q_ins = [ctx.Queue() for _ in range(n_procs)]
    procs = [ctx.Process(target=_worker, args=(p, n_procs, per_proc, tag, dtype, ftol, q_out, q_ins[p])) for p in range(n_procs)]
    for p in procs:
        p.start()
    try:
        got = {}
        for _ in procs:
            kind, proc, payload = q_out.get(timeout=420)
            assert kind == "handles", (kind, proc, payload)
            got[proc] = payload
        handles = [h for p in range(n_procs) for h in got[p]]
        for q in q_ins:
            q.put(handles)
        results = []
        for _ in procs:
            kind, proc, payload = q_out.get(timeout=600)
            assert kind == "done", (kind, proc, payload)
            results += payload
        for q in q_ins:
            q.put("close")
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    results.sort(key=lambda t: t[0])
    assert all(len(r) > 2 for r in results), results
    return results


def _check_ranks_agree(res, world):
    assert [r[0] for r in res] == list(range(world))
    r0 = res[0]
    for r in res[1:]:
        assert r[1] == r0[1] and r[3] == r0[3] and r[8] == r0[8]
        assert r[2] == r0[2] and r[7] == r0[7]                       # whole-job cost and optimality: the same bits
        assert np.array_equal(r[4], r0[4])                           # replicated cameras: the same bits
    assert r0[1] in (2, 3, 4)
    return r0, np.vstack([r[5] for r in res])


def _single_rank(tag, dtype, ftol=1e-4):
    from lasercalib_amd import _native
    a = _rig(tag)
    with _native.Problem(*a, dtype=dtype) as prob:
        cams, pts, rep, _ = prob.solve_lm(prob.make_opts(ftol=ftol))
    return cams, pts, rep


@pytest.mark.parametrize("tag,dtype,n_procs,per_proc", [
    ("c64_dense", "f32", 4, 2),        # config 4's shape, 8 ranks: bf16 pair kernels per shard, 2 MB exchange, k_chol_big at n = 704
    ("c64_sparse", "f64", 4, 1),       # 4 ranks in fp64 (the class's default dtype), sparse visibility
    ("c64_sparse", "f32", 4, 2),
    ("c128t_dense", "f32", 4, 2),      # config 5's shape, 8 ranks: 13-parameter rows, 11 MB exchange, k_chol_big at n = 1664
    ("c128t_sparse", "f64", 4, 1),
])
def test_configs_4_and_5_sharded_over_4_and_8_ranks(tag, dtype, n_procs, per_proc):
    world = n_procs * per_proc
    res = _run(tag, dtype, n_procs, per_proc)
    r0, pts_all = _check_ranks_agree(res, world)
    cams1, pts1, rep1 = _single_rank(tag, dtype)
    if dtype == "f64":
        assert rep1.nfev == r0[3] and abs(rep1.cost - r0[2]) <= 1e-9 * rep1.cost            # the single-rank trajectory
        assert np.max(np.abs(cams1 - r0[4])) <= 1e-6 and np.max(np.abs(pts1 - pts_all)) <= 1e-6
    else:
        assert abs(rep1.cost - r0[2]) <= 1e-4 * rep1.cost                                   # fp32: the shards sum in another order
    # the points every rank returned are its own slice, in order
    assert pts_all.shape == pts1.shape
    assert [r[6] for r in res] == sorted(r[6] for r in res)


def test_config_4_shape_sharded_matches_the_reference_optimum():
    """The smallest 64-camera rig against scipy (the reference's call) at ftol 1e-4, 8 ranks in fp64: two-sided 1e-5 on the cost."""
    from oracle import sba_oracle as orc
    res = _run("c64_dense", "f64", 4, 2)
    r0, pts_all = _check_ranks_agree(res, 8)
    a = _rig("c64_dense")
    ref, _, _ = orc.bundle_adjust(*a, ftol=1e-4)
    assert r0[2] <= ref.cost * (1 + 1e-9) and ref.cost - r0[2] <= 1e-5 * ref.cost
    x = np.hstack((r0[4].ravel(), pts_all.ravel()))
    f = orc.fun(x, 64, a[1].shape[0], a[3], a[4], a[2], 1.0)
    assert abs(0.5 * float(f @ f) - r0[2]) <= 1e-9 * r0[2]          # the job-wide cost the ranks agreed on IS the reference's fun at the returned x


def _housekeeping_worker(q):
    """Runs in a fresh process (its own HIP runtime with one hardware queue per stream, see _worker)."""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    os.environ["SBA_IPC_TIMEOUT_S"] = "0.5"
    import time
    try:
        from lasercalib_amd import _native, dist
        from lasercalib_amd.synth import make_rig
        rig = make_rig(8, 400, seed=5)
        shards = [dist.make_shard(rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], None, 2, r) for r in range(2)]
        out = {}
        probs = [_native.Problem(rig["cams0"], s["pts"], s["uv"], s["ci"], s["pi_local"], dtype="f64") for s in shards]
        try:
            handles = [p.ipc_export(2) for p in probs]

            def refused(fn):
                try:
                    fn()
                except _native.SbaError as e:
                    return str(e)
                return None
            out["comm_init_after_export"] = refused(lambda: probs[0].comm_init(b"\0" * _native.COMM_ID_BYTES, 0, 1))
            out["second_export"] = refused(lambda: probs[0].ipc_export(2))
            for r, p in enumerate(probs):
                p.ipc_attach(r, handles)                                # both peers live in this process: no hipIpcOpenMemHandle involved
            out["second_attach"] = refused(lambda: probs[0].ipc_attach(0, handles))
            t0 = time.perf_counter()                                    # rank 1 never joins: rank 0's first exchange (lm_begin) gives up
            out["peer_absent"] = refused(lambda: probs[0].solve_lm(probs[0].make_opts(ftol=1e-4)))
            out["peer_absent_seconds"] = time.perf_counter() - t0
            out["after_timeout"] = refused(lambda: probs[0].solve_lm(probs[0].make_opts(ftol=1e-4)))
        finally:
            for p in probs:
                p.close()
        # the same two ranks, both solving (two threads): bit-identical, and the single-rank trajectory
        probs = [_native.Problem(rig["cams0"], s["pts"], s["uv"], s["ci"], s["pi_local"], dtype="f64") for s in shards]
        try:
            handles = [p.ipc_export(2) for p in probs]
            for r, p in enumerate(probs):
                p.ipc_attach(r, handles)
            res = [None, None]

            def run(k):
                res[k] = probs[k].solve_lm(probs[k].make_opts(ftol=1e-4))
            th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            with _native.Problem(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"]) as solo:
                _, _, rep, _ = solo.solve_lm(solo.make_opts(ftol=1e-4))
            out["pair"] = (res[0] is not None and res[1] is not None and bool(np.array_equal(res[0][0], res[1][0])),
                           res[0][2].cost, res[1][2].cost, int(res[0][2].status), int(res[0][2].nfev), rep.cost, int(rep.nfev))
        finally:
            for p in probs:
                p.close()
        q.put(out)
    except BaseException as e:      # noqa: BLE001
        q.put({"error": repr(e)})
        raise


def test_exchange_setup_rules_and_bounded_wait():
    """The one-shot exchange's housekeeping (round 4): peers inside ONE process are found in the library's table of local areas
    (hipIpcOpenMemHandle refuses the exporting process); the two exchanges exclude each other in both directions; a second export or
    attach is refused; a peer that never arrives costs SBA_IPC_TIMEOUT_S (0.5 s here), not a hang, and leaves the handle unusable for
    further exchanges; two ranks of one process that both solve agree bit for bit and follow the single-rank trajectory."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_housekeeping_worker, args=(q,))
    p.start()
    out = q.get(timeout=300)
    p.join(timeout=60)
    assert "error" not in out, out
    assert out["comm_init_after_export"] and "exclusive" in out["comm_init_after_export"]
    assert out["second_export"] and out["second_attach"]
    assert out["peer_absent"] and "did not reach the exchange" in out["peer_absent"] and out["peer_absent_seconds"] < 5.0
    assert out["after_timeout"] and "timed out" in out["after_timeout"]
    same, c0, c1, status, nfev, c_solo, nfev_solo = out["pair"]
    assert same and c0 == c1 and status == 2 and nfev == nfev_solo and abs(c0 - c_solo) <= 1e-9 * c_solo
