"""CPU, world_size 2 over gloo: the sharded LM driver (lasercalib_amd.dist.run_lm) with the numpy model engine
reproduces the single-rank solve and the reference's optimum.  This is the N > 1 control path the GPU run uses,
with the HIP phase calls replaced by their numpy model."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tag, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lasercalib_amd import dist as sdist
        from oracle import lm_schur_model as model
        g = np.load(os.path.join(ROOT, "tests", "golden", "f4_solves.npz"))
        cams, pts, uv, ci, pi = (g[f"{tag}_{k}"] for k in ("cams0", "pts0", "uv", "ci", "pi"))
        assert sdist.world_size() == world and sdist.rank() == rank
        sh = sdist.make_shard(pts, uv, ci, pi, None, world, rank)
        eng = model.ModelEngine(cams, sh["pts"], sh["uv"], sh["ci"], sh["pi_local"])
        eng.begin(ftol=1e-4)
        comm = sdist.TorchComm()
        status, iters = sdist.run_lm(eng, comm)
        parts = comm.all_gather_var((sh["p0"], eng.pts, 0.5 * float(np.sum(eng.res ** 2))))
        full = np.empty_like(pts)
        cost = 0.0
        for p0, pl, c in parts:
            full[p0:p0 + pl.shape[0]] = pl
            cost += c
        q.put((rank, status, iters, eng.cams.copy(), full, cost, eng.nfev))
    except BaseException as e:          # never leave the parent waiting on the queue
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tag", ["cfg1", "sparse"])
def test_two_rank_sharded_lm_matches_single_rank_and_reference(tag):
    from oracle import lm_schur_model as model
    g = np.load(os.path.join(ROOT, "tests", "golden", "f4_solves.npz"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tag, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    assert all(len(r) > 2 for r in results), results
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, s0, it0, cams0, pts0, cost0, nfev0), (_, s1, it1, cams1, pts1, cost1, nfev1) = results
    # both ranks take identical decisions and hold identical cameras / gathered points
    assert s0 == s1 == 2 and it0 == it1 and nfev0 == nfev1
    assert np.array_equal(cams0, cams1) and np.array_equal(pts0, pts1) and cost0 == cost1
    # same trajectory as the unsharded model (summation order differs => tiny rounding differences)
    eng = model.ModelEngine(*(g[f"{tag}_{k}"] for k in ("cams0", "pts0", "uv", "ci", "pi")))
    out = model.run_lm_single(eng, ftol=1e-4)
    assert out["iterations"] == it0 and abs(out["cost"] - cost0) <= 1e-9 * cost0
    assert np.max(np.abs(out["cams"] - cams0)) <= 1e-6 and np.max(np.abs(out["pts"] - pts0)) <= 1e-6
    ref = float(g[f"{tag}_loose_cost"])
    assert cost0 <= ref * (1 + 1e-9) and ref - cost0 <= 1e-5 * ref


# ----------------------------------------------------------------------------- termination must be rank-invariant
def _worker_budget(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lasercalib_amd import dist as sdist
        from lasercalib_amd.synth import make_rig
        from oracle import lm_schur_model as model
        # sparse visibility: shards balanced by observation count own DIFFERENT numbers of points
        rig = make_rig(5, 90, seed=3, visibility=0.45)
        sh = sdist.make_shard(rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], None, world, rank)
        n_global = sdist._n_params(0, 11, 5, 90)
        eng = model.ModelEngine(rig["cams0"], sh["pts"], sh["uv"], sh["ci"], sh["pi_local"])
        eng.begin(ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=4)          # explicit, job-wide budget
        comm = sdist.TorchComm()
        status, iters = sdist.run_lm(eng, comm)
        # the "one rank fails, everybody raises" helper
        try:
            sdist.raise_everywhere(comm, "Residuals are not finite in the initial point." if rank == 1 else None)
            raised = None
        except ValueError as e:
            raised = str(e)
        q.put((rank, status, iters, eng.nfev, sh["pts"].shape[0], n_global, raised, sdist.sharding_requested()))
    except BaseException as e:
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_budget_and_failures_are_job_wide_not_per_shard():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 90)
    procs = [ctx.Process(target=_worker_budget, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    assert all(len(r) > 2 for r in results), results
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = results
    assert r0[4] != r1[4]                                  # unequal shards ...
    assert r0[1] == r1[1] == 0 and r0[2] == r1[2] and r0[3] == r1[3] == 4     # ... same status 0 at the same evaluation count
    assert r0[5] == r1[5] == 11 * 5 + 3 * 90               # default budget basis: parameters of the whole job
    assert r0[6] == r1[6] and "not finite" in r0[6] and "rank 1" in r0[6]      # both ranks raise the same error
    assert r0[7] is False and r1[7] is False               # a live process group alone never turns on sharding
