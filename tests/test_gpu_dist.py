"""GPU, 2 ranks sharing the one card (gloo carries the collectives; RCCL refuses two ranks on one device):
the sharded product path -- PySBA.bundleAdjust -> dist.solve_sharded -> HipEngine phase calls -- gives every rank
the same result as the single-rank device solve and the reference."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tag, q, dtype="f64"):
    sys.path.insert(0, ROOT)
    os.environ["LASERCALIB_SBA_DTYPE"] = dtype
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", LASERCALIB_SBA_DEVICE="0", LASERCALIB_SBA_SHARD="1", LASERCALIB_SBA_COMM="torch")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lasercalib_amd.pySBA import PySBA
        g = np.load(os.path.join(ROOT, "tests", "golden", "f4_solves.npz"))
        sba = PySBA(g[f"{tag}_cams0"].copy(), g[f"{tag}_pts0"].copy(), g[f"{tag}_uv"], g[f"{tag}_ci"], g[f"{tag}_pi"])
        res = sba.bundleAdjust(1e-4)
        q.put((rank, res.status, res.cost, res.nfev, sba.cameraArray.copy(), sba.points3D.copy(), res.fun.copy(), res.optimality))
    except BaseException as e:          # never leave the parent waiting on the queue
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tag", ["sparse", "mid"])
def test_sharded_gpu_solve_two_ranks(tag):
    import torch.multiprocessing as mp
    from lasercalib_amd import _native
    from oracle import sba_oracle as orc
    g = np.load(os.path.join(ROOT, "tests", "golden", "f4_solves.npz"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tag, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    assert all(len(r) > 2 for r in results), results
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0, r1 = results
    assert r0[1] == r1[1] == 2 and r0[3] == r1[3]
    assert np.array_equal(r0[4], r1[4]) and np.array_equal(r0[5], r1[5]) and r0[2] == r1[2]
    # against the single-rank device solve
    with _native.Problem(g[f"{tag}_cams0"], g[f"{tag}_pts0"], g[f"{tag}_uv"], g[f"{tag}_ci"], g[f"{tag}_pi"]) as prob:
        cams, pts, rep, _ = prob.solve_lm(prob.make_opts(ftol=1e-4))
    assert rep.nfev == r0[3] and abs(rep.cost - r0[2]) <= 1e-9 * rep.cost
    assert np.max(np.abs(cams - r0[4])) <= 1e-6 and np.max(np.abs(pts - r0[5])) <= 1e-6
    # residual vector comes back in the caller's observation order
    x = np.hstack((r0[4].ravel(), r0[5].ravel()))
    ref_f = orc.fun(x, cams.shape[0], pts.shape[0], g[f"{tag}_ci"], g[f"{tag}_pi"], g[f"{tag}_uv"], 1.0)
    assert np.max(np.abs(ref_f - r0[6])) <= 1e-8
    ref = float(g[f"{tag}_loose_cost"])
    assert r0[2] <= ref * (1 + 1e-9) and ref - r0[2] <= 1e-5 * ref
    assert abs(r0[7] - rep.optimality) <= 1e-6 * max(1.0, rep.optimality)


def test_sharded_gpu_solve_two_ranks_f32_fused_path():
    """Same sharded product path in fp32: every rank's shard of the dense 8x2000 rig runs the fused linearise+Schur
    kernel and the dense back substitution; both ranks must agree bit for bit and match the single-rank fp32 solve."""
    import torch.multiprocessing as mp
    from lasercalib_amd import _native
    g = np.load(os.path.join(ROOT, "tests", "golden", "f4_solves.npz"))
    tag = "mid"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tag, q, "f32")) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    assert all(len(r) > 2 for r in results), results
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0, r1 = results
    assert r0[1] == r1[1] and r0[1] in (2, 3, 4) and r0[3] == r1[3]
    assert np.array_equal(r0[4], r1[4]) and np.array_equal(r0[5], r1[5]) and r0[2] == r1[2]
    with _native.Problem(g[f"{tag}_cams0"], g[f"{tag}_pts0"], g[f"{tag}_uv"], g[f"{tag}_ci"], g[f"{tag}_pi"], dtype="f32") as prob:
        cams, pts, rep, _ = prob.solve_lm(prob.make_opts(ftol=1e-4))
    assert abs(rep.cost - r0[2]) <= 1e-4 * rep.cost          # fp32: the shards sum in a different order
    ref = float(g[f"{tag}_loose_cost"])
    assert abs(r0[2] - ref) <= 1e-4 * ref                     # fp32 bar of SURVEY 8(d)
