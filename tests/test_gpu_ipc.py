"""GPU: the IN-LIBRARY sharded LM loop (sba_solve_lm with n_ranks > 1) executed for real -- two processes sharing the one card,
exchanging the reduced camera system and the trial scalars through peer-mapped device buffers (sba_ipc_export / sba_ipc_attach,
csrc/sba_ipc.hpp).  RCCL refuses two ranks on one device, so until this path existed the library-owned multi-rank loop had only
ever run on a 1-rank communicator; here every rank runs it and must agree with its peer bit for bit and with the single-rank solve.
torch.distributed (gloo) only carries the 64-byte handles and the final gathers."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rig17():
    from lasercalib_amd.synth import make_rig
    rig = make_rig(17, 900, seed=3, visibility=0.6, min_cams_per_point=4)
    return rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"]


def _worker(rank, world, port, tag, q, dtype):
    sys.path.insert(0, ROOT)
    os.environ["LASERCALIB_SBA_DTYPE"] = dtype
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", LASERCALIB_SBA_DEVICE="0", LASERCALIB_SBA_SHARD="1", LASERCALIB_SBA_COMM="ipc")
    import torch  # noqa: F401
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lasercalib_amd.pySBA import PySBA
        if tag == "rig17":
            a = _rig17()
            sba = PySBA(a[0].copy(), a[1].copy(), a[2], a[3], a[4])
        else:
            g = np.load(os.path.join(ROOT, "tests", "golden", "f4_solves.npz"))
            sba = PySBA(g[f"{tag}_cams0"].copy(), g[f"{tag}_pts0"].copy(), g[f"{tag}_uv"], g[f"{tag}_ci"], g[f"{tag}_pi"])
        res = sba.bundleAdjust(1e-4)
        q.put((rank, res.status, res.cost, res.nfev, sba.cameraArray.copy(), sba.points3D.copy(), res.fun.copy(), res.optimality))
    except BaseException as e:          # never leave the parent waiting on the queue
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def _run(tag, dtype, world=2):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, world, port, tag, q, dtype)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    assert all(len(r) > 2 for r in results), results
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return results


@pytest.mark.parametrize("tag", ["sparse", "mid"])
def test_in_library_sharded_loop_two_ranks_one_card_f64(tag):
    from lasercalib_amd import _native
    from oracle import sba_oracle as orc
    g = np.load(os.path.join(ROOT, "tests", "golden", "f4_solves.npz"))
    r0, r1 = _run(tag, "f64")
    assert r0[1] == r1[1] == 2 and r0[3] == r1[3]
    assert np.array_equal(r0[4], r1[4]) and np.array_equal(r0[5], r1[5]) and r0[2] == r1[2]       # same bits on both ranks
    with _native.Problem(g[f"{tag}_cams0"], g[f"{tag}_pts0"], g[f"{tag}_uv"], g[f"{tag}_ci"], g[f"{tag}_pi"]) as prob:
        cams, pts, rep, _ = prob.solve_lm(prob.make_opts(ftol=1e-4))
    assert rep.nfev == r0[3] and abs(rep.cost - r0[2]) <= 1e-9 * rep.cost                          # the single-rank trajectory
    assert np.max(np.abs(cams - r0[4])) <= 1e-6 and np.max(np.abs(pts - r0[5])) <= 1e-6
    x = np.hstack((r0[4].ravel(), r0[5].ravel()))
    ref_f = orc.fun(x, cams.shape[0], pts.shape[0], g[f"{tag}_ci"], g[f"{tag}_pi"], g[f"{tag}_uv"], 1.0)
    assert np.max(np.abs(ref_f - r0[6])) <= 1e-8
    ref = float(g[f"{tag}_loose_cost"])
    assert r0[2] <= ref * (1 + 1e-9) and ref - r0[2] <= 1e-5 * ref                                 # and the reference's optimum


def test_in_library_sharded_loop_three_ranks_f32_fused_path():
    """fp32, three ranks on the card: every shard of the dense 8 x 2000 rig runs the fused kernel with the decision in its
    prologue; the exchanges sit between the fused kernel, the Cholesky and the back substitution of every trial."""
    from lasercalib_amd import _native
    g = np.load(os.path.join(ROOT, "tests", "golden", "f4_solves.npz"))
    tag = "mid"
    res = _run(tag, "f32", world=3)
    r0 = res[0]
    for r in res[1:]:
        assert r[1] == r0[1] and r[3] == r0[3] and r[2] == r0[2]
        assert np.array_equal(r[4], r0[4]) and np.array_equal(r[5], r0[5])
    assert r0[1] in (2, 3, 4)
    with _native.Problem(g[f"{tag}_cams0"], g[f"{tag}_pts0"], g[f"{tag}_uv"], g[f"{tag}_ci"], g[f"{tag}_pi"], dtype="f32") as prob:
        _, _, rep, _ = prob.solve_lm(prob.make_opts(ftol=1e-4))
    assert abs(rep.cost - r0[2]) <= 1e-4 * rep.cost          # fp32: the shards sum in a different order
    ref = float(g[f"{tag}_loose_cost"])
    assert abs(r0[2] - ref) <= 1e-4 * ref                     # fp32 bar of SURVEY 8(d)


def test_in_library_sharded_loop_17_cameras_f32_wide_kernel():
    """The reference's rig shape, sharded: every rank's shard runs k_schur_fused_wide (decision in its prologue), the exchanged
    187 x 187 system goes through k_cholesky_ll on every rank -- two ranks on the card over the one-shot exchange."""
    sys.path.insert(0, ROOT)
    from lasercalib_amd import _native
    from oracle import sba_oracle as orc
    res = _run("rig17", "f32", world=2)
    r0, r1 = res
    assert r0[1] == r1[1] and r0[1] in (2, 3, 4) and r0[3] == r1[3] and r0[2] == r1[2]
    assert np.array_equal(r0[4], r1[4]) and np.array_equal(r0[5], r1[5])
    a = _rig17()
    with _native.Problem(*a, dtype="f32") as prob:
        _, _, rep, _ = prob.solve_lm(prob.make_opts(ftol=1e-4))
    assert abs(rep.cost - r0[2]) <= 1e-4 * rep.cost
    ref, _, _ = orc.bundle_adjust(*a, ftol=1e-4)
    assert abs(r0[2] - ref.cost) <= 1e-4 * ref.cost


def test_in_library_sharded_loop_17_cameras_f64_wide_kernel():
    """The same rig in the class's default dtype: every rank's shard runs k_schur_fused_wide_f64 (csrc/sba_schur_f64.hpp), the
    exchanged system goes through k_cholesky_ll<double>; the two ranks agree bit for bit, and with the single-rank fp64 solve to
    the rounding of the different summation order."""
    sys.path.insert(0, ROOT)
    from lasercalib_amd import _native
    res = _run("rig17", "f64", world=2)
    r0, r1 = res
    assert r0[1] == r1[1] and r0[1] in (2, 3, 4) and r0[3] == r1[3] and r0[2] == r1[2]
    assert np.array_equal(r0[4], r1[4]) and np.array_equal(r0[5], r1[5])
    a = _rig17()
    with _native.Problem(*a, dtype="f64") as prob:
        _, _, rep, _ = prob.solve_lm(prob.make_opts(ftol=1e-4))
    assert abs(rep.cost - r0[2]) <= 1e-9 * rep.cost

