import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch, torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lasercalib_amd import dist as sdist, _native
    g = np.load(os.path.join(ROOT, "tests", "golden", "f4_solves.npz"))
    tag = "sparse"
    cams, pts, uv, ci, pi = (g[f"{tag}_{k}"] for k in ("cams0", "pts0", "uv", "ci", "pi"))
    comm = sdist.TorchComm()
    sh = sdist.make_shard(pts, uv, ci, pi, None, world, rank)
    eng = sdist.HipEngine(cams, sh, 0, 0, dict(ftol=1e-4, xtol=1e-8, gtol=1e-8, max_nfev=40, mode=0, verbose=0))
    log = open(os.path.join(ROOT, "gpurun_out", f"dbg_rank{rank}.log"), "w")
    status, it = None, 0
    for outer in range(6):
        for _ in range(eng.batch):
            eng.linearize()
            E = eng.form_reduced()
            comm.all_reduce_sum(E)
            sc = eng.solve_trial(E)
            sc_all = comm.all_gather_rows(sc)
            eng.decide_async(sc_all, comm.n)
        status, it = eng.poll()
        rows = eng.prob.iteration_log()
        print(outer, "status", status, "iters", it, [(r.iteration, r.accepted, r.cost, r.rho, r.lambda_) for r in rows[-4:]], file=log, flush=True)
        if status is not None:
            break
    print("done", file=log, flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, 29655)) for r in range(2)]
    for p in ps: p.start()
    for p in ps: p.join(timeout=100)
    for p in ps:
        if p.is_alive(): p.terminate()
