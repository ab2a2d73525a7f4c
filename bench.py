#!/usr/bin/env python3
"""bench.py -- LM-iteration throughput of the MI355X bundle-adjustment engine.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N = 1:  one process, one GPU.   N > 1: launched by torchrun, one rank per GPU, RCCL over xGMI: the library owns the
          collectives (sba_comm_init + sba_solve_lm: one ncclAllReduce of the packed reduced camera system and one
          ncclAllGather of 8 scalars per LM trial, on the engine's stream); torch.distributed only carries the
          ncclUniqueId, the barriers around the timed region and the max-over-ranks of the elapsed time.
One "step" = one full Levenberg-Marquardt iteration of the hot path on synthetic data already resident
in HBM: linearize (analytic residual + Jacobian blocks -> normal-equation blocks), Schur complement,
reduced camera solve, back-substitution, trial residual, accept/reject.  Nothing is skipped: steps
run with always_relinearize so a rejected step costs the same as an accepted one.

Workload (BASELINE.json configs[2], the config its metric is quoted on): 16 cameras x 50,000 points,
full visibility => 800,000 observations PER GPU (weak scaling: N GPUs solve one 16 x 50,000*N problem,
points sharded, cameras replicated, all-reduce of the reduced camera system every step).

Prints ONE JSON line (rank 0).  `value` = observations x LM-iterations per second over the whole job
(M_total * K / t / 1e6, "Mobs/s"); `lm_iters_per_s` = K / t is reported beside it, as are the
residual+Jacobian kernel rates.  `roofline` is for the kernel with the largest share of a step;
`cpu_baseline` times the reference's algorithm (oracle/sba_oracle.py: numpy model + scipy
least_squares with the reference's exact arguments) on this host.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}   # f32-in MFMA = vector rate 157.3 TF; f64 MFMA is half that on CDNA4
# HBM bytes per launch come from the rocprofv3 PMC summaries committed under profiles/ (separate --pmc FETCH_SIZE / WRITE_SIZE
# passes folded by tools/pmc_summary.py; traffic = 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950).
# They are read at run time, newest round first, and only for the shape they were measured on; any other shape reports null.
PMC_KERNEL_OF_SLOT = {"schur_fused": ("k_schur_fused_bf3", "k_schur_fused_f64", "k_schur_fused_wide", "k_schur_fused"), "schur": ("k_schur<", "k_schur_sym<"),
                      "resjac": ("k_resjac<",), "linearize_points": ("k_linearize_points<",), "linearize_cams": ("k_linearize_cams<",),
                      "backsub": ("k_backsub_dense<", "k_backsub_trial<"), "residual": ("k_residual<",)}


def pmc_traffic_bytes(slot, C, Np, dtype, tangential=False, visibility=1.0):
    """(bytes per launch, source file) of the kernel behind a profile slot, or (None, None) when profiles/ holds no PMC
    summary for this exact shape."""
    import glob
    import re
    if tangential or visibility < 1.0:
        return None, None
    tag = f"{C}x{Np // 1000}k_{dtype}"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_{tag}.json")),
                   key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)), reverse=True)
    for f in files:
        try:
            tab = json.load(open(f))
        except Exception:
            continue
        for want in PMC_KERNEL_OF_SLOT.get(slot, ()):
            for name, v in tab.items():
                if want in name:
                    return (2.0 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024.0, os.path.relpath(f, ROOT)
    return None, None


def rocprof_kernel_us(kname, C, Np, dtype, tangential=False, visibility=1.0):
    """(average launch duration in us, source file) of a kernel from the newest committed rocprofv3 --kernel-trace --stats summary
    of this bench command on this shape (profiles/r*_bench_<shape>_kernel_stats.csv), or (None, None)."""
    import csv
    import glob
    import re
    if tangential or visibility < 1.0:
        return None, None
    tag = f"{C}x{Np // 1000}k_{dtype}"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_bench_{tag}_kernel_stats.csv")),
                   key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)), reverse=True)
    for f in files:
        try:
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    name = row["Name"]
                    if "::" + kname + "(" in name or "::" + kname + "<" in name:
                        return float(row["AverageNs"]) * 1e-3, os.path.relpath(f, ROOT)
        except Exception:      # noqa: BLE001
            continue
    return None, None


def e2e_block(rig, dtype, calls=7):
    """Wall time of the drop-in call as the reference's caller makes it (scripts/calibrate_camera.py:62-72): construct PySBA on the
    host arrays, bundleAdjust(1e-4) -- handle creation, upload (PCIe), device layout, the LM solve to ITS OWN convergence, download
    of cameras and points, result packaging -- in a warm process.  Median of `calls` calls after one untimed call."""
    import contextlib
    import io
    from lasercalib_amd.pySBA import PySBA
    old = os.environ.get("LASERCALIB_SBA_DTYPE")
    os.environ["LASERCALIB_SBA_DTYPE"] = dtype
    try:
        times, res = [], None
        for k in range(calls + 1):
            cams, pts = rig["cams0"].copy(), rig["pts0"].copy()
            t0 = time.perf_counter()
            sba = PySBA(cams, pts, rig["points_2d"], rig["camera_ind"], rig["point_ind"])
            with contextlib.redirect_stdout(io.StringIO()):
                res = sba.bundleAdjust(1e-4)
            dt = time.perf_counter() - t0
            if k:
                times.append(dt)
    finally:
        if old is None:
            os.environ.pop("LASERCALIB_SBA_DTYPE", None)
        else:
            os.environ["LASERCALIB_SBA_DTYPE"] = old
    med = float(np.median(times))
    iters = max(1, int(res.nfev) - 1)
    return {"dtype": dtype, "seconds": med, "seconds_min": float(min(times)), "seconds_max": float(max(times)), "calls": calls,
            "lm_iterations": iters, "nfev": int(res.nfev), "status": int(res.status), "final_cost": float(res.cost),
            "lm_iters_per_s_incl_upload": iters / med,
            "what": "PySBA(...) + bundleAdjust(1e-4): handle, host-to-device upload, layout, LM solve to convergence, download, packaging"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--cams", type=int, default=16)
    ap.add_argument("--points", type=int, default=50000, help="points PER GPU")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--tangential", action="store_true", help="13-parameter camera rows (radial + tangential: BASELINE config 5)")
    ap.add_argument("--visibility", type=float, default=1.0, help="probability that a camera sees a point (1.0 = dense, BASELINE's shape)")
    ap.add_argument("--min-views", type=int, default=2, help="cameras every point keeps at least (the reference's example uses 4)")
    ap.add_argument("--f64-record", default="auto", choices=["auto", "off"], help="also time the f64 engine (PySBA's default dtype) "
                    "on the same workload and report it as the `f64` block (N=1, f32 runs only)")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off"])
    ap.add_argument("--e2e", default="auto", choices=["auto", "off"], help="also time the drop-in call PySBA(...).bundleAdjust(1e-4) "
                    "end to end (upload included) on the same rig, f32 and f64 (N=1)")
    ap.add_argument("--cpu-points", type=int, default=0, help="points of the CPU baseline solve; 0 = the GPU workload's own point count "
                    "(16 x 50,000: about 2 minutes of scipy); pass e.g. 5000 for a short sample")
    return ap.parse_args()


def algorithmic_bytes_per_obs(kernel, s, C, N, M):
    """SURVEY.md section 8(d).  s = sizeof(real); indices are int32 on the device."""
    if kernel == "resjac":            # materialising: read ci,pi,uv,point ; write r + 28 J values
        return 8 + 35 * s
    if kernel == "linearize_points":  # fused: read ci(4) + uv(2s) + point(3s) ; write V(6)+gp(3) doubles per point
        return 4 + 5 * s + 9 * 8 * N / M
    if kernel == "linearize_cams":    # camera-major pass: read pi(4) + uv(2s) + point(3s) per observation
        return 4 + 5 * s
    if kernel == "backsub":           # read ci(4)+uv(2s)+point(3s) ; per point read V,gp,D2p (12 doubles) + X (3 doubles), write X (3 doubles + 3 s)
        return 4 + 5 * s + (15 * 8 + 3 * 8 + 3 * s) * N / M
    if kernel == "residual":
        return 8 + 5 * s
    raise KeyError(kernel)


def cpu_baseline(C, n_points_sample, ftol=1e-4, rig_kw=None):
    """Reference algorithm on the host: scipy TRF + 3-point FD Jacobian through oracle.bundle_adjust (the reference's exact
    least_squares call, pySBA.py:141), run to its own convergence at the caller's ftol on the SAME rig recipe and, by
    default, the same size as the GPU workload (16 x 50,000 = 800k observations: ~1 min, 3-4 LM iterations).
    The path is numpy elementwise + scipy sparse FD/LSMR: effectively one core, whatever the host has.
    13-parameter rigs (--tangential) use oracle/sba_oracle_tangential.py: the reference's call around the extended model."""
    from lasercalib_amd.synth import make_rig
    rig_kw = dict(rig_kw or {})
    if rig_kw.get("tangential"):
        from oracle import sba_oracle_tangential as orc
    else:
        from oracle import sba_oracle as orc
    rig = make_rig(C, n_points_sample, seed=0, **rig_kw)
    M = rig["camera_ind"].size
    t0 = time.perf_counter()
    res, _, _ = orc.bundle_adjust(rig["cams0"], rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], ftol=ftol)
    dt = time.perf_counter() - t0
    iters = max(1, res.nfev - 1)
    threads = 1
    try:
        from threadpoolctl import threadpool_info
        threads = max([1] + [p.get("num_threads", 1) for p in threadpool_info()])
    except Exception:
        pass
    return {"value": M * iters / dt / 1e6, "unit": "Mobs/s", "cores": 1, "kind": "port",
            "sample": f"{C} cams x {n_points_sample} points ({M} obs), oracle.bundle_adjust = scipy least_squares(trf, 3-point FD, "
                      f"x_scale=jac, ftol={ftol}) to convergence: {iters} LM iterations in {dt:.2f} s",
            "lm_iters_per_s": iters / dt, "seconds": dt, "nfev": int(res.nfev), "njev": int(res.njev),
            "final_cost": float(res.cost), "host_cpus": os.cpu_count(), "blas_threads": threads,
            "note": "numpy elementwise + scipy sparse FD/LSMR: effectively one core"}


def roofline_of_step(kt, dtype, P, C, Nloc, M_local, shape):
    """`roofline` object for the heaviest streaming / MFMA kernel of a step (kt = in-loop kernel durations in us)."""
    s = 4 if dtype == "f32" else 8
    n = P * C
    traffic = lambda slot: pmc_traffic_bytes(slot, *shape)
    # one camera group + f32 (+ dense or mask-able visibility): the linearisation runs inside the Schur kernel
    fused = kt["linearize_points"] == 0 and kt["linearize_cams"] == 0 and kt["schur"] > 0
    streaming = ("schur", "linearize_cams", "linearize_points", "backsub")
    # latency-bound single-workgroup / reduction stages (cholesky_solve, schur_reduce) have no meaningful bandwidth roofline:
    # the heaviest streaming or MFMA kernel is reported, those are listed in kernel_us
    dominant = max(streaming, key=lambda k: kt[k])
    if dominant == "schur":
        flops_mfma = (n * (n + 1) / 2) * 3 * Nloc * 2                      # symmetric S: n(n+1)/2 entries x K=3N x 2
        # the fused kernel also carries the whole linearisation on the VALU (60 residual + 330 Jacobian +
        # 480 block accumulation flops per observation, 50 + 198 C per point); on gfx950 the f32 MFMA and the f32 VALU
        # share the SIMD's FMA lanes (tools/micro/mix_waves.hip: their times add), reported beside `frac`, not in it
        flops_valu = (870.0 * M_local + (50.0 + 198.0 * C) * Nloc) if fused else 0.0
        ach = flops_mfma / (kt["schur"] * 1e-6) / 1e12                     # roofline fraction = MFMA work only
        bf3 = fused and dtype == "f32" and os.environ.get("SBA_FUSED_MFMA", "bf3") != "f32"
        if fused:
            wide = C > 16 or P == 13        # csrc/sba_schur_wide.hpp: 17 .. 23 cameras, and every one-launch rig of the 13-parameter model
            kname = (("k_schur_fused_wide_f64" if C > 16 else "k_schur_fused_f64") if dtype == "f64" else      # csrc/sba_schur_f64.hpp
                     "k_schur_fused_wide" if (wide and bf3) else "k_schur_fused_bf3" if bf3 else "k_schur_fused")
        elif dtype == "f64":
            kname = "k_schur_sym<double>"
        else:
            kname = "k_schur_diag_bf3 + k_schur_offdiag_bf3" if C > 16 else "k_schur<float>"
        tb, tsrc = traffic("schur_fused" if fused else "schur")
        # the same fraction from the committed rocprofv3 kernel-trace summary of this command (its average duration for the kernel);
        # `frac` itself stays the live HIP-event figure of THIS run
        rp_us, rp_src = rocprof_kernel_us(kname, *shape) if " " not in kname and "<" not in kname else (None, None)
        return {"kernel": kname, "bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS[dtype],
                "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS[dtype], "traffic": tb, "traffic_source": tsrc,
                "frac_rocprof": (flops_mfma / (rp_us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS[dtype]) if rp_us else None,
                "rocprof_launch_us": rp_us, "rocprof_source": rp_src,
                "algorithmic_flops_per_launch": flops_mfma, "launch_us": kt["schur"],
                # the linearisation the fused kernel also carries on the VALU (SURVEY 8d estimate), kept apart from `frac`
                "valu_flops_estimate": flops_valu,
                "frac_with_valu_estimate": (flops_mfma + flops_valu) / (kt["schur"] * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS[dtype],
                "note": ("algorithmic f32 flops of the symmetric Schur product against the f32-input MFMA peak (= f32 vector peak). "
                         "k_schur_fused_bf3 forms every f32 product exactly from six bf16 partial products on the bf16 matrix pipe "
                         "(3-way split of the f32 panel), so the f32 MFMA peak is the yardstick BASELINE/SURVEY name, not a hard ceiling "
                         "for this kernel; its own limit is the producers' VALU work (DESIGN.md 4.2).  launch_us is the kernel as it runs "
                         "in the loop: its prologue also takes the accept/reject decision of the previous LM step (about 3 us; a separate "
                         "6.6 us k_decide launch before, SBA_DECIDE_KERNEL=1 restores it)") if bf3 else None}
    by = algorithmic_bytes_per_obs(dominant, s, C, Nloc, M_local) * M_local
    ach = by / (kt[dominant] * 1e-6) / 1e9
    tb, tsrc = traffic(dominant)
    return {"kernel": "k_" + dominant, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": tb, "traffic_source": tsrc, "algorithmic_bytes_per_launch": by, "launch_us": kt[dominant]}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SBA_BENCH_BACKEND", "nccl") != "nccl":
        local = 0      # rehearsal on a one-GPU box: every rank uses device 0
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    import torch
    import torch.distributed as dist
    from lasercalib_amd import _native
    from lasercalib_amd import dist as sdist
    from lasercalib_amd.synth import make_rig

    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SBA_BENCH_BACKEND", "nccl")      # "gloo" only to rehearse N>1 on a one-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    C, Np = a.cams, a.points
    P = 13 if a.tangential else 11
    N_total = Np * world
    rig_kw = dict(visibility=a.visibility, min_cams_per_point=a.min_views, tangential=a.tangential)
    rig = make_rig(C, N_total, seed=0, **rig_kw)       # every rank builds the same job, then takes its slice
    M_total = rig["camera_ind"].size
    shard = sdist.make_shard(rig["pts0"], rig["points_2d"], rig["camera_ind"], rig["point_ind"], None, world, rank)
    M_local = shard["ci"].size
    Nloc = shard["pts"].shape[0]
    shape = (C, Np, a.dtype, a.tangential, a.visibility)
    stream = torch.cuda.current_stream().cuda_stream
    prob = _native.Problem(rig["cams0"], shard["pts"], shard["uv"], shard["ci"], shard["pi_local"], dtype=a.dtype,
                           device=local, stream=stream)
    x0 = np.hstack((rig["cams0"].ravel(), shard["pts"].ravel()))
    # SBA_BENCH_COMM=ipc: the library's one-shot exchange through peer-mapped buffers (csrc/sba_ipc.hpp) instead of RCCL; it also
    # works with every rank on ONE card (SBA_BENCH_BACKEND=gloo), which is how the in-library sharded loop is timed on a one-GPU box
    use_ipc = world > 1 and os.environ.get("SBA_BENCH_COMM", "") == "ipc"
    rehearsal = world > 1 and os.environ.get("SBA_BENCH_BACKEND", "nccl") != "nccl" and not use_ipc     # gloo on one card: phase API + torch collectives
    phase_api = rehearsal or bool(os.environ.get("SBA_BENCH_PHASE_API"))
    comm_note = None
    if use_ipc:
        handles = [None] * world
        dist.all_gather_object(handles, prob.ipc_export(world))
        prob.ipc_attach(rank, handles)
    elif world > 1 and not rehearsal:
        # the library's own RCCL communicator; should binding or initialising it fail on EVERY rank alike (library missing,
        # version mismatch: symmetric failures), every rank falls back to the phase-API loop with torch.distributed collectives
        # (slower, but the run still measures the sharded solve) and says so.  A ONE-SIDED failure inside sba_comm_init cannot
        # be rescued here: the other ranks are inside ncclCommInitRank by then and only RCCL's own timeout ends that.
        failed = 0
        try:
            ids = [_native.comm_unique_id() if rank == 0 else None]
        except Exception as e:       # noqa: BLE001
            ids, failed, comm_note = [None], 1, f"sba_comm_get_unique_id failed: {e}"
        dist.broadcast_object_list(ids, src=0)
        if ids[0] is None:
            failed = 1
        else:
            try:
                prob.comm_init(ids[0], rank, world)
            except Exception as e:   # noqa: BLE001
                failed, comm_note = 1, f"sba_comm_init failed: {e}"
        flag = torch.tensor([failed], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            phase_api = True
            comm_note = comm_note or "another rank could not initialise the library's RCCL communicator"
            prob.close()             # a half-initialised communicator must not take part in the solve
            prob = _native.Problem(rig["cams0"], shard["pts"], shard["uv"], shard["ci"], shard["pi_local"], dtype=a.dtype,
                                   device=local, stream=stream)
    elif os.environ.get("SBA_BENCH_RCCL_1"):          # one GPU through the same RCCL code path (a 1-rank communicator)
        prob.comm_init(_native.comm_unique_id(), 0, 1)
    comm = sdist.TorchComm() if (world > 1 and phase_api) else sdist.SoloComm()

    per_solve = {}

    def make_runner(pr):
        E = torch.empty(pr.exchange_size(), dtype=torch.float64, device="cuda")
        sc = torch.empty(sdist.NSCALARS, dtype=torch.float64, device="cuda")

        def run(iters, profile=False):
            """`iters` LM iterations from the initial guess; returns (seconds, costs)."""
            pr.set_params(x0)
            opts = pr.make_opts(ftol=0.0, xtol=0.0, gtol=0.0, max_iter=iters, always_relinearize=True, profile=profile)
            if not phase_api:
                # the library's own loop (N > 1: the sharded loop and its collectives).  A step is one LM iteration: the timed
                # region is sba_lm_run, exactly `iters` iterations; the per-solve work either side of it -- sba_lm_begin (reset +
                # cost of the initial point) and sba_lm_finish (gradient at the returned point, download of cameras and points into
                # pageable host arrays) -- is timed separately and reported as `per_solve_us`
                barrier()
                tb = time.perf_counter()
                pr.lm_begin(opts)
                barrier()
                t0 = time.perf_counter()
                pr.lm_run()
                barrier()
                t1 = time.perf_counter()
                costs = [r.cost for r in pr.iteration_log()]
                pr.lm_finish()
                barrier()
                per_solve["lm_begin"] = (t0 - tb) * 1e6
                per_solve["lm_finish"] = (time.perf_counter() - t1) * 1e6
                return t1 - t0, costs
            # phase API (torch.distributed carries the exchanges): same timed region, the iterations alone
            barrier()
            tb = time.perf_counter()
            pr.lm_begin(opts)
            barrier()
            t0 = time.perf_counter()
            done = 0
            while done < iters:
                for _ in range(min(32, iters - done)):      # 32 steps enqueued between two host polls
                    pr.lm_linearize()
                    pr.lm_form_reduced(E.data_ptr())
                    if world > 1:
                        dist.all_reduce(E)
                    pr.lm_solve_trial(E.data_ptr(), sc.data_ptr())
                    sc_all = comm.all_gather_rows(sc)
                    pr.lm_decide_async(sc_all.data_ptr(), world)
                status, done = pr.lm_poll()
            barrier()
            t1 = time.perf_counter()
            costs = [r.cost for r in pr.iteration_log()]
            pr.lm_finish()
            barrier()
            per_solve["lm_begin"] = (t0 - tb) * 1e6
            per_solve["lm_finish"] = (time.perf_counter() - t1) * 1e6
            return t1 - t0, costs
        return run

    run = make_runner(prob)
    if a.warmup > 0:
        run(a.warmup)
    dt, costs = run(a.steps)
    per_solve_main = dict(per_solve)
    run(min(a.steps, 20), profile=True)      # separate pass with one HIP event pair per kernel class per step
    t = torch.tensor([dt], dtype=torch.float64, device="cuda" if (world == 1 or dist.get_backend() == "nccl") else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # kernel durations: LIVE over the timed region (one HIP event pair per kernel class per step on the
    # engine's stream); residual / resjac are not part of a step and are timed back-to-back afterwards.
    kt = prob.kernel_profile()
    kt.update({k: prob.time_kernel(k, 20) for k in ("residual", "resjac")})
    step_us = dt / a.steps * 1e6
    out = None
    if rank == 0:
        n = P * C
        s = 4 if a.dtype == "f32" else 8
        fused = kt["linearize_points"] == 0 and kt["linearize_cams"] == 0 and kt["schur"] > 0
        roof = roofline_of_step(kt, a.dtype, P, C, Nloc, M_local, shape)
        rj_bytes = algorithmic_bytes_per_obs("resjac", s, C, Nloc, M_local) * M_local
        if P == 13:
            rj_bytes += 4 * s * M_local          # two more camera columns of the 2 x P block
        tb, tsrc = pmc_traffic_bytes("resjac", *shape)
        rj = {"kernel": "k_resjac", "bound": "hbm", "achieved": rj_bytes / (kt["resjac"] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
              "unit": "GB/s", "frac": rj_bytes / (kt["resjac"] * 1e-6) / 1e9 / HBM_PEAK_GBS, "launch_us": kt["resjac"],
              "algorithmic_bytes_per_launch": rj_bytes, "mobs_per_s": M_local / kt["resjac"], "traffic": tb, "traffic_source": tsrc}
        vis_txt = "full visibility" if a.visibility >= 1.0 else f"visibility {a.visibility:g} (>= {a.min_views} views per point)"
        out = {
            "metric": "LM iters/sec and residual+Jacobian Mobs/s at 16 cams x 50k points",
            "value": M_total * a.steps / dt / 1e6,
            "unit": "Mobs/s (observations x LM iterations per second, whole job)",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{C} cams x {Np} points per GPU, {vis_txt}, {P}-parameter cameras ({M_local} obs per GPU, {M_total} total), "
                                   "full on-device Schur-complement LM iteration", "cams": C, "points_per_gpu": Np,
                       "camera_parameters": P, "visibility": a.visibility,
                       "observations_total": int(M_total), "parallelism": f"points sharded x{world}, cameras replicated",
                       "collectives_fallback_reason": comm_note,
                       "collectives": ("none" if world == 1 else "torch.distributed through the phase C ABI (rehearsal / fallback)" if phase_api else
                                       "one-shot exchange through peer-mapped buffers inside libsba_hip.so (sba_ipc): per step n(n+1)/2+3n+1 doubles + 8 doubles written once per rank, read by every rank" if use_ipc else
                                       "RCCL inside libsba_hip.so: per step 1 all-reduce of n(n+1)/2+3n+1 doubles + 1 all-gather of 8 doubles per rank"),
                       "exchange_doubles_per_step": (0 if world == 1 else n * n + 3 * n + 1 if phase_api else n * (n + 1) // 2 + 3 * n + 1)},
            "lm_iters_per_s": a.steps / dt,
            "resjac_mobs_per_s": M_local / kt["resjac"],
            "fused_linearize_mobs_per_s": M_local / (kt["schur"] if fused else (kt["linearize_points"] + kt["linearize_cams"])),
            "linearize_fused_into_schur": bool(fused),
            "kernel_us": kt, "step_us": step_us,
            # the timed region is the K iterations (sba_lm_run); once per solve, outside it: sba_lm_begin (reset + cost of the initial
            # point) and sba_lm_finish (gradient at the returned point + download of cameras and points to pageable host memory)
            "timed_region": "K LM iterations (sba_lm_run) between sba_lm_begin and sba_lm_finish",
            # version 1 (rounds 1, 2 and the first half of 3): sba_lm_begin and sba_lm_finish inside the timer -- that figure is
            # `ms_per_step_with_begin_and_finish`, the one to compare with those rounds; version 2: the iterations alone
            "timed_region_version": 2,
            "per_solve_us": per_solve_main,
            "ms_per_step_with_begin_and_finish": (dt + (per_solve_main.get("lm_begin", 0.0) + per_solve_main.get("lm_finish", 0.0)) * 1e-6) / a.steps * 1e3,
            "cost_first_last": [costs[0], costs[-1]] if costs else None,
            "roofline": roof, "roofline_resjac": rj,
        }
    # the f64 engine on the same workload: PySBA's default dtype (the reference computes in f64), i.e. what an unmodified
    # calibrate_camera.py runs.  Outside the timed headline; same step definition, fewer steps.
    if world == 1 and a.dtype == "f32" and a.f64_record != "off":
        prob.close()
        prob = _native.Problem(rig["cams0"], shard["pts"], shard["uv"], shard["ci"], shard["pi_local"], dtype="f64",
                               device=local, stream=stream)
        run64 = make_runner(prob)
        k64 = min(a.steps, 20)
        run64(min(max(a.warmup, 1), 5))
        dt64, costs64 = run64(k64)
        run64(k64, profile=True)
        kt64 = prob.kernel_profile()
        out["f64"] = {"dtype": "f64", "steps": k64, "ms_per_step": dt64 / k64 * 1e3, "lm_iters_per_s": k64 / dt64,
                      "value": M_total * k64 / dt64 / 1e6, "kernel_us": kt64,
                      "cost_first_last": [costs64[0], costs64[-1]] if costs64 else None,
                      "roofline": roofline_of_step(kt64, "f64", P, C, Nloc, M_local, (C, Np, "f64", a.tangential, a.visibility)),
                      "note": "the drop-in class's default engine (LASERCALIB_SBA_DTYPE unset): same workload and step as the headline"}
    # end-to-end wall time of the drop-in call on the same rig (SURVEY 8(d) metric (1): "report both incl./excl." upload), N = 1
    if world == 1 and rank == 0 and a.e2e != "off":
        prob.close()
        out["e2e"] = {dt_: e2e_block(rig, dt_) for dt_ in ("f32", "f64")}
        out["lm_iters_per_s_incl_upload"] = out["e2e"][a.dtype]["lm_iters_per_s_incl_upload"]
    # SBA_BENCH_COMPARE_IPC=1 (N > 1, opt-in): the same steps once more over the one-shot exchange through peer-mapped buffers
    # (csrc/sba_ipc.hpp) on a second handle, reported beside the headline as `ipc_exchange` -- the comparison of the two exchange
    # mechanisms on whatever node this runs on.  Off by default: the headline is the RCCL path BASELINE.json names.
    if world > 1 and not use_ipc and os.environ.get("SBA_BENCH_COMPARE_IPC"):
        saved_phase = phase_api
        prob2 = _native.Problem(rig["cams0"], shard["pts"], shard["uv"], shard["ci"], shard["pi_local"], dtype=a.dtype,
                                device=local, stream=stream)
        handles = [None] * world
        dist.all_gather_object(handles, prob2.ipc_export(world))
        prob2.ipc_attach(rank, handles)
        phase_api = False                      # the library's own loop
        run2 = make_runner(prob2)
        run2(max(a.warmup, 1))
        dt2, costs2 = run2(a.steps)
        t2 = torch.tensor([dt2], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        dt2 = float(t2.item())
        phase_api = saved_phase
        dist.barrier()
        prob2.close()
        if rank == 0:
            out["ipc_exchange"] = {"ms_per_step": dt2 / a.steps * 1e3, "lm_iters_per_s": a.steps / dt2, "value": M_total * a.steps / dt2 / 1e6,
                                   "cost_first_last": [costs2[0], costs2[-1]] if costs2 else None,
                                   "note": "same workload and steps through sba_ipc_export / sba_ipc_attach instead of the headline's collectives"}
    if rank == 0:
        if a.cpu_baseline != "off" and world == 1:
            cpu_pts = a.cpu_points if a.cpu_points > 0 else Np
            cb = cpu_baseline(C, cpu_pts, rig_kw=rig_kw)
            out["cpu_baseline"] = cb
            if cpu_pts == Np:      # same configuration on both sides: the ratio of LM iterations per second is meaningful
                # steady-state device iterations against scipy's whole-solve wall / iterations: NOT like for like (stated);
                # `speedup_e2e` below is
                out["speedup_vs_cpu_lm_iters_per_s"] = out["lm_iters_per_s"] / cb["lm_iters_per_s"]
                if "e2e" in out:
                    # like for like: ONE bundleAdjust(1e-4) call on the same rig, both sides to their own convergence, everything
                    # included on both sides (scipy: sparsity build, grouping, FD Jacobians, LSMR; device: upload, solve, download)
                    out["speedup_e2e"] = {k: cb["seconds"] / v["seconds"] for k, v in out["e2e"].items()}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()          # (sba_ipc: nobody frees an area a peer may still be reading)
    prob.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
