#!/usr/bin/env python3
"""bench.py — GBP iterations/second on the synthetic 1M-factor BAL graph (BASELINE.json configs[3], "S1").

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one synchronous GBP iteration (GBP_PROG of the reference, ba/ba.cpp:895-905: prep ->
messages -> beliefs) over the whole factor graph, inputs resident in HBM, no host read-back inside the
timed region.  N = 1 runs S1 = 1 000 cameras x 100 000 landmarks x 1 000 000 factors.  N > 1 is WEAK
scaling: N x S1 (1000 N cameras, 100k N landmarks, 1M N factors), landmark-sharded, one all-gather of
camera partials per iteration.  `value` = iterations/s x (total factors / 1e6), i.e. "1M-factor-graph
GBP iterations per second": at N = 1 it is exactly BASELINE.json's metric, and it aggregates over
ranks like tokens/s does (raw iterations/s of the N x larger graph is in config.iters_per_sec).

The warm-up runs the reference's start of a BA run (LINEARISE, prior weakening on iterations 1,3,5,7,9)
so the timed iterations are steady-state sweeps of a converging problem.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
ALGO_BYTES_PER_FACTOR = 1112   # algorithmic bytes per factor-iteration of the sweep


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cams", type=int, default=1000, help="cameras per GPU")
    ap.add_argument("--lmks", type=int, default=100000, help="landmarks per GPU")
    ap.add_argument("--obs", type=int, default=10, help="observations per landmark")
    ap.add_argument("--seed", type=int, default=20200303)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--profile-steps", type=int, default=20, help="per-stage hipEvent-timed iterations for the roofline")
    ap.add_argument("--force-sharded", action="store_true",
                    help="diagnostic: run the N>1 code path (shard ctx, RCCL all_gather, overlap) even with one rank")
    ap.add_argument("--sharded-graph", type=int, default=0, help="capture sharded iterations (kernels + RCCL) in a hipGraph")
    ap.add_argument("--exchange-chunks", type=int, default=None,
                    help="camera ranges of the pipelined all-gather (default: 1 / 2 / 3 for 1 / 2-4 / 8 GPUs)")
    ap.add_argument("--tile-order", type=int, default=0, help="gbp_params.tile_order: 0 = default, 1 = sequential, 2 = sweep tiles XCD-aware too")
    return ap.parse_args()


def warm_start(eng, opts, warmup):
    """ba.cpp:1001-1008 for `warmup` iterations (weaken priors on 1,3,5,7,9).  Warm-up iterations beyond the
    prior-weakening phase are issued in bursts of 10 so that the one-off costs of the multi-iteration path
    (hipGraph capture + instantiation, first-use code-object loading on a fresh box) are paid here, not in the
    timed region."""
    it = 0
    while it < warmup:
        if it < opts.steps * 2 or warmup - it < 10:
            if ((it + 1) % 2 == 0) and (it < opts.steps * 2):
                eng.weaken_priors()
            eng.iterate(1)
            it += 1
        else:
            eng.iterate(10)
            it += 10


def cpu_baseline(bal, K, state, opts, budget_s):
    """The CPU oracle (OpenMP over factors / variables) timed on this host on the SAME graph."""
    from oracle import oracle as orc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # honour a cgroup CPU quota (containers): "max 100000" or "<quota> <period>"
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    orc.set_threads(cores)
    o = orc.Oracle(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K)
    o.upload(state)
    o.linearise()
    t0 = time.perf_counter()
    o.iterate(1)
    t1 = time.perf_counter() - t0
    # the sample: the reference's own start of a BA run (ba.cpp:1001-1008), as many iterations as the budget allows
    n = max(2, min(50, int(budget_s / max(t1, 1e-6))))
    t_iter = t1
    for it in range(1, n):
        if ((it + 1) % 2 == 0) and (it < opts.steps * 2):
            o.weaken_priors()
        t0 = time.perf_counter()
        o.iterate(1)
        t_iter += time.perf_counter() - t0
    ips = n / t_iter
    ev = o.eval()
    o.close()
    return {"value": ips * bal["n_edges"] / 1e6, "unit": "1M-factor GBP iters/s", "cores": cores, "kind": "port",
            "sample": "first %d iterations of the ./ba flow on the same %d-factor graph (oracle/, gcc -O2 -fopenmp, %d threads)"
                      % (n, bal["n_edges"], cores),
            "iterations": n, "rmse_px": float((2.0 * ev["sum_half_sq"] / max(ev["n_active"], 1)) ** 0.5),
            "mean_reproj_px": ev["sum_norm"] / max(ev["n_active"], 1)}


def gpu_accuracy_run(bal, K, state, opts, n):
    """A fresh GPU run of the first n iterations of the ./ba flow: the accuracy figure quoted next to the CPU one."""
    from gbp_poplar_amd import driver
    from gbp_poplar_amd.engine import GbpEngine
    eng = GbpEngine(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K)
    traj = driver.run_ba(eng, state, opts, n_iters=n, eval_every=n)
    ev = eng.eval()
    eng.close()
    return {"rmse_px": float((2.0 * ev["sum_half_sq"] / max(ev["n_active"], 1)) ** 0.5), "mean_reproj_px": traj[-1][1]}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank != 0:      # only rank 0 may write to stdout (libraries such as RCCL print banners through C stdio)
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if world != a.gpus and world != 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with torch.distributed.run (one process per GPU)")

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or a.force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "RANK" not in os.environ:     # --force-sharded without a launcher: a 1-rank group
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 1000))
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from gbp_poplar_amd import driver, hostlib
    from gbp_poplar_amd.distributed import ShardedGbp, landmark_partition
    from gbp_poplar_amd.engine import GbpEngine

    C, L = a.cams * world, a.lmks * world
    bal = hostlib.synth_generate(C, L, a.obs, a.seed)      # every rank generates the same global graph
    E = bal["n_edges"]
    opts = driver.Options()
    K, state, _ = driver.build_inputs(bal, opts, hostlib)

    from gbp_poplar_amd import _cabi
    prm = _cabi.GbpParams.defaults(tile_order=a.tile_order)
    if world == 1 and not a.force_sharded:
        eng = GbpEngine(bal["cam_id"], bal["lmk_id"], C, L, K, params=prm)
        run = eng
        e_local = E
    else:
        bounds = landmark_partition(bal["lmk_id"], L, world)
        eng = GbpEngine(bal["cam_id"], bal["lmk_id"], C, L, K, params=prm,
                        shard=(rank, world, int(bounds[rank]), int(bounds[rank + 1])))
        run = ShardedGbp(eng, C, rank, world, dist=dist, device="cuda", always_collective=a.force_sharded,
                         use_graph=bool(a.sharded_graph), chunks=a.exchange_chunks)
        e_local = int(((bal["lmk_id"] >= bounds[rank]) & (bal["lmk_id"] < bounds[rank + 1])).sum())
    run.upload(state)
    run.linearise()
    ev0 = run.eval()
    warm_start(run, opts, a.warmup)
    if getattr(run, "use_graph", False):
        run.iterate(run.graph_unroll + 3)      # un-timed: triggers the one-off capture of the sharded iteration graph

    def fence():
        run.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    run.iterate(a.steps)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ev1 = run.eval()

    # ---- roofline of the dominant kernel (k_sweep), measured live with hipEvents on its stream ----
    graph_used = (getattr(run, "graph", None) is not None) if hasattr(run, "use_graph") else None
    roof = None
    if a.profile_steps > 0:
        sharded = world > 1 or a.force_sharded
        eng.timing(reset=True)
        eng.set_profiling(True)
        if sharded:      # split-phase path: gbp_iterate_begin brackets its sweep launch; every rank runs the iterations
            if getattr(run, "use_graph", False):
                run.use_graph, run.graph = False, None
            run.iterate(a.profile_steps)
            fence()
        else:
            eng.iterate(a.profile_steps)
        eng.set_profiling(False)
        tm = eng.timing(reset=True)
        sweep_s = tm["sweep_ms"] / 1e3 / a.profile_steps
        achieved = ALGO_BYTES_PER_FACTOR * e_local / sweep_s / 1e9
        traffic = None     # HBM bytes per launch from the committed PMC passes of this same workload (profiles/run_profile.sh)
        tpath = os.path.join(ROOT, "profiles", "traffic_S1.json")
        if os.path.exists(tpath) and (a.cams, a.lmks, a.obs) == (1000, 100000, 10) and world == 1:
            try:
                traffic = int(json.load(open(tpath))["hbm_bytes_per_launch"])
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": "k_sweep", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_FACTOR * e_local,
                "avg_launch_us": round(sweep_s * 1e6, 2),
                "belief_kernels_avg_us": round(tm["belief_ms"] * 1e3 / a.profile_steps, 2) if not sharded else None,
                "measured_on": "rank 0" if world > 1 else "the GPU"}

    cpu = None
    if rank == 0 and world == 1 and a.cpu_seconds > 0:
        cpu = cpu_baseline(bal, K, state, opts, a.cpu_seconds)
        g = gpu_accuracy_run(bal, K, state, opts, cpu["iterations"])
        cpu["gpu_rmse_px_same_iterations"] = g["rmse_px"]
        cpu["rmse_rel_diff"] = abs(g["rmse_px"] - cpu["rmse_px"]) / cpu["rmse_px"]

    if rank == 0:
        ips = a.steps / dt
        m0, m1 = driver.metric(ev0), driver.metric(ev1)
        out = {
            "metric": "GBP iters/sec on the 1M-factor synthetic BAL graph (iterations/s x factors/1e6)",
            "value": round(ips * E / 1e6, 2), "unit": "1M-factor GBP iters/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "S1 synthetic BAL graph x%d: %d cams x %d lmks x %d factors (seed %d)%s"
                                   % (world, C, L, E, a.seed, ", landmark-sharded over %d GPUs" % world if world > 1 else ""),
                       "cams": C, "lmks": L, "factors": E, "iters_per_sec": round(ips, 2),
                       "parallelism": "1 GPU, hipGraph x10 iterations" if (world == 1 and not a.force_sharded) else "landmark shards x%d + all_gather(cam partials)" % world,
                       "reproj_rmse_px_initial": round(m0[2], 6), "reproj_rmse_px_final": round(m1[2], 6),
                       "mean_reproj_px_final": round(m1[0], 6), "iterations_run": a.warmup + a.steps,
                       "nonfinite_beliefs": int(ev1["n_nonfinite"]),
                       "exchange_chunks": getattr(run, "chunks", None),
                       "sharded_graph": graph_used,
                       "sharded_graph_error": getattr(run, "graph_error", None)},
        }
        if roof:
            out["roofline"] = roof
        if cpu:
            out["cpu_baseline"] = cpu
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # The JSON line must be the LAST line on stdout: RCCL prints a version banner through C stdio, which is fully
    # buffered when redirected and would otherwise be flushed at exit, after Python's own output.
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
