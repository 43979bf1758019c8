"""EXAMPLE, not the product path: a caller that brings its OWN collective (torch.distributed) around the split-phase
C-ABI (gbp_iterate_begin / gbp_iterate_local / gbp_iterate_end, gbp_poplar_amd/distributed.py), with the `ba` / `slam` flag
sets.  The product's multi-GPU path is `bin/ba --ipus N` / `bin/slam --ipus N`: one forked process per GPU, the exchange
owned by the C++ library (csrc/gbp_comm.cpp, RCCL).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           examples/ba_torchrun.py --bal_file F [--n_iters K] [--eval_every E] [... the ba flags]

One process per GPU; landmarks are sharded over the ranks (gbp_poplar_amd.distributed), rank 0 prints the
same lines as `./ba` (ba.cpp:996,1004,1026-1028) or, with `--slam`, as `./slam` (slam.cpp:1073-1076; keyframes
every `--iters_between_kfs` sweeps).  With WORLD_SIZE unset it runs on one GPU.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="ba_torchrun")
    ap.add_argument("--bal_file", required=True)
    ap.add_argument("--n_iters", type=int, default=1500)
    ap.add_argument("--reproj_meas_var", type=float, default=4.0)
    ap.add_argument("--prior_std_weaker_factor", type=float, default=100.0)
    ap.add_argument("--first_cam_prior_std", type=float, default=0.01)
    ap.add_argument("--steps", type=float, default=5.0)
    ap.add_argument("--undamped_start", type=int, default=15)
    ap.add_argument("--tn", type=float, default=0.0, help="std (m) of the noise on keyframe translation initialisations")
    ap.add_argument("--rn", type=float, default=0.0, help="std (degrees) of the noise on keyframe rotation initialisations")
    ap.add_argument("--ltn", type=float, default=0.0, help="std (m) of the noise on landmark initialisations")
    ap.add_argument("--avdepth_on", type=int, default=0, help="initialise landmarks one unit in front of their first keyframe")
    ap.add_argument("--avdepth", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=0, help="seed of the initialisation noise (0 = from the OS)")
    ap.add_argument("--eval_every", type=int, default=1)
    ap.add_argument("--slam", action="store_true", help="incremental SLAM flow of ./slam instead of batch BA")
    ap.add_argument("--iters_between_kfs", type=int, default=700)
    a = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if not torch.cuda.is_available():
        print("Could not find a device", file=sys.stderr)
        return 255
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from gbp_poplar_amd import driver, hostlib
    from gbp_poplar_amd.distributed import ShardedGbp, landmark_partition
    from gbp_poplar_amd.engine import GbpEngine

    log = print if rank == 0 else (lambda *x: None)
    try:
        bal = hostlib.bal_read(a.bal_file)
    except IOError as e:
        print(e, file=sys.stderr)
        return 1
    opts = driver.Options(n_iters=a.n_iters, reproj_meas_var=a.reproj_meas_var,
                          prior_std_weaker_factor=a.prior_std_weaker_factor,
                          first_cam_prior_std=a.first_cam_prior_std, steps=a.steps, undamped_start=a.undamped_start,
                          tn=a.tn, rn=a.rn, ltn=a.ltn, avdepth_on=bool(a.avdepth_on), seed=a.seed)
    if world > 1 and (a.tn or a.rn or a.ltn) and not a.seed:
        print("--tn/--rn/--ltn on several ranks need --seed (every rank must draw the same noise)", file=sys.stderr)
        return 2
    opts.iters_between_kfs = a.iters_between_kfs
    K, state, extra = driver.build_inputs(bal, opts, hostlib, slam=a.slam)
    log("Completed loading data!\n\n%s\n" % ("Incremental SLAM" if a.slam else "Bundle Adjustment"))
    log("Number of keyframe nodes in the graph: %d\nNumber of landmark nodes in the graph: %d\nNumber of edges in the graph: %d"
        % (bal["n_cams"], bal["n_lmks"], bal["n_edges"]))
    log("\nNumber of GPUs: %d" % world)
    C, L = bal["n_cams"], bal["n_lmks"]
    bounds = landmark_partition(bal["lmk_id"], L, world)
    eng = GbpEngine(bal["cam_id"], bal["lmk_id"], C, L, K, shard=(rank, world, int(bounds[rank]), int(bounds[rank + 1])))
    run = ShardedGbp(eng, C, rank, world, dist=dist, device="cuda")
    t0 = time.perf_counter()
    if a.slam:
        traj = driver.run_slam(run, hostlib, bal, state, extra, opts, eval_every=a.eval_every, log=log)
        n_done = (C - 1) * a.iters_between_kfs - 1
    else:
        traj = driver.run_ba(run, state, opts, n_iters=a.n_iters, eval_every=a.eval_every, log=log)
        n_done = a.n_iters
    run.sync()
    log("\n Finished GBP.\nTotal time: %.3f s (%d iterations, %d GPUs)" % (time.perf_counter() - t0, n_done, world))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if traj else 1


if __name__ == "__main__":
    sys.exit(main())
