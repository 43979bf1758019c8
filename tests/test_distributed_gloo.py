"""N > 1 path on CPU: two gloo processes run the product's ShardedGbp host logic (landmark partition,
per-iteration all_gather of camera partials, fixed-rank-order combine, metric reduction) over an
oracle-backed rank engine, and must reproduce the single-process oracle in 2-shard device order
BIT FOR BIT (the exchange adds nothing but a fixed-order sum)."""
import os
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, out_dir, slam=False):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gbp_poplar_amd import driver, hostlib
    from gbp_poplar_amd.distributed import ShardedGbp, landmark_partition
    from tests.oracle_shard_engine import OracleShardEngine
    bal = hostlib.synth_generate(14, 260, 5, 5)
    opts = driver.Options()
    K, state, extra = driver.build_inputs(bal, opts, hostlib, slam=slam)
    bounds = landmark_partition(bal["lmk_id"], bal["n_lmks"], world)
    eng = OracleShardEngine(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K,
                            (rank, world, int(bounds[rank]), int(bounds[rank + 1])))
    run = ShardedGbp(eng, bal["n_cams"], rank, world, dist=dist, device="cpu")
    if slam:
        traj = driver.run_slam(run, hostlib, bal, state, extra, opts, iters_between_kfs=6, max_iters=40, eval_every=1)
    else:
        traj = driver.run_ba(run, state, opts, n_iters=22, eval_every=1)
    r = run.read()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), traj=np.array(traj), bounds=bounds, **r)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_two_shard_oracle():
    import torch.multiprocessing as mp
    from gbp_poplar_amd import driver, hostlib
    from oracle import oracle as orc
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port + 1, d), nprocs=world, join=True)
        res = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(world)]
        res = [{k: x[k] for k in x.files} for x in res]
    bal = hostlib.synth_generate(14, 260, 5, 5)
    opts = driver.Options()
    K, state, _ = driver.build_inputs(bal, opts, hostlib)
    bounds = res[0]["bounds"]
    assert bounds[0] == 0 and bounds[-1] == bal["n_lmks"] and 0 < bounds[1] < bal["n_lmks"]
    ref = orc.Oracle(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K)
    ref.set_sum_order(1, bounds)
    tref = np.array(driver.run_ba(ref, state, opts, n_iters=22, eval_every=1))
    rr = ref.read()
    # camera beliefs are replicated and bit-identical on both ranks and equal to the reference
    for k in ("cam_beliefs_eta", "cam_beliefs_lambda"):
        assert np.array_equal(res[0][k], res[1][k]) and np.array_equal(res[0][k], rr[k]), k
    # landmark beliefs / per-factor state: each rank owns its range
    lmk = np.asarray(bal["lmk_id"])
    for r in range(world):
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        assert np.array_equal(res[r]["lmk_beliefs_eta"][3 * lo:3 * hi], rr["lmk_beliefs_eta"][3 * lo:3 * hi])
        assert np.array_equal(res[r]["lmk_beliefs_lambda"][9 * lo:9 * hi], rr["lmk_beliefs_lambda"][9 * lo:9 * hi])
        own = (lmk >= lo) & (lmk < hi)
        assert np.array_equal(res[r]["damping_count"][own], rr["damping_count"][own])
    # the reduced metric is the same on both ranks and matches the single-process run
    assert np.array_equal(res[0]["traj"], res[1]["traj"])
    assert np.allclose(res[0]["traj"][:, 1:3], tref[:, 1:3], rtol=1e-12, atol=0)
    assert np.array_equal(res[0]["traj"][:, 3:], tref[:, 3:])
    assert tref[-1, 1] < 0.5 * tref[0, 1]


def test_two_rank_gloo_slam_matches_two_shard_oracle():
    """The incremental SLAM flow (READ_PRIORS / NEW_KEYFRAME every 6 sweeps, slam.cpp:1018-1055) through the
    sharded host logic on two gloo ranks == the single-process oracle in 2-shard order, bit for bit."""
    import torch.multiprocessing as mp
    from gbp_poplar_amd import driver, hostlib
    from oracle import oracle as orc
    world = 2
    port = 31500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port, d, True), nprocs=world, join=True)
        res = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(world)]
        res = [{k: x[k] for k in x.files} for x in res]
    bal = hostlib.synth_generate(14, 260, 5, 5)
    opts = driver.Options()
    K, state, extra = driver.build_inputs(bal, opts, hostlib, slam=True)
    bounds = res[0]["bounds"]
    ref = orc.Oracle(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K)
    ref.set_sum_order(1, bounds)
    tref = np.array(driver.run_slam(ref, hostlib, bal, state, extra, opts, iters_between_kfs=6, max_iters=40, eval_every=1))
    rr = ref.read()
    for k in ("cam_beliefs_eta", "cam_beliefs_lambda"):
        assert np.array_equal(res[0][k], res[1][k], equal_nan=True) and np.array_equal(res[0][k], rr[k], equal_nan=True), k
    lmk = np.asarray(bal["lmk_id"])
    for r in range(world):
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        assert np.array_equal(res[r]["lmk_beliefs_eta"][3 * lo:3 * hi], rr["lmk_beliefs_eta"][3 * lo:3 * hi], equal_nan=True)
        own = (lmk >= lo) & (lmk < hi)
        assert np.array_equal(res[r]["damping_count"][own], rr["damping_count"][own])
    assert np.array_equal(res[0]["traj"], res[1]["traj"])
    assert np.allclose(res[0]["traj"][:, 1:3], tref[:, 1:3], rtol=1e-12, atol=0)
    assert np.array_equal(res[0]["traj"][:, 3:], tref[:, 3:])
    assert len(tref) == 41 and tref[-1, 0] == 39          # 6 keyframe insertions happened on the way


def test_landmark_partition_balances_factors():
    from gbp_poplar_amd.distributed import landmark_partition
    rng = np.random.default_rng(0)
    lmk = np.sort(rng.integers(0, 1000, 20000)).astype(np.uint32)
    for world in (1, 2, 3, 8):
        b = landmark_partition(lmk, 1000, world)
        assert b[0] == 0 and b[-1] == 1000 and len(b) == world + 1 and np.all(np.diff(b.astype(np.int64)) >= 0)
        cnt = [int(((lmk >= b[r]) & (lmk < b[r + 1])).sum()) for r in range(world)]
        assert sum(cnt) == 20000 and max(cnt) - min(cnt) <= 2 * 20000 / 1000 * 3 + 40
