"""bench.py --gpus N without a launcher starts the N ranks itself (torch.distributed.run children, before any GPU
call), relays rank 0's JSON line as the LAST line of its stdout and propagates a failing rank's status.  Exercised here
on the CPU with the --launch-selftest mode (gloo group, one all_reduce) — the same plumbing an N-GPU run goes through."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-selftest", *extra],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=240)


def test_self_launch_relays_rank0_json_as_last_stdout_line():
    p = _run("--steps", "3")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    out = json.loads(lines[-1])
    assert len(lines) == 1                                 # rank noise goes to stderr, stdout carries the JSON line only
    assert out == {"selftest": True, "n_gpus": 2, "sum_of_ranks_plus_1": 3.0, "steps": 3}


def test_self_launch_propagates_a_failing_rank():
    p = _run("--selftest-fail-rank", "1")
    assert p.returncode != 0


def test_default_workloads():
    sys.path.insert(0, ROOT)
    import bench
    a1, a8, a4 = bench.parse([]), bench.parse(["--gpus", "8"]), bench.parse(["--gpus", "4", "--weak-s1"])
    assert (a1.cams, a1.lmks) == (1000, 100000)                       # S1
    assert (a8.cams * 8, a8.lmks * 8, a8.obs) == (8000, 1000000, 10)  # BASELINE config 5: 8 000 x 1 000 000 x 10 M
    assert (a4.cams, a4.lmks) == (1000, 100000)
    assert "BASELINE config 5:" in bench.workload_name(a8, 8, 8000, 1000000, 10000000)
    assert bench.workload_name(a1, 1, 1000, 100000, 1000000).startswith("S1 ")
    s = bench.build_stamp(a1)
    assert len(s["source_sha16"]) == 16 and s["workload"] == [1000, 100000, 10, 20200303]
    assert bench.pmc_shape(a8, 8) == (8000, 125000)                   # the PMC child of an N > 1 run: rank 0's shard shape
    assert bench.build_stamp(a8, 8)["workload"] == [8000, 125000, 10, 20200303]


def _bench_gpu(*args):
    import pytest
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cams", "60", "--lmks", "4000", "--steps", "12",
                        "--warmup", "12", "--profile-steps", "5", "--cpu-seconds", "0", *args],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("comm", ["native", "torch"])
def test_bench_sharded_code_path_on_one_gpu(comm):
    """bench.py --force-sharded: the N > 1 code path (shard ctx, RCCL all-gather, overlap; the library's own communicator
    or torch.distributed) with one rank on one GPU — same contract fields, same convergence as the plain path."""
    plain = _bench_gpu("--pmc", "off")
    out = _bench_gpu("--force-sharded", "--comm", comm, "--sharded-graph", "1" if comm == "native" else "0", "--pmc", "off", "--preflight", "0")
    assert out["config"]["iterations_run"] == plain["config"]["iterations_run"] == 24
    assert out["n_gpus"] == 1 and out["steps"] == 12 and out["value"] > 0 and out["higher_is_better"] is True
    assert out["config"]["comm_error"] is None
    assert ("native" in out["config"]["exchange"]) == (comm == "native")
    assert out["config"]["reproj_rmse_px_final"] == plain["config"]["reproj_rmse_px_final"]      # bit-identical runs
    assert out["roofline"]["avg_launch_us"] > 0 and out["roofline"]["frac_algorithmic"] > 0
    if comm == "native":
        assert out["roofline"]["exchange_avg_us"] > 0          # measured on rank 0: partial sums + all-gather per iteration


@pytest.mark.gpu
def test_bench_live_pmc_traffic():
    """The default N = 1 bench measures the HBM-side traffic of its kernels itself (two rocprofv3 --pmc child passes)."""
    out = _bench_gpu()
    r = out["roofline"]
    assert r["traffic_source"].startswith("live rocprofv3") and r["traffic"] > 0 and 0 < r["frac"] < 1.0, r
    assert {k["kernel"] for k in r["kernels"]} == {"k_sweep", "k_beliefs"}
    assert r["profiled_ms_per_step"] * 1e3 >= 0.9 * (r["avg_launch_us"] + r["belief_kernels_avg_us"])
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3            # achieved, peak and frac are one statement


@pytest.mark.gpu
def test_bench_line_describes_the_run_it_times():
    """VERDICT r04 item 1, on a small graph (the driver's own `--steps 20 --warmup 5` shape of flags):
      * the timed region is iterations 5..24 of the ./ba flow, the prior weakenings in front of 5, 7, 9 inside it and on record;
      * `windows` {min, median, max} of five further windows and a `sustained` window of >= 1 s beside `value`;
      * the flow trace: the timed engine's metric at the end of the timed region and of the profiled window lies on the
        trajectory of a second engine running the reference's default loop; `configs.s1_default_loop` reports that loop;
      * the PMC child replayed the same launches (its n_relin per profiled iteration == the trajectory's, same final metric);
      * cpu_baseline: every belief and the per-factor state bit-exact against the oracle after the sample's iterations."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cams", "200", "--lmks", "20000", "--steps", "20", "--warmup", "5",
                        "--profile-steps", "12", "--cpu-seconds", "4", "--sustained-seconds", "1", "--small-configs", "off"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    tr = out["config"]["timed_region"]
    assert tr == {"first_iteration": 5, "iterations": 20, "prior_weakenings_inside": [5, 7, 9], "is": tr["is"]}
    assert out["config"]["iterations_run"] == 25 and out["steps"] == 20 and out["warmup"] == 5
    w, su = out["windows"], out["sustained"]
    assert w["n"] == 5 and 0 < w["min"] <= w["median"] <= w["max"] and w["iterations_each"] == 20
    assert su["seconds"] >= 1.0 and su["iterations"] >= 20 and su["value"] > 0
    ft = out["config"]["flow_trace"]
    assert ft["timed_run_on_trajectory"] is True and ft["profiled_run_on_trajectory"] is True, ft
    assert abs(ft["rmse_px_at_end_of_timed_region"] - out["config"]["reproj_rmse_px_final"]) < 1e-6
    rep = out["roofline"]["replay"]
    assert rep["child_replayed_the_same_launches"] is True, rep
    assert len(rep["n_relin_per_iteration"]) == 12 and out["roofline"]["window"]["first_iteration"] == 25
    cb = out["cpu_baseline"]
    assert cb["beliefs_bit_exact_vs_oracle"] is True and cb["max_rel_deviation"] == 0.0, cb
    assert cb["gpu_n_relin_same_iterations"] == cb["n_relin"] and cb["rmse_rel_diff"] < 1e-6
    dl = out["configs"]["s1_default_loop"]
    assert dl["iterations"] >= 37 and dl["iters_per_sec"] > 0 and dl["oracle_metric_same_iteration"]["rel_diff"] < 1e-6
    assert dl["oracle_metric_same_iteration"]["gpu_n_relin"] == dl["oracle_metric_same_iteration"]["oracle_n_relin"]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_with_real_ranks_sharing_one_gpu(world, oracle_mod):
    """`bench.py --gpus N --share-gpu` runs the N > 1 code path with N REAL processes on the one GPU of the
    box — self-launch through torch.distributed.run, per-rank shard build, rendezvous, preflight block, schedule choice, MAX-over-
    ranks timing, rank-0 relay — with the launcher's group on gloo and the library's communicator on its host-staged transport
    (RCCL refuses duplicate GPUs).  The line says so (all_ranks_on_distinct_gpus false), and the run is RIGHT: the final RMSE is
    the N-shard oracle's after the same iterations of the same flow."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from gbp_poplar_amd import driver, hostlib
    from gbp_poplar_amd.distributed import landmark_partition
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cams, lmks = 40, 1500                  # per rank
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--share-gpu", "--cams", str(cams), "--lmks", str(lmks),
                        "--steps", "8", "--warmup", "4", "--profile-steps", "4", "--cpu-seconds", "0", "--pmc", "off", "--windows", "2",
                        "--sustained-seconds", "0.2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    out = json.loads(lines[-1])
    assert out["n_gpus"] == world and out["value"] > 0 and out["steps"] == 8 and out["scaling"] == "weak"
    pre = out["config"]["preflight"]
    assert pre["all_ranks_on_distinct_gpus"] is False and pre["ranks_share_gpus"] is True and pre["distinct_pci_bus_ids"] == 1
    assert len(pre["ranks"]) == world and {r["rank"] for r in pre["ranks"]} == set(range(world))
    assert all(r["transport"] == "host-staged" for r in pre["ranks"]) and pre["exchange_probe_us"] > 0
    assert "host-staged" in out["config"]["exchange"] and out["config"]["comm_error"] is None
    assert out["roofline"]["rank_step_ms_max"] >= out["roofline"]["rank_step_ms_min"] > 0
    # the line diagnoses itself — per-rank step, sweep and exchange times, who can reach whom
    assert len(out["roofline"]["rank_step_ms"]) == world and len(out["roofline"]["sweep_avg_us_per_rank"]) == world
    assert len(out["roofline"]["exchange_avg_us_per_rank"]) == world and all(x > 0 for x in out["roofline"]["sweep_avg_us_per_rank"])
    assert all(len(r["peer_access_from_this_device"]) == r["visible_devices"] for r in pre["ranks"])
    n_it = out["config"]["iterations_run"]
    assert n_it == 4 + 50 + 8 and out["windows"]["n"] == 2 and out["sustained"]["iterations"] >= 8
    # the N-shard oracle through the same flow: RMSE after the timed region
    C, L = cams * world, lmks * world
    bal = hostlib.synth_generate(C, L, 10, 20200303)
    opts = driver.Options()
    K, state, _ = driver.build_inputs(bal, opts, hostlib)
    oracle_mod.set_trig_mode(1)
    try:
        orc = oracle_mod.Oracle(bal["cam_id"], bal["lmk_id"], C, L, K)
        orc.set_sum_order(1, landmark_partition(bal["lmk_id"], L, world))
        traj = driver.run_ba(orc, state, opts, n_iters=n_it, eval_every=n_it)
        ev = orc.eval()
    finally:
        oracle_mod.set_trig_mode(0)
    rmse = float(np.sqrt(2.0 * ev["sum_half_sq"] / ev["n_active"]))
    assert abs(out["config"]["reproj_rmse_px_final"] - rmse) <= 2e-6 * rmse, (out["config"]["reproj_rmse_px_final"], rmse, traj[-1])


@pytest.mark.gpu
def test_bench_sharded_line_is_complete_on_the_config5_shard_shape():
    """the line an N > 1 run prints must carry a MEASURED roofline.frac (PMC passes on rank 0's shard
    shape), the exchange time and the per-rank step times — checked here with one rank on the per-GPU shard of BASELINE
    config 5 (8 000 cameras x 125 000 landmarks x 1.25 M factors), the library's own communicator."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-sharded", "--cams", "8000", "--lmks", "125000", "--steps", "20",
                        "--warmup", "5", "--cpu-seconds", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    r = out["roofline"]
    assert out["config"]["comm_error"] is None and "native" in out["config"]["exchange"]
    assert r["traffic"] and 0.3 < r["frac"] < 1.0, r
    assert "shard shape" in r["traffic_source"]
    assert r["exchange_avg_us"] > 0 and r["rank_step_ms_max"] >= r["rank_step_ms_min"] > 0
    assert out["config"]["factors"] == 1250000 and out["steps"] == 20
    # the first multi-GPU run validates itself — the preflight block rides in the line
    pre = out["config"]["preflight"]
    assert pre["all_ranks_on_distinct_gpus"] is True and pre["distinct_pci_bus_ids"] == 1 and pre["same_library_on_every_rank"] is True
    rk = pre["ranks"][0]
    assert rk["transport"] == "rccl" and "librccl" in rk["library"] and rk["library_version"] > 20000 and rk["pci_bus_id"]
    assert rk["peer_access_from_this_device"][0] is True
    assert pre["exchange_probe_us"] > 0
    sched = pre["schedule_ms_per_iteration"]
    assert sched["one_stream"] > 0 and sched["two_streams"] > 0
    assert pre["stream_mode_chosen"] == min(sched, key=sched.get)
    assert out["config"]["iterations_run"] == 5 + 50 + 20
    assert out["config"]["timed_region"]["first_iteration"] == 55 and out["config"]["timed_region"]["prior_weakenings_inside"] == []
    assert out["windows"]["n"] == 5 and out["sustained"]["seconds"] >= 1.5


@pytest.mark.gpu
def test_bench_native_communicator_failure_is_fatal():
    """No silent downgrade: with the library's RCCL unloadable, `--comm native` (the default) exits non-zero; the line is still
    printed — value null, `comm_error` and what every rank saw — so that the record of a failed multi-GPU run
    says why; `--comm torch` is the explicit way to measure the torch.distributed exchange."""
    env = dict(os.environ, GBP_RCCL_LIB="/nonexistent/librccl-missing.so")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-sharded", "--cams", "60", "--lmks", "1500", "--steps", "4", "--warmup", "2",
                        "--cpu-seconds", "0", "--pmc", "off", "--profile-steps", "0"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 4, (p.returncode, p.stderr[-2000:])
    assert "could not be set up" in p.stderr
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["value"] is None and out["n_gpus"] == 1 and "could not be set up" in out["comm_error"]
    assert len(out["config"]["ranks"]) == 1 and "librccl-missing" in out["config"]["ranks"][0]["comm_error"]


def test_cli_metric_lines_are_parsed():
    sys.path.insert(0, ROOT)
    import bench
    rows = bench.parse_iter_lines("Initial Reprojection error: 20.4 Cost 1e6\n"
                                  "Iter 0 // Reprojection error 3.84 // Cost 636719 // n relins: 0 // n robust edges 1337\n"
                                  "Weakening priors \n"
                                  "Iters 701 (since last kf 1) // Reprojection error 0.7036 // Cost 363.876 // n relins: 5 // n robust edges 27\n")
    assert rows == [(0, 3.84, 636719.0, 0, 1337), (701, 0.7036, 363.876, 5, 27)]
    assert set(bench.SMALL_CONFIGS) == {"fr1xyz", "slam_fr2robot2"}


def test_small_config_cpu_baseline_runs_the_reference_loop(monkeypatch):
    """The CPU leg of bench.py's `configs` block: the oracle through driver.run_ba / run_slam on the shipped file (short prefix here)."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setitem(bench.SMALL_CONFIGS, "fr1xyz", ("ba", "fr1xyz", None, (1.42, 1.47), 12))
    monkeypatch.setitem(bench.SMALL_CONFIGS, "slam_fr2robot2", ("slam", "fr2robot2", "slam_fr2robot2", None, 40))
    for name in ("fr1xyz", "slam_fr2robot2"):
        c = bench.small_config_cpu(name, [])
        assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["iterations"] in (12, 40)
        assert "the first %d iterations of the run" % c["iterations"] in c["sample"]


@pytest.mark.gpu
def test_bench_line_carries_the_fr1xyz_and_slam_configs():
    """BASELINE.json's metric is "fr1xyz AND the 1M-factor synthetic graph" — the driver-run JSON line
    carries `configs.fr1xyz` (bin/ba, default flags) and `configs.slam_fr2robot2` (bin/slam, config 3): iterations/s, the
    path that ran, final mean reprojection error + RMSE with the band / golden check, and a CPU baseline each."""
    out = _bench_gpu("--small-configs", "on", "--cpu-seconds", "1", "--pmc", "off", "--profile-steps", "0")
    cfg = out["configs"]
    fx, sl = cfg["fr1xyz"], cfg["slam_fr2robot2"]
    for c in (fx, sl):
        assert "error" not in c, c
        for k in ("iters_per_sec", "iters_per_sec_device", "us_per_iter_device", "final_mean_reproj_px", "rmse_px", "graph_state",
                  "cpu_baseline", "eval_every_100", "loop_wall_ms"):
            assert k in c and c[k] is not None, k
        assert c["graph_state"] == 2 and c["iters_per_sec"] > 0 and c["nonfinite_beliefs"] == 0
        cb = c["cpu_baseline"]
        assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb
        assert cb["same_iteration_metric_matches"] is True, cb
        assert c["same_final_metric_with_eval_every_100"] is True
    assert fx["iterations"] == 1500 and fx["in_converged_band"] is True and 1.42 <= fx["final_mean_reproj_px"] <= 1.47
    assert abs(fx["rmse_px"] - (2.0 * fx["final_cost"] / fx["n_active"]) ** 0.5) < 1e-6
    assert sl["iterations"] == 13299 and sl["golden"]["within_1e-3"] is True, sl["golden"]
