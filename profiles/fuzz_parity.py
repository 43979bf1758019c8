#!/usr/bin/env python3
"""Randomised parity soak on the GPU: random graphs x random parameters x random call patterns, three runs of each in lock step.

    python profiles/fuzz_parity.py [seconds=600] [first_seed=0] [max_edges=300000]

Per seed (everything below drawn from the seed):
  * a graph the shipped files never show — 2..120 cameras, 3..30 000 landmarks (log-uniform), up to 12 factors per landmark, the edge
    list UNSORTED (slot order = file order, ba.cpp:267-279), duplicate (camera, landmark) factors, hub landmarks, landmarks without a
    factor, ~10 % inactive factors; parameters that relinearise early and often; per_factor_mu 0 / 1; tile_order 0..3; the all-pad
    segment skipping of the sweep forced on / off / left to the library;
  * A: the two-kernel path, one gbp_iterate(1) at a time (gbp_weaken_priors where ba.cpp:1001-1008 weakens) — against the ORACLE
    (oracle/, device summation order) after every one of the first 10 iterations and after the last: every belief, both message sets,
    damping state, and at the end the factor potentials, bit for bit;
  * B: the path the library chooses by itself for that size (k_persist_flow in one launch per burst, or the hipGraph replay), driven in
    bursts of random length by one of {gbp_iterate(k) + gbp_weaken_priors, gbp_ba_loop with the metric, gbp_ba_loop without}; the
    persistent kernel with tagged records or barriers, with or without its redundant-record check — against A after every burst:
    every tensor incl. the hoisted means, bit for bit; the metrics gbp_ba_loop returns against A's gbp_eval() after each iteration.
Every fourth seed runs ./slam's flow instead (slam_seed below), every eighth a landmark-sharded group of contexts (shard_seed).
The first mismatch stops the run with the seed and what differed; the summary line goes to stdout (copied to profiles/ by hand).
test infrastructure: the oracle is the checker here, never the product.
"""
import os
import sys
import time

import numpy as np

os.environ.setdefault("OMP_NUM_THREADS", "16")      # the oracle's OpenMP team: a box of 256 cores spins on small graphs otherwise
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gbp_poplar_amd import _cabi, _lib, driver, hostlib  # noqa: E402
from gbp_poplar_amd.engine import GbpEngine  # noqa: E402
from oracle import oracle as orc_mod  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_edges = int(sys.argv[3]) if len(sys.argv) > 3 else 300000


def random_problem(rng, slam=False, big=False):
    if slam:      # a keyframe sequence: sorted by camera (slam.cpp relies on it), every keyframe sees a window of the landmarks that moves on
        C = int(rng.integers(3, 31))
        L = int(np.exp(rng.uniform(np.log(20), np.log(3000))))
        per_cam = rng.integers(4, max(5, min(L, 250)), C)
        if big:      # one SLAM seed in 250: beyond 4 M positions — NEW_KEYFRAME's streams and READ_PROG's state no longer fit the staging buffer
            C, L = int(rng.integers(12, 16)), int(rng.integers(300000, 400000))
            per_cam = rng.integers(300000, 380000, C)
        width = max(4, int(L * rng.uniform(0.1, 0.6)))
        cam_id = np.repeat(np.arange(C), per_cam)
        centre = (np.arange(C) / max(C - 1, 1) * max(L - width, 1)).astype(np.int64)
        lmk_id = np.concatenate([np.sort((centre[c] + rng.integers(0, width, per_cam[c])) % L) for c in range(C)])
        E = int(cam_id.size)
    else:
        C = int(rng.integers(2, 121))
        L = int(np.exp(rng.uniform(np.log(3), np.log(30000))))
        if big:      # one seed in fifty: beyond 1.6 M positions (gbp_upload's device-buffer path), thousands of tiles (the shape rules of the sweep)
            C, L = int(rng.integers(200, 3000)), int(rng.integers(150000, 220000))
        E = int(min(max_edges, rng.integers(max(C, L), max(C, L) + 11 * L + 1)))
        if big:
            E = int(rng.integers(1700000, 2100000))
        cam_id = rng.integers(0, C, E)
        if rng.random() < 0.5:                                   # half of the graphs: a few cameras see almost everything
            m = rng.random(E) < 0.5
            cam_id[m] = rng.integers(0, min(C, 3), int(m.sum()))
        lmk_id = rng.integers(0, max(1, L - 2), E)               # the last two landmarks stay factor-less
        for _ in range(int(rng.integers(0, 3))):
            lmk_id[rng.random(E) < rng.uniform(0.005, 0.1)] = int(rng.integers(0, max(1, L - 2)))      # hub landmarks
        cam_id[:C] = np.arange(C)                                # every camera has a factor (else its prior is NaN by design)
        if rng.random() < 0.3:                                   # some files ARE sorted by camera
            o = np.argsort(cam_id, kind="stable")
            cam_id, lmk_id = cam_id[o], lmk_id[o]
    cams = np.zeros((C, 6))
    c = np.arange(C)
    cams[:, 0], cams[:, 1], cams[:, 2] = 0.1 * c, -0.05 * c, 5.0 + 0.2 * c
    cams[:, 3], cams[:, 4], cams[:, 5] = 0.05 + 0.01 * (c % 40), -0.04, 0.03 * ((c % 25) + 1)
    pts = rng.uniform(-1, 1, (L, 3))
    w = cams[cam_id, 3:]
    th = np.linalg.norm(w, axis=1, keepdims=True)
    k = w / th
    y = pts[lmk_id]
    Ry = y * np.cos(th) + np.cross(k, y) * np.sin(th) + k * np.sum(k * y, axis=1, keepdims=True) * (1 - np.cos(th))
    p = Ry + cams[cam_id, :3]
    obs = np.stack([500 * p[:, 0] / p[:, 2] + 320, 500 * p[:, 1] / p[:, 2] + 240], axis=1) + rng.normal(0, 1, (E, 2))
    return {"n_cams": C, "n_lmks": L, "n_edges": E, "fx": 500.0, "fy": 500.0, "cx": 320.0, "cy": 240.0,
            "cam_id": np.asarray(cam_id, np.uint32), "lmk_id": np.asarray(lmk_id, np.uint32), "observations": obs.ravel(),
            "cameras": (cams + rng.normal(0, 0.01, cams.shape) * (np.arange(C)[:, None] >= 2)).ravel(),
            "points": (pts + rng.normal(0, 0.05, pts.shape)).ravel()}


def snapshot(eng):
    d = dict(eng.read())
    d.update({"msg_" + k: v for k, v in eng.messages().items()})
    fe, fl = eng.factor_potentials()
    d.update(fac_eta=fe, fac_lambda=fl, mu=eng.mu()[0])
    return d


class Mismatch(AssertionError):
    pass


def ev_equal(a, b):
    """two metric dicts (or lists of them), NaN == NaN (a graph that diverges does so on both sides, bit for bit)"""
    if isinstance(a, list):
        return len(a) == len(b) and all(ev_equal(x, y) for x, y in zip(a, b))
    return all(a[k] == b[k] or (a[k] != a[k] and b[k] != b[k]) for k in a)


def same(a, b, what):
    if not np.array_equal(a, b, equal_nan=True):
        a, b = np.asarray(a), np.asarray(b)
        bad = np.flatnonzero(~((a == b) | (np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64)))))
        raise Mismatch("%s differs in %d of %d entries (first at %d: %r vs %r)" % (what, bad.size, a.size, bad[0], a.ravel()[bad[0]], b.ravel()[bad[0]]))


def against_oracle(eng, orc, it, potentials=False):
    g, o = eng.read(), orc.read()
    for k in ("cam_beliefs_eta", "cam_beliefs_lambda", "lmk_beliefs_eta", "lmk_beliefs_lambda", "damping", "damping_count", "robust_flag"):
        same(g[k], o[k], "A vs oracle after iteration %d: %s" % (it, k))
    gm, om = eng.messages(), orc.messages()
    mask = np.tile(np.tril(np.ones((6, 6), bool)).ravel(), eng.E)      # cam message Lambda: the lower triangle is stored
    same(gm["cam_eta"], om["cam_eta"], "A vs oracle after iteration %d: cam message eta" % it)
    same(gm["cam_lambda"][mask], om["cam_lambda"][mask], "A vs oracle after iteration %d: cam message lambda" % it)
    same(gm["lmk_eta"], om["lmk_eta"], "A vs oracle after iteration %d: lmk message eta" % it)
    same(gm["lmk_lambda"], om["lmk_lambda"], "A vs oracle after iteration %d: lmk message lambda" % it)
    if potentials:
        fe, fl = eng.factor_potentials()
        oe, ol = orc.factor_potentials()
        same(fe, oe, "A vs oracle: factor eta")
        same(fl, ol, "A vs oracle: factor lambda")


class Clock:
    def __init__(self):
        self.t, self.acc = time.perf_counter(), {}

    def lap(self, name):
        n = time.perf_counter()
        self.acc[name] = self.acc.get(name, 0.0) + n - self.t
        self.t = n


def one_seed(seed, lib):
    ck = Clock()
    rng = np.random.default_rng(7000 + seed)
    bal = random_problem(rng, big=(seed % 50 == 49))
    C, L, E = bal["n_cams"], bal["n_lmks"], bal["n_edges"]
    opts = driver.Options()
    opts.undamped_start = int(rng.integers(1, 4))
    K, state, _ = driver.build_inputs(bal, opts, hostlib)
    state["active_flag"] = (rng.random(E) < 0.9).astype(np.uint32)
    kw = dict(dmu_threshold=float(rng.choice([0.05, 0.02, 3e-3])), min_linear_iters=int(rng.integers(2, 6)), num_undamped_iters=int(rng.integers(1, 4)),
              per_factor_mu=int(rng.integers(0, 2)), tile_order=int(rng.integers(0, 4)))
    seg = int(rng.integers(-1, 2))
    steps = int(opts.steps)
    assert lib.gbp_debug_force_seg_skip(seg) == 0
    try:
        A = GbpEngine(bal["cam_id"], bal["lmk_id"], C, L, K, hooks=True, params=_cabi.GbpParams.defaults(persistent=-1, **kw))
        B = GbpEngine(bal["cam_id"], bal["lmk_id"], C, L, K, hooks=True, params=_cabi.GbpParams.defaults(persistent=0, **kw))
    finally:
        lib.gbp_debug_force_seg_skip(-1)
    okw = {k: v for k, v in kw.items() if k not in ("per_factor_mu", "tile_order")}
    O = orc_mod.Oracle(bal["cam_id"], bal["lmk_id"], C, L, K, params=_cabi.GbpParams.defaults(**okw))
    O.set_sum_order(1)
    path = B.graph_state()
    flow = verify = -1
    if path == 2:
        flow = int(rng.integers(0, 2))
        B.persist_flow(flow)
        if flow == 1:
            verify = int(rng.integers(0, 2))
            B.persist_verify(verify)
    drive = int(rng.integers(0, 3))      # 0: gbp_iterate(k) + gbp_weaken_priors; 1: gbp_ba_loop with the metric; 2: gbp_ba_loop without
    for x in (A, B, O):
        x.upload(state)
        x.linearise()
    ck.lap("setup")
    total = int(rng.integers(24, 60)) if E < 1000000 else int(rng.integers(12, 20))
    it = n_relin = 0
    while it < total:
        burst = int(min(total - it, rng.choice([1, 1, 2, 3, 5, 8, 13, 21])))
        evA = []
        for i in range(it, it + burst):
            if (i + 1) % 2 == 0 and i < 2 * steps:
                A.weaken_priors()
                O.weaken_priors()
            A.iterate(1)
            ck.lap("A")
            O.iterate(1)
            ck.lap("oracle")
            if drive == 1 or i < 10:
                evA.append(A.eval())
            ck.lap("A")
            if i < 10:
                against_oracle(A, O, i)
                ck.lap("compare")
                eo = O.eval()
                if (evA[-1]["n_relin"], evA[-1]["n_robust"], evA[-1]["n_active"]) != (eo["n_relin"], eo["n_robust"], eo["n_active"]):
                    raise Mismatch("counts after iteration %d: %r vs oracle %r" % (i, evA[-1], eo))
        if drive == 0:
            left, i = burst, it
            while left > 0:
                if (i + 1) % 2 == 0 and i < 2 * steps:
                    B.weaken_priors()
                k = 1
                while k < left and not ((i + k + 1) % 2 == 0 and i + k < 2 * steps):
                    k += 1
                B.iterate(k)
                i += k
                left -= k
        elif drive == 1:
            evB = B.ba_loop(burst, it, steps)
            if not ev_equal(evB, evA[-burst:]):
                j = next(j for j in range(burst) if not ev_equal(evB[j], evA[-burst:][j]))
                raise Mismatch("metric of iteration %d (burst of %d from %d; C %d L %d E %d, %r, B's path %d flow %d verify %d): B %r vs A %r; n_nonfinite of the burst B %r A %r"
                               % (it + j, burst, it, C, L, E, kw, B.graph_state(), flow, verify, evB[j], evA[-burst:][j], [e["n_nonfinite"] for e in evB], [e["n_nonfinite"] for e in evA[-burst:]]))
        else:
            B.ba_loop(burst, it, steps, metrics=False)
        it += burst
        B.sync()
        ck.lap("B")
        sa, sb = snapshot(A), snapshot(B)
        for k in sa:
            same(sb[k], sa[k], "B vs A after %d iterations: %s" % (it, k))
        ea, eb = A.eval(), B.eval()
        if not ev_equal(ea, eb):
            raise Mismatch("gbp_eval after %d iterations: B %r vs A %r" % (it, eb, ea))
        n_relin += ea["n_relin"]
        ck.lap("compare")
    against_oracle(A, O, it - 1, potentials=True)
    path = B.graph_state()
    if verify == 1:
        bad = B.persist_verify(0)
        if bad:
            raise Mismatch("the redundant-record check counted %d records that differed from their copy" % bad)
    desc = "C %d L %d E %d | mu %d order %d seg %+d | B: %s%s, driven by %s | %d iterations, relinearisations seen %d" % (
        C, L, E, kw["per_factor_mu"], kw["tile_order"], seg,
        {2: "k_persist_flow", 1: "hipGraph", 0: "direct", -1: "direct"}[path], "" if path != 2 else (" (tagged records%s)" % (", redundant check" if verify == 1 else "") if flow == 1 else " (barriers)"),
        ("gbp_iterate bursts", "gbp_ba_loop + metric", "gbp_ba_loop")[drive], it, n_relin)
    ck.lap("compare")
    desc += " | s: " + ", ".join("%s %.2f" % kv for kv in ck.acc.items())
    A.close()
    B.close()
    return desc, n_relin


def slam_seed(seed):
    """./slam's flow (slam.cpp:1013-1103 as gbp_poplar_amd/driver.py drives it: NEW_KEYFRAME / READ_PRIORS, activation of factors, re-armed
    damping counts, the weakening schedule restarting at every keyframe) on a random keyframe sequence: the PRODUCT library on the path it
    chooses (or kept off the persistent kernel) against the oracle — counts of every printed line exact, the metric to 1e-5 relative + 1e-4 px, every belief and per-factor state bit for bit at the end, the priors (READ_PRIORS) too."""
    rng = np.random.default_rng(9000 + seed)
    big = seed % 1000 == 999
    bal = random_problem(rng, slam=True, big=big)
    C, L, E = bal["n_cams"], bal["n_lmks"], bal["n_edges"]
    opts = driver.Options()
    K, state, extra = driver.build_inputs(bal, opts, hostlib, slam=True)
    kw = dict(dmu_threshold=float(rng.choice([0.05, 0.02, 3e-3])), min_linear_iters=int(rng.integers(2, 8)), num_undamped_iters=int(rng.integers(1, 6)))
    pf, persistent = int(rng.integers(0, 2)), int(rng.choice([0, 0, -1]))
    ibk = int(rng.integers(4, 40))
    every = int(rng.choice([1, 1, 3, 10]))
    max_iters = int(rng.integers(40, 260)) if not big else int(rng.integers(30, 50))
    if big:
        ibk = int(rng.integers(4, 9))
        pf = (seed // 1000) % 2
    G = GbpEngine(bal["cam_id"], bal["lmk_id"], C, L, K, hooks=False, params=_cabi.GbpParams.defaults(per_factor_mu=pf, persistent=persistent, **kw))
    O = orc_mod.Oracle(bal["cam_id"], bal["lmk_id"], C, L, K, params=_cabi.GbpParams.defaults(**kw))
    O.set_sum_order(1)
    tg = driver.run_slam(G, hostlib, bal, state, extra, opts, iters_between_kfs=ibk, max_iters=max_iters, eval_every=every)
    to = driver.run_slam(O, hostlib, bal, state, extra, opts, iters_between_kfs=ibk, max_iters=max_iters, eval_every=every)
    to = {r[0]: r for r in to}
    n_relin = 0
    worst = (0.0, None)
    for (i, mg, cg, rg, bg) in tg:
        if i not in to:
            continue
        _, mo, co, ro_, bo = to[i]
        if (rg, bg) != (ro_, bo):
            raise Mismatch("slam: line of iteration %d: GPU (%.9g, relins %d, robust %d) vs oracle (%.9g, %d, %d)" % (i, mg, rg, bg, mo, ro_, bo))
        # The metric is not a bit-exact quantity between the two (the means are solved differently, z - h(x) cancels ~500 px coordinates in fp32,
        # and a point near a camera's plane amplifies an ulp without bound): 1e-5 relative + 1e-4 px as in tests/test_gpu_parity.py is what
        # well-posed sequences keep; a line beyond it is reported, and it is the STATE below that decides — bit for bit.
        if mg == mg and mo == mo:
            dev = abs(mg - mo) / (1e-5 * mo + 1e-4)
            if dev > worst[0]:
                worst = (dev, (i, mg, mo))
        n_relin += rg
    g, o = G.read(), O.read()
    for k in o:
        same(g[k], o[k], "slam: final %s" % k)
    pg, po = G.read_priors(), O.read_priors()
    for k in po:
        same(pg[k], po[k], "slam: final %s" % k)
    path = G.graph_state()
    desc = "SLAM C %d L %d E %d | mu %d | keyframe every %d, metric every %d, %d iterations | path %s | relinearisations seen %d" % (
        C, L, E, pf, ibk, every, min((C - 1) * ibk - 1, max_iters), {2: "k_persist_flow", 1: "hipGraph", 0: "direct", -1: "direct"}[path], n_relin)
    if worst[0] > 1.0:
        desc += " | metric of iteration %d beyond 1e-5 rel + 1e-4 px (GPU %.9g, oracle %.9g) with every tensor bit-equal at the end" % worst[1]
    G.close()
    return desc, n_relin


class GatherAll:
    """the all-gather between shard contexts living in one process (what RCCL does between GPUs): every rank's send buffer into every
    rank's receive buffer, by device copies"""

    def __init__(self):
        self.members = []

    def __call__(self):
        import torch
        for m in self.members:
            m.stream.synchronize()
        for dst in self.members:
            for r, src in enumerate(self.members):
                n = src.send.numel()
                dst.recv[r * n:(r + 1) * n].copy_(src.send)
        torch.cuda.synchronize()


def shard_seed(seed):
    """`world` landmark-shard contexts (2..8, ranges balanced by factor count — ranks without a landmark included) on the one GPU, the
    exchange by device copies: the sharded C-ABI verbs (gbp_iterate_begin / _local / _end, gbp_refresh_*, gbp_linearise_factors,
    gbp_weaken_priors on a sharded ctx) against the oracle summing in `world`-shard device order — camera beliefs identical on every rank
    and equal to the oracle's, every rank's own landmark beliefs and per-factor state, bit for bit."""
    from gbp_poplar_amd.distributed import ShardedGbp, landmark_partition
    rng = np.random.default_rng(11000 + seed)
    bal = random_problem(rng)
    C, L, E = bal["n_cams"], bal["n_lmks"], bal["n_edges"]
    opts = driver.Options()
    opts.undamped_start = int(rng.integers(1, 4))
    K, state, _ = driver.build_inputs(bal, opts, hostlib)
    state["active_flag"] = (rng.random(E) < 0.9).astype(np.uint32)
    kw = dict(dmu_threshold=float(rng.choice([0.05, 0.02, 3e-3])), min_linear_iters=int(rng.integers(2, 6)), num_undamped_iters=int(rng.integers(1, 4)))
    world = int(rng.choice([2, 3, 4, 5, 8]))
    bounds = landmark_partition(bal["lmk_id"], L, world)
    gather = GatherAll()
    shards = []
    for r in range(world):
        eng = GbpEngine(bal["cam_id"], bal["lmk_id"], C, L, K, shard=(r, world, int(bounds[r]), int(bounds[r + 1])), params=_cabi.GbpParams.defaults(**kw))
        sh = ShardedGbp(eng, C, r, world, dist=None, device="cuda")
        gather.members.append(sh)
        shards.append(sh)

    def all_do(name, *a):
        for sh in shards:
            getattr(sh.e, name)(*a)

    O = orc_mod.Oracle(bal["cam_id"], bal["lmk_id"], C, L, K, params=_cabi.GbpParams.defaults(**kw))
    O.set_sum_order(1, bounds)
    O.upload(state)
    O.linearise()
    all_do("upload", state)
    all_do("refresh_begin"); gather(); all_do("refresh_end"); all_do("linearise_factors")
    total = int(rng.integers(12, 40))
    steps = int(opts.steps)
    lmk = np.asarray(bal["lmk_id"])
    n_relin = 0
    for it in range(total):
        if (it + 1) % 2 == 0 and it < 2 * steps:
            all_do("weaken_priors")
            O.weaken_priors()
        all_do("iterate_begin")
        if rng.random() < 0.5:
            all_do("iterate_local")          # the landmark half first (what overlaps the all-gather on N GPUs)
        gather()
        all_do("iterate_end")
        O.iterate(1)
        if it < 6 or it == total - 1:
            ro = O.read()
            for r, sh in enumerate(shards):
                g = sh.read()
                same(g["cam_beliefs_eta"], ro["cam_beliefs_eta"], "world %d rank %d after iteration %d: cam_beliefs_eta" % (world, r, it))
                same(g["cam_beliefs_lambda"], ro["cam_beliefs_lambda"], "world %d rank %d after iteration %d: cam_beliefs_lambda" % (world, r, it))
                lo, hi = int(bounds[r]), int(bounds[r + 1])
                same(g["lmk_beliefs_eta"][3 * lo:3 * hi], ro["lmk_beliefs_eta"][3 * lo:3 * hi], "world %d rank %d after iteration %d: lmk_beliefs_eta" % (world, r, it))
                same(g["lmk_beliefs_lambda"][9 * lo:9 * hi], ro["lmk_beliefs_lambda"][9 * lo:9 * hi], "world %d rank %d after iteration %d: lmk_beliefs_lambda" % (world, r, it))
                own = (lmk >= lo) & (lmk < hi)
                for k in ("damping", "damping_count", "robust_flag"):
                    same(g[k][own], ro[k][own], "world %d rank %d after iteration %d: %s" % (world, r, it, k))
            evs = [sh.e.eval() for sh in shards]
            eo = O.eval()
            for k in ("n_active", "n_relin", "n_robust"):
                if sum(e[k] for e in evs) != eo[k]:
                    raise Mismatch("world %d after iteration %d: %s summed over the ranks %d vs oracle %d" % (world, it, k, sum(e[k] for e in evs), eo[k]))
            n_relin += eo["n_relin"]
    desc = "SHARDED world %d (landmark ranges %s) C %d L %d E %d | %d iterations | relinearisations seen %d" % (
        world, "/".join(str(int(bounds[r + 1] - bounds[r])) for r in range(world)), C, L, E, total, n_relin)
    for sh in shards:
        sh.e.close()
    return desc, n_relin


def main():
    orc_mod.load("restatement")
    orc_mod.set_trig_mode(1)
    lib = _lib.load(hooks=True)
    t0 = time.time()
    seed, done, relin_runs, paths = first, 0, 0, {}
    while time.time() - t0 < budget:
        try:
            desc, nr = slam_seed(seed) if seed % 4 == 3 else shard_seed(seed) if seed % 8 == 5 else one_seed(seed, lib)
        except Mismatch as e:
            print("seed %d: MISMATCH — %s" % (seed, e), flush=True)
            return 1
        print("seed %d ok: %s" % (seed, desc), flush=True)
        relin_runs += nr > 0
        key = (desc.split("| B: ")[1].split(" |")[0] if "| B: " in desc else "sharded, world " + desc.split()[2] if desc.startswith("SHARDED")
               else "SLAM flow on " + desc.split("| path ")[1].split(" |")[0])
        paths[key] = paths.get(key, 0) + 1
        done += 1
        seed += 1
    print("SUMMARY: %d seeds (%d..%d) in %.0f s, no mismatch; %d of them relinearised; B's path and driver: %s"
          % (done, first, seed - 1, time.time() - t0, relin_runs, ", ".join("%s x %d" % kv for kv in sorted(paths.items()))), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
