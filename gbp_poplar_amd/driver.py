"""Host-side driver logic of the reference's two programs, engine-agnostic.

Mirrors what `main()` does around the Poplar engine in the reference:
  ba/ba.cpp:479-604   load, K, measurements, prior means, priors, scalings, per-factor state
  ba/ba.cpp:979-1053  WRITE, LINEARISE, READ, eval, loop {WEAKEN_PRIORS?, GBP, READ, eval}
  ba/slam.cpp:589-595,1013-1103  the incremental (keyframe-by-keyframe) variant

`engine` is anything with the program-list verbs (upload, linearise, iterate, weaken_priors, read,
read_priors, new_keyframe, eval) — the product passes gbp_poplar_amd.engine.GbpEngine; `host`
supplies the CPU-side helper functions (the product passes gbp_poplar_amd.hostlib).  The C++ CLIs
(csrc/ba_main.cpp, csrc/slam_main.cpp) implement the same flow natively; this module exists so that
parity tests and bench.py can drive the C-ABI the way the reference's main() drives Poplar.
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class Options:
    """CLI flags and defaults of ba.cpp:394-476 / slam.cpp:394-476."""
    n_iters: int = 1500
    iters_between_kfs: int = 700
    reproj_meas_var: float = 4.0
    prior_std_weaker_factor: float = 100.0
    first_cam_prior_std: float = 0.01
    steps: float = 5.0
    undamped_start: int = 15
    # initialisation options, ba.cpp:422-441 (+ --seed: the reference seeds its noise from the clock)
    tn: float = 0.0
    rn: float = 0.0
    ltn: float = 0.0
    avdepth_on: bool = False
    seed: int = 0


def k_matrix(bal):
    """ba.cpp:494-495"""
    return np.array([bal["fx"], 0.0, bal["cx"], 0.0, bal["fy"], bal["cy"], 0.0, 0.0, 1.0], dtype=np.float32)


def build_inputs(bal, opts, host, slam=False):
    """Everything WRITE_PROG uploads, as a dict of numpy arrays (ba.cpp:494-604, slam.cpp:575-595)."""
    C, L, E = int(bal["n_cams"]), int(bal["n_lmks"]), int(bal["n_edges"])
    K = k_matrix(bal)
    cam_file = np.asarray(bal["cameras"], dtype=np.float64).astype(np.float32)   # ba.cpp:523-528
    lmk_file = np.asarray(bal["points"], dtype=np.float64).astype(np.float32)    # ba.cpp:529-534
    cam_mean, lmk_mean = cam_file.copy(), lmk_file.copy()                       # no init noise by default
    if getattr(opts, "tn", 0) or getattr(opts, "rn", 0) or (getattr(opts, "ltn", 0) and not opts.avdepth_on):   # ba.cpp:536-545
        seed = opts.seed or int.from_bytes(__import__("os").urandom(8), "little")
        cam_mean, lmk_mean = host.init_add_noise(C, L, cam_mean, lmk_mean, opts.tn, opts.rn,
                                                 0.0 if opts.avdepth_on else opts.ltn, seed)
    if getattr(opts, "avdepth_on", False):                                       # ba.cpp:546-548
        lmk_mean = host.init_av_depth(bal["cam_id"], bal["lmk_id"], C, L, cam_mean, lmk_mean)
    cpe, cpl, lpe, lpl = host.set_prior_lambda(bal["cam_id"], bal["lmk_id"], C, L, K, opts.reproj_meas_var,
                                               cam_file, lmk_file, cam_mean, lmk_mean)
    cs, ls = host.prior_scalings(C, L, cpl, opts.steps, opts.prior_std_weaker_factor, opts.first_cam_prior_std)
    steps = int(opts.steps)
    state = {
        "damping": np.zeros(E, np.float32),
        "damping_count": np.full(E, -opts.undamped_start, np.int32),        # ba.cpp:581
        # mu / oldmu are zero in the reference (ba.cpp:582-583): left out = NULL = zeros in the C-ABI
        "cam_scaling": cs, "lmk_scaling": ls,
        "cam_priors_eta": cpe, "cam_priors_lambda": cpl, "lmk_priors_eta": lpe, "lmk_priors_lambda": lpl,
        "measurements": np.asarray(bal["observations"], dtype=np.float64).astype(np.float32),
        "meas_variances": np.full(E, opts.reproj_meas_var, np.float32),
    }
    extra = {}
    if not slam:                                                             # ba.cpp:588-590
        state["active_flag"] = np.ones(E, np.uint32)
        state["cam_weaken_flag"] = np.full(C, steps, np.uint32)
        state["lmk_weaken_flag"] = np.full(L, steps, np.uint32)
    else:                                                                    # slam.cpp:589-595
        active, cwf, lwf, laf = host.slam_create_flags(bal["cam_id"], bal["lmk_id"], C, L, steps)
        state["active_flag"], state["cam_weaken_flag"], state["lmk_weaken_flag"] = active, cwf, lwf
        extra["lmk_active_flag"] = laf
    return K, state, extra


def metric(ev):
    """(mean reprojection error, cost, rmse) from raw eval sums: util.cpp:143, SURVEY 8d."""
    n = max(int(ev["n_active"]), 1)
    return ev["sum_norm"] / n, ev["sum_half_sq"], float(np.sqrt(2.0 * ev["sum_half_sq"] / n))


def run_ba(engine, state, opts, n_iters=None, eval_every=1, log=None):
    """ba.cpp:979-1053.  Returns [(iter, mean_reproj, cost, n_relin, n_robust)], iter -1 = initial."""
    n_iters = opts.n_iters if n_iters is None else n_iters
    engine.upload(state)
    engine.linearise()
    traj = []
    ev = engine.eval()
    m = metric(ev)
    traj.append((-1, m[0], m[1], int(ev["n_relin"]), int(ev["n_robust"])))
    if log:
        log("Initial Reprojection error: %.6f Cost %.6f" % (m[0], m[1]))
    it = 0
    while it < n_iters:
        if ((it + 1) % 2 == 0) and (it < opts.steps * 2):                   # ba.cpp:1003-1006
            if log:
                log("Weakening priors ")
            engine.weaken_priors()
        # up to the next host event (prior weakening, metric read-back, end) in ONE call, like csrc/ba_main.cpp: the
        # engine then runs the burst without leaving the device (hipGraph replay, or k_persist on small graphs)
        if eval_every == 1 and hasattr(engine, "iterate_eval_each"):
            # the reference's default, the metric after EVERY iteration: everything up to the next prior weakening in one call
            burst = 1
            while it + burst < n_iters and not (((it + burst + 1) % 2 == 0) and (it + burst < opts.steps * 2)):
                burst += 1
            for k, ev in enumerate(engine.iterate_eval_each(burst)):
                m = metric(ev)
                traj.append((it + k, m[0], m[1], int(ev["n_relin"]), int(ev["n_robust"])))
                if log:
                    log("Iter %d // Reprojection error %.6f // Cost %.6f // n relins: %d // n robust edges %d"
                        % (it + k, m[0], m[1], ev["n_relin"], ev["n_robust"]))
            it += burst
            continue
        burst = 1
        while (it + burst < n_iters and not (eval_every and (it + burst) % eval_every == 0)
               and not (((it + burst + 1) % 2 == 0) and (it + burst < opts.steps * 2))):
            burst += 1
        engine.iterate(burst)
        it += burst - 1
        if eval_every and ((it + 1) % eval_every == 0 or it == n_iters - 1):
            ev = engine.eval()
            m = metric(ev)
            traj.append((it, m[0], m[1], int(ev["n_relin"]), int(ev["n_robust"])))
            if log:
                log("Iter %d // Reprojection error %.6f // Cost %.6f // n relins: %d // n robust edges %d"
                    % (it, m[0], m[1], ev["n_relin"], ev["n_robust"]))
        it += 1
    return traj


def run_slam(engine, host, bal, state, extra, opts, iters_between_kfs=None, max_iters=None, eval_every=1, log=None):
    """slam.cpp:1013-1103."""
    C, L, E = int(bal["n_cams"]), int(bal["n_lmks"]), int(bal["n_edges"])
    ibk = opts.iters_between_kfs if iters_between_kfs is None else iters_between_kfs
    steps = int(opts.steps)
    active = state["active_flag"].copy()
    cwf, lwf = state["cam_weaken_flag"].copy(), state["lmk_weaken_flag"].copy()
    laf = extra["lmk_active_flag"].copy()
    engine.upload(state)
    engine.linearise()
    traj = []
    ev = engine.eval()
    m = metric(ev)
    traj.append((-1, m[0], m[1], int(ev["n_relin"]), int(ev["n_robust"])))
    if log:
        log("Initial Reprojection error: %.6f Cost %.6f" % (m[0], m[1]))
    niters = (C - 1) * ibk - 1                                              # slam.cpp:1013
    if max_iters is not None:
        niters = min(niters, max_iters)
    it, data_counter = 0, 0
    i = 0
    while i < niters:
        if (i + 1) % ibk == 0:                                              # slam.cpp:1020-1046
            it = 0
            data_counter += 1
            n_new = host.slam_update_flags(bal["cam_id"], bal["lmk_id"], C, L, steps, data_counter,
                                           active, lwf, cwf, laf)
            pri = engine.read_priors()
            bel = engine.read()
            host.slam_initialise_new_kf(data_counter, bel["cam_beliefs_eta"], bel["cam_beliefs_lambda"],
                                        pri["cam_priors_lambda"], pri["cam_priors_eta"])
            engine.new_keyframe({"damping_count": np.full(E, -15, np.int32),   # literal -15, slam.cpp:1040
                                 "cam_priors_eta": pri["cam_priors_eta"], "cam_priors_lambda": pri["cam_priors_lambda"],
                                 "lmk_priors_eta": pri["lmk_priors_eta"], "lmk_priors_lambda": pri["lmk_priors_lambda"],
                                 "active_flag": active, "cam_weaken_flag": cwf, "lmk_weaken_flag": lwf})
            if log:
                log("Adding keyframe %d, %d new landmarks" % (data_counter + 1, n_new))
        if ((it + 1) % 2 == 0) and (it < opts.steps * 2):
            engine.weaken_priors()
        if eval_every == 1 and hasattr(engine, "iterate_eval_each"):
            burst = 1                   # the metric after EVERY iteration, up to the next keyframe / prior weakening in one call
            while (i + burst < niters and (i + burst + 1) % ibk != 0
                   and not (((it + burst + 1) % 2 == 0) and (it + burst < opts.steps * 2))):
                burst += 1
            for k, ev in enumerate(engine.iterate_eval_each(burst)):
                m = metric(ev)
                traj.append((i + k, m[0], m[1], int(ev["n_relin"]), int(ev["n_robust"])))
                if log:
                    log("Iters %d (since last kf %d) // Reprojection error %.6f // Cost %.6f // n relins: %d // n robust edges %d"
                        % (ibk * data_counter + it + k, it + k, m[0], m[1], ev["n_relin"], ev["n_robust"]))
            i += burst
            it += burst
            continue
        burst = 1                       # up to the next keyframe / prior weakening / read-back in one call
        while (i + burst < niters and (i + burst + 1) % ibk != 0 and not (eval_every and (i + burst) % eval_every == 0)
               and not (((it + burst + 1) % 2 == 0) and (it + burst < opts.steps * 2))):
            burst += 1
        engine.iterate(burst)
        i += burst - 1
        it += burst - 1
        if eval_every and ((i + 1) % eval_every == 0 or i == niters - 1):
            ev = engine.eval()
            m = metric(ev)
            traj.append((i, m[0], m[1], int(ev["n_relin"]), int(ev["n_robust"])))
            if log:
                log("Iters %d (since last kf %d) // Reprojection error %.6f // Cost %.6f // n relins: %d // n robust edges %d"
                    % (ibk * data_counter + it, it, m[0], m[1], ev["n_relin"], ev["n_robust"]))
        it += 1
        i += 1
    return traj
