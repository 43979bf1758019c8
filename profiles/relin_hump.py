#!/usr/bin/env python3
"""Why ordinary k_sweep launches of the S1 run take 94 us early and late, but 100-106 us in between, with the same bytes.

The ./ba flow on the 1M-factor graph, one sweep at a time with the stage brackets on: per sweep the k_sweep duration (hipEvents
on its stream), the number of factors that relinearised in it (gbp_eval's counter) and the share of 64-factor groups of the
camera-sorted factor list that hold at least one of them (a wavefront whose tile holds ONE relinearising lane runs the whole
relinearisation path, ~1 500 more instructions, for that lane).
    python3 profiles/relin_hump.py [sweeps] > gpurun_out/r04_relin_hump.csv"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gbp_poplar_amd import driver, hostlib          # noqa: E402
from gbp_poplar_amd.engine import GbpEngine         # noqa: E402

n_sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 170
bal = hostlib.synth_generate(1000, 100000, 10, 20200303)
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
eng = GbpEngine(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K)
eng.upload(state)
eng.linearise()
eng.set_profiling(True)
E = bal["n_edges"]
print("sweep,k_sweep_us,k_beliefs_us,n_relin,groups_of_64_with_a_relinearising_factor_pct,mean_relinearising_factors_in_such_a_group")
for it in range(n_sweeps):
    if (it + 1) % 2 == 0 and it < opts.steps * 2:
        eng.weaken_priors()
    eng.sync()
    eng.timing(reset=True)
    eng.iterate(1)
    eng.sync()
    tm = eng.timing(reset=True)
    ev = eng.eval()
    cnt = eng.read()["damping_count"]
    # a relinearisation re-arms the counter to -num_undamped_iters = -8 (the initial count-up from -15 passes -8 once: masked)
    just = (cnt == -8) if ev["n_relin"] > 0 else np.zeros(E, bool)
    pad = (-E) % 64
    g = np.concatenate([just, np.zeros(pad, bool)]).reshape(-1, 64)
    per = g.sum(axis=1)
    hit = per > 0
    print("%d,%.2f,%.2f,%d,%.2f,%.2f" % (it, 1e3 * tm["sweep_ms"], 1e3 * tm["belief_ms"], ev["n_relin"], 100.0 * hit.mean(),
                                         per[hit].mean() if hit.any() else 0.0), flush=True)
