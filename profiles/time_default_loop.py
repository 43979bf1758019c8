#!/usr/bin/env python3
"""The reference's default loop (the metric after EVERY iteration, ba.cpp:1001-1053) against the iterations alone, on a synthetic
graph:  python profiles/time_default_loop.py [CAMS LMKS] [N]
  gbp_iterate(N)             the hipGraph replay of k_sweep + k_beliefs
  gbp_iterate_eval_each(N)   the same with the metric riding along (k_sweep<EV> + k_beliefs<EV>, one k_eval_fold per piece)
  N x {gbp_iterate(1); gbp_eval()}   what the entry point did on large graphs before round 5 (direct launches, a host hand-shake per iteration)
us per iteration, best of 5 and median, on one engine in the steady state of the ./ba flow."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gbp_poplar_amd import driver, hostlib          # noqa: E402
from gbp_poplar_amd.engine import GbpEngine          # noqa: E402

cams, lmks = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 100000)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
bal = hostlib.synth_generate(cams, lmks, 10, 20200303)
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
eng = GbpEngine(bal["cam_id"], bal["lmk_id"], cams, lmks, K)
eng.upload(state)
eng.linearise()
for it in range(10):
    if (it + 1) % 2 == 0:
        eng.weaken_priors()
    eng.iterate(1)
eng.iterate(40)
eng.iterate_eval_each(40)


def old_loop(k):
    for _ in range(k):
        eng.iterate(1)
        eng.eval()


def timed(f):
    eng.sync()
    t0 = time.perf_counter()
    f(n)
    eng.sync()
    return (time.perf_counter() - t0) / n * 1e6


rows = {}
for _ in range(5):          # alternating, so that a drift of the box hits all three alike
    for name, f in (("gbp_iterate", eng.iterate), ("gbp_iterate_eval_each", eng.iterate_eval_each), ("iterate(1)+eval", old_loop)):
        rows.setdefault(name, []).append(timed(f))
base = min(rows["gbp_iterate"])
print("%d cameras x %d landmarks x %d factors, %d iterations per call, graph_state %d" % (cams, lmks, bal["n_edges"], n, eng.graph_state()))
for name, v in rows.items():
    print("%-24s best %8.2f us  median %8.2f us  = %.3f x gbp_iterate" % (name, min(v), statistics.median(v), min(v) / base))
ev = eng.iterate_eval_each(1)[0]
print("last metric: mean reproj %.6f px, n_relin %d" % (ev["sum_norm"] / ev["n_active"], ev["n_relin"]))
