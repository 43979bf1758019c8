#!/usr/bin/env python3
"""us per iteration of k_persist bursts on a shipped sequence: plain gbp_iterate(100) bursts (no metric in the launch), bursts with
the metric once at the end (gbp_iterate_eval(100)) and with the metric after every iteration (gbp_iterate_eval_each(100)).
    [GBP_LIB=<variant .so>] python3 profiles/time_bursts.py [fr1xyz] [reps] [flow=0]
flow=0: plain bursts in k_persist<false> (counter barriers) instead of k_persist_flow (tagged records), through the test-hooks build."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gbp_poplar_amd import driver, hostlib          # noqa: E402
from gbp_poplar_amd.engine import GbpEngine         # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "fr1xyz"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bal = hostlib.bal_read(os.path.join(ROOT, "data", "sequences", name + ".txt"))
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
flow_off = "flow=0" in sys.argv[3:]
eng = GbpEngine(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K, hooks=flow_off)
if flow_off:
    eng.persist_flow(0)
eng.upload(state)
eng.linearise()
for it in range(10):
    if (it + 1) % 2 == 0:
        eng.weaken_priors()
    eng.iterate(1)
eng.iterate(190)
eng.sync()
out = []
for label, fn in (("iterate(100)", lambda: eng.iterate(100)),
                  ("iterate_eval(100)", lambda: (eng.iterate_eval(100), eng.eval_end())),
                  ("iterate_eval_each(100)", lambda: eng.iterate_eval_each(100))):
    fn()
    eng.sync()
    eng.timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    eng.sync()
    wall = time.perf_counter() - t0
    tm = eng.timing(reset=True)
    out.append("%s: %.2f us/iteration on the device, %.2f wall" % (label, 1e3 * tm["total_ms"] / (100 * reps), 1e6 * wall / (100 * reps)))
print("%s (graph_state %d%s) | " % (name, eng.graph_state(), ", barriers" if flow_off else "") + " | ".join(out))
