#!/usr/bin/env python3
"""The persistent kernels (one launch per burst: k_persist with barriers, k_persist_flow with tagged records) against the two-kernel path (hipGraph of k_sweep + k_beliefs) over graph size:
device microseconds per iteration of gbp_iterate(100) in the steady state of the ./ba flow, synthetic graphs
(10 observations per landmark) and the shipped sequences.   python profiles/persist_crossover.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gbp_poplar_amd import _cabi, driver, hostlib          # noqa: E402
from gbp_poplar_amd.engine import GbpEngine         # noqa: E402


def run(bal, mode, flow=1):
    opts = driver.Options()
    K, state, _ = driver.build_inputs(bal, opts, hostlib)
    eng = GbpEngine(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K, hooks=True, params=_cabi.GbpParams.defaults(persistent=mode))
    st = eng.graph_state()
    if st == 2:
        eng.persist_flow(flow)
    eng.upload(state)
    eng.linearise()
    for it in range(10):
        if (it + 1) % 2 == 0:
            eng.weaken_priors()
        eng.iterate(1)
    eng.iterate(100)
    eng.sync()
    eng.timing(reset=True)
    for _ in range(4):
        eng.iterate(100)
    eng.sync()
    t = eng.timing(reset=True)
    eng.close()
    return 1e3 * t["total_ms"] / 400, st


print("| graph | factors | two kernels (hipGraph) us/it | k_persist (barriers) us/it | k_persist_flow (tagged records) us/it | two kernels / flow |")
print("|---|---|---|---|---|---|")
cases = [(name, hostlib.bal_read(os.path.join(ROOT, "data", "sequences", name + ".txt"))) for name in ("fr2robot2", "fr1xyz", "fr1desk")]
for cams, lmks in ((10, 400), (20, 800), (40, 1600), (60, 3200), (100, 6400), (150, 9600), (200, 12800), (250, 16000), (300, 19200)):
    cases.append(("synthetic %d x %d" % (cams, lmks), hostlib.synth_generate(cams, lmks, 10, 7)))
for name, bal in cases:
    a, _ = run(bal, -1)
    b, st = run(bal, 1, 0)
    f, st = run(bal, 1, 1)
    print("| %s | %d | %.2f | %s | %s | %s |" % (name, bal["n_edges"], a, "%.2f" % b if st == 2 else "n/a", "%.2f" % f if st == 2 else "n/a",
                                           "%.2fx" % (a / f) if st == 2 else "-"), flush=True)
