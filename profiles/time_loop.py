#!/usr/bin/env python3
"""Mapping experiment (experiments build): the sweep as RESIDENT waves looping over the tiles (k_sweep_loop,
gbp_params.reserved[0] = 2 -> 512 workgroups, or the number of workgroups) against one wave per tile (k_sweep), S1,
per-iteration time of bursts of 100 in the steady state and per-stage times.     python profiles/time_loop.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GBP_LIB", os.path.join(ROOT, "gbp_poplar_amd", "libgbp_mi355x_exp.so"))
import numpy as np                                    # noqa: E402
from gbp_poplar_amd import _cabi as cabi, driver, hostlib          # noqa: E402
from gbp_poplar_amd.engine import GbpEngine          # noqa: E402
import bench                                          # noqa: E402

bal = hostlib.synth_generate(1000, 100000, 10, 20200303)
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
ref = None
print("| sweep | ms per iteration (bursts of 100) | sweep us (per-stage events) | beliefs us | bit-identical |")
print("|---|---|---|---|---|")
for variant in (0, 2, 256, 384, 768, 1024, 0):
    prm = cabi.GbpParams.defaults()
    prm.reserved[0] = variant
    eng = GbpEngine(bal["cam_id"], bal["lmk_id"], 1000, 100000, K, params=prm)
    eng.upload(state)
    eng.linearise()
    bench.warm_start(eng, opts, 40)
    eng.iterate(100)
    eng.sync()
    eng.timing(reset=True)
    eng.iterate(300)
    eng.sync()
    t = eng.timing(reset=True)
    ms = t["total_ms"] / 300
    eng.set_profiling(True)
    eng.iterate(20)
    eng.sync()
    tp = eng.timing(reset=True)
    eng.set_profiling(False)
    out = eng.read()
    same = "-"
    if ref is None:
        ref = out
    else:
        same = str(all(np.array_equal(out[k], ref[k], equal_nan=True) for k in out))
    label = "k_sweep (one wave per tile)" if variant == 0 else "k_sweep_loop, %d workgroups" % (512 if variant == 2 else variant)
    print("| %s | %.4f | %.1f | %.1f | %s |" % (label, ms, 1e3 * tp["sweep_ms"] / 20, 1e3 * tp["belief_ms"] / 20, same))
    eng.close()
