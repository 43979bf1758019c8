#!/usr/bin/env python3
"""Companion of relin_hump.py: the same ./ba flow on S1 BACK TO BACK (no host wait between sweeps), three times from the same
uploaded state in one process (second pass right behind the first, third after one second of idle GPU), under
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python3 profiles/relin_hump_b2b.py [sweeps]
If the slower launches sit at the same SWEEP of every pass they follow the data; if they sit at the same time after the GPU
started working, they follow the clocks."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gbp_poplar_amd import driver, hostlib          # noqa: E402
from gbp_poplar_amd.engine import GbpEngine         # noqa: E402

n_sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
bal = hostlib.synth_generate(1000, 100000, 10, 20200303)
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
eng = GbpEngine(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K)
for p in range(3):
    eng.upload(state)
    eng.linearise()
    eng.sync()
    if p == 2:
        time.sleep(1.0)
    for it in range(n_sweeps):
        if (it + 1) % 2 == 0 and it < opts.steps * 2:
            eng.weaken_priors()
        eng.iterate(1)
    eng.sync()
