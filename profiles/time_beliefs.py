#!/usr/bin/env python3
"""k_beliefs alone on a synthetic graph: whole kernel / camera part only / landmark part only (us per launch)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GBP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gbp_poplar_amd", "libgbp_mi355x_exp.so"))
from gbp_poplar_amd import driver, hostlib
from gbp_poplar_amd.engine import GbpEngine
cams, lmks = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 100000)
bal = hostlib.synth_generate(cams, lmks, 10, 20200303)
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
eng = GbpEngine(bal["cam_id"], bal["lmk_id"], cams, lmks, K, hooks=True)
eng.upload(state)
eng.linearise()
eng.iterate(12)
for abl, name in ((100, "k_beliefs"), (101, "camera part"), (102, "landmark part")):
    us = C.c_double()
    rc = eng.lib.gbp_debug_time_sweep(eng.h, abl, 50, C.byref(us))
    print("%-14s %8.2f us  rc=%d" % (name, us.value, rc))
