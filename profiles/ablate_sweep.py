#!/usr/bin/env python3
"""Timing-only ablations of k_sweep on the S1 graph (see gbp_debug_time_sweep): which stream costs what.
The ablated instantiations live in the experiments build only (csrc/experiments/, -DGBP_BUILD_EXPERIMENTS):
    python -m gbp_poplar_amd.build --experiments
    GBP_LIB=gbp_poplar_amd/libgbp_mi355x_exp.so python profiles/ablate_sweep.py
(the product and the test-hooks library answer gbp_debug_time_sweep(ablation != 0) with GBP_ERR_INVALID)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbp_poplar_amd import driver, hostlib          # noqa: E402
from gbp_poplar_amd.engine import GbpEngine         # noqa: E402

NC = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
bal = hostlib.synth_generate(NC, NL, 10, 20200303)
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
eng = GbpEngine(bal["cam_id"], bal["lmk_id"], NC, NL, K, hooks=True)
eng.upload(state)
eng.linearise()
for it in range(20):
    if (it + 1) % 2 == 0 and it < 10:
        eng.weaken_priors()
    eng.iterate(1)
names = {0: "full kernel", 1: "no LMSG gather/scatter", 2: "no landmark-belief gather", 3: "no LMSG, no lmk-belief gather",
         4: "no arithmetic", 7: "streams only (no LMSG, no lmkb, no arithmetic)", 8: "LMSG by position (streaming)",
         12: "LMSG by position, no arithmetic", 16: "no LMSG gather (scatter stays)", 32: "no LMSG scatter (gather stays)"}
names.update({100: "k_beliefs (camera + landmark parts)", 101: "k_beliefs camera part only", 102: "k_beliefs landmark part only"})
for abl in (0, 1, 16, 32, 2, 3, 4, 7, 0, 100, 101, 102):
    us = C.c_double()
    rc = eng.lib.gbp_debug_time_sweep(eng.h, abl, 50, C.byref(us))
    print("ablation %2d  %-48s %8.2f us  rc=%d" % (abl, names[abl], us.value, rc))
