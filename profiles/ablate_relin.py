#!/usr/bin/env python3
"""What does the in-sweep relinearisation cost, and which part of it?  Timing-only ablations of k_sweep (results are
garbage) from the experiments build:
    python -m gbp_poplar_amd.build --experiments
    GBP_LIB=gbp_poplar_amd/libgbp_mi355x_exp.so python profiles/ablate_relin.py [fr1xyz | CAMS LMKS]
us per launch, 50 back-to-back launches (small graphs: includes ~2 us of launch overhead per launch)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GBP_LIB", os.path.join(ROOT, "gbp_poplar_amd", "libgbp_mi355x_exp.so"))
from gbp_poplar_amd import driver, hostlib          # noqa: E402
from gbp_poplar_amd.engine import GbpEngine         # noqa: E402

args = sys.argv[1:]
if len(args) == 1:
    bal = hostlib.bal_read(os.path.join(ROOT, "data", "sequences", args[0] + ".txt"))
    name = args[0]
else:
    nc, nl = (int(args[0]), int(args[1])) if len(args) == 2 else (1000, 100000)
    bal = hostlib.synth_generate(nc, nl, 10, 20200303)
    name = "synthetic %d x %d" % (nc, nl)
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
eng = GbpEngine(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K, hooks=True)
rows = [(64, "no lane relinearises"), (128, "every lane relinearises"), (128 + 256, "  ... hardware sin/cos instead of fp64"),
        (128 + 512, "  ... 54 x multiply by 1/var' instead of 54 IEEE divisions"),
        (128 + 1024, "  ... linearisation point without the dependent mean loads"),
        (128 + 2048, "  ... without the potential write-back"), (128 + 256 + 512, "  ... hardware trig + reciprocal multiply"),
        (128 + 256 + 512 + 1024 + 2048, "  ... all four"), (0, "product kernel (data-dependent relinearisation)"),
        (100, "k_beliefs"), (101, "k_beliefs camera part"), (102, "k_beliefs landmark part")]
print("### %s: %d factors" % (name, bal["n_edges"]))
print("| ablation | us per launch |\n|---|---|")
for abl, label in rows:
    eng.upload(state)
    eng.linearise()
    for it in range(12):
        if (it + 1) % 2 == 0 and it < 10:
            eng.weaken_priors()
        eng.iterate(1)
    us = C.c_double()
    rc = eng.lib.gbp_debug_time_sweep(eng.h, abl, 50, C.byref(us))
    print("| %d %s | %.2f |" % (abl, label, us.value) if rc == 0 else "| %d %s | rc=%d |" % (abl, label, rc))
