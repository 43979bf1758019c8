#!/usr/bin/env python3
"""Cost of the in-sweep relinearisation IN SITU: the ./ba flow of a small sequence (LINEARISE, prior weakening, N sweeps,
beliefs after every sweep) with an ablated sweep kernel swapped in (experiments build, GBP_SWEEP_ABL), device time per
iteration from gbp_timing.  One process per variant (the variable is read once):
    for a in 0 64 256 512 768 1024 2048 3840; do GBP_SWEEP_ABL=$a python profiles/insitu_relin.py fr1xyz 600; done"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GBP_LIB", os.path.join(ROOT, "gbp_poplar_amd", "libgbp_mi355x_exp.so"))
from gbp_poplar_amd import driver, hostlib          # noqa: E402
from gbp_poplar_amd.engine import GbpEngine         # noqa: E402

name, n = sys.argv[1], int(sys.argv[2])
bal = hostlib.bal_read(os.path.join(ROOT, "data", "sequences", name + ".txt"))
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
eng = GbpEngine(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K, hooks=True)
eng.upload(state)
eng.linearise()
for it in range(100):
    if (it + 1) % 2 == 0 and it < 10:
        eng.weaken_priors()
    eng.iterate(1)
eng.sync()
eng.timing(reset=True)
eng.iterate(n)
eng.sync()
t = eng.timing(reset=True)
ev = eng.eval()
labels = {0: "product kernel", 64: "no lane ever relinearises", 256: "hardware sin/cos", 512: "reciprocal multiply instead of 54 divisions",
          768: "hardware trig + reciprocal multiply", 1024: "no dependent mean loads", 2048: "no potential write-back", 3840: "all four"}
abl = int(os.environ.get("GBP_SWEEP_ABL", "0"))
print("| %s | %d %s | %.2f us/iteration | relins %d |" % (name, abl, labels.get(abl, "?"), 1e3 * t["total_ms"] / n, ev["n_relin"]))
