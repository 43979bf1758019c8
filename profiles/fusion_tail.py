#!/usr/bin/env python3
"""What would folding the sharded iteration's partial-sum launch into the sweep cost the sweep?  (profiles/r06_sharded_timeline.md)
The experiments build's ablation 256 gives EVERY sweep wave the tail that fusion needs — row sums written through (sc1), the wave's stores
acknowledged (s_waitcnt vmcnt(0)), one agent-scope arrival per row on its camera's counter, the wave waiting for what they return — on the
config-5 shard shape; timing only (gbp_debug_time_sweep), interleaved with the product sweep.
    python -m gbp_poplar_amd.build --experiments && python profiles/fusion_tail.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GBP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gbp_poplar_amd", "libgbp_mi355x_exp.so"))
from gbp_poplar_amd import _cabi, driver, hostlib          # noqa: E402
from gbp_poplar_amd.engine import GbpEngine         # noqa: E402

NC = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 125000
bal = hostlib.synth_generate(NC, NL, 10, 20200303)
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
os.environ["GBP_SEG_SKIP"] = "0"          # the ablated kernel is a copy of the plain sweep: compare like with like
eng = GbpEngine(bal["cam_id"], bal["lmk_id"], NC, NL, K, hooks=True, params=_cabi.GbpParams.defaults(persistent=-1))
for rep in range(3):
    eng.upload(state)
    eng.linearise()
    for it in range(20):
        if (it + 1) % 2 == 0 and it < 10:
            eng.weaken_priors()
        eng.iterate(1)
    # (sweeps without belief updates between them drift — the messages settle —, so the two kernels alternate and a tail launch is
    # compared with the mean of its two neighbours)
    seq = []
    for k in range(9):
        abl = 64 + (256 if k % 2 else 0)
        us = C.c_double()
        rc = eng.lib.gbp_debug_time_sweep(eng.h, abl, 60, C.byref(us))
        assert rc == 0
        seq.append(us.value)
    extra = [seq[k] - 0.5 * (seq[k - 1] + seq[k + 1]) for k in range(1, 8, 2)]
    print("%d x %d, run %d: product / tail alternating: %s | tail - mean of its neighbours: %s -> %.2f us" %
          (NC, NL, rep, " ".join("%.1f" % x for x in seq), " ".join("%+.2f" % x for x in extra), sum(extra) / len(extra)), flush=True)
