#!/usr/bin/env python3
"""Does the driver's short bench window (5 warm-up + 20 timed iterations = 3 ms of GPU work after seconds of host-side
set-up) run at full clocks?  The same window — S1, iterations 5..24 of a run — timed right after the set-up, after
300 ms of unrelated GPU work, and (for reference) iterations 200..219.     python profiles/clock_ramp.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                   # noqa: E402
import bench                                                   # noqa: E402
from gbp_poplar_amd import _cabi, driver, hostlib              # noqa: E402
from gbp_poplar_amd.engine import GbpEngine                    # noqa: E402

bal = hostlib.synth_generate(1000, 100000, 10, 20200303)
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)


def window(spin_ms, warm, idle_s):
    eng = GbpEngine(bal["cam_id"], bal["lmk_id"], 1000, 100000, K)
    eng.upload(state)
    eng.linearise()
    eng.eval()
    time.sleep(idle_s)                                         # what the host-side set-up of bench.py leaves: an idle GPU
    if spin_ms:
        x = torch.ones(64 << 20, device="cuda")
        t = time.perf_counter()
        while time.perf_counter() - t < spin_ms * 1e-3:
            x.mul_(1.0001)
        torch.cuda.synchronize()
    bench.warm_start(eng, opts, warm)
    eng.prepare()
    eng.sync()
    t0 = time.perf_counter()
    eng.iterate(20)
    eng.sync()
    dt = time.perf_counter() - t0
    eng.close()
    return 1e3 * dt / 20


for label, spin, warm, idle in (("5 warm-up, right after set-up", 0, 5, 0.0), ("5 warm-up, GPU idle 2 s before", 0, 5, 2.0),
                                ("5 warm-up, idle 2 s then 300 ms of other GPU work", 300, 5, 2.0),
                                ("200 warm-up, idle 2 s before", 0, 200, 2.0)):
    r = [window(spin, warm, idle) for _ in range(4)]
    print("%-55s ms/step %s" % (label, " ".join("%.4f" % v for v in r)))
