#!/usr/bin/env python3
"""Largest-size sanity run on one GPU: python profiles/big_graph.py [n_cams n_lmks]  (default 4000 x 4 000 000 =
40 M factors, ~17 GB of HBM).  Checks that the first sweeps reduce the error, nothing goes non-finite, and reports
time per iteration and bytes allocated (linear in the factor count: the S1 figure x40)."""
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from gbp_poplar_amd import driver, hostlib          # noqa: E402
from gbp_poplar_amd.engine import GbpEngine         # noqa: E402

nc, nl = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4000, 4000000)
t = time.time()
bal = hostlib.synth_generate(nc, nl, 10, 20200303)
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
print("host setup %.1f s: %d cams, %d lmks, %d factors" % (time.time() - t, nc, nl, bal["n_edges"]), flush=True)
t = time.time()
eng = GbpEngine(bal["cam_id"], bal["lmk_id"], nc, nl, K)
eng.upload(state)
eng.linearise()
ev0 = eng.eval()
print("create+upload+linearise %.1f s; initial rmse %.4f px" % (time.time() - t, driver.metric(ev0)[2]), flush=True)
for it in range(10):
    if (it + 1) % 2 == 0:
        eng.weaken_priors()
    eng.iterate(1)
eng.iterate(10)
eng.sync()
t = time.time()
eng.iterate(100)
eng.sync()
dt = (time.time() - t) / 100
ev = eng.eval()
tm = eng.timing()
print("%.3f ms/iteration (%.3f ns per factor-iteration); rmse after 120 sweeps %.4f px; nonfinite %d nonpd %d; device bytes %.2f GB"
      % (dt * 1e3, dt * 1e9 / bal["n_edges"], driver.metric(ev)[2], ev["n_nonfinite"], ev["n_nonpd"],
         tm["device_bytes_allocated"] / 1e9))
assert ev["n_nonfinite"] == 0 and driver.metric(ev)[2] < 0.2 * driver.metric(ev0)[2]
