import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gbp_poplar_amd import driver, hostlib
from gbp_poplar_amd.engine import GbpEngine
bal = hostlib.synth_generate(1000, 100000, 10, 20200303)
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
free0 = torch.cuda.mem_get_info()[0]
for i in range(30):
    e = GbpEngine(bal["cam_id"], bal["lmk_id"], 1000, 100000, K)
    e.upload(state); e.linearise(); e.iterate(12); e.close()
free1 = torch.cuda.mem_get_info()[0]
print("create/destroy x30: free before %.1f MB after %.1f MB" % (free0 / 1e6, free1 / 1e6))
e = GbpEngine(bal["cam_id"], bal["lmk_id"], 1000, 100000, K)
driver.run_ba(e, state, opts, n_iters=10, eval_every=0)
t = time.time(); e.iterate(20000); dt = time.time() - t
ev = e.eval()
print("20000 iterations in %.2f s (%.1f it/s); rmse %.6f nonfinite %d nonpd %d relin %d" % (
    dt, 20000 / dt, (2 * ev["sum_half_sq"] / ev["n_active"]) ** 0.5, ev["n_nonfinite"], ev["n_nonpd"], ev["n_relin"]))
