import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gbp_poplar_amd import driver, hostlib
from gbp_poplar_amd.engine import GbpEngine
bal = hostlib.synth_generate(1000, 100000, 10, 20200303)
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
def cycle(n):
    for i in range(n):
        e = GbpEngine(bal["cam_id"], bal["lmk_id"], 1000, 100000, K)
        e.upload(state); e.linearise(); e.iterate(12); e.eval(); e.close()
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]
cycle(3)                                   # runtime pools (events, kernel arguments, graph memory) fill up once
free0 = cycle(1)
free1 = cycle(30)
free2 = cycle(60)
print("create/destroy: free after warm-up %.1f MB, +30 cycles %.1f MB, +60 more %.1f MB" % (free0 / 1e6, free1 / 1e6, free2 / 1e6))
e = GbpEngine(bal["cam_id"], bal["lmk_id"], 1000, 100000, K)
driver.run_ba(e, state, opts, n_iters=10, eval_every=0)
t = time.time(); e.iterate(20000); e.sync(); dt = time.time() - t     # gbp_iterate only queues the work
ev = e.eval()
print("20000 iterations in %.2f s (%.1f it/s); rmse %.6f nonfinite %d nonpd %d relin %d" % (
    dt, 20000 / dt, (2 * ev["sum_half_sq"] / ev["n_active"]) ** 0.5, ev["n_nonfinite"], ev["n_nonpd"], ev["n_relin"]))
