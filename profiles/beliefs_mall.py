#!/usr/bin/env python3
"""Does the landmark gather of k_beliefs find the message records in the Infinity Cache (MALL, 256 MiB)?  
The 64 MB of factor->landmark message records it gathers were stored by the sweep microseconds earlier — but the sweep also
streams 570 MB of other tiles through the same cache.  Three situations, the SAME k_beliefs launch (gbp_refresh_end /
gbp_iterate_end: camera part from stored partial sums, landmark part as in every iteration), timed with events on its stream:
  A  back to back        nothing else runs in between: every line it touches was touched 17 us ago (cache-resident if anything is)
  B  behind the sweep    the situation of every iteration (gbp_iterate_begin = k_sweep + partial sums, then the launch)
  C  behind a flush      2 GiB of unrelated traffic in front of it: records, index records and priors come from DRAM
If B sits at A and far from C, the gather already hits the cache inside the iteration and there is nothing left to win from
cache policy or block order.       python profiles/beliefs_mall.py [CAMS LMKS]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                          # noqa: E402
import bench                                          # noqa: E402
from gbp_poplar_amd import driver, hostlib          # noqa: E402
from gbp_poplar_amd.engine import GbpEngine          # noqa: E402

cams, lmks = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 100000)
bal = hostlib.synth_generate(cams, lmks, 10, 20200303)
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
eng = GbpEngine(bal["cam_id"], bal["lmk_id"], cams, lmks, K, shard=(0, 1, 0, lmks))     # the split-phase verbs need a shard ctx (1 of 1)
s = torch.cuda.Stream()
eng.set_stream(s.cuda_stream)
send = torch.zeros(cams * 44, dtype=torch.float32, device="cuda")
recv = torch.zeros(cams * 44, dtype=torch.float32, device="cuda")
eng.set_exchange_buffers(send.data_ptr(), recv.data_ptr())
torch.cuda.synchronize()
eng.upload(state)


def exchange():
    with torch.cuda.stream(s):
        recv.copy_(send)


eng.refresh_begin(); exchange(); eng.refresh_end(); eng.linearise_factors()
for it in range(30):
    if (it + 1) % 2 == 0 and it < 10:
        eng.weaken_priors()
    eng.iterate_begin(); exchange(); eng.iterate_end()
flush = torch.zeros(512 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda")     # 512 MiB: read + write = 1 GiB per pass, two passes


def timed(before, reps=30):
    out = []
    for _ in range(reps):
        before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        eng.refresh_end()
        e1.record(s)
        e1.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    return out


def nothing():
    pass


def sweep():
    eng.iterate_begin()
    exchange()


def flushed():
    with torch.cuda.stream(s):
        flush.mul_(1.0001)
        flush.mul_(0.9999)


def sweep_then_flush():
    sweep()
    flushed()


timed(nothing, 5)
rows = [("A  back to back", timed(nothing)), ("B  behind the sweep (every iteration)", timed(sweep)), ("C  behind a 2 GiB flush", timed(flushed)),
        ("D  behind the sweep AND a flush", timed(sweep_then_flush))]
print("%d cameras x %d landmarks x %d factors; k_beliefs (camera part from partial sums + landmark part), us per launch" % (cams, lmks, bal["n_edges"]))
for name, v in rows:
    print("%-40s min %6.2f  median %6.2f  max %6.2f" % (name, min(v), statistics.median(v), max(v)))
