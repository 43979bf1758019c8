#!/usr/bin/env python3
"""Host-side cost of ONE rank of the real `bench.py --gpus N` line up to (and including the host half of) gbp_create: the
synthetic graph of N x (1 000 cameras, 125 000 landmarks, 1.25 M factors), driver.build_inputs (priors, scalings, state), the
landmark partition, and the device order of the rank's shard (csrc/gbp_layout.cpp — what gbp_create builds before it touches
the GPU).  No GPU needed.      python profiles/host_cost.py [N=8] [rank=0]
Wall time per stage and the process's peak RSS; the driver gives the 8-GPU bench 1 800 s."""
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GBP_NO_TORCH", "1")
from gbp_poplar_amd import driver, hostlib          # noqa: E402
from gbp_poplar_amd.distributed import landmark_partition          # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
C, L = 1000 * world, 125000 * world
rss = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0     # MiB
t0 = time.perf_counter()
bal = hostlib.synth_generate(C, L, 10, 20200303)
t1 = time.perf_counter()
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
t2 = time.perf_counter()
bounds = landmark_partition(bal["lmk_id"], L, world)
t3 = time.perf_counter()
y = hostlib.layout_build(bal["cam_id"], bal["lmk_id"], C, L, shard=(rank, world, int(bounds[rank]), int(bounds[rank + 1])))
t4 = time.perf_counter()
print("N = %d, rank %d: %d cameras x %d landmarks x %d factors; shard: %d landmarks, %d factors, %d tiles, rows placed: %s, tiles permuted: %s"
      % (world, rank, C, L, bal["n_edges"], y["L_loc"], y["E_loc"], y["n_tiles"], bool(y["row_slot"].size), bool(y["tile_perm"].size)))
print("synth_generate %.1f s | build_inputs %.1f s | landmark_partition %.2f s | device order of the shard (gbp_layout.cpp) %.2f s | total %.1f s | peak RSS %.0f MiB"
      % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0, rss()))
