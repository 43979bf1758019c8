#!/usr/bin/env python3
"""One variant of the device order / the sweep's cache policy on one graph shape (test-hooks library: gbp_debug_layout_options,
gbp_debug_force_sweep_policy):
    python profiles/shape_variant.py CAMS LMKS [key=value ...] [policy=P] [iters=N] [direct=1]
keys: row_placement row_window row_place_max_deg row_key_lane classes tile_window tile_min_tiles;  P: SweepArgs.policy bits
(1 camera messages loaded cached, 2 / 4 landmark messages loaded / stored non-temporal; default: by shape).
Prints us per iteration of gbp_iterate(N) (hipGraph replays; best of 5) — or, with direct=1, runs N direct-launch iterations once
and prints nothing: the form `rocprofv3 --pmc` passes are made over (profiles/shape_pmc.sh)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                          # noqa: E402
from gbp_poplar_amd import _cabi, _lib, driver, hostlib          # noqa: E402
from gbp_poplar_amd.engine import GbpEngine          # noqa: E402

cams, lmks = int(sys.argv[1]), int(sys.argv[2])
kv = dict(a.split("=") for a in sys.argv[3:])
iters = int(kv.pop("iters", 200))
direct = int(kv.pop("direct", 0))
policy = int(kv.pop("policy", -1))
lib = _lib.load(hooks=True)
opt = hostlib.layout_options(**{k: int(v) for k, v in kv.items()})
lib.gbp_debug_layout_options(opt)
lib.gbp_debug_force_sweep_policy(policy)
bal = hostlib.synth_generate(cams, lmks, 10, 20200303)
opts = driver.Options()
K, state, _ = driver.build_inputs(bal, opts, hostlib)
eng = GbpEngine(bal["cam_id"], bal["lmk_id"], cams, lmks, K, hooks=True, params=_cabi.GbpParams.defaults(graph_unroll=-1 if direct else 20))
eng.upload(state)
eng.linearise()
bench.ba_flow(eng, opts, 0, 20)
if direct:
    for _ in range(iters):
        eng.iterate(1)
    eng.sync()
    sys.exit(0)
eng.iterate(40)
best = 1e9
for _ in range(5):
    eng.sync()
    t0 = time.perf_counter()
    eng.iterate(iters)
    eng.sync()
    best = min(best, (time.perf_counter() - t0) / iters)
print("%d x %d %s policy=%d : %.2f us per iteration" % (cams, lmks, " ".join("%s=%s" % x for x in sorted(kv.items())) or "(product options)", policy, best * 1e6))
