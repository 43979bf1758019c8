#!/usr/bin/env python3
"""Device us per iteration of gbp_iterate(100) through k_persist on the shipped sequences, for the placement given by
GBP_PERSIST_SPREAD (experiments build; one process per setting)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GBP_LIB", os.path.join(ROOT, "gbp_poplar_amd", "libgbp_mi355x_exp.so"))
from gbp_poplar_amd import _cabi, driver, hostlib          # noqa: E402
from gbp_poplar_amd.engine import GbpEngine         # noqa: E402
out = []
for name in ("fr2robot2", "fr1xyz", "fr1desk"):
    bal = hostlib.bal_read(os.path.join(ROOT, "data", "sequences", name + ".txt"))
    opts = driver.Options()
    K, state, _ = driver.build_inputs(bal, opts, hostlib)
    eng = GbpEngine(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K, hooks=True, params=_cabi.GbpParams.defaults(persistent=1))
    eng.upload(state)
    eng.linearise()
    for it in range(10):
        if (it + 1) % 2 == 0:
            eng.weaken_priors()
        eng.iterate(1)
    eng.iterate(190)
    eng.sync()
    eng.timing(reset=True)
    for _ in range(6):
        eng.iterate(100)
    eng.sync()
    t = eng.timing(reset=True)
    out.append("%s %.2f" % (name, 1e3 * t["total_ms"] / 600))
    eng.close()
print("spread %s: %s us/iteration" % (os.environ.get("GBP_PERSIST_SPREAD", "1"), ", ".join(out)))
