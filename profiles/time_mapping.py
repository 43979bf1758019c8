#!/usr/bin/env python3
"""Mapping experiments behind DESIGN.md 2 (one lane per factor, two wavefronts per SIMD) — everything is built from this
tree (`python -m gbp_poplar_amd.build --experiments`) and timed here; output = the tables of profiles/r03_mapping.md.

 A. the dominant routine, the un-pivoted 6x6 LDL^T inverse (matlib.cpp:180-222), lane per matrix (inv6x6_lower as k_sweep
    uses it) against the north star's sub-wave mapping (k_inv6_coop: 16 lanes per matrix, operands in LDS, lane = output
    element, reference order) — both bit-identical to the reference (tests/test_gpu_device_math.py);
 B. the WHOLE sweep: the product kernel (242 VGPRs, 2 waves per SIMD), the same kernel forced to 3 waves per SIMD
    (k_sweep_w3: amdgpu_waves_per_eu(3,3) => 168 VGPRs + spills), and k_sweep_coop16 — the sweep in the sub-wave mapping:
    16 lanes per factor, the factor's blocks staged in LDS, every product / inverse with lane = output element;
    us per launch, 50 back-to-back launches on the state after 12 iterations (no lane relinearises yet)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GBP_LIB", os.path.join(ROOT, "gbp_poplar_amd", "libgbp_mi355x_exp.so"))
import numpy as np                                    # noqa: E402
from gbp_poplar_amd import _cabi as cabi, driver, hostlib          # noqa: E402
from gbp_poplar_amd._lib import load                 # noqa: E402
from gbp_poplar_amd.engine import GbpEngine          # noqa: E402

lib = load(hooks=True)
rng = np.random.default_rng(1)
print("### A. 6x6 inverse alone\n")
print("| matrices | lane per matrix (us) | 16 lanes per matrix (us) | ratio |")
print("|---|---|---|---|")
for n in (3551, 12908, 100000, 1000000):
    a = rng.standard_normal((n, 6, 6)).astype(np.float32)
    m = np.ascontiguousarray((a @ a.transpose(0, 2, 1) + 0.5 * np.eye(6, dtype=np.float32)).reshape(n, 36))
    t = []
    for op in (1, 9):
        out = np.zeros((n, 36), np.float32)
        us = C.c_double()
        rc = lib.gbp_debug_math_timed(op, cabi.ptr(m.reshape(-1), cabi.c_f32p), cabi.ptr(out.reshape(-1), cabi.c_f32p), n, 200, C.byref(us))
        assert rc == 0
        t.append(us.value)
    print("| %d | %.2f | %.2f | %.2fx |" % (n, t[0], t[1], t[1] / t[0]))

print("\n### B. the whole sweep\n")
print("| graph | factors | product sweep, 2 waves/SIMD (us) | forced to 3 waves/SIMD (us) | 16 lanes per factor (us) |")
print("|---|---|---|---|---|")
cases = [(name, hostlib.bal_read(os.path.join(ROOT, "data", "sequences", name + ".txt"))) for name in ("fr2robot2", "fr1xyz")]
cases += [("synthetic 100 x 10000", hostlib.synth_generate(100, 10000, 10, 20200303)), ("S1", hostlib.synth_generate(1000, 100000, 10, 20200303))]
for name, bal in cases:
    opts = driver.Options()
    K, state, _ = driver.build_inputs(bal, opts, hostlib)
    eng = GbpEngine(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K, hooks=True, params=cabi.GbpParams.defaults(persistent=-1))
    row = []
    for abl in (0, 3000, 3001):
        eng.upload(state)
        eng.linearise()
        for it in range(12):
            if (it + 1) % 2 == 0 and it < 10:
                eng.weaken_priors()
            eng.iterate(1)
        us = C.c_double()
        rc = eng.lib.gbp_debug_time_sweep(eng.h, abl, 50, C.byref(us))
        row.append("%.2f" % us.value if rc == 0 else "n/a")
    print("| %s | %d | %s | %s | %s |" % (name, bal["n_edges"], row[0], row[1], row[2]))
    eng.close()
