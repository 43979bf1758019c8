#!/usr/bin/env python3
"""Where does an iteration of k_persist spend its time?  Wall-clock stamps (100 MHz) written by lane 0 of every wave
around the two phases and the two barriers (experiments build), averaged over iterations 2..15 of a burst in the
steady state of the ./ba flow.   python profiles/persist_trace.py [fr1xyz] [each]
(`each`: the traced burst is gbp_iterate_eval_each — the metric after every iteration rides in the launch)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GBP_LIB", os.path.join(ROOT, "gbp_poplar_amd", "libgbp_mi355x_exp.so"))
from gbp_poplar_amd import _cabi, driver, hostlib          # noqa: E402
from gbp_poplar_amd.engine import GbpEngine         # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "fr1xyz"
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
lib = eng.lib
lib.gbp_debug_persist_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
waves = lib.gbp_debug_persist_trace(eng.h, None, 0)
lib.gbp_debug_ticks.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.gbp_debug_ticks(eng.h, None, 0)
redo = (C.c_ulonglong * 4)()
lib.gbp_debug_div_redo(redo, 1)
if "each" in sys.argv[2:]:
    eng.iterate_eval_each(32)
else:
    eng.iterate(32)
eng.sync()
lib.gbp_debug_div_redo(redo, 0)
print("div_shared slow path in 32 iterations: %d x (54 values), %d x (9 values), %d x (other)" % (redo[0], redo[1], redo[2]))
buf = np.zeros((waves, 16, 8), np.uint64)
lib.gbp_debug_persist_trace(eng.h, buf.ctypes.data_as(C.c_void_p), waves)
# the buffer is sized for the larger of the two grids a graph is launched with (metric roles: + one wave per camera); the waves
# of THIS launch are the ones that left stamps
waves = (int(np.nonzero(buf[:, 2, 0])[0].max()) + 4) // 4 * 4
buf = buf[:waves]
xcc = ((buf[:, 2, 7] >> np.uint64(8)) & np.uint64(0xff)).astype(int)
hwid = ((buf[:, 2, 7] >> np.uint64(16)) & np.uint64(0xffff)).astype(int)
buf[:, :, 7] &= np.uint64(0xff)
relin_wave = buf[:, :, 7] != 0
t = buf[:, :, :5].astype(np.float64) / 100.0          # us
t_loads = buf[:, :, 5].astype(np.float64) / 100.0
t_upd = buf[:, :, 6].astype(np.float64) / 100.0
Cn, L = bal["n_cams"], bal["n_lmks"]
n_tiles = (eng.timing()["device_bytes_allocated"] and None)
d = np.diff(t, axis=2)[:, 2:, :]            # [wave][iteration][phase A, barrier 1, phase B, barrier 2]
it_time = (t[:, 3:, 0] - t[:, 2:-1, 0])
print("### %s: %d waves; iteration (start to start) %.2f us" % (name, waves, it_time.mean()))
print("| waves | sweep phase us | wait at barrier 1 us | belief phase us | wait at barrier 2 us |")
print("|---|---|---|---|---|")
role = (np.arange(waves) % 4) * (waves // 4) + np.arange(waves) // 4        # belief-phase role of wave w (k_persist: v = wib * workgroups + workgroup)
groups = (("camera waves (%d)" % Cn, role < Cn), ("landmark waves", (role >= Cn) & (role < Cn + (L + 15) // 16)), ("all", slice(0, waves)))
for label, sl in groups:
    m = d[sl].mean(axis=(0, 1))
    print("| %s | %.2f | %.2f | %.2f | %.2f |" % (label, m[0], m[1], m[2], m[3]))
mx = d.max(axis=0).mean(axis=0)
print("| slowest wave per iteration (mean) | %.2f | - | %.2f | - |" % (mx[0], mx[2]))
has_tile = buf[:, 2, 5] != 0
n_relin = buf[has_tile, 2:, 7].astype(np.int64)
ld = (t_loads[has_tile, 2:] - t[has_tile, 2:, 0])
up = (t_upd[has_tile, 2:] - t_loads[has_tile, 2:])
st = (t[has_tile, 2:, 1] - t_upd[has_tile, 2:])
print("sweep waves: %.0f %% have a relinearising lane (mean %.1f lanes)" % (100.0 * (n_relin > 0).mean(), n_relin.mean()))
print("| sweep-phase part | min | median | mean | 90 % | max |\n|---|---|---|---|---|---|")
for label, x in (("loads", ld), ("factor update, no relinearising lane", up[n_relin == 0]), ("factor update, >= 1 relinearising lane", up[n_relin > 0]),
                 ("stores + row sums", st)):
    if x.size:
        print("| %s | %.2f | %.2f | %.2f | %.2f | %.2f |" % (label, x.min(), np.median(x), x.mean(), np.percentile(x, 90), x.max()))
bp = d[:, :, 2]
for label, sl in groups[:2]:
    x = bp[sl]
    print("| belief phase, %s | %.2f | %.2f | %.2f | %.2f | %.2f |" % (label, x.min(), np.median(x), x.mean(), np.percentile(x, 90), x.max()))
# belief phase of the landmark waves against the largest degree among their 16 landmarks (index record: 15 slots; 16..30: second batch)
deg = np.bincount(bal["lmk_id"], minlength=L)
lw = np.nonzero((role >= Cn) & (role < Cn + (L + 15) // 16))[0]
gmax = np.array([deg[(role[w] - Cn) * 16:(role[w] - Cn) * 16 + 16].max() for w in lw])
gsum = np.array([deg[(role[w] - Cn) * 16:(role[w] - Cn) * 16 + 16].sum() for w in lw])
for lo, hi in ((0, 8), (9, 15), (16, 22), (23, 30), (31, 10 ** 6)):
    sel = (gmax >= lo) & (gmax <= hi)
    if sel.any():
        x = bp[lw[sel]]
        print("| landmark waves, largest degree %d..%d (%d waves, %.0f factors per wave) | %.2f | %.2f | %.2f | %.2f | %.2f |"
              % (lo, min(hi, gmax.max()), sel.sum(), gsum[sel].mean(), x.min(), np.median(x), x.mean(), np.percentile(x, 90), x.max()))
tk = np.zeros((waves, 16), np.uint64)
lib.gbp_debug_ticks(eng.h, tk.ctypes.data_as(C.c_void_p), waves)
tk = tk.astype(np.float64) / 100.0
ok = tk[:, 0] > 0
dd = np.diff(tk[ok, :7], axis=1)
print("inside the relinearisation (waves that relinearised in the last iteration: %d), us: min / median / max" % ok.sum())
for i, label in enumerate(("entry -> so3exp", "so3exp (fp64 sin, cos)", "rest of jac_hfunc", "J^T J accumulation", "Huber", "54 divisions")):
    print("  %-24s %.2f / %.2f / %.2f" % (label, dd[:, i].min(), np.median(dd[:, i]), dd[:, i].max()))
slow = up > 1.4 * np.median(up[n_relin > 0])
waves_with_tile = np.nonzero(has_tile)[0]
per_wave = slow.sum(axis=1)
print("outliers (factor update > 1.4 x median): %d wave-iterations of %d; waves that are outliers in >= half of the iterations: %s"
      % (slow.sum(), slow.size, [int(waves_with_tile[i]) for i in np.nonzero(per_wave >= slow.shape[1] // 2)[0]][:40]))
print("outlier count per iteration:", slow.sum(axis=0).tolist())
print("relinearising lanes in outliers: mean %.1f, in the others %.1f" % (n_relin[slow].mean() if slow.any() else 0, n_relin[~slow].mean()))
wid = waves_with_tile
place = ["xcc%d se%d sh%d cu%d simd%d" % (xcc[w], (hwid[w] >> 13) & 7, (hwid[w] >> 12) & 1, (hwid[w] >> 8) & 15, (hwid[w] >> 4) & 3) for w in range(waves)]
cu_key = [(xcc[w], (hwid[w] >> 12) & 0xf, (hwid[w] >> 8) & 15) for w in range(waves)]
from collections import Counter
cnt = Counter(cu_key)
print("waves per (xcc, se/sh, cu): histogram of occupancy:", sorted(Counter(cnt.values()).items()))
simd_key = Counter((xcc[w], (hwid[w] >> 12) & 0xf, (hwid[w] >> 8) & 15, (hwid[w] >> 4) & 3) for w in range(waves))
print("waves per SIMD: histogram:", sorted(Counter(simd_key.values()).items()))
print("blocks per XCC:", sorted(Counter(xcc[::4]).items()))
for i in np.argsort(-per_wave)[:12]:
    w = int(wid[i])
    print("  wave %3d (block %2d): outlier in %2d of %d iterations; %s; waves on its CU: %d" % (w, w // 4, per_wave[i], slow.shape[1], place[w], cnt[cu_key[w]]))
tot = dd.sum(axis=1)
order = np.argsort(-tot)
idx_ok = np.nonzero(ok)[0]
print("relinearisation sections of the 6 slowest and 3 median waves of the last iteration (us): entry, so3exp, rest of jac, JtJ, Huber, 54 div | total | place")
for i in list(order[:6]) + list(order[len(order) // 2 - 1: len(order) // 2 + 2]):
    w = int(idx_ok[i])
    print("  wave %3d: %s | %.2f | %s" % (w, " ".join("%.2f" % x for x in dd[i]), tot[i], place[w]))
if os.environ.get("GBP_TRACE_PLACEMENT"):
    print("block -> placement:", "; ".join("%d: %s" % (b, place[4 * b].replace(" simd%d" % ((hwid[4 * b] >> 4) & 3), "")) for b in range(min(waves // 4, 64))))
