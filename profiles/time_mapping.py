#!/usr/bin/env python3
"""Lane-per-matrix vs the north star's sub-wave mapping (16 lanes cooperate on one matrix, operands in LDS), measured on
the dominant routine of the sweep — the un-pivoted 6x6 LDL^T inverse (matlib.cpp:180-222) — at the matrix counts of
BASELINE's configs: 3 551 (fr2robot2), 12 908 (fr1xyz), 100 000 and 1 000 000 (S1).  Both kernels are bit-identical
(tests/test_gpu_device_math.py); microseconds per launch, 200 back-to-back launches."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gbp_poplar_amd import _cabi as cabi
from gbp_poplar_amd._lib import load
lib = load(hooks=True)
rng = np.random.default_rng(1)
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
