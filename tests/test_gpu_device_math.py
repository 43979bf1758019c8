"""HIP device math vs the REFERENCE's own matlib.cpp / bafuncs.cpp — no restated layer in between.

tests/golden/math_vectors.npz holds inputs and the outputs of the reference's templates compiled here
(oracle/ref_adapter.cpp -> `make -C oracle ref`, out of tree; generator tests/golden/make_golden.py).  gbp_debug_math runs the routines of
csrc/gbp_device_math.hpp — the ones k_sweep / k_linearise / k_beliefs are built from — on the GPU, one lane per vector.

Bars: everything without a transcendental is BIT-EXACT (matMul modes, inv3x3, inv6x6, inf2mean).  so3exp / hfunc / Jac
call sin/cos: the device evaluates them correctly rounded, the golden vectors hold glibc's sinf/cosf (not correctly
rounded for ~1-2 % of arguments), so those are bit-exact on the vectors where the two libms agree (>= 85 %) and within
2 ulp of the rotation entries / 2e-6 of the Jacobian's largest entry everywhere.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "math_vectors.npz"))


def _run(op, inp, out_w):
    from gbp_poplar_amd import _cabi as cabi
    from gbp_poplar_amd._lib import load
    lib = load(hooks=True)      # gbp_debug_math lives in the test-hooks build of the same sources
    inp = np.ascontiguousarray(inp, np.float32)
    n = inp.shape[0]
    out = np.zeros((n, out_w), np.float32)
    rc = lib.gbp_debug_math(op, cabi.ptr(inp.reshape(-1), cabi.c_f32p), cabi.ptr(out.reshape(-1), cabi.c_f32p), n)
    assert rc == 0, lib.gbp_last_error(None)
    return out


def test_inv3x3_bit_exact_vs_reference():
    out = _run(0, G["inv3_in"].reshape(-1, 9), 9)
    assert np.array_equal(out, G["inv3_out"].reshape(-1, 9))


def test_inv6x6_bit_exact_vs_reference():
    out = _run(1, G["inv6_in"].reshape(-1, 36), 36)
    assert np.array_equal(out, G["inv6_out"].reshape(-1, 36))
    # the routine reads the lower triangle only (matlib.cpp:195-201): garbage above the diagonal changes nothing
    a = G["inv6_in"].copy()
    a[:, np.triu_indices(6, 1)[0], np.triu_indices(6, 1)[1]] = 1e30
    assert np.array_equal(_run(1, a.reshape(-1, 36), 36), out)


def test_inf2mean_bit_exact_vs_reference():
    m6 = _run(7, np.concatenate([G["mean6_eta"], G["mean6_lambda"].reshape(-1, 36)], axis=1), 6)
    m3 = _run(8, np.concatenate([G["mean3_eta"], G["mean3_lambda"].reshape(-1, 9)], axis=1), 3)
    assert np.array_equal(m6, G["mean6_out"]) and np.array_equal(m3, G["mean3_out"])


def test_accumulating_matmul_modes_bit_exact_vs_reference():
    """matMul accumulates (matlib.cpp:54,64,74): P starts at 0.25 in the golden vectors."""
    A, B = G["mm_A"].ravel(), G["mm_B"].ravel()
    p18, p36 = np.full(18, 0.25, np.float32), np.full(36, 0.25, np.float32)
    nn = _run(4, np.concatenate([A, B, p18])[None, :], 18)
    tn = _run(5, np.concatenate([A, B, p18])[None, :], 18)
    nt = _run(6, np.concatenate([A, p36])[None, :], 36)
    assert np.array_equal(nn.reshape(6, 3), G["mm_nn"])
    assert np.array_equal(tn.reshape(3, 6), G["mm_tn"])
    assert np.array_equal(nt.reshape(6, 6), G["mm_nt"])


def test_so3exp_vs_reference():
    out = _run(2, G["so3_in"], 9)
    ref = G["so3_out"]
    assert np.array_equal(out[:2], ref[:2])                         # identity branch below 1e-6 (bafuncs.cpp:38)
    same = (out == ref).all(axis=1)
    assert same.mean() >= 0.85, same.mean()
    assert np.max(np.abs(out.astype(np.float64) - ref)) <= 2.4e-7   # 2 ulp of an entry of magnitude <= 1


def test_hfunc_and_jac_vs_reference():
    n = G["proj_cam"].shape[0]
    inp = np.concatenate([G["proj_cam"], G["proj_lmk"], np.tile(G["proj_K"], (n, 1))], axis=1)
    out = _run(3, inp, 20)
    hx, jk, jl = out[:, :2], out[:, 2:14], out[:, 14:]
    same = (hx == G["hfunc_out"]).all(axis=1) & (jk == G["jac_kf"]).all(axis=1) & (jl == G["jac_lmk"]).all(axis=1)
    assert same.mean() >= 0.85, same.mean()
    assert np.max(np.abs(hx.astype(np.float64) - G["hfunc_out"])) <= 2e-4                      # pixels (values ~ 300)
    for a, b in ((jk, G["jac_kf"]), (jl, G["jac_lmk"])):
        den = np.max(np.abs(b), axis=1, keepdims=True).astype(np.float64)
        assert np.max(np.abs(a.astype(np.float64) - b) / den) <= 2e-6
    assert np.all(jk[:, 1] == 0) and np.all(jk[:, 6] == 0)                                       # structural zeros


def test_subwave_mapping_of_inv6x6_is_bit_identical():
    """op 9: the north star's sub-wave mapping (16 lanes per matrix, operands in LDS, lane = output element) built
    for the dominant routine: bit-identical to the reference's inv6x6 (golden) and to the lane-per-matrix routine on
    5 000 fresh SPD matrices with ragged tails (n not a multiple of 4, 16 or 64)."""
    out = _run(9, G["inv6_in"].reshape(-1, 36), 36)
    assert np.array_equal(out, G["inv6_out"].reshape(-1, 36))
    rng = np.random.default_rng(5)
    for n in (1, 3, 17, 4999):
        a = rng.standard_normal((n, 6, 6))
        m = (a @ a.transpose(0, 2, 1) + 0.5 * np.eye(6)).astype(np.float32).reshape(n, 36)
        assert np.array_equal(_run(9, m, 36), _run(1, m, 36)), n
