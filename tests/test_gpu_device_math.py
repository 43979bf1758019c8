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


def _run(op, inp, out_w, hooks=True):
    from gbp_poplar_amd import _cabi as cabi
    from gbp_poplar_amd._lib import load
    lib = load(hooks=hooks)     # gbp_debug_math lives in the test-hooks build of the same sources
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


def test_so3exp_is_the_correctly_rounded_trig_oracle_bit_for_bit(oracle_mod):
    """so3exp on the device evaluates sin / cos in fp64 (ONE argument reduction for both: the device library's sincos) and
    rounds once to fp32; the oracle's trig mode 1 does the same with glibc's fp64 sin / cos.  200 000 rotation vectors over the
    range the sequences and the synthetic graphs use (|w| from 1e-5 to 3.3 rad, plus a few hundred up to 50 rad): every one of
    the nine entries bit for bit (what lets whole chaotic runs with millions of relinearisations be bit-identical)."""
    import ctypes as C
    rng = np.random.default_rng(21)
    n = 200000
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    mag = np.concatenate([10.0 ** rng.uniform(-5, 0, n // 4), rng.uniform(0.0, 3.3, n - n // 4 - 400), rng.uniform(3.3, 50.0, 400)])
    w = (d * mag[:, None]).astype(np.float32)
    dev = _run(2, w, 9)
    lib = oracle_mod.load("restatement")
    ref = np.zeros((n, 9), np.float32)
    oracle_mod.set_trig_mode(1)
    try:
        fp = C.POINTER(C.c_float)
        for i in range(n):
            lib.om_so3exp(w[i].ctypes.data_as(fp), ref[i].ctypes.data_as(fp))
    finally:
        oracle_mod.set_trig_mode(0)
    bad = np.nonzero((dev != ref).any(axis=1))[0]
    assert bad.size == 0, (bad[:5], w[bad[:5]], dev[bad[:5]], ref[bad[:5]])


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


def _exp_lib_present():
    from gbp_poplar_amd import _lib
    return os.path.exists(_lib.EXP_LIB_PATH)


@pytest.mark.skipif(not _exp_lib_present(), reason="experiments build absent (python -m gbp_poplar_amd.build --experiments)")
def test_subwave_mapping_of_inv6x6_is_bit_identical():
    """op 9 (experiments build, csrc/experiments/): the north star's sub-wave mapping (16 lanes per matrix, operands in LDS, lane =
    output element) built for the dominant routine: bit-identical to the reference's inv6x6 (golden) and to the lane-per-matrix
    routine on 5 000 fresh SPD matrices with ragged tails (n not a multiple of 4, 16 or 64)."""
    out = _run(9, G["inv6_in"].reshape(-1, 36), 36, hooks="exp")
    assert np.array_equal(out, G["inv6_out"].reshape(-1, 36))
    rng = np.random.default_rng(5)
    for n in (1, 3, 17, 4999):
        a = rng.standard_normal((n, 6, 6))
        m = (a @ a.transpose(0, 2, 1) + 0.5 * np.eye(6)).astype(np.float32).reshape(n, 36)
        assert np.array_equal(_run(9, m, 36, hooks="exp"), _run(1, m, 36)), n


def test_shared_reciprocal_division_is_the_ieee_division():
    """div_shared (one fp64 reciprocal per divisor, 3 instructions per quotient) against numpy's IEEE fp32 division, bit for
    bit: random pairs over the whole exponent range, a dense sweep of exponents in [-60, 60] with extreme significands and
    the edges of that range, quotients engineered to sit next to fp32 rounding midpoints, divisors with
    all-ones / all-zeros significands, powers of two, signed zeros, infinities, NaNs and the subnormal range (where the fast
    path must hand over to the slow one)."""
    rng = np.random.default_rng(12)
    cases = []
    n = 200000
    # (a) random significands, moderate exponents (the regime of the potentials)
    cases.append((rng.standard_normal((n, 9)).astype(np.float32) * np.float32(1e3), (rng.random(n).astype(np.float32) + np.float32(0.1)) * np.float32(8)))
    # (b) random bit patterns: every exponent, subnormals, infs, NaNs
    cases.append((rng.integers(0, 2 ** 32, (n, 9), dtype=np.uint64).astype(np.uint32).view(np.float32),
                  rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)))
    # (c) quotients next to midpoints: x = fl(m * (q +- half an ulp of q)) for random q, m
    m = (rng.random(n).astype(np.float32) + np.float32(1.0)) * np.float32(3.0)
    q = (rng.random((n, 9)).astype(np.float32) + np.float32(1.0))
    half = np.spacing(q) * np.float32(0.5)
    x = ((q.astype(np.float64) + half.astype(np.float64) * rng.choice([-1.0, 1.0], (n, 9))) * m[:, None].astype(np.float64)).astype(np.float32)
    cases.append((x, m))
    # (e) moderate exponents, uniform in [-60, 60], random and extreme significands; the edges of that range
    def spread(shape):
        sig = rng.integers(0, 2 ** 23, shape, dtype=np.uint64)
        sig = np.where(rng.random(shape) < 0.05, rng.choice(np.array([0, 1, 2 ** 23 - 1, 2 ** 23 - 2, 2 ** 22], np.uint64), shape), sig)
        bits = (rng.integers(0, 2, shape, dtype=np.uint64) << 31) | (rng.integers(127 - 60, 127 + 60, shape, dtype=np.uint64) << 23) | sig
        return bits.astype(np.uint32).view(np.float32)
    for _ in range(3):
        cases.append((spread((n, 9)), spread(n)))
    lo, hi = np.float32(2.0 ** -60), np.float32(2.0 ** 60)
    edge = np.array([lo, np.nextafter(lo, np.float32(0)), np.nextafter(lo, np.float32(1)), hi, np.nextafter(hi, np.float32(0)),
                     np.nextafter(hi, np.float32(np.inf)), -lo, -hi, 0.0, -0.0, 1.5, -3.25, 7.0], np.float32)
    ex = np.resize(np.concatenate([edge, -edge]), 27).reshape(3, 9)
    for mm in edge:
        if mm != 0:
            cases.append((ex, np.full(3, mm, np.float32)))
    # (f) near-midpoint quotients again, now with divisors whose significand is all ones / nearly so, and negative ones
    for top in (np.float32(2.0) - np.spacing(np.float32(1.0)), np.float32(2.0) - 2 * np.spacing(np.float32(1.0)), np.float32(-1.0) - np.spacing(np.float32(1.0))):
        mq = np.full(n, top, np.float32) * np.float32(2.0) ** rng.integers(-20, 20, n).astype(np.float32)
        q = (rng.random((n, 9)).astype(np.float32) + np.float32(1.0))
        half = np.spacing(q) * np.float32(0.5)
        x = ((q.astype(np.float64) + half.astype(np.float64) * rng.choice([-1.0, 1.0], (n, 9))) * mq[:, None].astype(np.float64)).astype(np.float32)
        cases.append((x, mq))
    # (d) special divisors: significand all ones / all zeros, powers of two, tiny, huge
    special = np.array([1.0, 2.0, 0.5, np.float32(2.0) - np.spacing(np.float32(1.0)), np.float32(1.0) + np.spacing(np.float32(1.0)), 3.0,
                        1e-30, 1e30, 1e-38, 3e38, 1e-45, 0.0, -0.0, np.inf, -np.inf, np.nan, -7.0, 4.0], np.float32)
    xs = np.concatenate([special, rng.standard_normal(200).astype(np.float32), np.float32(1e-38) * rng.standard_normal(50).astype(np.float32),
                         np.float32(1e-44) * np.arange(1, 20, dtype=np.float32)])
    xs = np.resize(xs, (len(xs) + 8) // 9 * 9).reshape(-1, 9)
    for mm in special:
        cases.append((xs, np.full(xs.shape[0], mm, np.float32)))
    with np.errstate(all="ignore"):
        for x, m in cases:
            x, m = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(m, np.float32)
            out = _run(10, np.concatenate([x, m[:, None]], axis=1), 9)
            ref = x / m[:, None]
            same = (out.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(out) & np.isnan(ref))
            assert same.all(), (x[~same][:4], np.broadcast_to(m[:, None], x.shape)[~same][:4], out[~same][:4], ref[~same][:4])
