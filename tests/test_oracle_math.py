"""Pins of the oracle's dense-math layer (reference ba/matlib.cpp + ba/bafuncs.cpp).

1. against the committed golden vectors (tests/golden/math_vectors.npz, produced by the REFERENCE's own
   code through the out-of-tree reference build, see tests/golden/make_golden.py) — runs everywhere, bit for bit;
2. against the reference code itself when the out-of-tree reference build ($TMPDIR/gbp_oracle_ref/libref_math.so)
   is present (build container only — it never travels to the GPU box) on fresh random inputs — bit for bit."""
import os

import numpy as np
import pytest

from gbp_poplar_amd import _cabi as cabi
from oracle import oracle as orc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "math_vectors.npz"))
P = lambda a: cabi.ptr(a, cabi.c_f32p)


def _apply(lib, g=G):
    n = g["inv3_in"].shape[0]
    o = {k: np.zeros_like(g[k]) for k in ("inv3_out", "inv6_out", "so3_out", "hfunc_out", "jac_kf", "jac_lmk")}
    for k in range(n):
        lib.om_inv3x3(P(np.ascontiguousarray(g["inv3_in"][k])), P(o["inv3_out"][k]))
        lib.om_inv6x6(P(np.ascontiguousarray(g["inv6_in"][k])), P(o["inv6_out"][k]))
        lib.om_so3exp(P(np.ascontiguousarray(g["so3_in"][k])), P(o["so3_out"][k]))
        cam, lmk, K = (np.ascontiguousarray(g["proj_cam"][k]), np.ascontiguousarray(g["proj_lmk"][k]),
                       np.ascontiguousarray(g["proj_K"]))
        lib.om_hfunc(P(cam), P(lmk), P(K), P(o["hfunc_out"][k]))
        lib.om_jac(P(cam), P(lmk), P(K), P(o["jac_kf"][k]), P(o["jac_lmk"][k]))
    if "mean6_eta" in getattr(g, "files", g):
        o["mean6_out"], o["mean3_out"] = np.zeros_like(g["mean6_eta"]), np.zeros_like(g["mean3_eta"])
        for k in range(g["mean6_eta"].shape[0]):
            lib.om_inf2mean6x6(P(np.ascontiguousarray(g["mean6_eta"][k])), P(np.ascontiguousarray(g["mean6_lambda"][k])), P(o["mean6_out"][k]))
            lib.om_inf2mean3x3(P(np.ascontiguousarray(g["mean3_eta"][k])), P(np.ascontiguousarray(g["mean3_lambda"][k])), P(o["mean3_out"][k]))
    A, B = np.ascontiguousarray(g["mm_A"]), np.ascontiguousarray(g["mm_B"])
    for name, (x, y, ta, tb, pr, pc) in {"nn": (B, A, 0, 0, 6, 3), "tn": (A, B, 1, 0, 3, 6), "nt": (A, A, 0, 1, 6, 6)}.items():
        p = np.full((pr, pc), 0.25, np.float32)
        lib.om_matmul(P(x), x.shape[0], x.shape[1], P(y), y.shape[0], y.shape[1], P(p), pc, ta, tb)
        o["mm_" + name] = p
    return o


def test_restatement_reproduces_golden_vectors_bit_for_bit():
    lib = orc.load("restatement")
    assert lib.om_impl_name() == b"restatement"
    o = _apply(lib)
    for k, v in o.items():
        assert np.array_equal(v, G[k]), k


def test_golden_vectors_are_sane():
    # inverses really invert, rotations are orthonormal, identity branch below 1e-6 (bafuncs.cpp:38)
    for k in range(G["inv6_in"].shape[0]):
        assert np.allclose(G["inv6_in"][k].astype(np.float64) @ G["inv6_out"][k], np.eye(6), atol=2e-3)
        assert np.allclose(G["inv3_in"][k].astype(np.float64) @ G["inv3_out"][k], np.eye(3), atol=1e-3)
        R = G["so3_out"][k].reshape(3, 3).astype(np.float64)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-5)
    assert np.array_equal(G["so3_out"][0], np.eye(3, dtype=np.float32).ravel())
    assert np.array_equal(G["so3_out"][1], np.eye(3, dtype=np.float32).ravel())
    assert np.all(G["jac_kf"][:, 1] == 0) and np.all(G["jac_kf"][:, 6] == 0)


@pytest.mark.skipif(not orc.have("ref_math"), reason="reference-math build absent (make -C oracle ref; needs /root/reference)")
def test_restatement_equals_reference_code_on_fresh_inputs():
    ref, mine = orc.load("ref_math"), orc.load("restatement")
    assert ref.om_impl_name() == b"reference"
    rng = np.random.default_rng(1234)
    n = 400
    g = {}
    a = rng.standard_normal((n, 3, 3)); g["inv3_in"] = (a @ a.transpose(0, 2, 1) + 0.5 * np.eye(3)).astype(np.float32)
    a = rng.standard_normal((n, 6, 6)); g["inv6_in"] = (a @ a.transpose(0, 2, 1) + 0.5 * np.eye(6)).astype(np.float32)
    g["so3_in"] = (rng.standard_normal((n, 3)) * rng.uniform(1e-4, 2.5, (n, 1))).astype(np.float32)
    cam = np.concatenate([rng.standard_normal((n, 3)), rng.standard_normal((n, 3)) * 0.6], 1).astype(np.float32)
    cam[:, 2] += 5.0
    g["proj_cam"], g["proj_lmk"] = cam, rng.standard_normal((n, 3)).astype(np.float32)
    g["proj_K"] = np.array([500, 0, 320, 0, 500, 240, 0, 0, 1], np.float32)
    g["mm_A"], g["mm_B"] = rng.standard_normal((6, 3)).astype(np.float32), rng.standard_normal((6, 6)).astype(np.float32)
    g["mean6_eta"], g["mean3_eta"] = rng.standard_normal((n, 6)).astype(np.float32), rng.standard_normal((n, 3)).astype(np.float32)
    g["mean6_lambda"], g["mean3_lambda"] = g["inv6_in"], g["inv3_in"]
    for k in ("inv3_out", "inv6_out"):
        g[k] = np.zeros_like(g[k.replace("out", "in")])
    g["so3_out"] = np.zeros((n, 9), np.float32)
    g["hfunc_out"], g["jac_kf"], g["jac_lmk"] = np.zeros((n, 2), np.float32), np.zeros((n, 12), np.float32), np.zeros((n, 6), np.float32)
    a, b = _apply(ref, g), _apply(mine, g)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_correctly_rounded_trig_mode_differs_only_by_ulps():
    lib = orc.load("restatement")
    w = (np.random.default_rng(3).standard_normal((500, 3)) * 0.7).astype(np.float32)
    r0, r1 = np.zeros((500, 9), np.float32), np.zeros((500, 9), np.float32)
    for k in range(500):
        lib.om_so3exp(P(w[k]), P(r0[k]))
    orc.set_trig_mode(1)
    try:
        for k in range(500):
            lib.om_so3exp(P(w[k]), P(r1[k]))
    finally:
        orc.set_trig_mode(0)
    assert np.max(np.abs(r0 - r1)) <= 3e-7
    assert (r0 == r1).all(axis=1).mean() > 0.8      # glibc sinf/cosf are correctly rounded most of the time
