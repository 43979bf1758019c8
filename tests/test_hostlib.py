"""Product host code (csrc/gbp_host.cpp through the C-ABI) against the oracle's independent restatement
of the same reference functions, plus the synthetic generator's contract and error behaviour."""
import os

import numpy as np
import pytest

from gbp_poplar_amd import driver, hostlib
from oracle import oracle as orc
from tests.conftest import seq_path
from tests.oracle_host import OracleHost


@pytest.mark.parametrize("name,dims", [("fr1xyz", (42, 2194, 12908)), ("fr2robot2", (20, 862, 3551)),
                                       ("fr1desk", (63, 2869, 13298))])
def test_bal_loader(name, dims):
    a, b = hostlib.bal_read(seq_path(name)), orc.bal_read(seq_path(name))
    assert (a["n_cams"], a["n_lmks"], a["n_edges"]) == dims        # SURVEY section 2, component 10
    for k in ("cam_id", "lmk_id", "observations", "cameras", "points"):
        assert np.array_equal(a[k], b[k]), k
    assert (a["fx"], a["fy"], a["cx"], a["cy"]) == (b["fx"], b["fy"], b["cx"], b["cy"])
    assert np.all(np.diff(a["cam_id"].astype(np.int64)) >= 0)      # files are camera-sorted (util.cpp:95-99 relies on it)


def test_bal_loader_errors(tmp_path):
    with pytest.raises(IOError):
        hostlib.bal_read(str(tmp_path / "missing.txt"))             # ba.cpp:484-487
    bad = tmp_path / "bad.txt"
    bad.write_text("2 3 4\n500 500 320 240\n0 0 1.0\n")
    with pytest.raises(IOError):
        hostlib.bal_read(str(bad))                                  # truncated: an error, not a silent print


def test_bal_write_read_round_trip(tmp_path):
    bal = hostlib.synth_generate(5, 40, 3, 9)
    p = str(tmp_path / "s.txt")
    hostlib.bal_write(p, bal)
    back = hostlib.bal_read(p)
    for k in ("cam_id", "lmk_id", "observations", "cameras", "points"):
        assert np.array_equal(bal[k], back[k]), k                   # %.16e round-trips doubles exactly


@pytest.mark.parametrize("name", ["fr2robot2", "fr1xyz"])
def test_priors_scalings_and_state_match_oracle(name):
    bal = hostlib.bal_read(seq_path(name))
    opts = driver.Options()
    Ka, sa, _ = driver.build_inputs(bal, opts, hostlib)
    Kb, sb, _ = driver.build_inputs(bal, opts, OracleHost())
    assert np.array_equal(Ka, Kb)
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    lam = sa["cam_priors_lambda"].reshape(-1, 6, 6)
    assert np.all(lam[:, np.arange(6), np.arange(6)] > 0) and np.count_nonzero(lam) == 6 * bal["n_cams"]
    # weakening: 5 steps take cameras 0,1 to 1/std^2 and everything else down by weaker^2 (ba.cpp:561-572)
    s = sa["cam_scaling"].astype(np.float64)
    assert np.isclose(lam[0, 0, 0] * s[0] ** 5, 1 / 0.01 ** 2, rtol=1e-5)
    assert np.isclose(s[2] ** 5, 1e-4, rtol=1e-5) and np.isclose(float(sa["lmk_scaling"][0]) ** 5, 1e-4, rtol=1e-5)


def test_slam_flags_and_new_kf_match_oracle():
    bal = hostlib.bal_read(seq_path("fr2robot2"))
    C, L = bal["n_cams"], bal["n_lmks"]
    a = hostlib.slam_create_flags(bal["cam_id"], bal["lmk_id"], C, L, 5)
    b = orc.slam_create_flags(bal["cam_id"], bal["lmk_id"], C, L, 5)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert a[0].sum() == np.sum(bal["cam_id"] <= 1) and a[1][:2].tolist() == [5, 5] and a[1][2:].sum() == 0
    fa, fb = [x.copy() for x in a], [x.copy() for x in b]
    for dc in (1, 2, 3):
        na = hostlib.slam_update_flags(bal["cam_id"], bal["lmk_id"], C, L, 5, dc, fa[0], fa[2], fa[1], fa[3])
        nb = orc.slam_update_flags(bal["cam_id"], bal["lmk_id"], C, L, 5, dc, fb[0], fb[2], fb[1], fb[3])
        assert na == nb
        for x, y in zip(fa, fb):
            assert np.array_equal(x, y)
        assert fa[1][dc + 1] == 5 and fa[1].sum() == 5 and set(np.unique(fa[2])) <= {0, 5}
    rng = np.random.default_rng(0)
    m = rng.standard_normal((C, 6, 6))
    cbl = (m @ m.transpose(0, 2, 1) + 6 * np.eye(6)).astype(np.float32).ravel()
    cbe = rng.standard_normal(6 * C).astype(np.float32)
    cpl = (np.tile(np.eye(6), (C, 1, 1)) * 3.5).astype(np.float32).ravel()
    ea, eb = np.zeros(6 * C, np.float32), np.zeros(6 * C, np.float32)
    hostlib.slam_initialise_new_kf(2, cbe, cbl, cpl, ea)
    orc.slam_initialise_new_kf(2, cbe, cbl, cpl, eb)
    assert np.array_equal(ea, eb) and np.any(ea[18:24] != 0) and not np.any(ea[:18])
    mu = np.linalg.solve(cbl.reshape(C, 6, 6)[2].astype(np.float64), cbe[12:18].astype(np.float64))
    assert np.allclose(ea[18:24], 3.5 * mu, rtol=1e-5)


def test_host_metric_matches_oracle_and_reference_accumulation():
    host = OracleHost()
    bal = hostlib.bal_read(seq_path("fr2robot2"))
    opts = driver.Options()
    K, state, _ = driver.build_inputs(bal, opts, hostlib)
    o = orc.Oracle(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K)
    driver.run_ba(o, state, opts, n_iters=3, eval_every=0)
    r = o.read()
    args = (bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K, state["active_flag"], state["measurements"],
            r["cam_beliefs_eta"], r["cam_beliefs_lambda"], r["lmk_beliefs_eta"], r["lmk_beliefs_lambda"])
    a, b = hostlib.eval_host(*args), orc.eval_host(*args)
    assert a == b
    f32 = orc.eval_host_f32(*args)                    # the reference's own fp32 sequential accumulation
    assert abs(f32[0] - a[0] / a[2]) <= 2e-6 * f32[0] and abs(f32[1] - a[1]) <= 2e-5 * f32[1]


def test_synthetic_generator_contract():
    a = hostlib.synth_generate(50, 2000, 10, 20200303, ground_truth=True)
    b = hostlib.synth_generate(50, 2000, 10, 20200303)
    for k in ("cam_id", "lmk_id", "observations", "cameras", "points"):
        assert np.array_equal(a[k], b[k])                           # deterministic (counter-based PRNG)
    assert a["n_edges"] == 20000 and (a["fx"], a["fy"], a["cx"], a["cy"]) == (500, 500, 320, 240)
    key = a["cam_id"].astype(np.int64) * 2000 + a["lmk_id"]
    assert np.all(np.diff(key) > 0)                                 # sorted by (camera, landmark), no duplicates
    assert np.all(np.bincount(a["lmk_id"], minlength=2000) == 10)
    gc, gp = a["gt_cameras"].reshape(-1, 6), a["gt_points"].reshape(-1, 3)
    assert np.all(np.abs(gp) <= 2.0) and np.all(np.sum(gc[:, 3:] ** 2, axis=1) >= 1e-3)
    assert np.array_equal(a["cameras"][:12], a["gt_cameras"][:12])  # cameras 0,1 are the exact gauge anchors
    # every observation has positive depth and lands within ~5 sigma of the exact projection
    w = gc[a["cam_id"], 3:]
    th = np.linalg.norm(w, axis=1, keepdims=True)
    y = gp[a["lmk_id"]]
    kx = w / th
    Ry = y * np.cos(th) + np.cross(kx, y) * np.sin(th) + kx * np.sum(kx * y, 1, keepdims=True) * (1 - np.cos(th))
    pc = Ry + gc[a["cam_id"], :3]
    assert np.all(pc[:, 2] > 3.0) and np.all(pc[:, 2] < 17.0)
    uv = np.stack([500 * pc[:, 0] / pc[:, 2] + 320, 500 * pc[:, 1] / pc[:, 2] + 240], 1)
    res = a["observations"].reshape(-1, 2) - uv
    assert np.all(np.abs(res) < 6.0) and 0.9 < res.std() < 1.1
    c = hostlib.synth_generate(50, 2000, 10, 1)
    assert not np.array_equal(a["observations"], c["observations"])


def test_synthetic_graph_converges_under_the_oracle():
    """a graph drawn per the 8(d) spec converges smoothly (7.3 px -> ~1.15 px) with no blow-up."""
    bal = hostlib.synth_generate(30, 1000, 10, 20200303)
    opts = driver.Options()
    K, state, _ = driver.build_inputs(bal, opts, hostlib)
    o = orc.Oracle(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K)
    traj = driver.run_ba(o, state, opts, n_iters=80, eval_every=80)
    assert 5.0 < traj[0][1] < 12.0 and 1.0 < traj[-1][1] < 1.35, traj
    assert o.eval()["n_nonfinite"] == 0
