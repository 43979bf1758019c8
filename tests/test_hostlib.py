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


@pytest.mark.parametrize("threads", [1, 3, 7, 32])
def test_big_files_are_read_by_every_core_with_the_serial_result(tmp_path, monkeypatch, threads):
    """Files above 1 MB are cut at whitespace and converted by one thread per piece (gbp_host.cpp: read_number_file) — the same
    strtol / strtod conversions fscanf makes (dataio.cpp:17-57), so the arrays are those of the serial reader, for any thread count
    and any whitespace between the tokens; what follows the last number is ignored as fscanf ignores it."""
    bal = hostlib.synth_generate(60, 9000, 8, 4)
    p = str(tmp_path / "big.txt")
    hostlib.bal_write(p, bal)
    assert os.path.getsize(p) > (1 << 20)
    monkeypatch.setenv("GBP_HOST_THREADS", str(threads))
    back = hostlib.bal_read(p)
    for k in ("cam_id", "lmk_id", "observations", "cameras", "points", "fx", "fy", "cx", "cy"):
        assert np.array_equal(np.asarray(bal[k]), np.asarray(back[k])), k
    # the same tokens with other whitespace (tabs, CRLF, blank lines, leading blanks) and a tail that no conversion is asked for
    text = open(p).read().split("\n")
    mixed = str(tmp_path / "mixed.txt")
    with open(mixed, "w", newline="") as f:
        f.write("  \n\t" + text[0].replace(" ", "\t") + "\r\n\r\n" + text[1] + "\n")
        for i, line in enumerate(text[2:]):
            f.write(line.replace(" ", "  \t" if i % 3 == 0 else " ") + ("\r\n" if i % 5 == 0 else "\n\n" if i % 7 == 0 else "\n"))
        f.write("trailing words 1 2 3\n")
    again = hostlib.bal_read(mixed)
    for k in ("cam_id", "lmk_id", "observations", "cameras", "points", "fx", "fy", "cx", "cy"):
        assert np.array_equal(np.asarray(bal[k]), np.asarray(again[k])), k


@pytest.mark.parametrize("threads", [1, 4, 9])
def test_number_formats_fscanf_accepts_are_read_as_strtod_reads_them(tmp_path, monkeypatch, threads):
    """Every spelling %d / %lf take — explicit '+', leading zeros, no digit before or after the point, exponents with either case and
    sign, 17 significant digits, hexadecimal floats, "inf" — converted by the threaded reader exactly as float() (= strtod) converts the
    token: a 1.5 MB file of mixed spellings, 1 / 4 / 9 threads."""
    rng = np.random.default_rng(17)
    C, L, E = 7, 300, 40000

    def spell(v):
        k = int(rng.integers(0, 9))
        if k == 0:
            return repr(float(v))
        if k == 1:
            return "%+.17g" % v
        if k == 2:
            return ("%.9E" % v)
        if k == 3:
            return "%.6f" % v
        if k == 4:
            return float(v).hex()
        if k == 5:
            return ("%.3f" % (abs(v) % 1.0)).lstrip("0") or ".0"          # ".250"
        if k == 6:
            return "%d." % int(v)                                          # "12."
        if k == 7:
            return "%.12e" % (v * 1e-300)                                  # tiny
        return "%de%+d" % (int(v * 1000), int(rng.integers(-3, 4)))         # "12345e-2"

    vals = rng.normal(0, 300, 2 * E + 6 * C + 3 * L)
    toks = [spell(v) for v in vals]
    toks[5] = "inf"
    cam = rng.integers(0, C, E)
    lmk = rng.integers(0, L, E)
    p = str(tmp_path / "spell.txt")
    with open(p, "w") as f:
        f.write("%d +%d 0%d\n500.0 5e2 .32e3 240.\n" % (C, L, E))
        for i in range(E):
            f.write("%s%d %s %s %s\n" % ("+" if i % 5 == 0 else "00" if i % 7 == 0 else "", cam[i], lmk[i], toks[2 * i], toks[2 * i + 1]))
        for t in toks[2 * E:]:
            f.write(t + "\n")
    assert os.path.getsize(p) > (1 << 20)
    monkeypatch.setenv("GBP_HOST_THREADS", str(threads))
    bal = hostlib.bal_read(p)
    want = np.array([float.fromhex(t) if "0x" in t else float(t) for t in toks])
    assert (bal["n_cams"], bal["n_lmks"], bal["n_edges"]) == (C, L, E) and (bal["fx"], bal["fy"], bal["cx"], bal["cy"]) == (500.0, 500.0, 320.0, 240.0)
    assert np.array_equal(bal["cam_id"], cam) and np.array_equal(bal["lmk_id"], lmk)
    assert np.array_equal(bal["observations"], want[:2 * E]) and np.isinf(bal["observations"][5])
    assert np.array_equal(bal["cameras"], want[2 * E:2 * E + 6 * C]) and np.array_equal(bal["points"], want[2 * E + 6 * C:])


@pytest.mark.parametrize("lmks", [40, 9000])      # below and above the size where the threads start
def test_irregular_tokens_keep_fscanfs_meaning(tmp_path, monkeypatch, lmks):
    """A token that strtod does not take whole is where fscanf's result depends on what follows: such files go through the fscanf
    chain itself.  "1.5.25" is two numbers for fscanf (1.5 and .25); "7x" ends the file for it."""
    monkeypatch.setenv("GBP_HOST_THREADS", "4")
    bal = hostlib.synth_generate(60, lmks, 8, 4)
    p = str(tmp_path / "s.txt")
    hostlib.bal_write(p, bal)
    lines = open(p).read().split("\n")
    e = bal["n_edges"] // 2
    ci, li = lines[2 + e].split()[:2]
    lines[2 + e] = "%s %s 1.5.25" % (ci, li)
    q = str(tmp_path / "glued.txt")
    open(q, "w").write("\n".join(lines))
    back = hostlib.bal_read(q)
    want = np.array(bal["observations"], np.float64).copy()
    want[2 * e:2 * e + 2] = (1.5, 0.25)
    assert np.array_equal(back["observations"], want) and np.array_equal(back["points"], bal["points"])
    lines[2 + e] = "%s %s 7x 1.0" % (ci, li)
    open(q, "w").write("\n".join(lines))
    with pytest.raises(IOError):
        hostlib.bal_read(q)
    # truncated in the middle of the landmarks; an index out of range
    lines[2 + e] = "%s %s 1.0 2.0" % (ci, li)
    open(q, "w").write("\n".join(lines[:len(lines) - 20]))
    with pytest.raises(IOError):
        hostlib.bal_read(q)
    lines[2 + e] = "%d %s 1.0 2.0" % (bal["n_cams"], li)
    open(q, "w").write("\n".join(lines))
    with pytest.raises(IOError):
        hostlib.bal_read(q)
    lines[2 + e] = "%s -1 1.0 2.0" % ci
    open(q, "w").write("\n".join(lines))
    with pytest.raises(IOError):
        hostlib.bal_read(q)


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


def test_prior_strengths_do_not_depend_on_the_host_thread_count(monkeypatch):
    """gbp_set_prior_lambda cuts large graphs into one range of factors per thread; a maximum is the same in any order."""
    bal = hostlib.synth_generate(80, 30000, 8, 11)
    opts = driver.Options()
    out = []
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("GBP_HOST_THREADS", threads)
        _, st, _ = driver.build_inputs(bal, opts, hostlib)
        out.append(st)
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k]) and np.array_equal(out[0][k], out[2][k]), k


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


# ---- standard "Bundle Adjustment in the Large" import ------------------------------------------

def _rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def _write_standard_bal(path, rng, n_cams=7, n_lmks=40, point_noise=0.0):
    """Synthetic file in the published BAL model: P = R X + t, p = -P/P.z, pixel = f (1 + k1 r^2 + k2 r^4) p.
    Landmark-major edge order (as the public files), per-camera f / k1 / k2, cameras looking down -z."""
    pts = rng.uniform(-1, 1, (n_lmks, 3))
    cams = np.zeros((n_cams, 9))
    for c in range(n_cams):
        w = rng.normal(0, 0.15, 3)
        if c == 0:
            w[:] = 0                       # identity: becomes a rotation by exactly pi after the frame flip
        if c == 1:
            w[:] = [np.pi - 1e-3, 0, 0]    # becomes a near-zero rotation after the flip
        R = _rodrigues(w)
        centre = R.T @ np.array([0.0, 0.0, 6.0 + rng.uniform(-1, 1)])   # scene sits at z = -6 in camera coordinates
        cams[c, :3], cams[c, 3:6] = w, -R @ centre
        cams[c, 6:] = [rng.uniform(400, 650), rng.normal(0, 5e-2), rng.normal(0, 5e-3)]
    edges = []
    for l in range(n_lmks):
        for c in sorted(rng.choice(n_cams, 4, replace=False)):
            P = _rodrigues(cams[c, :3]) @ pts[l] + cams[c, 3:6]
            assert P[2] < 0
            p = -P[:2] / P[2]
            r2 = p @ p
            edges.append((c, l) + tuple(cams[c, 6] * (1 + cams[c, 7] * r2 + cams[c, 8] * r2 * r2) * p))
    with open(path, "w") as f:
        f.write("%d %d %d\n" % (n_cams, n_lmks, len(edges)))
        for c, l, x, y in edges:
            f.write("%d %d %.17e %.17e\n" % (c, l, x, y))
        for v in cams.ravel():
            f.write("%.17e\n" % v)
        for v in (pts + rng.normal(0, point_noise, pts.shape) if point_noise else pts).ravel():
            f.write("%.17e\n" % v)
    return cams, pts, edges


def test_standard_bal_import(tmp_path):
    rng = np.random.default_rng(5)
    src = str(tmp_path / "standard.txt")
    cams, pts, edges = _write_standard_bal(src, rng)
    bal = hostlib.bal_import_standard(src)
    assert (bal["n_cams"], bal["n_lmks"], bal["n_edges"]) == (7, 40, len(edges))
    assert bal["fx"] == bal["fy"] == pytest.approx(cams[:, 6].mean(), rel=1e-15) and bal["cx"] == bal["cy"] == 0.0
    assert np.array_equal(bal["points"].reshape(-1, 3), pts)
    # edges: same multiset, now sorted by (camera, landmark) as the reference's SLAM mode / metric assume
    key = bal["cam_id"].astype(np.int64) * 10**6 + bal["lmk_id"]
    assert np.all(np.diff(key) > 0)
    assert sorted((int(c), int(l)) for c, l in zip(bal["cam_id"], bal["lmk_id"])) == sorted((int(c), l) for c, l, _, _ in edges)
    # the converted cameras, used the REFERENCE's way (x_cam = exp(w) y + t, u = fx X/Z + cx; bafuncs.cpp:58-103),
    # reproduce the converted observations: the undistortion and the frame flip are consistent
    cam6, obs = bal["cameras"].reshape(-1, 6), bal["observations"].reshape(-1, 2)
    S = np.diag([1.0, -1.0, -1.0])
    for c in range(7):
        assert np.allclose(_rodrigues(cam6[c, 3:]), S @ _rodrigues(cams[c, :3]), atol=1e-12)
        assert np.allclose(cam6[c, :3], S @ cams[c, 3:6], atol=0)
    assert np.linalg.norm(cam6[0, 3:]) == pytest.approx(np.pi, abs=1e-12)      # identity -> rotation by pi about x
    assert np.linalg.norm(cam6[1, 3:]) == pytest.approx(1e-3, rel=1e-9)
    for e in range(bal["n_edges"]):
        c, l = bal["cam_id"][e], bal["lmk_id"][e]
        X = _rodrigues(cam6[c, 3:]) @ pts[l] + cam6[c, :3]
        assert X[2] > 0
        assert np.allclose(bal["fx"] * X[:2] / X[2], obs[e], rtol=1e-10, atol=1e-9), (e, c, l)
    # round trip through the reference's text format
    out = str(tmp_path / "converted.txt")
    hostlib.bal_write(out, bal)
    back = hostlib.bal_read(out)
    for k in ("cam_id", "lmk_id", "observations", "cameras", "points"):
        assert np.array_equal(back[k], bal[k]), k


def test_standard_bal_import_errors(tmp_path):
    with pytest.raises(IOError):
        hostlib.bal_import_standard(str(tmp_path / "missing.txt"))
    p = tmp_path / "truncated.txt"
    p.write_text("2 3 4\n0 0 1.0 2.0\n0 1 1.0 2.0\n")
    with pytest.raises(IOError):
        hostlib.bal_import_standard(str(p))
    p = tmp_path / "bad_index.txt"
    p.write_text("1 1 1\n0 5 1.0 2.0\n" + "0.0\n" * 12)
    with pytest.raises(IOError):
        hostlib.bal_import_standard(str(p))


# ---- initialisation options (ba.cpp:536-548 -> dataio.cpp:330-453), SURVEY 8f-3 ------------------------------------

def _rodrigues(w):
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) if th < 1e-6 else np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * (W @ W)


def _means(name="fr1xyz"):
    bal = hostlib.bal_read(seq_path(name))
    return bal, bal["cameras"].astype(np.float32), bal["points"].astype(np.float32)


def test_init_noise_is_reproducible_and_spares_the_anchors():
    bal, cam, lmk = _means()
    C, L = bal["n_cams"], bal["n_lmks"]
    a = hostlib.init_add_noise(C, L, cam, lmk, 0.05, 2.0, 0.1, seed=7)
    b = hostlib.init_add_noise(C, L, cam, lmk, 0.05, 2.0, 0.1, seed=7)
    c = hostlib.init_add_noise(C, L, cam, lmk, 0.05, 2.0, 0.1, seed=8)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])            # --seed: same draw
    assert not np.array_equal(a[0], c[0]) and not np.array_equal(a[1], c[1])
    assert np.array_equal(a[0][:12], cam[:12])                                  # cameras 0 and 1 anchor the gauge (k = 2)
    assert np.all(a[0].reshape(-1, 6)[2:, :3] != cam.reshape(-1, 6)[2:, :3])
    z = hostlib.init_add_noise(C, L, cam, lmk, 0.0, 0.0, 0.0, seed=7)
    assert np.array_equal(z[0], cam) and np.array_equal(z[1], lmk)              # zero std: nothing is drawn
    d = (a[1] - lmk).astype(np.float64)
    assert abs(d.std() - 0.1) < 0.01 and abs(d.mean()) < 0.01                   # N(0, ltn^2) on 6 582 coordinates
    t_only = hostlib.init_add_noise(C, L, cam, lmk, 0.05, 0.0, 0.0, seed=7)
    dt = (t_only[0] - cam).reshape(-1, 6)
    assert np.all(dt[:, 3:] == 0) and abs(dt[2:, :3].std() - 0.05) < 0.02       # --tn touches translations only


def test_rotation_noise_keeps_the_camera_centre_and_turns_about_one_axis():
    """add_cam_rot_noise (dataio.cpp:345-400): Tc2w's rotation block is pre-multiplied by a rotation about a coordinate
    axis, its translation (the camera centre) is kept, and the result goes back through so3log (util.cpp:34-46)."""
    bal, cam, lmk = _means("fr2robot2")
    C = bal["n_cams"]
    out, _ = hostlib.init_add_noise(C, bal["n_lmks"], cam, lmk, 0.0, 3.0, 0.0, seed=11)
    angles = []
    for c in range(2, C):
        x0, x1 = cam[6 * c:6 * c + 6].astype(np.float64), out[6 * c:6 * c + 6].astype(np.float64)
        R0, R1 = _rodrigues(x0[3:]), _rodrigues(x1[3:])
        assert np.allclose(-R0.T @ x0[:3], -R1.T @ x1[:3], atol=2e-5)           # camera centre unchanged
        Rn = R1.T @ R0                                                          # Rc2w' Rc2w^T = the injected rotation
        assert np.allclose(Rn @ Rn.T, np.eye(3), atol=1e-5)                     # so3log -> so3exp round trip stays on SO(3)
        ang = np.degrees(np.arccos(np.clip((np.trace(Rn) - 1) / 2, -1, 1)))
        axis = np.array([Rn[2, 1] - Rn[1, 2], Rn[0, 2] - Rn[2, 0], Rn[1, 0] - Rn[0, 1]])
        if ang > 0.05:
            assert np.sort(np.abs(axis / np.linalg.norm(axis)))[1] < 2e-2       # a coordinate axis
        angles.append(ang)
    assert 1.0 < np.sqrt(np.mean(np.square(angles))) < 6.0                      # ~ N(0, 3 deg)


@pytest.mark.parametrize("shuffle", [False, True])
def test_av_depth_init_equals_the_reference_loops(shuffle):
    """A literal numpy restatement of av_depth_init's camera-major double loop (dataio.cpp:417-453) against the O(E)
    implementation, also on an UNSORTED edge list."""
    bal, cam, lmk = _means("fr2robot2")
    C, L, E = bal["n_cams"], bal["n_lmks"], bal["n_edges"]
    cid, lid = np.asarray(bal["cam_id"]).copy(), np.asarray(bal["lmk_id"]).copy()
    if shuffle:
        p = np.random.default_rng(0).permutation(E)
        cid, lid = cid[p], lid[p]
    got = hostlib.init_av_depth(cid, lid, C, L, cam, lmk)
    want = lmk.copy()
    done = np.zeros(L, bool)
    for c in range(C):                                                          # dataio.cpp:424-452
        x = cam[6 * c:6 * c + 6].astype(np.float64)
        R = _rodrigues(x[3:])
        spot = R.T @ (np.array([0.0, 0.0, 1.0]) - x[:3])                        # Tw2c^-1 (0,0,1,1)
        for e in np.nonzero(cid == c)[0]:
            if not done[lid[e]]:
                want[3 * lid[e]:3 * lid[e] + 3] = spot
                done[lid[e]] = True
    assert done.all()
    assert np.allclose(got, want, atol=5e-6)
    # every landmark now projects to the principal point of its first observer, at depth 1
    c0 = np.array([np.min(cid[lid == l]) for l in range(L)])
    for l in (0, L // 2, L - 1):
        x = cam[6 * c0[l]:6 * c0[l] + 6].astype(np.float64)
        assert np.allclose(_rodrigues(x[3:]) @ got[3 * l:3 * l + 3] + x[:3], [0, 0, 1], atol=1e-5)


def test_build_inputs_applies_the_init_options():
    bal = hostlib.bal_read(seq_path("fr2robot2"))
    base = driver.build_inputs(bal, driver.Options(), hostlib)[1]
    a = driver.build_inputs(bal, driver.Options(tn=0.05, ltn=0.05, seed=3), hostlib)[1]
    b = driver.build_inputs(bal, driver.Options(tn=0.05, ltn=0.05, seed=3), hostlib)[1]
    assert np.array_equal(a["cam_priors_eta"], b["cam_priors_eta"]) and np.array_equal(a["lmk_priors_eta"], b["lmk_priors_eta"])
    assert not np.array_equal(a["cam_priors_eta"], base["cam_priors_eta"])
    # prior STRENGTH is evaluated at the file values (dataio.cpp:76-116), so Lambda is untouched by the noise
    assert np.array_equal(a["cam_priors_lambda"], base["cam_priors_lambda"])
    d = driver.build_inputs(bal, driver.Options(avdepth_on=True, ltn=0.5, seed=3), hostlib)[1]    # --avdepth_on wins over --ltn
    e = driver.build_inputs(bal, driver.Options(avdepth_on=True), hostlib)[1]
    assert np.array_equal(d["lmk_priors_eta"], e["lmk_priors_eta"]) and not np.array_equal(e["lmk_priors_eta"], base["lmk_priors_eta"])


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_landmark_partition_native_equals_python(world):
    """gbp_landmark_partition (C++ launcher) and distributed.landmark_partition (Python launcher) must cut the
    landmarks identically: a shard layout is part of the result (summation order of the camera partials)."""
    from gbp_poplar_amd.distributed import landmark_partition
    for bal in (hostlib.bal_read(seq_path("fr2robot2")), hostlib.synth_generate(9, 70, 4, 5)):
        a = hostlib.landmark_partition(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], world)
        b = landmark_partition(bal["lmk_id"], bal["n_lmks"], world)
        assert np.array_equal(a, b) and a[0] == 0 and a[-1] == bal["n_lmks"] and np.all(np.diff(a.astype(np.int64)) >= 0)
    # more ranks than landmarks with factors: trailing shards are empty, never out of range
    a = hostlib.landmark_partition(np.array([0, 1], np.uint32), np.array([0, 0], np.uint32), 2, 3, 4)
    assert a[-1] == 3 and np.all(np.diff(a.astype(np.int64)) >= 0)
    assert np.array_equal(a, landmark_partition(np.array([0, 0]), 3, 4))


def test_comm_region_contract():
    """The rendezvous region of a forked launcher: size grows with cameras x ranks, init validates its arguments."""
    import ctypes
    from gbp_poplar_amd._lib import load
    lib = load()
    n1, n2 = lib.gbp_comm_region_bytes(1000, 2), lib.gbp_comm_region_bytes(1000, 8)
    assert n2 - n1 == 2 * 6 * 1000 * 44 * 4 and n1 > 2 * 2 * 1000 * 44 * 4
    buf = ctypes.create_string_buffer(n1)
    assert lib.gbp_comm_region_init(buf, n1, 1000, 2) == 0
    assert lib.gbp_comm_region_init(buf, n1, 1000, 8) != 0          # too small for 8 ranks
    assert lib.gbp_comm_region_init(buf, n1, 1000, 0) != 0
    lib.gbp_comm_region_abort(buf)                                  # supervisor call: must not crash


def test_belief_means_solve_the_information_form():
    rng = np.random.default_rng(8)
    C, L = 5, 7
    cam_mu, lmk_mu = rng.standard_normal((C, 6)), rng.standard_normal((L, 3))
    a = rng.standard_normal((C, 6, 6)); cl = a @ a.transpose(0, 2, 1) + 6 * np.eye(6)
    a = rng.standard_normal((L, 3, 3)); ll = a @ a.transpose(0, 2, 1) + 3 * np.eye(3)
    ce = np.einsum("cij,cj->ci", cl, cam_mu)
    le = np.einsum("lij,lj->li", ll, lmk_mu)
    cams, pts = hostlib.belief_means(C, L, ce.ravel(), cl.ravel(), le.ravel(), ll.ravel())
    assert cams.dtype == np.float64 and np.allclose(cams.reshape(C, 6), cam_mu, atol=2e-5) and np.allclose(pts.reshape(L, 3), lmk_mu, atol=2e-5)


def test_local_tile_order_is_a_bounded_bijection():
    """gbp_tile_order_local (the XCD-aware sweep order, gbp_params.tile_order = 3): every tile exactly once; no tile runs more
    than `window` slots away from its place in the sequential order; on camera-major classes (a camera's 16 tiles walk through
    the 8 landmark octiles, as on the 1M-factor graph) nearly every wave slot gets a tile of its workgroup's XCD class; on
    adversarial classes (all tiles of one class) it degenerates to the sequential order."""
    from gbp_poplar_amd import hostlib
    rng = np.random.default_rng(3)
    n = 16 * 1000 + 7                                            # not a multiple of anything
    cls = ((np.arange(n) % 16) // 2).astype(np.uint8)            # 0 0 1 1 ... 7 7 per "camera"
    jitter = rng.random(n) < 0.1                                  # 10 % of the tiles straddle into a neighbouring octile
    cls[jitter] = (cls[jitter] + rng.integers(0, 2, jitter.sum()) * 2 - 1) % 8
    for window in (8, 32, 96, 500):
        perm = hostlib.tile_order_local(cls, window)
        assert np.array_equal(np.sort(perm), np.arange(n)), window
        slot = np.arange(n)
        assert np.max(np.abs(perm.astype(np.int64) - slot)) <= window + 32, window      # (+ 32: the oldest tile waits for the next workgroup of its class)
        want = (slot // 4) % 8
        hit = (cls[perm] == want).mean()
        assert hit > (0.85 if window >= 32 else 0.3), (window, hit)
    same = np.zeros(n, np.uint8)
    perm = hostlib.tile_order_local(same, 96)
    assert np.array_equal(np.sort(perm), np.arange(n))
    assert np.max(np.abs(perm.astype(np.int64) - np.arange(n))) <= 96 + 32
    assert np.array_equal(hostlib.tile_order_local(np.zeros(0, np.uint8)), np.zeros(0, np.uint32))
