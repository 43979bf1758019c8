#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/ (run in the BUILD container, where
/root/reference exists):   python tests/golden/make_golden.py

Sources of truth:
  * variant "ref": $TMPDIR/gbp_oracle_ref/liboracle_ref.so (out of tree) — the vertex/schedule restatement linked against the
    REFERENCE's own matlib.cpp / bafuncs.cpp (compiled where they lie).  Everything tagged `ref_` below
    comes from it (literal std::sin/std::cos, ascending slot-order sums).
  * `dev_` entries: the same restatement in the device's arithmetic conventions (correctly rounded trig,
    row-of-16 tree order for camera sums) — what the HIP kernels must reproduce bit for bit.
The fixtures are data only (inputs + expected outputs); no reference source text is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as orc            # noqa: E402
from gbp_poplar_amd import _cabi as cabi    # noqa: E402
from gbp_poplar_amd import driver           # noqa: E402
from tests.oracle_host import OracleHost    # noqa: E402
from tests.traj_util import EvalAt, wanted   # noqa: E402

P = lambda a: cabi.ptr(a, cabi.c_f32p)


def spd(rng, n, scale):
    a = rng.standard_normal((n, n))
    m = a @ a.T + n * np.eye(n)
    d = np.exp(rng.uniform(-1, 1, n) * scale)
    return (m * d[:, None] * d[None, :]).astype(np.float32)


def math_vectors(lib):
    rng = np.random.default_rng(20200303)
    n = 64
    out = {}
    a3 = np.stack([spd(rng, 3, 2.0) for _ in range(n)])
    a6 = np.stack([spd(rng, 6, 2.0) for _ in range(n)])
    i3, i6 = np.zeros_like(a3), np.zeros_like(a6)
    for k in range(n):
        lib.om_inv3x3(P(a3[k]), P(i3[k]))
        lib.om_inv6x6(P(a6[k]), P(i6[k]))
    out.update(inv3_in=a3, inv3_out=i3, inv6_in=a6, inv6_out=i6)
    w = (rng.standard_normal((n, 3)) * 0.8).astype(np.float32)
    w[0] = 0
    w[1] = 1e-8
    w[2] = [3.0, 0.5, -0.2]
    R = np.zeros((n, 9), np.float32)
    for k in range(n):
        lib.om_so3exp(P(w[k]), P(R[k]))
    out.update(so3_in=w, so3_out=R)
    cam = np.concatenate([rng.standard_normal((n, 3)) * 0.5, rng.standard_normal((n, 3)) * 0.4], axis=1).astype(np.float32)
    cam[:, 2] += 4.0
    lmk = (rng.standard_normal((n, 3)) * 0.7).astype(np.float32)
    K = np.array([520.9, 0, 325.1, 0, 521.0, 249.7, 0, 0, 1], np.float32)
    hx, jk, jl = np.zeros((n, 2), np.float32), np.zeros((n, 12), np.float32), np.zeros((n, 6), np.float32)
    for k in range(n):
        lib.om_hfunc(P(cam[k]), P(lmk[k]), P(K), P(hx[k]))
        lib.om_jac(P(cam[k]), P(lmk[k]), P(K), P(jk[k]), P(jl[k]))
    out.update(proj_cam=cam, proj_lmk=lmk, proj_K=K, hfunc_out=hx, jac_kf=jk, jac_lmk=jl)
    A, B = rng.standard_normal((6, 3)).astype(np.float32), rng.standard_normal((6, 6)).astype(np.float32)
    for name, (x, y, ta, tb, pr, pc) in {"nn": (B, A, 0, 0, 6, 3), "tn": (A, B, 1, 0, 3, 6), "nt": (A, A, 0, 1, 6, 6)}.items():
        p = np.full((pr, pc), 0.25, np.float32)   # non-zero start: matMul accumulates
        lib.om_matmul(P(x), x.shape[0], x.shape[1], P(y), y.shape[0], y.shape[1], P(p), pc, ta, tb)
        out["mm_" + name] = p
    out.update(mm_A=A, mm_B=B)
    # inf2mean6x6 / inf2mean3x3 (bafuncs.cpp:2-15) — drawn AFTER everything above so the older vectors keep their values
    e6, e3 = rng.standard_normal((n, 6)).astype(np.float32), rng.standard_normal((n, 3)).astype(np.float32)
    l6 = np.stack([spd(rng, 6, 1.5) for _ in range(n)])
    l3 = np.stack([spd(rng, 3, 1.5) for _ in range(n)])
    m6, m3 = np.zeros((n, 6), np.float32), np.zeros((n, 3), np.float32)
    for k in range(n):
        lib.om_inf2mean6x6(P(e6[k]), P(l6[k]), P(m6[k]))
        lib.om_inf2mean3x3(P(e3[k]), P(l3[k]), P(m3[k]))
    out.update(mean6_eta=e6, mean6_lambda=l6, mean6_out=m6, mean3_eta=e3, mean3_lambda=l3, mean3_out=m3)
    return out


def snapshot(o):
    r = o.read()
    m = o.messages()
    fe, fl = o.factor_potentials()
    d = {k: r[k] for k in r}
    d.update({"msg_" + k: v for k, v in m.items()})
    d.update(fac_eta=fe, fac_lambda=fl)
    return d


def run_states(bal, variant, trig, sum_order, n_sweeps, prefix, out):
    host = OracleHost("restatement")
    opts = driver.Options()
    K, state, _ = driver.build_inputs(bal, opts, host)
    orc.set_trig_mode(trig, variant)
    o = orc.Oracle(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K, variant=variant)
    o.set_sum_order(sum_order)
    o.upload(state)
    o.linearise()
    for k, v in snapshot(o).items():
        out["%s_lin_%s" % (prefix, k)] = v
    for it in range(n_sweeps):
        if (it + 1) % 2 == 0 and it < 10:
            o.weaken_priors()
        o.iterate(1)
        for k, v in snapshot(o).items():
            out["%s_it%d_%s" % (prefix, it, k)] = v
    orc.set_trig_mode(0, variant)
    return K, state


VERTEX_SWEEPS = (0, 1, 17, 18, 100)
VERTEX_WIDTH = {"fac_eta": 9, "fac_lambda": 81, "msg_cam_eta": 6, "msg_cam_lambda": 36, "msg_lmk_eta": 3,
                "msg_lmk_lambda": 9, "mu": 9, "dmu": 1, "damping": 1, "damping_count": 1, "robust_flag": 1}


def vertex_sample(o, ids):
    """Per-factor outputs of the five vertices (relinearise/prep + four message vertices) for factors `ids`."""
    r = o.read()
    m = o.messages()
    fe, fl = o.factor_potentials()
    mu, dmu = o.mu()
    full = {"fac_eta": fe, "fac_lambda": fl, "mu": mu, "dmu": dmu, "damping": r["damping"],
            "damping_count": r["damping_count"], "robust_flag": r["robust_flag"]}
    full.update({"msg_" + k: v for k, v in m.items()})
    return {k: full[k].reshape(-1, w)[ids].copy() for k, w in VERTEX_WIDTH.items()}


def vertex_vectors(name, n_sample=64):
    """Vertex known answers on a real sequence at the sweeps where relinearisation first fires (17, 18 with
    the default --undamped_start 15) plus 0, 1 and 100."""
    host = OracleHost("ref")
    bal = host.bal_read(os.path.join(ROOT, "data", "sequences", name + ".txt"))
    K, state, _ = driver.build_inputs(bal, driver.Options(), host)
    ids = np.sort(np.random.default_rng(7).choice(bal["n_edges"], n_sample, replace=False)).astype(np.int64)
    out = {"ids": ids, "sweeps": np.array(VERTEX_SWEEPS)}
    for prefix, variant, trig, order in (("ref", "ref", 0, 0), ("dev", "restatement", 1, 1)):
        orc.set_trig_mode(trig, variant)
        o = orc.Oracle(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K, variant=variant)
        o.set_sum_order(order)
        o.upload(state)
        o.linearise()
        n_relin = []
        for it in range(max(VERTEX_SWEEPS) + 1):
            if (it + 1) % 2 == 0 and it < 10:
                o.weaken_priors()
            o.iterate(1)
            n_relin.append(o.eval()["n_relin"])
            if it in VERTEX_SWEEPS:
                for k, v in vertex_sample(o, ids).items():
                    out["%s_it%d_%s" % (prefix, it, k)] = v
        out[prefix + "_n_relin"] = np.array(n_relin)
        orc.set_trig_mode(0, variant)
    return out


def trajectories():
    """printed metric trajectory (iteration, mean reproj, cost, RMSE, n_relins, n_robust, n_active) of
    `./ba` on fr1xyz / fr2robot2 (1500 sweeps) and `./slam` on fr2robot2 (700 sweeps per keyframe), from the
    reference-math oracle in the literal conventions (libm trig, slot-order sums)."""
    host = OracleHost("ref")
    opts = driver.Options()
    out = {}
    for name in ("fr2robot2", "fr1xyz"):
        bal = host.bal_read(os.path.join(ROOT, "data", "sequences", name + ".txt"))
        K, state, _ = driver.build_inputs(bal, opts, host)
        o = orc.Oracle(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K, variant="ref")
        w = EvalAt(o, wanted(opts.n_iters))
        t = driver.run_ba(w, state, opts, eval_every=0)
        out["ba_%s" % name] = np.array(w.rows, dtype=np.float64)
        out["ba_%s_initial" % name] = np.array(t[0][1:3], dtype=np.float64)
        if name == "fr2robot2":
            K, state, extra = driver.build_inputs(bal, opts, host, slam=True)
            o = orc.Oracle(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K, variant="ref")
            n_total = (bal["n_cams"] - 1) * opts.iters_between_kfs - 1
            w = EvalAt(o, wanted(n_total, head=30, every=700, tail=50))
            t = driver.run_slam(w, host, bal, state, extra, opts, eval_every=0)
            out["slam_%s" % name] = np.array(w.rows, dtype=np.float64)
            out["slam_%s_initial" % name] = np.array(t[0][1:3], dtype=np.float64)
    return out


def main():
    if not orc.have("ref"):
        raise SystemExit("the reference-math build is missing: run `make -C oracle ref` in the build container first")
    ref = orc.load("ref")
    assert ref.om_impl_name() == b"reference"
    if sys.argv[1:] == ["vertex"]:          # only (re)write the vertex-level fixture
        np.savez_compressed(os.path.join(HERE, "vertex_vectors.npz"), **vertex_vectors("fr2robot2"))
        return
    if sys.argv[1:] == ["math"]:
        np.savez_compressed(os.path.join(HERE, "math_vectors.npz"), **math_vectors(ref))
        return
    if sys.argv[1:] == ["traj"]:
        np.savez_compressed(os.path.join(HERE, "trajectories.npz"), **trajectories())
        return
    np.savez_compressed(os.path.join(HERE, "math_vectors.npz"), **math_vectors(ref))

    # tiny synthetic graph: one-sweep state pairs (4 cams x 24 lmks)
    from gbp_poplar_amd import hostlib
    bal = hostlib.synth_generate(4, 24, 3, 42)
    out = {"bal_" + k: np.asarray(v) for k, v in bal.items()}
    K, state = run_states(bal, "ref", 0, 0, 4, "ref", out)
    run_states(bal, "restatement", 1, 1, 4, "dev", out)
    out["K"] = K
    out.update({"state_" + k: v for k, v in state.items()})
    np.savez_compressed(os.path.join(HERE, "state_tiny.npz"), **out)

    # fr2robot2 belief snapshots + metric trajectories of both real sequences
    host = OracleHost("ref")
    opts = driver.Options()
    snaps = {}
    for name, n_it in (("fr2robot2", 30), ("fr1xyz", 30)):
        bal = host.bal_read(os.path.join(ROOT, "data", "sequences", name + ".txt"))
        K, state, _ = driver.build_inputs(bal, opts, host)
        o = orc.Oracle(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K, variant="ref")
        o.upload(state)
        o.linearise()
        traj = []
        ev = o.eval()
        traj.append((-1,) + driver.metric(ev)[:2] + (ev["n_relin"], ev["n_robust"]))
        if name == "fr2robot2":
            for k in ("cam_beliefs_eta", "cam_beliefs_lambda", "lmk_beliefs_eta", "lmk_beliefs_lambda"):
                snaps["lin_" + k] = o.read()[k]
        for it in range(n_it):
            if (it + 1) % 2 == 0 and it < 10:
                o.weaken_priors()
            o.iterate(1)
            ev = o.eval()
            traj.append((it,) + driver.metric(ev)[:2] + (ev["n_relin"], ev["n_robust"]))
            if name == "fr2robot2" and it in (0, 1, 2, 5):
                for k in ("cam_beliefs_eta", "cam_beliefs_lambda", "lmk_beliefs_eta", "lmk_beliefs_lambda"):
                    snaps["it%d_%s" % (it, k)] = o.read()[k]
        snaps["traj_" + name] = np.array(traj, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "sequence_snapshots.npz"), **snaps)
    np.savez_compressed(os.path.join(HERE, "vertex_vectors.npz"), **vertex_vectors("fr2robot2"))
    np.savez_compressed(os.path.join(HERE, "trajectories.npz"), **trajectories())
    print("golden fixtures written:", [f for f in os.listdir(HERE) if f.endswith(".npz")])


if __name__ == "__main__":
    main()
