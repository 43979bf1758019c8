"""Mapping experiments (libgbp_mi355x_exp.so, `python -m gbp_poplar_amd.build --experiments`): the kernels DESIGN.md 2's
mapping decision is measured against must compute the same thing as the product kernel, or their timings mean nothing."""
import os

import numpy as np
import pytest

from tests.conftest import seq_path

pytestmark = pytest.mark.gpu


def _exp_lib_present():
    from gbp_poplar_amd import _lib
    return os.path.exists(_lib.EXP_LIB_PATH)


@pytest.mark.skipif(not _exp_lib_present(), reason="experiments build absent (python -m gbp_poplar_amd.build --experiments)")
@pytest.mark.parametrize("name", ["fr2robot2", "fr1xyz"])
def test_subwave_sweep_is_bit_identical_to_the_product_sweep(name):
    """k_sweep_coop16 — the north star's mapping: 16 lanes per factor, blocks staged in LDS, lane = output element,
    cooperative 6x6 LDL^T — against k_sweep (one lane per factor): 60 iterations of the ./ba flow, relinearisations
    included, every tensor equal bit for bit."""
    from gbp_poplar_amd import _cabi, driver, hostlib
    from gbp_poplar_amd.engine import GbpEngine
    bal = hostlib.bal_read(seq_path(name))
    opts = driver.Options()
    K, state, _ = driver.build_inputs(bal, opts, hostlib)
    engs = []
    for variant in (1, 0):
        prm = _cabi.GbpParams.defaults(persistent=-1, graph_unroll=-1)
        prm.reserved[0] = variant
        engs.append(GbpEngine(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K, params=prm, hooks="exp"))
    n_relin = 0
    for e in engs:
        e.upload(state)
        e.linearise()
    for it in range(60):
        for e in engs:
            if (it + 1) % 2 == 0 and it < 10:
                e.weaken_priors()
            e.iterate(1)
        if it % 6 == 5 or it > 50:
            a, b = engs[0], engs[1]
            ra, rb = a.read(), b.read()
            ma, mb = a.messages(), b.messages()
            fa, fb = a.factor_potentials(), b.factor_potentials()
            for k in ra:
                assert np.array_equal(ra[k], rb[k], equal_nan=True), (it, k)
            for k in ma:
                assert np.array_equal(ma[k], mb[k], equal_nan=True), (it, k)
            assert np.array_equal(fa[0], fb[0]) and np.array_equal(fa[1], fb[1]), it
            n_relin += a.eval()["n_relin"]
    assert n_relin > 0


@pytest.mark.skipif(not _exp_lib_present(), reason="experiments build absent (python -m gbp_poplar_amd.build --experiments)")
def test_persistent_kernel_gives_up_instead_of_hanging():
    """A k_persist launch whose workgroups can NOT all be resident (forced here: every 8th dispatch slot = one XCD = 32 CUs for
    the 52 workgroups of fr1xyz) must end by itself — bounded barrier wait, abort word — and surface as an error at the
    next synchronisation, not hang the GPU."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time\n"
            "sys.path.insert(0, %r)\n"
            "from gbp_poplar_amd import _cabi, driver, hostlib\n"
            "from gbp_poplar_amd.engine import GbpEngine, GbpError\n"
            "bal = hostlib.bal_read(%r)\n"
            "K, state, _ = driver.build_inputs(bal, driver.Options(), hostlib)\n"
            "eng = GbpEngine(bal['cam_id'], bal['lmk_id'], bal['n_cams'], bal['n_lmks'], K, hooks='exp', params=_cabi.GbpParams.defaults(persistent=1))\n"
            "eng.upload(state); eng.linearise(); eng.iterate(1)\n"
            "t0 = time.time()\n"
            "try:\n"
            "    eng.iterate(50); eng.sync(); print('NO ERROR')\n"
            "except GbpError as e:\n"
            "    print('ERROR', round(time.time() - t0, 1), e)\n") % (root, seq_path("fr1xyz"))
    env = dict(os.environ, GBP_PERSIST_SPREAD="8")
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "ERROR" in p.stdout and "barrier timed out" in p.stdout, p.stdout
    assert time.time() - t0 < 60
