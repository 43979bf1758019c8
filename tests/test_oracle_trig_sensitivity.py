"""Why 'relinearisation <= 1e-5'  is the right bar between the device (sin/cos correctly rounded) and a
literal libm build of the reference (std::sin(float) -> glibc sinf here, Poplar's libm on the IPU).

Measured with the oracle's two trig modes on identical inputs: glibc 2.35 sinf / cosf return a value that is not the
correctly rounded one for ~0.8 % / ~1.8 % of arguments, a camera is hit with probability ~3 %, and a 1-ulp different
sin or cos moves the relinearised potentials of that camera's factors by <= 5e-6 of the factor's largest entry —
Jac's division by |w|^2 (bafuncs.cpp:197-204) does not amplify it further on the shipped sequences (|w| >= 0.28)."""
import numpy as np
import pytest

from gbp_poplar_amd import driver
from tests.conftest import seq_path


@pytest.mark.parametrize("name", ["fr2robot2", "fr1xyz", "fr1desk"])
def test_linearise_in_both_trig_modes(name, oracle_mod, oracle_host):
    bal = oracle_host.bal_read(seq_path(name))
    K, state, _ = driver.build_inputs(bal, driver.Options(), oracle_host)
    pots = []
    try:
        for trig in (0, 1):
            oracle_mod.set_trig_mode(trig)
            o = oracle_mod.Oracle(bal["cam_id"], bal["lmk_id"], bal["n_cams"], bal["n_lmks"], K)
            o.upload(state)
            o.linearise()
            fe, fl = o.factor_potentials()
            pots.append((fe.reshape(-1, 9).astype(np.float64), fl.reshape(-1, 81).astype(np.float64)))
    finally:
        oracle_mod.set_trig_mode(0)
    (e0, l0), (e1, l1) = pots
    err_l = np.abs(l0 - l1).max(axis=1) / np.abs(l1).max(axis=1)
    err_e = np.abs(e0 - e1).max(axis=1) / np.abs(e1).max(axis=1)
    assert err_l.max() <= 1e-5 and err_e.max() <= 1e-5, (err_l.max(), err_e.max())
    cams_hit = len(np.unique(np.asarray(bal["cam_id"])[err_l > 0]))
    assert cams_hit <= max(2, bal["n_cams"] // 8), cams_hit          # most cameras: bit-identical in both modes


def test_glibc_sinf_is_not_always_correctly_rounded():
    """The literal reference has no single answer: its std::sin(float) is whatever libm the target ships."""
    x = np.random.default_rng(1).uniform(0.25, 2.25, 200000).astype(np.float32)
    cr = np.sin(x.astype(np.float64)).astype(np.float32)
    libm = np.sin(x)                          # numpy float32 sin: its own SIMD kernel, yet another libm
    frac = float(np.mean(cr != libm))
    assert 0.0 <= frac < 0.2
