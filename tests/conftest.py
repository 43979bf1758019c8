import os
import sys

import numpy as np
import pytest

def _effective_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 16))


# the oracle's OpenMP loops: never oversubscribe a CPU-quota'd container (the GPU box shows 256 CPUs, quota 16)
os.environ.setdefault("OMP_NUM_THREADS", str(_effective_cores()))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEQ = os.path.join(ROOT, "data", "sequences")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as orc
    orc.load("restatement")
    return orc


@pytest.fixture(scope="session")
def oracle_host():
    from tests.oracle_host import OracleHost
    return OracleHost("restatement")


def seq_path(name):
    return os.path.join(SEQ, name + ".txt")


def small_synth(n_cams=12, n_lmks=300, obs=6, seed=7):
    """Small synthetic BAL graph through the product's generator (host code, no GPU needed)."""
    from gbp_poplar_amd import hostlib
    return hostlib.synth_generate(n_cams, n_lmks, obs, seed)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def per_var_rel(a, b, width):
    """max over variables of ||a_v - b_v||_inf / ||b_v||_inf (the per-variable inf-norm of SURVEY 8c)."""
    a = np.asarray(a, np.float64).reshape(-1, width)
    b = np.asarray(b, np.float64).reshape(-1, width)
    den = np.maximum(np.max(np.abs(b), axis=1), 1e-30)
    return float(np.max(np.max(np.abs(a - b), axis=1) / den))
