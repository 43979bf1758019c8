"""ctypes binding of the CPU oracle — TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, as the checker /
reported baseline.  The product package (gbp_poplar_amd) never imports this module.

Two builds of the same vertex/schedule restatement (oracle_gbp.c):
  variant "restatement": oracle/_build/liboracle.so  — dense math from oracle_math.c (travels)
  variant "ref"        : $TMPDIR/gbp_oracle_ref/liboracle_ref.so — dense math is the reference's own
                         matlib.cpp / bafuncs.cpp compiled in the build container, OUTSIDE the repository:
                         nothing made from the reference's sources sits in the work tree or travels to the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from gbp_poplar_amd import _cabi as cabi  # noqa: E402  (struct declarations only)

_LIBS = {}


def ref_dir():
    """Where `make -C oracle ref` puts the reference-math builds (out of tree; GBP_ORACLE_REF_DIR overrides)."""
    return os.environ.get("GBP_ORACLE_REF_DIR") or os.path.join(os.environ.get("TMPDIR") or "/tmp", "gbp_oracle_ref")


def lib_path(variant="restatement"):
    if variant == "restatement":
        return os.path.join(_HERE, "_build", "liboracle.so")
    if variant == "ref":
        return os.path.join(ref_dir(), "liboracle_ref.so")
    if variant == "ref_math":
        return os.path.join(ref_dir(), "libref_math.so")
    raise ValueError(variant)


def build(ref=None):
    """Compile the oracle (and, when /root/reference is present, the reference-math variant)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])
    if ref is None:
        ref = os.path.isdir("/root/reference/ba")
    if ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref", "REFOUT=" + ref_dir()])


def have(variant):
    return os.path.exists(lib_path(variant))


def load(variant="restatement"):
    if variant in _LIBS:
        return _LIBS[variant]
    path = lib_path(variant)
    if not os.path.exists(path) and variant == "restatement":
        build(ref=False)
    lib = C.CDLL(path)
    if variant != "ref_math":
        lib.orc_create.restype = C.c_void_p
        lib.orc_create.argtypes = [C.POINTER(cabi.GbpProblem), C.POINTER(cabi.GbpParams)]
        lib.orc_destroy.argtypes = [C.c_void_p]
        lib.orc_math_impl.restype = C.c_char_p
        for name in ("orc_linearise", "orc_weaken_priors"):
            getattr(lib, name).argtypes = [C.c_void_p]
        lib.orc_iterate.argtypes = [C.c_void_p, C.c_int]
        lib.orc_upload.argtypes = [C.c_void_p, C.POINTER(cabi.GbpStateIn)]
        lib.orc_read.argtypes = [C.c_void_p, C.POINTER(cabi.GbpStateOut)]
        lib.orc_read_priors.argtypes = [C.c_void_p, C.POINTER(cabi.GbpPriorsOut)]
        lib.orc_new_keyframe.argtypes = [C.c_void_p, C.POINTER(cabi.GbpKfUpdate)]
        lib.orc_eval.argtypes = [C.c_void_p, C.POINTER(cabi.GbpEvalOut)]
        lib.orc_set_sum_order.argtypes = [C.c_void_p, C.c_int, C.c_int, cabi.c_u32p]
        lib.orc_get_factor_potentials.argtypes = [C.c_void_p, cabi.c_f32p, cabi.c_f32p]
        lib.orc_get_messages.argtypes = [C.c_void_p] + [cabi.c_f32p] * 4
        lib.orc_get_mu.argtypes = [C.c_void_p, cabi.c_f32p, cabi.c_f32p]
        lib.orc_set_threads.argtypes = [C.c_int]
        lib.orc_bal_read_header.argtypes = [C.c_char_p, C.POINTER(cabi.GbpBal)]
        lib.orc_bal_read.argtypes = [C.c_char_p, C.POINTER(cabi.GbpBal)]
        lib.orc_set_prior_lambda.argtypes = [C.POINTER(cabi.GbpProblem), C.c_float] + [cabi.c_f32p] * 8
        lib.orc_prior_scalings.argtypes = [C.c_uint32, C.c_uint32, cabi.c_f32p, C.c_float, C.c_float,
                                           C.c_float, cabi.c_f32p, cabi.c_f32p]
        lib.orc_slam_create_flags.argtypes = [C.POINTER(cabi.GbpProblem), C.c_uint32] + [cabi.c_u32p] * 4
        lib.orc_slam_update_flags.argtypes = ([C.POINTER(cabi.GbpProblem), C.c_uint32, C.c_uint32]
                                              + [cabi.c_u32p] * 4 + [cabi.c_i32p])
        lib.orc_slam_initialise_new_kf.argtypes = [C.c_uint32] + [cabi.c_f32p] * 4
        lib.orc_eval_host.argtypes = ([C.POINTER(cabi.GbpProblem), cabi.c_u32p] + [cabi.c_f32p] * 5
                                      + [cabi.c_f64p, cabi.c_f64p, C.POINTER(C.c_uint64)])
        lib.orc_eval_host_f32.argtypes = ([C.POINTER(cabi.GbpProblem), cabi.c_u32p] + [cabi.c_f32p] * 6)
    for name, args in (("om_matmul", [cabi.c_f32p, C.c_int, C.c_int, cabi.c_f32p, C.c_int, C.c_int,
                                      cabi.c_f32p, C.c_int, C.c_int, C.c_int]),
                       ("om_inv3x3", [cabi.c_f32p] * 2), ("om_inv6x6", [cabi.c_f32p] * 2),
                       ("om_so3exp", [cabi.c_f32p] * 2),
                       ("om_inf2mean6x6", [cabi.c_f32p] * 3), ("om_inf2mean3x3", [cabi.c_f32p] * 3), ("om_hfunc", [cabi.c_f32p] * 4),
                       ("om_jac", [cabi.c_f32p] * 5)):
        getattr(lib, name).argtypes = args
        getattr(lib, name).restype = None
    lib.om_impl_name.restype = C.c_char_p
    lib.om_set_trig_mode.argtypes = [C.c_int]
    _LIBS[variant] = lib
    return lib


def set_threads(n, variant="restatement"):
    load(variant).orc_set_threads(int(n))


def set_trig_mode(mode, variant="restatement"):
    """0 = host libm sinf/cosf (literal), 1 = correctly rounded (what the HIP kernels compute)."""
    load(variant).om_set_trig_mode(int(mode))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Oracle:
    """Same verbs as gbp_poplar_amd.engine.GbpEngine (the Poplar program list)."""

    def __init__(self, cam_id, lmk_id, n_cams, n_lmks, K9, params=None, variant="restatement"):
        self.lib = load(variant)
        self._keep = []
        self.problem = cabi.make_problem(cam_id, lmk_id, n_cams, n_lmks, K9, self._keep)
        self.C, self.L, self.E = int(n_cams), int(n_lmks), self.problem.n_edges
        self.params = params if params is not None else cabi.GbpParams.defaults()
        self.h = self.lib.orc_create(C.byref(self.problem), C.byref(self.params))
        if not self.h:
            raise RuntimeError("orc_create failed")

    def close(self):
        if self.h:
            self.lib.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("oracle %s failed: %d" % (what, rc))

    def set_sum_order(self, mode, bounds=None):
        if bounds is None:
            self._chk(self.lib.orc_set_sum_order(self.h, mode, 1, cabi.c_u32p()), "set_sum_order")
        else:
            b = np.ascontiguousarray(bounds, dtype=np.uint32)
            self._chk(self.lib.orc_set_sum_order(self.h, mode, len(b) - 1, cabi.ptr(b, cabi.c_u32p)), "set_sum_order")

    def upload(self, state):
        keep = []
        s = cabi.fill_struct(cabi.GbpStateIn(), state, keep)
        self._chk(self.lib.orc_upload(self.h, C.byref(s)), "upload")

    def linearise(self):
        self._chk(self.lib.orc_linearise(self.h), "linearise")

    def iterate(self, n=1):
        self._chk(self.lib.orc_iterate(self.h, int(n)), "iterate")

    def weaken_priors(self):
        self._chk(self.lib.orc_weaken_priors(self.h), "weaken_priors")

    def read(self):
        out = {"cam_beliefs_eta": np.zeros(6 * self.C, np.float32),
               "cam_beliefs_lambda": np.zeros(36 * self.C, np.float32),
               "lmk_beliefs_eta": np.zeros(3 * self.L, np.float32),
               "lmk_beliefs_lambda": np.zeros(9 * self.L, np.float32),
               "damping": np.zeros(self.E, np.float32),
               "damping_count": np.zeros(self.E, np.int32),
               "robust_flag": np.zeros(self.E, np.uint32)}
        keep = []
        s = cabi.fill_struct(cabi.GbpStateOut(), out, keep)
        self._chk(self.lib.orc_read(self.h, C.byref(s)), "read")
        return out

    def read_priors(self):
        out = {"cam_priors_eta": np.zeros(6 * self.C, np.float32),
               "cam_priors_lambda": np.zeros(36 * self.C, np.float32),
               "lmk_priors_eta": np.zeros(3 * self.L, np.float32),
               "lmk_priors_lambda": np.zeros(9 * self.L, np.float32)}
        keep = []
        s = cabi.fill_struct(cabi.GbpPriorsOut(), out, keep)
        self._chk(self.lib.orc_read_priors(self.h, C.byref(s)), "read_priors")
        return out

    def new_keyframe(self, upd):
        keep = []
        s = cabi.fill_struct(cabi.GbpKfUpdate(), upd, keep)
        self._chk(self.lib.orc_new_keyframe(self.h, C.byref(s)), "new_keyframe")

    def eval(self):
        o = cabi.GbpEvalOut()
        self._chk(self.lib.orc_eval(self.h, C.byref(o)), "eval")
        return {k: getattr(o, k) for k, _ in o._fields_}

    def factor_potentials(self):
        eta = np.zeros(9 * self.E, np.float32)
        lam = np.zeros(81 * self.E, np.float32)
        self.lib.orc_get_factor_potentials(self.h, cabi.ptr(eta, cabi.c_f32p), cabi.ptr(lam, cabi.c_f32p))
        return eta, lam

    def messages(self):
        a = [np.zeros(n * self.E, np.float32) for n in (6, 36, 3, 9)]
        self.lib.orc_get_messages(self.h, *[cabi.ptr(x, cabi.c_f32p) for x in a])
        return dict(zip(("cam_eta", "cam_lambda", "lmk_eta", "lmk_lambda"), a))

    def mu(self):
        mu = np.zeros(9 * self.E, np.float32)
        dmu = np.zeros(self.E, np.float32)
        self.lib.orc_get_mu(self.h, cabi.ptr(mu, cabi.c_f32p), cabi.ptr(dmu, cabi.c_f32p))
        return mu, dmu


# ---- host-side restatements --------------------------------------------------------------------

def bal_read(path, variant="restatement"):
    lib = load(variant)
    h = cabi.GbpBal()
    rc = lib.orc_bal_read_header(path.encode(), C.byref(h))
    if rc != 0:
        raise IOError("oracle: cannot read %s" % path)
    out = {"cam_id": np.zeros(h.n_edges, np.uint32), "lmk_id": np.zeros(h.n_edges, np.uint32),
           "observations": np.zeros(2 * h.n_edges, np.float64),
           "cameras": np.zeros(6 * h.n_cams, np.float64), "points": np.zeros(3 * h.n_lmks, np.float64)}
    keep = []
    cabi.fill_struct(h, out, keep)
    for k, a in zip(("cam_id", "lmk_id", "observations", "cameras", "points"), keep):
        out[k] = a
    rc = lib.orc_bal_read(path.encode(), C.byref(h))
    if rc != 0:
        raise IOError("oracle: malformed %s" % path)
    out.update(n_cams=h.n_cams, n_lmks=h.n_lmks, n_edges=h.n_edges, fx=h.fx, fy=h.fy, cx=h.cx, cy=h.cy)
    return out


def set_prior_lambda(cam_id, lmk_id, n_cams, n_lmks, K9, var, cam_file, lmk_file, cam_mean, lmk_mean,
                     variant="restatement"):
    lib = load(variant)
    keep = []
    p = cabi.make_problem(cam_id, lmk_id, n_cams, n_lmks, K9, keep)
    outs = [np.zeros(6 * n_cams, np.float32), np.zeros(36 * n_cams, np.float32),
            np.zeros(3 * n_lmks, np.float32), np.zeros(9 * n_lmks, np.float32)]
    ins = [_f32(cam_file), _f32(lmk_file), _f32(cam_mean), _f32(lmk_mean)]
    lib.orc_set_prior_lambda(C.byref(p), C.c_float(var), *[cabi.ptr(a, cabi.c_f32p) for a in ins + outs])
    return outs


def prior_scalings(n_cams, n_lmks, cam_priors_lambda, steps, weaker, first_std, variant="restatement"):
    lib = load(variant)
    cs, ls = np.zeros(n_cams, np.float32), np.zeros(n_lmks, np.float32)
    cpl = _f32(cam_priors_lambda)
    lib.orc_prior_scalings(n_cams, n_lmks, cabi.ptr(cpl, cabi.c_f32p), steps, weaker, first_std,
                           cabi.ptr(cs, cabi.c_f32p), cabi.ptr(ls, cabi.c_f32p))
    return cs, ls


def eval_host(cam_id, lmk_id, n_cams, n_lmks, K9, active, meas, cbe, cbl, lbe, lbl, variant="restatement"):
    lib = load(variant)
    keep = []
    p = cabi.make_problem(cam_id, lmk_id, n_cams, n_lmks, K9, keep)
    a = np.ascontiguousarray(active, dtype=np.uint32)
    arrs = [_f32(x) for x in (meas, cbe, cbl, lbe, lbl)]
    sn, sh, na = C.c_double(), C.c_double(), C.c_uint64()
    lib.orc_eval_host(C.byref(p), cabi.ptr(a, cabi.c_u32p), *[cabi.ptr(x, cabi.c_f32p) for x in arrs],
                      C.byref(sn), C.byref(sh), C.byref(na))
    return sn.value, sh.value, na.value


def eval_host_f32(cam_id, lmk_id, n_cams, n_lmks, K9, active, meas, cbe, cbl, lbe, lbl, variant="restatement"):
    lib = load(variant)
    keep = []
    p = cabi.make_problem(cam_id, lmk_id, n_cams, n_lmks, K9, keep)
    a = np.ascontiguousarray(active, dtype=np.uint32)
    arrs = [_f32(x) for x in (meas, cbe, cbl, lbe, lbl)]
    out = np.zeros(2, np.float32)
    lib.orc_eval_host_f32(C.byref(p), cabi.ptr(a, cabi.c_u32p), *[cabi.ptr(x, cabi.c_f32p) for x in arrs],
                          cabi.ptr(out, cabi.c_f32p))
    return float(out[0]), float(out[1])


def slam_create_flags(cam_id, lmk_id, n_cams, n_lmks, steps, variant="restatement"):
    lib = load(variant)
    keep = []
    p = cabi.make_problem(cam_id, lmk_id, n_cams, n_lmks, [0] * 9, keep)
    active = np.zeros(p.n_edges, np.uint32)
    cwf, lwf, laf = np.zeros(n_cams, np.uint32), np.zeros(n_lmks, np.uint32), np.zeros(n_lmks, np.uint32)
    lib.orc_slam_create_flags(C.byref(p), steps, *[cabi.ptr(x, cabi.c_u32p) for x in (active, cwf, lwf, laf)])
    return active, cwf, lwf, laf


def slam_update_flags(cam_id, lmk_id, n_cams, n_lmks, steps, data_counter, active, lwf, cwf, laf,
                      variant="restatement"):
    lib = load(variant)
    keep = []
    p = cabi.make_problem(cam_id, lmk_id, n_cams, n_lmks, [0] * 9, keep)
    n_new = np.zeros(1, np.int32)
    lib.orc_slam_update_flags(C.byref(p), steps, data_counter,
                              *[cabi.ptr(x, cabi.c_u32p) for x in (active, lwf, cwf, laf)],
                              cabi.ptr(n_new, cabi.c_i32p))
    return int(n_new[0])


def slam_initialise_new_kf(data_counter, cbe, cbl, cpl, cpe, variant="restatement"):
    lib = load(variant)
    lib.orc_slam_initialise_new_kf(data_counter, *[cabi.ptr(x, cabi.c_f32p) for x in (cbe, cbl, cpl, cpe)])
