"""Python face of the CPU-side helper exports of the C-ABI (csrc/gbp_host.cpp): BAL loader, prior
strength, weakening scales, SLAM flag bookkeeping, host metric, synthetic generator.  No device needed."""
import ctypes as C

import numpy as np

from . import _cabi as cabi
from ._lib import load


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with status %d" % (what, rc))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _alloc_bal(h):
    out = {"cam_id": np.zeros(h.n_edges, np.uint32), "lmk_id": np.zeros(h.n_edges, np.uint32),
           "observations": np.zeros(2 * h.n_edges, np.float64),
           "cameras": np.zeros(6 * h.n_cams, np.float64), "points": np.zeros(3 * h.n_lmks, np.float64)}
    h.cam_id, h.lmk_id = cabi.ptr(out["cam_id"], cabi.c_u32p), cabi.ptr(out["lmk_id"], cabi.c_u32p)
    h.observations = cabi.ptr(out["observations"], cabi.c_f64p)
    h.cameras, h.points = cabi.ptr(out["cameras"], cabi.c_f64p), cabi.ptr(out["points"], cabi.c_f64p)
    return out


def _finish_bal(h, out):
    out.update(n_cams=int(h.n_cams), n_lmks=int(h.n_lmks), n_edges=int(h.n_edges),
               fx=h.fx, fy=h.fy, cx=h.cx, cy=h.cy)
    return out


def bal_read(path):
    """BALProblem::LoadFile (dataio.cpp:17-57); raises IOError like `./ba` returns 1 (ba.cpp:484-487)."""
    lib = load()
    h = cabi.GbpBal()
    if lib.gbp_bal_read_header(path.encode(), C.byref(h)) != 0:
        raise IOError("ERROR: unable to open file %s" % path)
    out = _alloc_bal(h)
    if lib.gbp_bal_read(path.encode(), C.byref(h)) != 0:
        raise IOError("Invalid UW data file: %s" % path)
    return _finish_bal(h, out)


def bal_import_standard(path):
    """Standard 9-parameter "Bundle Adjustment in the Large" file -> the reference's format (shared pin-hole K,
    +z cameras [t, w], edges sorted by camera then landmark); see gbp_bal_import_standard in the header."""
    lib = load()
    h = cabi.GbpBal()
    if lib.gbp_bal_import_standard_header(path.encode(), C.byref(h)) != 0:
        raise IOError("ERROR: unable to open file %s" % path)
    out = _alloc_bal(h)
    if lib.gbp_bal_import_standard(path.encode(), C.byref(h)) != 0:
        raise IOError("Invalid BAL data file: %s" % path)
    return _finish_bal(h, out)


def bal_write(path, bal):
    lib = load()
    h = cabi.GbpBal()
    h.n_cams, h.n_lmks, h.n_edges = bal["n_cams"], bal["n_lmks"], bal["n_edges"]
    h.fx, h.fy, h.cx, h.cy = bal["fx"], bal["fy"], bal["cx"], bal["cy"]
    keep = [np.ascontiguousarray(bal["cam_id"], np.uint32), np.ascontiguousarray(bal["lmk_id"], np.uint32),
            np.ascontiguousarray(bal["observations"], np.float64), np.ascontiguousarray(bal["cameras"], np.float64),
            np.ascontiguousarray(bal["points"], np.float64)]
    h.cam_id, h.lmk_id = cabi.ptr(keep[0], cabi.c_u32p), cabi.ptr(keep[1], cabi.c_u32p)
    h.observations, h.cameras, h.points = (cabi.ptr(keep[2], cabi.c_f64p), cabi.ptr(keep[3], cabi.c_f64p),
                                           cabi.ptr(keep[4], cabi.c_f64p))
    _chk(lib.gbp_bal_write(path.encode(), C.byref(h)), "gbp_bal_write")


def synth_generate(n_cams, n_lmks, obs_per_lmk=10, seed=20200303, ground_truth=False):
    """Synthetic BAL graph (SURVEY 8d spec)."""
    lib = load()
    h = cabi.GbpBal()
    h.n_cams, h.n_lmks, h.n_edges = n_cams, n_lmks, n_lmks * min(obs_per_lmk, n_cams)
    out = _alloc_bal(h)
    gtc = np.zeros(6 * n_cams, np.float64) if ground_truth else None
    gtp = np.zeros(3 * n_lmks, np.float64) if ground_truth else None
    _chk(lib.gbp_synth_generate(n_cams, n_lmks, obs_per_lmk, seed, C.byref(h),
                                cabi.ptr(gtc, cabi.c_f64p), cabi.ptr(gtp, cabi.c_f64p)), "gbp_synth_generate")
    out = _finish_bal(h, out)
    if ground_truth:
        out["gt_cameras"], out["gt_points"] = gtc, gtp
    return out


def set_prior_lambda(cam_id, lmk_id, n_cams, n_lmks, K9, var, cam_file, lmk_file, cam_mean, lmk_mean):
    lib = load()
    keep = []
    p = cabi.make_problem(cam_id, lmk_id, n_cams, n_lmks, K9, keep)
    outs = [np.zeros(6 * n_cams, np.float32), np.zeros(36 * n_cams, np.float32),
            np.zeros(3 * n_lmks, np.float32), np.zeros(9 * n_lmks, np.float32)]
    ins = [_f32(cam_file), _f32(lmk_file), _f32(cam_mean), _f32(lmk_mean)]
    _chk(lib.gbp_set_prior_lambda(C.byref(p), C.c_float(var), *[cabi.ptr(a, cabi.c_f32p) for a in ins + outs]),
         "gbp_set_prior_lambda")
    return outs


def prior_scalings(n_cams, n_lmks, cam_priors_lambda, steps, weaker, first_std):
    lib = load()
    cs, ls = np.zeros(n_cams, np.float32), np.zeros(n_lmks, np.float32)
    cpl = _f32(cam_priors_lambda)
    _chk(lib.gbp_prior_scalings(n_cams, n_lmks, cabi.ptr(cpl, cabi.c_f32p), steps, weaker, first_std,
                                cabi.ptr(cs, cabi.c_f32p), cabi.ptr(ls, cabi.c_f32p)), "gbp_prior_scalings")
    return cs, ls


def init_add_noise(n_cams, n_lmks, cam_mean, lmk_mean, trans_std=0.0, rot_std_deg=0.0, lmk_std=0.0, seed=1):
    """ba.cpp:536-545 -> dataio.cpp:330-415 with an explicit seed; returns noised copies of the prior means."""
    cam, lmk = _f32(cam_mean).copy(), _f32(lmk_mean).copy()
    _chk(load().gbp_init_add_noise(int(n_cams), int(n_lmks), float(trans_std), float(rot_std_deg), float(lmk_std),
                                   int(seed), cabi.ptr(cam, cabi.c_f32p), cabi.ptr(lmk, cabi.c_f32p)), "gbp_init_add_noise")
    return cam, lmk


def init_av_depth(cam_id, lmk_id, n_cams, n_lmks, cam_mean, lmk_mean):
    """ba.cpp:546-548 -> av_depth_init, dataio.cpp:417-453; returns the new landmark means."""
    keep = []
    p = cabi.make_problem(cam_id, lmk_id, n_cams, n_lmks, [0] * 9, keep)
    cam, lmk = _f32(cam_mean), _f32(lmk_mean).copy()
    _chk(load().gbp_init_av_depth(C.byref(p), cabi.ptr(cam, cabi.c_f32p), cabi.ptr(lmk, cabi.c_f32p)), "gbp_init_av_depth")
    return lmk


def belief_means(n_cams, n_lmks, cbe, cbl, lbe, lbl):
    """gbp_belief_means: (cameras [6C], points [3L]) = Lambda^-1 eta of every variable, float64 arrays."""
    cams, pts = np.zeros(6 * n_cams, np.float64), np.zeros(3 * n_lmks, np.float64)
    a = [_f32(x) for x in (cbe, cbl, lbe, lbl)]
    _chk(load().gbp_belief_means(int(n_cams), int(n_lmks), *[cabi.ptr(x, cabi.c_f32p) for x in a],
                                 cabi.ptr(cams, cabi.c_f64p), cabi.ptr(pts, cabi.c_f64p)), "gbp_belief_means")
    return cams, pts


def landmark_partition(cam_id, lmk_id, n_cams, n_lmks, world):
    """gbp_landmark_partition: contiguous landmark ranges balanced by factor count -> bounds[world + 1]."""
    keep = []
    p = cabi.make_problem(cam_id, lmk_id, n_cams, n_lmks, [0] * 9, keep)
    bounds = np.zeros(world + 1, np.uint32)
    _chk(load().gbp_landmark_partition(C.byref(p), int(world), cabi.ptr(bounds, cabi.c_u32p)), "gbp_landmark_partition")
    return bounds


def slam_create_flags(cam_id, lmk_id, n_cams, n_lmks, steps):
    lib = load()
    keep = []
    p = cabi.make_problem(cam_id, lmk_id, n_cams, n_lmks, [0] * 9, keep)
    active = np.zeros(p.n_edges, np.uint32)
    cwf, lwf, laf = np.zeros(n_cams, np.uint32), np.zeros(n_lmks, np.uint32), np.zeros(n_lmks, np.uint32)
    _chk(lib.gbp_slam_create_flags(C.byref(p), steps, *[cabi.ptr(x, cabi.c_u32p) for x in (active, cwf, lwf, laf)]),
         "gbp_slam_create_flags")
    return active, cwf, lwf, laf


def slam_update_flags(cam_id, lmk_id, n_cams, n_lmks, steps, data_counter, active, lwf, cwf, laf):
    lib = load()
    keep = []
    p = cabi.make_problem(cam_id, lmk_id, n_cams, n_lmks, [0] * 9, keep)
    n_new = np.zeros(1, np.int32)
    _chk(lib.gbp_slam_update_flags(C.byref(p), steps, data_counter,
                                   *[cabi.ptr(x, cabi.c_u32p) for x in (active, lwf, cwf, laf)],
                                   cabi.ptr(n_new, cabi.c_i32p)), "gbp_slam_update_flags")
    return int(n_new[0])


def slam_initialise_new_kf(data_counter, cbe, cbl, cpl, cpe):
    lib = load()
    _chk(lib.gbp_slam_initialise_new_kf(data_counter, *[cabi.ptr(x, cabi.c_f32p) for x in (cbe, cbl, cpl, cpe)]),
         "gbp_slam_initialise_new_kf")


def eval_host(cam_id, lmk_id, n_cams, n_lmks, K9, active, meas, cbe, cbl, lbe, lbl):
    lib = load()
    keep = []
    p = cabi.make_problem(cam_id, lmk_id, n_cams, n_lmks, K9, keep)
    a = np.ascontiguousarray(active, dtype=np.uint32)
    arrs = [_f32(x) for x in (meas, cbe, cbl, lbe, lbl)]
    sn, sh, na = C.c_double(), C.c_double(), C.c_uint64()
    _chk(lib.gbp_eval_host(C.byref(p), cabi.ptr(a, cabi.c_u32p), *[cabi.ptr(x, cabi.c_f32p) for x in arrs],
                           C.byref(sn), C.byref(sh), C.byref(na)), "gbp_eval_host")
    return sn.value, sh.value, na.value


# ---- test hooks (include/gbp_mi355x_debug.h; libgbp_mi355x_test.so): the device order without a device ----

def tile_order_local(tile_class, window=96, n_classes=8):
    """perm[wave slot] = tile of the local XCD-aware sweep order (gbp_params.tile_order = 3) for tiles of the given classes."""
    cls = np.ascontiguousarray(tile_class, np.uint8)
    perm = np.zeros(cls.size, np.uint32)
    _chk(load(hooks=True).gbp_debug_tile_order_local(cls.ctypes.data_as(C.POINTER(C.c_uint8)), cls.size, int(window), int(n_classes),
                                                     cabi.ptr(perm, cabi.c_u32p)), "gbp_debug_tile_order_local")
    return perm


LAYOUT_DIMS = ("C", "L", "E", "lmk_begin", "lmk_end", "L_loc", "E_loc", "n_rows", "n_tiles", "Ep", "row_window")
LAYOUT_ARRAYS = ("pos_edge", "pos_cam", "pos_lmk_loc", "pos_lpos", "cam_row_ptr", "row_slot", "row_cam", "lmk_ptr", "lmk_fpos",
                 "lmk_ix", "tile_perm")


def layout_options(**kw):
    """The construction knobs of the device order; defaults = the product's, keywords override (gbp_layout_options)."""
    o = cabi.GbpLayoutOptions()
    load(hooks=True).gbp_debug_layout_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, int(v))
    return o


def layout_build(cam_id, lmk_id, n_cams, n_lmks, tile_order=0, shard=None, options=None):
    """What gbp_create builds before it touches the GPU (csrc/gbp_layout.cpp), as a dict of numpy arrays + dims.  No device needed."""
    lib = load(hooks=True)
    keep = []
    p = cabi.make_problem(cam_id, lmk_id, n_cams, n_lmks, np.zeros(9, np.float32), keep)
    sh = cabi.GbpShard(int(shard[0]), int(shard[1]), int(shard[2]), int(shard[3])) if shard is not None else None
    h = C.c_void_p()
    rc = lib.gbp_debug_layout_build(C.byref(p), int(tile_order), C.byref(sh) if sh is not None else None,
                                    C.byref(options) if options is not None else None, C.byref(h))
    if rc != 0:
        raise RuntimeError("gbp_debug_layout_build: %s (status %d)" % (lib.gbp_last_error(None).decode(), rc))
    try:
        dims = np.zeros(len(LAYOUT_DIMS), np.uint32)
        _chk(lib.gbp_debug_layout_dims(h, cabi.ptr(dims, cabi.c_u32p)), "gbp_debug_layout_dims")
        out = {k: int(v) for k, v in zip(LAYOUT_DIMS, dims)}
        for i, name in enumerate(LAYOUT_ARRAYS):
            data, n = cabi.c_u32p(), C.c_size_t()
            _chk(lib.gbp_debug_layout_array(h, i, C.byref(data), C.byref(n)), "gbp_debug_layout_array")
            out[name] = np.ctypeslib.as_array(data, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint32)
    finally:
        lib.gbp_debug_layout_free(h)
    return out
