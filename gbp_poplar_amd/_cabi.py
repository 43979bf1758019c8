"""ctypes mirror of include/gbp_mi355x.h (struct layouts + argument helpers).

Pure declarations: no library is loaded here, so the CPU oracle binding (oracle/oracle.py, test
infrastructure) can reuse the same structs without touching the product library.
"""
import ctypes as C

import numpy as np

GBP_ABI_VERSION = 6          # include/gbp_mi355x.h

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_u32p = C.POINTER(C.c_uint32)
c_i32p = C.POINTER(C.c_int32)


class GbpProblem(C.Structure):
    _fields_ = [("n_cams", C.c_uint32), ("n_lmks", C.c_uint32), ("n_edges", C.c_uint32),
                ("cam_id", c_u32p), ("lmk_id", c_u32p), ("K", C.c_float * 9)]


class GbpParams(C.Structure):
    _fields_ = [("maxeta_damping", C.c_float), ("num_undamped_iters", C.c_int32),
                ("dmu_threshold", C.c_float), ("min_linear_iters", C.c_int32),
                ("nstds", C.c_float), ("relin_mode", C.c_int32), ("graph_unroll", C.c_int32),
                ("per_factor_mu", C.c_int32), ("tile_order", C.c_int32), ("persistent", C.c_int32), ("reserved", C.c_int32 * 1),
                ("persist_coop", C.c_int32)]

    @classmethod
    def defaults(cls, **kw):
        """Reference globals, gbp_codelets.cpp:11-16."""
        p = cls(0.4, 8, 3e-3, 10, 2.5, 0, 0)
        for k, v in kw.items():
            setattr(p, k, v)
        return p


class GbpLayoutOptions(C.Structure):
    """include/gbp_mi355x_debug.h: knobs of the device-order construction (test-hooks build only)."""
    _fields_ = [("row_placement", C.c_uint32), ("row_window", C.c_uint32), ("row_place_max_deg", C.c_uint32),
                ("row_key_lane", C.c_uint32), ("classes", C.c_uint32), ("tile_window", C.c_uint32),
                ("tile_min_tiles", C.c_uint32), ("tile_identity", C.c_uint32), ("row_sort_in_class", C.c_uint32)]


class GbpShard(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32),
                ("lmk_begin", C.c_uint32), ("lmk_end", C.c_uint32)]


class GbpStateIn(C.Structure):
    _fields_ = [("damping", c_f32p), ("damping_count", c_i32p), ("mu", c_f32p), ("oldmu", c_f32p),
                ("active_flag", c_u32p), ("cam_scaling", c_f32p), ("lmk_scaling", c_f32p),
                ("cam_weaken_flag", c_u32p), ("lmk_weaken_flag", c_u32p),
                ("cam_priors_eta", c_f32p), ("cam_priors_lambda", c_f32p),
                ("lmk_priors_eta", c_f32p), ("lmk_priors_lambda", c_f32p),
                ("measurements", c_f32p), ("meas_variances", c_f32p)]


class GbpStateOut(C.Structure):
    _fields_ = [("cam_beliefs_eta", c_f32p), ("cam_beliefs_lambda", c_f32p),
                ("lmk_beliefs_eta", c_f32p), ("lmk_beliefs_lambda", c_f32p),
                ("damping", c_f32p), ("damping_count", c_i32p), ("robust_flag", c_u32p)]


class GbpPriorsOut(C.Structure):
    _fields_ = [("cam_priors_eta", c_f32p), ("cam_priors_lambda", c_f32p),
                ("lmk_priors_eta", c_f32p), ("lmk_priors_lambda", c_f32p)]


class GbpKfUpdate(C.Structure):
    _fields_ = [("damping_count", c_i32p), ("cam_priors_eta", c_f32p), ("cam_priors_lambda", c_f32p),
                ("lmk_priors_eta", c_f32p), ("lmk_priors_lambda", c_f32p), ("active_flag", c_u32p),
                ("cam_weaken_flag", c_u32p), ("lmk_weaken_flag", c_u32p)]


class GbpEvalOut(C.Structure):
    _fields_ = [("sum_norm", C.c_double), ("sum_half_sq", C.c_double), ("n_active", C.c_uint64),
                ("n_relin", C.c_uint64), ("n_robust", C.c_uint64), ("n_nonfinite", C.c_uint64),
                ("n_nonpd", C.c_uint64)]


class GbpTimingOut(C.Structure):
    _fields_ = [("sweep_ms", C.c_double), ("belief_ms", C.c_double), ("total_ms", C.c_double),
                ("iterations", C.c_uint64), ("algorithmic_bytes_per_iter", C.c_uint64),
                ("device_bytes_allocated", C.c_uint64), ("exchange_ms", C.c_double)]


class GbpBal(C.Structure):
    _fields_ = [("n_cams", C.c_uint32), ("n_lmks", C.c_uint32), ("n_edges", C.c_uint32),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("cam_id", c_u32p), ("lmk_id", c_u32p), ("observations", c_f64p),
                ("cameras", c_f64p), ("points", c_f64p)]


_DT = {c_f32p: np.float32, c_f64p: np.float64, c_u32p: np.uint32, c_i32p: np.int32}


def ptr(arr, ctype):
    """numpy array -> typed pointer (None -> NULL). The array must be C-contiguous of the right dtype."""
    if arr is None:
        return ctype()
    assert arr.flags["C_CONTIGUOUS"] and arr.dtype == _DT[ctype], (arr.dtype, ctype)
    return arr.ctypes.data_as(ctype)


def fill_struct(struct, arrays, keep):
    """Set every pointer field of `struct` from dict `arrays` (missing -> NULL); hold refs in `keep`."""
    for name, ctype in struct._fields_:
        if ctype in _DT:
            a = arrays.get(name)
            if a is not None:
                a = np.ascontiguousarray(a, dtype=_DT[ctype])
                keep.append(a)
            setattr(struct, name, ptr(a, ctype))
    return struct


def make_problem(cam_id, lmk_id, n_cams, n_lmks, K9, keep):
    cam_id = np.ascontiguousarray(cam_id, dtype=np.uint32)
    lmk_id = np.ascontiguousarray(lmk_id, dtype=np.uint32)
    keep += [cam_id, lmk_id]
    p = GbpProblem()
    p.n_cams, p.n_lmks, p.n_edges = int(n_cams), int(n_lmks), int(cam_id.shape[0])
    p.cam_id, p.lmk_id = ptr(cam_id, c_u32p), ptr(lmk_id, c_u32p)
    for i in range(9):
        p.K[i] = float(K9[i])
    return p
