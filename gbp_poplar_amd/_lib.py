"""Loader of the native library (HIP kernels + C-ABI).  There is no CPU fallback: if the HIP
extension is missing or cannot be loaded, importing callers fail loudly."""
import ctypes as C
import os

from . import _cabi as cabi

HERE = os.path.dirname(os.path.abspath(__file__))
# GBP_LIB: measurement scripts under profiles/ point this at the experiments build (build.py --experiments)
LIB_PATH = os.environ.get("GBP_LIB") or os.path.join(HERE, "libgbp_mi355x.so")
# the product sources + the gbp_debug_* hooks of include/gbp_mi355x_debug.h; loaded by tests that look at internal state
TEST_LIB_PATH = os.environ.get("GBP_LIB") or os.path.join(HERE, "libgbp_mi355x_test.so")
EXP_LIB_PATH = os.path.join(HERE, "libgbp_mi355x_exp.so")    # build.py --experiments: + ablations / mapping experiments
_lib = None
_test_lib = None
_exp_lib = None

# every symbol include/gbp_mi355x.h, gbp_mi355x_multi.h and gbp_mi355x_compat.h declare (tests/test_cabi_symbols.py parses the headers and checks this list)
_SIGS = {
    "gbp_abi_version": (C.c_int, []),
    "gbp_default_params": (None, [C.POINTER(cabi.GbpParams)]),
    "gbp_create": (C.c_int, [C.POINTER(cabi.GbpProblem), C.POINTER(cabi.GbpParams), C.POINTER(cabi.GbpShard),
                             C.POINTER(C.c_void_p)]),
    "gbp_destroy": (None, [C.c_void_p]),
    "gbp_last_error": (C.c_char_p, [C.c_void_p]),
    "gbp_upload": (C.c_int, [C.c_void_p, C.POINTER(cabi.GbpStateIn)]),
    "gbp_linearise": (C.c_int, [C.c_void_p]),
    "gbp_iterate": (C.c_int, [C.c_void_p, C.c_int]),
    "gbp_prepare": (C.c_int, [C.c_void_p]),
    "gbp_weaken_priors": (C.c_int, [C.c_void_p]),
    "gbp_read": (C.c_int, [C.c_void_p, C.POINTER(cabi.GbpStateOut)]),
    "gbp_read_priors": (C.c_int, [C.c_void_p, C.POINTER(cabi.GbpPriorsOut)]),
    "gbp_new_keyframe": (C.c_int, [C.c_void_p, C.POINTER(cabi.GbpKfUpdate)]),
    "gbp_eval": (C.c_int, [C.c_void_p, C.POINTER(cabi.GbpEvalOut)]),
    "gbp_eval_begin": (C.c_int, [C.c_void_p]),
    "gbp_eval_end": (C.c_int, [C.c_void_p, C.POINTER(cabi.GbpEvalOut)]),
    "gbp_iterate_eval": (C.c_int, [C.c_void_p, C.c_int]),
    "gbp_iterate_eval_each": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(cabi.GbpEvalOut)]),
    "gbp_ba_loop": (C.c_int, [C.c_void_p, C.c_int, C.c_uint, C.c_uint, C.POINTER(cabi.GbpEvalOut)]),
    "gbp_sync": (C.c_int, [C.c_void_p]),
    "gbp_timing": (C.c_int, [C.c_void_p, C.POINTER(cabi.GbpTimingOut), C.c_int]),
    "gbp_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gbp_set_exchange_buffers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gbp_iterate_begin": (C.c_int, [C.c_void_p]),
    "gbp_iterate_local": (C.c_int, [C.c_void_p]),
    "gbp_iterate_end": (C.c_int, [C.c_void_p]),
    "gbp_refresh_begin": (C.c_int, [C.c_void_p]),
    "gbp_refresh_end": (C.c_int, [C.c_void_p]),
    "gbp_linearise_factors": (C.c_int, [C.c_void_p]),
    "gbp_device_count": (C.c_int, []),
    "gbp_set_device": (C.c_int, [C.c_int]),
    "gbp_landmark_partition": (C.c_int, [C.POINTER(cabi.GbpProblem), C.c_int, cabi.c_u32p]),
    "gbp_comm_region_abort": (None, [C.c_void_p]),
    "gbp_comm_region_selftest": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "gbp_comm_region_bytes": (C.c_size_t, [C.c_uint32, C.c_int]),
    "gbp_comm_region_init": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int]),
    "gbp_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "gbp_comm_unique_id": (C.c_int, [C.c_void_p]),
    "gbp_comm_init_rccl": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gbp_comm_transport": (C.c_char_p, [C.c_void_p]),
    "gbp_graph_state": (C.c_int, [C.c_void_p]),
    "gbp_comm_describe": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "gbp_comm_set_schedule": (C.c_int, [C.c_void_p, C.c_int]),
    "gbp_comm_probe": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "gbp_comm_barrier": (C.c_int, [C.c_void_p]),
    "gbp_eval_global": (C.c_int, [C.c_void_p, C.POINTER(cabi.GbpEvalOut)]),
    "gbp_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "gbp_bal_read_header": (C.c_int, [C.c_char_p, C.POINTER(cabi.GbpBal)]),
    "gbp_bal_read": (C.c_int, [C.c_char_p, C.POINTER(cabi.GbpBal)]),
    "gbp_bal_write": (C.c_int, [C.c_char_p, C.POINTER(cabi.GbpBal)]),
    "gbp_bal_import_standard_header": (C.c_int, [C.c_char_p, C.POINTER(cabi.GbpBal)]),
    "gbp_bal_import_standard": (C.c_int, [C.c_char_p, C.POINTER(cabi.GbpBal)]),
    "gbp_set_prior_lambda": (C.c_int, [C.POINTER(cabi.GbpProblem), C.c_float] + [cabi.c_f32p] * 8),
    "gbp_prior_scalings": (C.c_int, [C.c_uint32, C.c_uint32, cabi.c_f32p, C.c_float, C.c_float, C.c_float,
                                     cabi.c_f32p, cabi.c_f32p]),
    "gbp_init_add_noise": (C.c_int, [C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_uint64, cabi.c_f32p, cabi.c_f32p]),
    "gbp_init_av_depth": (C.c_int, [C.POINTER(cabi.GbpProblem), cabi.c_f32p, cabi.c_f32p]),
    "gbp_slam_create_flags": (C.c_int, [C.POINTER(cabi.GbpProblem), C.c_uint32] + [cabi.c_u32p] * 4),
    "gbp_slam_update_flags": (C.c_int, [C.POINTER(cabi.GbpProblem), C.c_uint32, C.c_uint32] + [cabi.c_u32p] * 4
                              + [cabi.c_i32p]),
    "gbp_slam_initialise_new_kf": (C.c_int, [C.c_uint32] + [cabi.c_f32p] * 4),
    "gbp_eval_host": (C.c_int, [C.POINTER(cabi.GbpProblem), cabi.c_u32p] + [cabi.c_f32p] * 5
                      + [cabi.c_f64p, cabi.c_f64p, C.POINTER(C.c_uint64)]),
    "gbp_belief_means": (C.c_int, [C.c_uint32, C.c_uint32] + [cabi.c_f32p] * 4 + [cabi.c_f64p, cabi.c_f64p]),
    "gbp_synth_generate": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(cabi.GbpBal),
                                     cabi.c_f64p, cabi.c_f64p]),
}

# include/gbp_mi355x_debug.h: exported by libgbp_mi355x_test.so only
_DEBUG_SIGS = {
    "gbp_debug_get": (C.c_int, [C.c_void_p, C.c_int, cabi.c_f32p, cabi.c_f32p]),
    "gbp_debug_time_sweep": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "gbp_debug_set_factor_potentials": (C.c_int, [C.c_void_p, cabi.c_f32p, cabi.c_f32p]),
    "gbp_debug_math": (C.c_int, [C.c_int, cabi.c_f32p, cabi.c_f32p, C.c_int]),
    "gbp_debug_math_timed": (C.c_int, [C.c_int, cabi.c_f32p, cabi.c_f32p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "gbp_debug_layout_default_options": (None, [C.POINTER(cabi.GbpLayoutOptions)]),
    "gbp_debug_layout_options": (C.c_int, [C.POINTER(cabi.GbpLayoutOptions)]),
    "gbp_debug_force_sweep_policy": (C.c_int, [C.c_int]),
    "gbp_debug_force_seg_skip": (C.c_int, [C.c_int]),
    "gbp_debug_persist_flow": (C.c_int, [C.c_void_p, C.c_int]),
    "gbp_debug_persist_verify": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]),
    "gbp_debug_flow_torture": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.POINTER(C.c_uint64)]),
    "gbp_debug_persist_roles": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32]),
    "gbp_debug_layout_build": (C.c_int, [C.POINTER(cabi.GbpProblem), C.c_int, C.POINTER(cabi.GbpShard),
                                         C.POINTER(cabi.GbpLayoutOptions), C.POINTER(C.c_void_p)]),
    "gbp_debug_layout_dims": (C.c_int, [C.c_void_p, cabi.c_u32p]),
    "gbp_debug_layout_array": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(cabi.c_u32p), C.POINTER(C.c_size_t)]),
    "gbp_debug_layout_free": (None, [C.c_void_p]),
    "gbp_debug_tile_order_local": (C.c_int, [C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.c_uint32, cabi.c_u32p]),
}


def symbols():
    return sorted(_SIGS)


def debug_symbols():
    return sorted(_DEBUG_SIGS)


def load(hooks=False):
    """The product library (hooks=False) or the test-hooks build of the same sources (hooks=True)."""
    global _lib, _test_lib, _exp_lib
    if hooks == "exp":               # the experiments build (mapping experiments: tests/test_gpu_experiments.py, profiles/)
        if _exp_lib is not None:
            return _exp_lib
    elif hooks and _test_lib is not None:
        return _test_lib
    elif not hooks and _lib is not None:
        return _lib
    path = EXP_LIB_PATH if hooks == "exp" else (TEST_LIB_PATH if hooks else LIB_PATH)
    if not os.path.exists(path):
        raise RuntimeError(
            "native library %s is missing — build it with `python -m gbp_poplar_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback." % path)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and dlopens it by path, so
    # if this library (linked against /opt/rocm's copy) initialised the GPU first, a later `import torch`
    # would bring up a second runtime that finds "No HIP GPUs".  Loading torch's runtime first makes our
    # DT_NEEDED libamdhip64.so.7 resolve to the copy already in the process.  (The C++ CLIs never load torch.)
    if os.environ.get("GBP_NO_TORCH") != "1":
        try:
            import torch
            torch.cuda.is_available()
        except ImportError:
            pass
    lib = C.CDLL(path)
    lib.gbp_abi_version.restype = C.c_int        # first: a stale .so gets the rebuild message, not an AttributeError for a new symbol
    lib.gbp_abi_version.argtypes = []
    if lib.gbp_abi_version() != cabi.GBP_ABI_VERSION:
        raise RuntimeError("%s has ABI version %d, these bindings were written for %d — rebuild (python -m gbp_poplar_amd.build)"
                           % (path, lib.gbp_abi_version(), cabi.GBP_ABI_VERSION))
    sigs = dict(_SIGS)
    if hooks:
        sigs.update(_DEBUG_SIGS)
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if hooks == "exp":
        _exp_lib = lib
    elif hooks:
        _test_lib = lib
    else:
        _lib = lib
    return lib
