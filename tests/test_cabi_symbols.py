"""The C-ABI library loads on a CPU-only box and exports every symbol the headers under include/ declare — gbp_mi355x.h (the
single-GPU boundary), gbp_mi355x_multi.h (landmark shards), gbp_mi355x_compat.h (earlier forms of the loop) — and nothing else;
every one of them is defined through the GBP_EXPORT macros, i.e. runs inside the exception guard (no device compute is called here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


PRODUCT_HEADERS = ("gbp_mi355x.h", "gbp_mi355x_multi.h", "gbp_mi355x_compat.h")
CSRC = os.path.join(ROOT, "gbp_poplar_amd", "csrc")


def declared_functions(header=PRODUCT_HEADERS):
    names = []
    for h in ([header] if isinstance(header, str) else header):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"^\s*(?:GBP_API\s+)?(?:const\s+char\s*\*|int|void|size_t)\s+(gbp_\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_declares_the_program_list():
    """The core header alone is the single-GPU boundary (one export per Poplar program + the loop body, the metric, timing
    and the host helpers); shards and the exchange live in the multi header, the earlier loop forms in the compatibility header."""
    core = declared_functions("gbp_mi355x.h")
    for must in ("gbp_create", "gbp_destroy", "gbp_last_error", "gbp_upload", "gbp_linearise", "gbp_iterate",
                 "gbp_weaken_priors", "gbp_read", "gbp_read_priors", "gbp_new_keyframe", "gbp_eval", "gbp_timing", "gbp_sync",
                 "gbp_ba_loop", "gbp_bal_read", "gbp_set_prior_lambda", "gbp_synth_generate"):
        assert must in core
    assert not [n for n in core if n.startswith("gbp_comm_") or n in ("gbp_iterate_begin", "gbp_iterate_end", "gbp_eval_begin", "gbp_iterate_eval_each")]
    multi = declared_functions("gbp_mi355x_multi.h")
    assert {"gbp_iterate_begin", "gbp_iterate_end", "gbp_comm_init", "gbp_comm_init_rccl", "gbp_eval_global", "gbp_landmark_partition"} <= set(multi)
    assert declared_functions("gbp_mi355x_compat.h") == ["gbp_eval_begin", "gbp_eval_end", "gbp_iterate_eval", "gbp_iterate_eval_each"]
    assert len(core) == 34 and len(multi) == 24 and len(declared_functions()) == 62      # ABI 6: the 62 exports of round 5, regrouped


def test_every_export_is_defined_through_the_guard_macro():
    """include/gbp_mi355x.h: "no C++ exception ever crosses this ABI".  Every declared function is DEFINED with GBP_EXPORT /
    GBP_EXPORT_T / GBP_EXPORT_VOID (csrc/gbp_export.hpp: the exported symbol is a shell that runs the body inside a try block), and
    no translation unit of the library defines a gbp_* function with C linkage in any other way."""
    defined, stray = {}, []
    for dirpath, _, files in os.walk(CSRC):
        for f in files:
            if not f.endswith((".cpp", ".hip")) or f.endswith("_main.cpp"):
                continue
            src = open(os.path.join(dirpath, f)).read()
            src = re.sub(r"//[^\n]*", "", src)
            for m in re.finditer(r"^GBP_EXPORT(?:_T|_VOID)?\(\s*(?:[^,()]+,\s*[^,]+,\s*)?(gbp_\w+)\s*,", src, flags=re.M):
                defined.setdefault(m.group(1), []).append(f)
            if 'extern "C"' in src:
                stray.append(f)
            stray += ["%s: %s" % (f, m.group(1)) for m in
                      re.finditer(r"^(?:int|void|size_t|const char\s*\*)\s+(gbp_\w+)\s*\(", src, flags=re.M)]
    assert not stray, stray
    want = set(declared_functions()) | set(declared_functions("gbp_mi355x_debug.h"))
    assert set(defined) == want, set(defined) ^ want
    assert all(len(v) == 1 for v in defined.values()), {k: v for k, v in defined.items() if len(v) > 1}
    macro = open(os.path.join(CSRC, "gbp_export.hpp")).read()
    assert macro.count("catch (...)") >= 3 and "guarded(ctx, #name" in macro


def test_library_exports_every_declared_symbol():
    from gbp_poplar_amd import _cabi, _lib
    lib = _lib.load()
    names = declared_functions()
    assert len(names) == 62
    for n in names:
        assert hasattr(lib, n), "libgbp_mi355x.so does not export %s" % n
    assert sorted(_lib.symbols()) == names, set(names) ^ set(_lib.symbols())
    assert lib.gbp_abi_version() == _cabi.GBP_ABI_VERSION == 6


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if l.strip()}


def test_product_library_exports_no_test_hooks():
    """The default build carries the program list, the host helpers and gbp_comm_* — no gbp_debug_* entry point and
    none of the test-only kernels."""
    from gbp_poplar_amd import _lib
    sym = _exported(_lib.LIB_PATH)
    assert not [s for s in sym if "gbp_debug" in s], [s for s in sym if "gbp_debug" in s]
    assert not [s for s in sym if "k_debug_math" in s or "k_inv6_coop" in s]
    exported_api = {s for s in sym if s.startswith("gbp_")}
    assert exported_api == set(declared_functions()), exported_api ^ set(declared_functions())


def test_libraries_export_the_c_abi_and_nothing_else():
    """-fvisibility=hidden + csrc/gbp_exports.map: `nm -D` of the product shows the functions include/gbp_mi355x.h declares and
    nothing else — no C++ symbol of the library's own (_ZN3gbp...), no kernel stub, no weak libstdc++ instantiation; the
    test-hooks build adds exactly the functions of include/gbp_mi355x_debug.h."""
    from gbp_poplar_amd import _lib
    assert _exported(_lib.LIB_PATH) == set(declared_functions())
    assert _exported(_lib.TEST_LIB_PATH) == set(declared_functions()) | set(declared_functions("gbp_mi355x_debug.h"))


def test_test_hooks_library_exports_the_debug_header():
    from gbp_poplar_amd import _lib
    lib = _lib.load(hooks=True)
    names = declared_functions("gbp_mi355x_debug.h")
    assert names == sorted(_lib.debug_symbols()) and len(names) == 18
    for n in names + declared_functions():
        assert hasattr(lib, n), "libgbp_mi355x_test.so does not export %s" % n


def test_struct_sizes_match_the_header():
    from gbp_poplar_amd import _cabi as cabi
    assert ctypes.sizeof(cabi.GbpParams) == 4 * 12
    assert ctypes.sizeof(cabi.GbpShard) == 16
    assert ctypes.sizeof(cabi.GbpProblem) == 16 + 16 + 36 + 4   # 3 x u32 (+pad), 2 pointers, K[9] (+pad)
    assert ctypes.sizeof(cabi.GbpEvalOut) == 56
    assert ctypes.sizeof(cabi.GbpStateIn) == 15 * 8 and ctypes.sizeof(cabi.GbpStateOut) == 7 * 8


def test_no_cpu_fallback_without_a_gpu():
    """gbp_create must fail loudly (GBP_ERR_NO_DEVICE) instead of falling back to host code."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from gbp_poplar_amd import hostlib
    from gbp_poplar_amd.engine import GbpEngine, GbpError
    bal = hostlib.synth_generate(4, 24, 3, 1)
    with pytest.raises(GbpError, match="no HIP device"):
        GbpEngine(bal["cam_id"], bal["lmk_id"], 4, 24, [1] * 9)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "gbp_poplar_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "oracle/" not in txt.replace("oracle/oracle.py, test", "").replace("under oracle/", "") or f in ("_cabi.py", "__init__.py", "distributed.py"), f
