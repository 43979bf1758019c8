"""Build recipe of the native library (in-tree, gfx950 only).

    python -m gbp_poplar_amd.build        -> gbp_poplar_amd/libgbp_mi355x.so       the product (+ bin/ba, bin/slam, bin/bal_convert)
                                             gbp_poplar_amd/libgbp_mi355x_test.so  the same sources + the test hooks of
                                                                                   include/gbp_mi355x_debug.h (tests/ only)
    python -m gbp_poplar_amd.build --experiments
                                          -> gbp_poplar_amd/libgbp_mi355x_exp.so   + csrc/experiments/: timing ablations / mapping
                                                                                   experiments (profiles/*.py only)

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off: results are compared
bit-for-bit with the CPU oracle, so no FMA contraction on either side.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgbp_mi355x.so")
TEST_LIB = os.path.join(HERE, "libgbp_mi355x_test.so")  # + gbp_debug_* (include/gbp_mi355x_debug.h): what tests/ load for stage-level checks
EXP_LIB = os.path.join(HERE, "libgbp_mi355x_exp.so")    # + timing ablations and mapping experiments (profiles/*.py only)
BIN = os.path.join(HERE, "bin")
ARCH = "gfx950"
# -fvisibility=hidden: the library exports the gbp_* functions of include/*.h (GBP_API) and nothing else
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function"]
LIB_SRCS = ["gbp_kernels.hip", "gbp_capi.cpp", "gbp_layout.cpp", "gbp_comm.cpp", "gbp_host.cpp"]
CLI_SRCS = {"ba": "ba_main.cpp", "slam": "slam_main.cpp", "bal_convert": "bal_convert_main.cpp"}


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps():
    inc = os.path.join(HERE, "..", "include")
    srcs = [os.path.join(d, f) for d, _, fs in os.walk(CSRC) for f in fs]          # csrc/, csrc/hooks/, csrc/experiments/
    return srcs + [os.path.join(inc, f) for f in os.listdir(inc)]


def _build_lib(target, defines, force, verbose):
    if force or _stale(target, _deps()):
        cmd = [hipcc(), "-shared", "-o", target] + FLAGS + defines + ["-x", "hip"] + \
              [os.path.join(CSRC, s) for s in LIB_SRCS] + ["-ldl", "-Wl,--version-script=" + os.path.join(CSRC, "gbp_exports.map")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return target


def build(force=False, verbose=False, test_hooks=True):
    deps = _deps()
    _build_lib(LIB, [], force, verbose)
    if test_hooks:
        _build_lib(TEST_LIB, ["-DGBP_BUILD_TEST_HOOKS"], force, verbose)
    if os.path.exists(EXP_LIB):          # keep an existing measurement build in step with the sources (never created here)
        build_experiments(force, verbose)
    os.makedirs(BIN, exist_ok=True)
    for name, src in CLI_SRCS.items():
        path = os.path.join(CSRC, src)
        exe = os.path.join(BIN, name)
        if os.path.exists(path) and (force or _stale(exe, deps + [LIB])):
            # the CLIs are plain C++ on top of the C-ABI: host compiler, linked against the in-tree library
            cmd = [shutil.which("g++") or "g++", "-o", exe, "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", path,
                   "-L" + HERE, "-lgbp_mi355x", "-Wl,-rpath,$ORIGIN/.."]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    return LIB


def build_experiments(force=False, verbose=False):
    """The measurement build: the product sources + test hooks + the ablated / experimental kernel instantiations.
    Loaded only by profiles/*.py (GBP_LIB)."""
    return _build_lib(EXP_LIB, ["-DGBP_BUILD_EXPERIMENTS", "-DGBP_BUILD_TEST_HOOKS"], force, verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--experiments" in sys.argv:
        print(build_experiments(force="--force" in sys.argv, verbose=True))
