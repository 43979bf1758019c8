"""Build recipe of the native library (in-tree, gfx950 only).

    python -m gbp_poplar_amd.build        -> gbp_poplar_amd/libgbp_mi355x.so       the product (+ bin/ba, bin/slam, bin/bal_convert)
                                             gbp_poplar_amd/libgbp_mi355x_test.so  the same sources + the test hooks of
                                                                                   include/gbp_mi355x_debug.h (tests/ only)
    python -m gbp_poplar_amd.build --experiments
                                          -> gbp_poplar_amd/libgbp_mi355x_exp.so   + csrc/experiments/: timing ablations / mapping
                                                                                   experiments (profiles/*.py only)

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off: results are compared
bit-for-bit with the CPU oracle, so no FMA contraction on either side.

Every translation unit is compiled to an object of its own (gbp_poplar_amd/_obj/<variant>/, in parallel, only when it or a
header changed) and the objects are linked: the device code (gbp_kernels.hip, ~25 s) is not recompiled for an edit of the host
side of the C-ABI, whose translation units are plain C++ (no device pass).
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libgbp_mi355x.so")
TEST_LIB = os.path.join(HERE, "libgbp_mi355x_test.so")  # + gbp_debug_* (include/gbp_mi355x_debug.h): what tests/ load for stage-level checks
EXP_LIB = os.path.join(HERE, "libgbp_mi355x_exp.so")    # + timing ablations and mapping experiments (profiles/*.py only)
BIN = os.path.join(HERE, "bin")
ARCH = "gfx950"
# -fvisibility=hidden: the library exports the gbp_* functions of include/*.h (GBP_API) and nothing else
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
DEVICE_SRCS = ["gbp_kernels.hip"]                        # HIP: host + gfx950 device pass
# the C-ABI (gbp_ctx.hpp names what each holds), the device order, the exchange transports, the host helpers: plain C++
HOST_SRCS = ["gbp_api_ctx.cpp", "gbp_api_launch.cpp", "gbp_api_persist.cpp", "gbp_api_eval.cpp", "gbp_api_comm.cpp", "gbp_api_debug.cpp",
             "gbp_layout.cpp", "gbp_comm.cpp", "gbp_host.cpp"]
LIB_SRCS = DEVICE_SRCS + HOST_SRCS
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


def _headers():
    inc = os.path.join(HERE, "..", "include")
    hdr = [os.path.join(d, f) for d, _, fs in os.walk(CSRC) for f in fs if f.endswith((".h", ".hpp"))]
    hdr += [os.path.join(d, f) for d in (os.path.join(CSRC, "hooks"), os.path.join(CSRC, "experiments")) if os.path.isdir(d)
            for f in os.listdir(d)]          # device sources #included by gbp_kernels.hip
    return hdr + [os.path.join(inc, f) for f in os.listdir(inc)] + [os.path.abspath(__file__)]


def _deps():
    """every file a library depends on (bench.py stamps its traffic figures with a hash over these)"""
    return sorted(set([os.path.join(CSRC, s) for s in LIB_SRCS] + _headers()))


def _compile(src, obj, defines, verbose):
    cmd = [hipcc(), "-c", "-o", obj] + FLAGS + defines
    if src.endswith(".hip"):
        cmd += ["--offload-arch=" + ARCH, "-x", "hip"]
    else:
        cmd += ["-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc()))), "include")]
    cmd.append(os.path.join(CSRC, src))
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _build_lib(target, variant, defines, force, verbose):
    odir = os.path.join(OBJ, variant)
    os.makedirs(odir, exist_ok=True)
    hdr = _headers()
    objs, todo = [], []
    for s in LIB_SRCS:
        obj = os.path.join(odir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, hdr + [os.path.join(CSRC, s)]):
            todo.append((s, obj))
    if todo:
        workers = max(1, min(len(todo), (os.cpu_count() or 2)))
        with concurrent.futures.ThreadPoolExecutor(workers) as ex:
            for f in [ex.submit(_compile, s, o, defines, verbose) for s, o in todo]:
                f.result()
    if todo or force or _stale(target, objs):
        cmd = [hipcc(), "-shared", "-o", target, "--offload-arch=" + ARCH] + objs + \
              ["-ldl", "-pthread", "-Wl,--version-script=" + os.path.join(CSRC, "gbp_exports.map")]      # (-pthread: the host helpers' reader threads, for a libc that still keeps them in libpthread)
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return target


def build(force=False, verbose=False, test_hooks=True):
    _build_lib(LIB, "product", [], force, verbose)
    if test_hooks:
        _build_lib(TEST_LIB, "test", ["-DGBP_BUILD_TEST_HOOKS"], force, verbose)
    if os.path.exists(EXP_LIB):          # keep an existing measurement build in step with the sources (never created here)
        build_experiments(force, verbose)
    os.makedirs(BIN, exist_ok=True)
    inc = os.path.join(HERE, "..", "include")
    cli_deps = [os.path.join(CSRC, "cli_common.hpp")] + [os.path.join(inc, f) for f in os.listdir(inc)] + [LIB]
    for name, src in CLI_SRCS.items():
        path = os.path.join(CSRC, src)
        exe = os.path.join(BIN, name)
        if os.path.exists(path) and (force or _stale(exe, cli_deps + [path])):
            # the CLIs are plain C++ on top of the C-ABI: host compiler, linked against the in-tree library
            cmd = [shutil.which("g++") or "g++", "-o", exe, "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", path,
                   "-L" + HERE, "-lgbp_mi355x", "-Wl,-rpath,$ORIGIN/.."]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return LIB


def build_experiments(force=False, verbose=False):
    """The measurement build: the product sources + test hooks + the ablated / experimental kernel instantiations.
    Loaded only by profiles/*.py (GBP_LIB)."""
    return _build_lib(EXP_LIB, "exp", ["-DGBP_BUILD_EXPERIMENTS", "-DGBP_BUILD_TEST_HOOKS"], force, verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--experiments" in sys.argv:
        print(build_experiments(force="--force" in sys.argv, verbose=True))
