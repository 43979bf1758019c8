"""AddressSanitizer + UBSan over the CPU-side C/C++ (the whole host side of the C-ABI library — gbp_api_*.cpp with the device code stubbed out, negative
tests of every export that needs no device —, the host helpers, the device-order builder of gbp_create and the oracle).  GPU sanitizers are
not available on this pool, so this is the memory-safety gate of everything that runs on the host; the harness is
tests/sanitize/host_sanitize_main.cpp."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_and_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1",
           "-ffp-contract=off"]
    objs = []
    for src in ("oracle/oracle_gbp.c", "oracle/oracle_math.c"):
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.check_call(["gcc", "-std=c11", "-c", os.path.join(ROOT, src), "-o", obj] + san, cwd=ROOT)
        objs.append(obj)
    # The host side of the library itself — the C-ABI (gbp_api_*.cpp), the device order, the transports of the multi-rank exchange, the
    # host helpers — compiled with g++ under the sanitizers; the device code is replaced by tests/sanitize/kernel_stubs.cpp (every
    # launcher aborts: nothing here may reach a launch), the HIP / RCCL headers and libamdhip64 are on the image.  api_negative.cpp
    # calls every export that needs no device with NULL / negative / out-of-order arguments.
    csrc = os.path.join(ROOT, "gbp_poplar_amd", "csrc")
    lib_srcs = [os.path.join(csrc, f) for f in ("gbp_api_ctx.cpp", "gbp_api_launch.cpp", "gbp_api_persist.cpp", "gbp_api_eval.cpp", "gbp_api_comm.cpp",
                                                "gbp_api_debug.cpp", "gbp_host.cpp", "gbp_comm.cpp", "gbp_layout.cpp")]
    harness = [os.path.join(ROOT, "tests", "sanitize", f) for f in ("layout_sanitize.cpp", "kernel_stubs.cpp", "api_negative.cpp", "host_sanitize_main.cpp")]
    cxx = ["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-DGBP_BUILD_TEST_HOOKS", "-I/opt/rocm/include", "-c"] + san
    cxx_objs = []
    from concurrent.futures import ThreadPoolExecutor
    def one(src):
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.check_call(cxx + [src, "-o", obj], cwd=ROOT)
        return obj
    with ThreadPoolExecutor(4) as ex:
        cxx_objs = list(ex.map(one, lib_srcs + harness))
    subprocess.check_call(["g++"] + cxx_objs + objs + san + ["-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-lm", "-pthread", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], cwd=ROOT)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode == 0 and "sanitize: ok" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-3000:])


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_threaded_host_code_under_tsan(tmp_path):
    """The host code that runs on several threads — the file reader (one piece of the file per thread) and the prior strengths (one range
    of factors per thread) — under ThreadSanitizer: results equal to the single-threaded ones, no report (tests/sanitize/host_threads_main.cpp)."""
    exe = str(tmp_path / "host_threads")
    subprocess.check_call(["g++", "-std=c++17", "-fsanitize=thread", "-O1", "-g", "-pthread", "-ffp-contract=off",
                           os.path.join(ROOT, "gbp_poplar_amd", "csrc", "gbp_host.cpp"), os.path.join(ROOT, "tests", "sanitize", "host_threads_main.cpp"), "-o", exe], cwd=ROOT)
    p = subprocess.run([exe, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    if "unexpected memory mapping" in p.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow on this kernel")
    assert p.returncode == 0 and "tsan: ok" in p.stdout and "ThreadSanitizer" not in p.stderr, (p.returncode, p.stdout[-300:], p.stderr[-3000:])
