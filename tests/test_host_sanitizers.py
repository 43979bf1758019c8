"""AddressSanitizer + UBSan over the CPU-side C/C++ (host helpers of the C-ABI, the device-order builder of gbp_create and the oracle).  GPU sanitizers are
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
    # gbp_comm.cpp (transports of the multi-rank exchange) is host code too; its HIP / RCCL headers are on the image, the
    # harness only exercises the parts that need no device (region layout, abort flag)
    subprocess.check_call(["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "gbp_poplar_amd/csrc/gbp_host.cpp"), os.path.join(ROOT, "gbp_poplar_amd/csrc/gbp_comm.cpp"),
                           os.path.join(ROOT, "gbp_poplar_amd/csrc/gbp_layout.cpp"), os.path.join(ROOT, "tests/sanitize/layout_sanitize.cpp"),
                           os.path.join(ROOT, "tests/sanitize/comm_glue.cpp"),
                           os.path.join(ROOT, "tests/sanitize/host_sanitize_main.cpp")] + objs + san
                          + ["-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-lm", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], cwd=ROOT)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode == 0 and "sanitize: ok" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-3000:])
