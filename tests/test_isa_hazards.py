"""The hand-placed DPP adds of the camera row sums (row16_sums_store, gbp_kernels.hip) sit in inline asm, which hipcc's hazard
recogniser does not look into: on gfx9-class hardware a DPP operand must have been written at least two instructions (wait
states) before the DPP instruction reads it — nothing interlocks, the lanes would read stale registers.  The kernel source
arranges that by construction (blocks of six behind an s_nop, operands produced before their block); this test holds the
SHIPPED code objects to it: it disassembles the gfx950 code object of every in-tree library and checks every v_add_f32_dpp.
No GPU needed (llvm-objdump from the ROCm image)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
LIBS = ["libgbp_mi355x.so", "libgbp_mi355x_test.so", "libgbp_mi355x_exp.so"]


def _disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="gbp_isa_")
    try:
        shutil.copy(lib, tmp)
        name = os.path.basename(lib)
        subprocess.run([OBJDUMP, "--offloading", name], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        assert len(co) == 1, os.listdir(tmp)
        return subprocess.run([OBJDUMP, "-d", co[0]], cwd=tmp, check=True, stdout=subprocess.PIPE, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _vregs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"^[av]\[(\d+):(\d+)\]$", tok)
    if m and tok[0] == "v":
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


@pytest.mark.parametrize("libname", LIBS)
def test_no_dpp_operand_is_read_within_two_wait_states_of_its_write(libname):
    lib = os.path.join(ROOT, "gbp_poplar_amd", libname)
    if not os.path.exists(lib) or not os.path.exists(OBJDUMP):
        pytest.skip("library or llvm-objdump not present")
    text = _disassemble(lib)
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        ins = line.split("//")[0].strip()
        if cur is not None and ins and not ins.endswith(":"):
            cur.append(ins)
    n_dpp, n_kernels = 0, 0
    for name, ins in kernels.items():
        here = 0
        for j, l in enumerate(ins):
            if not l.startswith("v_add_f32_dpp"):
                continue
            here += 1
            toks = l.split()
            src = _vregs(toks[2])                       # the operand that goes through the DPP network
            assert src, l
            waited = 0
            for k in range(j - 1, max(j - 3, -1), -1):  # the two instructions in front (an s_nop N counts N + 1 wait states)
                p = ins[k].split()
                if p[0] == "s_nop":
                    waited += int(p[1]) + 1
                    if waited >= 2:
                        break
                    continue
                if waited >= 2:
                    break
                if p[0].startswith(("v_", "global_load", "buffer_load", "ds_read", "scratch_load")) and len(p) > 1:
                    assert not (_vregs(p[1]) & src), "%s: DPP operand of `%s` written by `%s` %d instruction(s) earlier" % (name, l, ins[k], j - k)
                waited += 1
        if here:
            n_kernels += 1
            n_dpp += here
    if libname == "libgbp_mi355x.so":
        # k_sweep<..>, k_persist<true|false>, k_persist_flow<true|false>: 60 tree nodes each
        assert n_kernels >= 4 and n_dpp >= 240, (n_kernels, n_dpp)
