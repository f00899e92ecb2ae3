"""The shipped gfx950 code objects contain no packed-fp32 VALU instruction.

On MI355X a v_pk_{mul,fma,add}_f32 whose op_sel crosses register halves returns wrong results in lanes 48-63 while another
wave's MFMA is executing (tools/probe/pk_probe.hip, profiles/r01_h_pk_probe.txt).  The compiler forms such
instructions on its own, so the build switches the feature off (Makefile: NOPKF32); this test disassembles libstn.so and fails
if any came back (e.g. after a flag change)."""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not in this image")
def test_no_packed_fp32_in_device_code():
    lib = os.path.join(ROOT, "supertonic_amd", "libstn.so")
    assert os.path.exists(lib), "libstn.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, "libstn.so")
        shutil.copy(lib, copy)
        subprocess.run([OBJDUMP, "--offloading", copy], check=True, capture_output=True, cwd=tmp)  # writes <copy>.N.<triple>
        objs = sorted(glob.glob(copy + ".*gfx950"))
        assert objs, "no gfx950 code object found in libstn.so"
        n_mfma = 0
        for co in objs:
            asm = subprocess.run([OBJDUMP, "-d", co], check=True, capture_output=True, text=True).stdout
            bad = re.findall(r"^\s*v_pk_[a-z0-9]+_f32.*$", asm, flags=re.M)
            assert not bad, f"{len(bad)} packed-fp32 instructions in {os.path.basename(co)}, e.g. {bad[0].strip()}"
            n_mfma += len(re.findall(r"\bv_mfma_", asm))
        assert n_mfma > 100  # the disassembly really is the engine's kernels
