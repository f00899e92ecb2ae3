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


def _checker():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_hazards", os.path.join(ROOT, "tools", "check_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_hazard_checker_sees_the_exit_edge_pattern():
    """The checker itself, on the instruction pattern that shipped wrong and on its fixed forms."""
    ch = _checker()
    loop_tail = [(0x00, "v_mfma_f32_32x32x16_f16", "a[0:15], v[6:9], v[10:13], a[0:15]"),
                 (0x08, "s_andn2_b64", "vcc, exec, s[0:1]"),
                 (0x0c, "s_cbranch_vccz", "1"),          # exit edge -> 0x14
                 (0x10, "s_branch", "65531"),            # back edge -> 0x00
                 (0x14, "s_or_b64", "exec, exec, s[10:11]"),
                 (0x18, "s_waitcnt", "lgkmcnt(0)"),
                 (0x1c, "s_nop", "0")]
    bad = loop_tail + [(0x20, "v_accvgpr_read_b32", "v19, a15"), (0x24, "s_endpgm", "")]
    found = ch.check_function("k", bad)
    assert len(found) == 1 and "read after 5 wait states (need 11)" in found[0][2], found
    good = loop_tail + [(0x20, "s_nop", "7"), (0x24, "v_accvgpr_read_b32", "v19, a15"), (0x28, "s_endpgm", "")]
    assert ch.check_function("k", good) == []
    # a dependent MFMA (result as SrcC) is interlocked by the hardware; as an A/B operand it is not
    assert ch.check_function("k", [(0, "v_mfma_f32_32x32x16_bf16", "a[0:15], v[0:3], v[4:7], 0"),
                                   (8, "v_mfma_f32_32x32x16_bf16", "a[0:15], v[0:3], v[4:7], a[0:15]"), (16, "s_endpgm", "")]) == []
    assert len(ch.check_function("k", [(0, "v_mfma_f32_32x32x16_bf16", "v[0:15], v[20:23], v[24:27], 0"),
                                       (8, "v_mfma_f32_32x32x16_bf16", "v[32:47], v[0:3], v[24:27], 0"), (16, "s_endpgm", "")])) == 1
    # stores and packed fp32 are seen too; the fp32 MFMA needs 19 wait states
    assert len(ch.check_function("k", [(0, "v_mfma_f32_32x32x2_f32", "v[0:15], v20, v21, v[0:15]"), (8, "s_nop", "15"),
                                       (12, "global_store_dword", "v[30:31], v3, off"), (20, "s_endpgm", "")])) == 1
    assert len(ch.check_function("k", [(0, "v_pk_mul_f32", "v[0:1], v[2:3], v[4:5] op_sel:[0,1]")])) == 1


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not in this image")
def test_no_mfma_read_hazards_in_device_code():
    ch = _checker()
    found, n_mfma = ch.check_library(os.path.join(ROOT, "supertonic_amd", "libstn.so"))
    assert n_mfma > 1000
    assert not found, "\n".join(f"{n[:60]} @ {a:#x}: {w}: {d}" for n, a, w, d in found[:10])
