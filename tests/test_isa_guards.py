"""Guards on the SHIPPED device code (the in-tree libc2m_hip.so, disassembled): hardware lessons that a source edit can
silently undo.  No GPU needed; skipped when the ROCm llvm-objdump is not installed."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "c2-matching_amd", "csrc", "libc2m_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.fixture(scope="module")
def disassembly(tmp_path_factory):
    if not os.path.exists(OBJDUMP):
        pytest.skip("ROCm llvm-objdump not installed")
    if not os.path.exists(LIB):
        pytest.fail("libc2m_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    d = tmp_path_factory.mktemp("bundles")
    shutil.copy(LIB, d / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=d, check=True, capture_output=True)
    objs = sorted(f for f in os.listdir(d) if "amdgcn-amd-amdhsa--gfx950" in f)
    assert objs, "no gfx950 code objects in the library"
    text = []
    for f in objs:
        out = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f], cwd=d, check=True, capture_output=True, text=True).stdout
        text.append(out)
    return "\n".join(text)


def test_library_is_gfx950_only_and_has_matrix_kernels(disassembly):
    assert "v_mfma_f32_32x32x16_f16" in disassembly and "v_mfma_f32_32x32x2_f32" in disassembly


def test_no_128_bit_buffer_store_with_a_register_soffset(disassembly):
    """gfx950: `buffer_store_dwordx4 v[a:a+3], voff, s[..], sN offen` followed within two issue slots by a VALU write of
    v[a:a+3] stores the NEW value in lanes 12..15 / 28..31 of each half -- hipcc pads that hazard only for the immediate-
    soffset form (found twice: DESIGN.md 6.2 (round 4), 6.9 (round 5)).  The library keeps every wide buffer store on the
    immediate / zero soffset form; this test fails if an edit re-introduces the register form."""
    pat = re.compile(r"buffer_store_dwordx[34]\s+v\[\d+:\d+\],\s*\S+,\s*s\[\d+:\d+\],\s*(\S+)")
    wide = [m.group(1) for m in pat.finditer(disassembly)]
    assert len(wide) > 100        # the channels-last epilogues are there
    reg = [s for s in wide if re.fullmatch(r"(s\d+|m0|vcc_lo|vcc_hi|ttmp\d+)", s)]
    assert not reg, f"{len(reg)} wide buffer stores with a register soffset: {sorted(set(reg))[:8]}"
