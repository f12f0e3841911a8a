"""The device code must not contain packed-fp32 instructions (v_pk_mul/add/fma_f32): beside MFMA waves they produced wrong
values in the split GEMM's staging waves.  The Makefile disables them with a target feature; this test compiles
the GEMM kernels with the Makefile's own flags and looks at the ISA."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "candle_birefnet_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_no_packed_fp32_in_gemm_isa(tmp_path):
    mk = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^CXXFLAGS\s*=\s*(.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    assert "-packed-fp32-ops" in flags, "the Makefile no longer disables packed fp32"
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / "gemm.s"
    flags = [f for f in flags if f != "-fPIC"]
    subprocess.run([hipcc, *flags, "-x", "hip", "-S", "--cuda-device-only", os.path.join(CSRC, "kernels", "gemm_f32.hip"), "-o", str(out)],
                   check=True, capture_output=True, timeout=600)
    isa = out.read_text()
    assert "v_mfma_f32_32x32x16_bf16" in isa                      # it is the device ISA we are looking at
    packed = re.findall(r"^\s*(v_pk_(?:mul|add|fma)_f32)\b", isa, re.M)
    assert not packed, f"{len(packed)} packed-fp32 instructions in the GEMM kernels"
