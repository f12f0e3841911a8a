"""The device code must not contain packed-fp32 instructions (v_pk_mul/add/fma_f32).  Round 1 found wrong values in the split GEMM's
staging waves while MFMA waves shared their SIMD; round 2 traced it in the ISA (profiles/r02_packed_fp32_producer_before.s): hipcc
broadcasts the scalar mask with `op_sel_hi:[0,1]` — both halves of the packed multiply are to read the LOW register of the source
pair — and, treating the pair's HIGH register as unread, reuses it as scratch (v79 below holds packed bf16 bits or a 0 / 1.0 mask):

    v_pk_mul_f32 v[98:99], v[78:79], v[50:51] op_sel_hi:[0,1]
    v_cvt_pk_bf16_f32 v79, v98, s0                                  <- the "unused" high half is live scratch
    v_pk_fma_f32 v[96:97], v[78:79], v[52:53], v[96:97] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]

A high lane that reads v79 instead of v78 multiplies by that scratch value (0 or a denormal-sized bit pattern): exactly the symptom
(one operand of one row pair "multiplied by zero", always the op_sel'd source, last 16 lanes, only beside MFMA waves).  No
intra-wave hazard rule applies (the staging waves issue no MFMA), so this is an op_sel source-select problem of the packed-fp32 path
under matrix-pipe contention, not a missing wait state; the fix is not to emit packed fp32 at all (Makefile target feature + asm
helpers in kernels/split_planes.h).  This test compiles EVERY kernel file with the Makefile's own flags and looks at the ISA."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "candle_birefnet_amd", "csrc")
KERNELS = sorted(os.path.basename(f) for f in glob.glob(os.path.join(CSRC, "kernels", "*.hip")))


def _flags():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^CXXFLAGS\s*=\s*(.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    return [f for f in flags if f != "-fPIC"]


def test_makefile_disables_packed_fp32_and_lists_every_kernel():
    assert "-packed-fp32-ops" in _flags(), "the Makefile no longer disables packed fp32"
    mk = open(os.path.join(CSRC, "Makefile")).read()
    for k in KERNELS:
        assert f"kernels/{k}" in mk, f"{k} is not built by the Makefile"


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
@pytest.mark.parametrize("kernel", KERNELS)
def test_no_packed_fp32_in_isa(tmp_path, kernel):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / (kernel + ".s")
    extra = ["-ffp-contract=off"] if kernel == "imageproc.hip" else []
    subprocess.run([hipcc, *_flags(), *extra, "-x", "hip", "-S", "--cuda-device-only", os.path.join(CSRC, "kernels", kernel), "-o", str(out)],
                   check=True, capture_output=True, timeout=900)
    isa = out.read_text()
    assert ".amdhsa_kernel" in isa                                # it is the device ISA we are looking at
    if kernel.startswith("gemm"):
        assert "v_mfma_f32_32x32x" in isa
    packed = re.findall(r"^\s*(v_pk_(?:mul|add|fma)_f32)\b", isa, re.M)
    assert not packed, f"{len(packed)} packed-fp32 instructions in {kernel}"
