"""CPU test for DESIGN_LESSONS.md lesson 46: the shipped library must not contain the packed-fp32 instruction form that computes wrong
results beside fp16 / bf16 MFMA kernels on MI355X -- v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 whose SECOND source supplies its
HIGH register to the LOW half of the result (op_sel:[x,1] / op_sel:[x,1,x]; scripts/repro/pk_opsel_matrix.hip measured exactly
these forms wrong, profiles/r06_overlap/r06_pk_opsel_matrix.log).  The check is static: the library's gfx950 code objects are
disassembled with llvm-objdump (scripts/isa_pk_opsel.py) -- no GPU needed, and no reliance on a race showing up in a test run."""
import io
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import isa_pk_opsel as ISA  # noqa: E402

LIB = os.path.join(ROOT, "patchmatchnet_amd", "csrc", "libpmn_hip.so")
UNGUARDED = os.path.join(ROOT, "build", "wc", "libpmn_hip_nosettle.so")
needs_llvm = pytest.mark.skipif(not os.path.exists(os.path.join(ISA.LLVM, "llvm-objdump")), reason="llvm-objdump of the ROCm toolchain not found")


def test_the_checker_flags_exactly_the_measured_forms():
    text = "\n".join([
        "0000000000001000 <kernel_a>:",
        "\tv_pk_add_f32 v[34:35], v[78:79], v[34:35] op_sel:[0,1]        // 000000001000: D3B24022 10026F4E",
        "\tv_pk_mul_f32 v[8:9], s[2:3], v[82:83] op_sel:[0,1] op_sel_hi:[1,0]   // 0000",
        "\tv_pk_fma_f32 v[48:49], v[26:27], v[64:65], v[48:49] op_sel:[0,1,0]// 0000",
        "\tv_pk_fma_f32 v[48:49], v[26:27], v[64:65], v[48:49] op_sel:[1,1,0]// 0000",
        "0000000000002000 <kernel_b>:",
        "\tv_pk_add_f32 v[34:35], v[78:79], v[34:35] op_sel:[1,0]        // src0's high half: measured right",
        "\tv_pk_add_f32 v[34:35], v[78:79], v[34:35] op_sel_hi:[1,0]     // low half to both: measured right",
        "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[1,0,0] op_sel_hi:[0,1,1]",
        "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1]    // src2's high half: measured right",
        "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7]",
        "\tv_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]              // not arithmetic: measured right",
    ])
    found, seen = ISA.sites(text)
    assert seen == 9
    assert list(found) == ["kernel_a"] and len(found["kernel_a"]) == 4


@needs_llvm
def test_shipped_library_contains_no_affected_instruction():
    assert os.path.exists(LIB), "libpmn_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    out = io.StringIO()
    n = ISA.check([LIB], verbose=True, out=out)
    report = out.getvalue()
    assert "packed-fp32 instructions" in report and " 0 packed-fp32" not in report, report  # the disassembly really saw the kernels
    assert n == 0, "libpmn_hip.so contains packed-fp32 instructions of the form lesson 46 measured wrong beside MFMA kernels:\n" + report


@needs_llvm
def test_the_unguarded_build_contains_it():
    """-DPMN_NO_SETTLE (scripts/build_waitcnt_variants.sh) compiles the pins out: the form must come back, or the check proves nothing."""
    if not os.path.exists(UNGUARDED):
        pytest.skip("build/wc/libpmn_hip_nosettle.so not built (bash scripts/build_waitcnt_variants.sh)")
    out = io.StringIO()
    assert ISA.check([UNGUARDED], out=out) > 100, out.getvalue()
