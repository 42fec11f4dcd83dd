"""CPU build check: the hot kernels keep their state in registers.  hipcc's kernel-resource-usage remarks for the plain / strided
bf16x3 conv shapes (one translation unit, device code only, ~15 s): no scratch memory, no spills.  Round 4 found two shapes that
had run from scratch since round 2 because a lambda of the K loop was not inlined (tools/kernel_resources.py lists every kernel)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_plain_and_strided_bf16x3_conv_kernels_use_no_scratch():
    import kernel_resources as kr
    rows = kr.all_resources(["iaf_bf3p_4_1_4_1"])
    assert len(rows) >= 12, "forward, data-gradient and the two strided forms at NT = 2, 4, 5"
    bad = [(r["name"], r.get("scratch"), r.get("vspill")) for r in rows if r.get("scratch", 0) or r.get("vspill", 0)]
    assert not bad, bad


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_one_launch_step_kernel_and_no_bf16x3_conv_kernel_uses_scratch():
    """all nine depth-4 exchange-form instantiations of the one-launch step ran with 13-27 spilled VGPRs inside their
    MFMA regions, unseen because this file looked at one translation unit.  Now: EVERY iaf_step_fused_* and iaf_bf3* unit of the build
    (~140 + ~250 kernels, a few minutes of device-only compiles, eight at a time) -- no scratch, no spilled VGPR."""
    import kernel_resources as kr
    rows = kr.all_resources(["iaf_step_fused_", "iaf_bf3"])
    fused = [r for r in rows if "iaf_step_fused_kernel" in r["name"]]
    assert len(fused) >= 130 and len(rows) >= 250, (len(fused), len(rows))
    # (kernels that spill a few SGPRs -- into VGPR lanes, v_writelane / v_readlane, outside the K loop -- get a 20-byte private segment reserved
    #  for that VGPR although no scratch instruction is emitted: the stride-2 deconv forms S2 = 2 and two fp16-plane 512-thread forms.  What
    #  this test forbids is a spilled VECTOR register or a private segment beyond that frame.)
    bad = [(r["unit"], r["name"], r.get("scratch"), r.get("vspill")) for r in rows
           if r.get("vspill", 0) or (r.get("scratch", 0) > (32 if r.get("sspill", 0) else 0))]
    assert not bad, bad
