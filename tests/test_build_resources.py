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
