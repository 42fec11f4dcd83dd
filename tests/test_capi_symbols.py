"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/iaf_hip.h
declares, the ctypes table covers them all, argument validation answers before any device call, and the host
mirror of the mask functions matches the reference golden masks.  No compute calls (no GPU needed)."""
import ctypes
import os
import re

import numpy as np
import pytest

import golden_inputs as gi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from iaf_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    _capi.lib()
    return _capi


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "iaf_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(iaf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(capi):
    syms = declared_symbols()
    assert len(syms) >= 20
    lib = ctypes.CDLL(capi.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "library does not export %s" % s
    assert sorted(capi.SIGNATURES) == syms, "ctypes table and header disagree"


def test_abi_version_and_error_strings(capi):
    lib = capi.lib()
    assert lib.iaf_abi_version() == 8 == capi.IAF_ABI_VERSION
    assert b"halo exchange" in lib.iaf_error_string(capi.IAF_ERR_EXCHANGE)
    with pytest.raises(capi.ExchangeError):
        capi.check(capi.IAF_ERR_EXCHANGE)
    assert b"null" in lib.iaf_error_string(capi.IAF_ERR_NULL)
    assert b"multiple" in lib.iaf_error_string(capi.IAF_ERR_NOT_MULTIPLE)


def test_argument_validation_mirrors_reference_errors(capi):
    lib = capi.lib()
    h = ctypes.c_void_p()
    assert lib.iaf_stack_create(None, 32, 160, 2, 0) == capi.IAF_ERR_NULL
    assert lib.iaf_stack_create(ctypes.byref(h), 64, 160, 2, 0) == capi.IAF_ERR_NOT_MULTIPLE   # layers.py:116
    assert lib.iaf_stack_create(ctypes.byref(h), 0, 160, 2, 0) == capi.IAF_ERR_SHAPE
    assert lib.iaf_stack_create(ctypes.byref(h), 4, 8, 2, 1) == capi.IAF_ERR_UNSUPPORTED       # channels % 16: the generic
                                                                                               # fallback is TF-variant only
    assert lib.iaf_stack_create(ctypes.byref(h), 32, 160, 2, 7) == capi.IAF_ERR_UNSUPPORTED    # unknown variant
    with pytest.raises(AssertionError):
        capi.check(capi.IAF_ERR_NOT_MULTIPLE)
    with pytest.raises(ValueError):
        capi.check(capi.IAF_ERR_SHAPE)
    assert lib.iaf_gaussian_sample(None, None, None, None, 10, None) == capi.IAF_ERR_NULL
    assert lib.iaf_compute_lowerbound(None, None, None, 4, 1, None) == capi.IAF_ERR_NULL


def test_host_masks_match_reference_golden(golden_dir):
    import iaf_amd
    g = np.load(os.path.join(golden_dir, "masks.npz"))
    for n_in, n_out, zd in gi.MASK_CASES:
        key = "%d_%d_%d" % (n_in, n_out, int(zd))
        np.testing.assert_array_equal(iaf_amd.get_linear_ar_mask(n_in, n_out, zd), g["lin_" + key].astype(np.float32))
        np.testing.assert_array_equal(iaf_amd.get_conv_ar_mask(3, 3, n_in, n_out, zd), g["conv_" + key].astype(np.float32))
    np.testing.assert_array_equal(iaf_amd.get_conv_ar_mask(5, 5, 8, 16, False), g["conv5x5_8_16_0"].astype(np.float32))
    with pytest.raises(AssertionError):
        iaf_amd.get_linear_ar_mask(64, 160)


def test_product_never_imports_oracle():
    """the product path must not route through the CPU oracle (parity claims depend on it)"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "iaf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no CPU fallback", ""), "%s mentions the oracle" % f
