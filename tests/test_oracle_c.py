"""The plain-C restatement (oracle/iaf_oracle.c, direct loops) must agree with the NumPy oracle and with the
reference golden outputs.  CPU only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import golden_inputs as gi
from oracle import iaf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def clib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libiaf_oracle_c.so"))
    lib.iaf_oracle_c_step.restype = ctypes.c_int
    return lib


def c_step(lib, z, ctx, params, n_h):
    B, n_z, H, W = z.shape
    d = len(n_h)
    names = ["layer_%d" % i for i in range(d)] + ["layer_out_0", "layer_out_1"]
    keep = []

    def arr(kind):
        ptrs = (ctypes.c_void_p * len(names))()
        for i, nm in enumerate(names):
            a = np.ascontiguousarray(params[nm + "/" + kind], dtype=np.float64)
            keep.append(a)
            ptrs[i] = a.ctypes.data
        return ptrs

    z = np.ascontiguousarray(z, dtype=np.float64)
    ctx = np.ascontiguousarray(ctx, dtype=np.float64)
    outs = [np.empty_like(z) for _ in range(4)]
    rc = lib.iaf_oracle_c_step(ctypes.c_void_p(z.ctypes.data), ctypes.c_void_p(ctx.ctypes.data), arr("V"), arr("g"), arr("b"),
                               n_z, n_h[0] if d else n_z, d, B, H, W, *[ctypes.c_void_p(o.ctypes.data) for o in outs])
    assert rc == 0
    return outs


@pytest.mark.parametrize("name", ["ar_tiny", "ar_k_down", "ar_depth1", "ar_depth4", "ar_cfg1_4x4"])
def test_c_restatement_matches_reference_golden(clib, golden_dir, name):
    g = np.load(os.path.join(golden_dir, "ar_multiconv2d.npz"))
    c = gi.ar_case_inputs(name)
    z_new, logsd, m_raw, s_raw = c_step(clib, c["z"], c["context"], c["params"], c["n_h"])
    np.testing.assert_allclose(m_raw, g[name + "/m_raw"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(s_raw, g[name + "/s_raw"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(z_new, g[name + "/z_new"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(logsd, g[name + "/logsd"], rtol=1e-10, atol=1e-11)


def test_c_restatement_matches_numpy_oracle_random():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libiaf_oracle_c.so"))
    rng = np.random.RandomState(21)
    n_z, n_h, B, H, W = 16, [48, 48], 2, 5, 3
    params = gi.ar_multiconv2d_params(rng, n_z, n_h, [n_z, n_z])
    z = rng.standard_normal((B, n_z, H, W))
    ctx = rng.standard_normal((B, n_h[0], H, W))
    z_new, logsd, _, _ = c_step(lib, z, ctx, params, n_h)
    ez, es = O.iaf_step(z, ctx, params, n_h)
    np.testing.assert_allclose(z_new, ez, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(logsd, es, rtol=1e-10, atol=1e-11)
