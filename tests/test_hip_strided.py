"""GPU parity tests for the downsampling IAFLayer's two strided convs at their minimal work (include/iaf_hip.h:
iaf_conv3x3_forward_stride2, iaf_conv3x3_forward_deconv; kernels: iaf_amd/csrc/iaf_conv_bf3.hpp, template parameter S2):
conv2d(stride=[2,2], SAME) of tf_train.py:33,36 and deconv2d of tf_train.py:89-91 (tf_utils/layers.py:31-64, 83-112).
Checked against the CPU oracle's conv2d(stride (2,2)) / deconv2d (pinned to the reference's own outputs by tests/test_oracle*.py) and against the stride-1 formulation they replace (full-resolution conv + subsampling; zero-inserted input).
The whole layer against the reference's own outputs: tests/test_hip_layer.py (iaf_layer_ds.npz)."""
import ctypes

import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import iaf_oracle as O

pytestmark = pytest.mark.gpu

ATOL = 1e-4


@pytest.fixture(scope="module")
def amd():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    import iaf_amd
    iaf_amd._capi.lib()      # raises if the HIP extension is missing: no silent fallback
    return iaf_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy().astype(np.float64)


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def ref_conv_stride2(x, p, elu):
    """the CPU oracle's conv2d with stride (2,2) (oracle/iaf_oracle.py, tf_utils/layers.py:31-64; TF SAME padding)"""
    xx = f32(x)
    return O.conv2d(O.elu(xx) if elu else xx, f32(p["V"]), f32(p["g"]), f32(p["b"]), stride=(2, 2))


def ref_deconv(x, p, elu):
    """the CPU oracle's deconv2d (tf_utils/layers.py:83-112: weight norm per input channel, conv2d_transpose SAME stride 2, + b)"""
    xx = f32(x)
    return O.deconv2d(O.elu(xx) if elu else xx, f32(p["V"]), f32(p["g"]), f32(p["b"]))


S2_CASES = [   # B, n_in, n_out, split, H_out, W_out, elu
    (32, 160, 384, [32, 32, 160, 160], 8, 8, True),       # up_conv1 of the BASELINE run's downsampling layer
    (5, 64, 128, [128], 4, 4, False),
    (3, 32, 96, [32, 64], 8, 8, True),
    (2, 160, 384, [32, 32, 160, 160], 2, 2, True),
    (7, 32, 32, [32], 3, 5, False),                       # odd output grid, W != H
    (2, 160, 64, [64], 16, 16, False),                    # 16-pixel output rows: 163 staged slots, 158 KiB
]


@pytest.mark.parametrize("case", S2_CASES, ids=lambda c: "B%d_%dto%d_%dx%d" % (c[0], c[1], c[2], c[4], c[5]))
def test_stride2_conv_vs_fp64_definition_and_vs_the_subsampled_stride1_conv(amd, case):
    B, ci, co, split, H, W, elu = case
    rng = np.random.RandomState(100 + B)
    p = gi.conv_params(rng, ci, co)
    x = rng.standard_normal((B, ci, 2 * H, 2 * W))
    conv = amd.WNConv2d(ci, co)
    conv.prepare(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    outs = (ctypes.c_void_p * len(split))()
    got_t = [torch.empty((B, c, H, W), device="cuda") for c in split]
    for k, t in enumerate(got_t):
        outs[k] = t.data_ptr()
    xd = dev(x)
    rc = amd._capi.lib().iaf_conv3x3_forward_stride2(conv._h, amd.layers._ptr(xd), 1 if elu else 0, outs,
                                                     (ctypes.c_int * len(split))(*split), len(split), B, H, W, amd.layers._stream())
    assert rc == 0, "the strided kernel covers this shape (rc %d)" % rc
    got = np.concatenate([host(t) for t in got_t], axis=1)
    want = ref_conv_stride2(x, p, elu)
    np.testing.assert_allclose(got, want, rtol=0, atol=ATOL)
    # ... and the stride-1 formulation it replaces (the method's own fallback)
    old = [amd.resample2(t, "down_odd") for t in conv(dev(x), elu_input=elu, split=split)]
    np.testing.assert_allclose(got, np.concatenate([host(t) for t in old], axis=1), rtol=0, atol=ATOL)
    # the method returns the same tensors
    via = conv.stride2(dev(x), elu_input=elu, split=split)
    assert all(torch.equal(a, b) for a, b in zip(via, got_t))


def test_stride2_conv_falls_back_where_the_phase_tiles_do_not_fit(amd):
    """c_in = 160 at 24-pixel output rows: four phase tiles of 32 pixels + halos exceed 160 KiB -> IAF_ERR_UNSUPPORTED from the entry
    point, and WNConv2d.stride2 computes the same numbers through the stride-1 kernel"""
    B, ci, co, H, W = 2, 160, 64, 4, 24
    rng = np.random.RandomState(7)
    p = gi.conv_params(rng, ci, co)
    x = rng.standard_normal((B, ci, 2 * H, 2 * W))
    conv = amd.WNConv2d(ci, co)
    conv.prepare(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    o = torch.empty((B, co, H, W), device="cuda")
    xd = dev(x)
    rc = amd._capi.lib().iaf_conv3x3_forward_stride2(conv._h, amd.layers._ptr(xd), 0, (ctypes.c_void_p * 1)(o.data_ptr()),
                                                     (ctypes.c_int * 1)(co), 1, B, H, W, amd.layers._stream())
    assert rc == amd._capi.IAF_ERR_UNSUPPORTED
    got = host(conv.stride2(dev(x))[0])
    np.testing.assert_allclose(got, ref_conv_stride2(x, p, False), rtol=0, atol=ATOL)


DECONV_CASES = [   # B, c1, c2 (x2 channels, 0 = none), n_out, H_in, W_in, elu, residual
    (32, 32, 160, 160, 8, 8, True, True),                 # down_deconv2 of the BASELINE run's downsampling layer
    (5, 64, 0, 32, 4, 4, False, False),
    (3, 32, 32, 64, 8, 8, True, True),
    (2, 32, 160, 160, 1, 1, True, True),
    (4, 64, 0, 64, 3, 5, False, True),
]


@pytest.mark.parametrize("case", DECONV_CASES, ids=lambda c: "B%d_%d+%dto%d_%dx%d" % c[:6])
def test_deconv_by_phases_vs_fp64_definition_and_vs_the_zero_inserted_conv(amd, case):
    B, c1, c2, co, H, W, elu, with_res = case
    ci = c1 + c2
    rng = np.random.RandomState(200 + B)
    p = gi.deconv_params(rng, ci, co)
    x = rng.standard_normal((B, ci, H, W))
    res = rng.standard_normal((B, co, H, W)) if with_res else None
    conv = amd.WNConv2d(ci, co)
    conv.prepare_deconv(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    xa, xb = dev(x[:, :c1]), (dev(x[:, c1:]) if c2 else None)
    out = torch.empty((B, co, 2 * H, 2 * W), device="cuda")
    P = amd.layers._ptr
    resd = dev(res) if with_res else None
    rc = amd._capi.lib().iaf_conv3x3_forward_deconv(conv._h, P(xa), P(xb), c1 if c2 else 0, 1 if elu else 0,
                                                    P(resd), P(out), B, H, W, amd.layers._stream())
    assert rc == 0, "the phase kernel covers this shape (rc %d)" % rc
    want = ref_deconv(x, p, elu)
    if with_res:
        want = np.repeat(np.repeat(f32(res), 2, axis=2), 2, axis=3) + 0.1 * want
    np.testing.assert_allclose(host(out), want, rtol=0, atol=ATOL)
    # the zero-inserted formulation it replaces
    old = conv(amd.resample2(xa, "up_zero_odd"), x2=amd.resample2(xb, "up_zero_odd") if c2 else None, elu_input=elu,
               residual=amd.resample2(dev(res), "up_nearest") if with_res else None)[0]
    np.testing.assert_allclose(host(out), host(old), rtol=0, atol=ATOL)
    via = conv.deconv(xa, x2=xb, elu_input=elu, residual=dev(res) if with_res else None)
    assert torch.equal(via, out)


def test_deconv_falls_back_for_channel_counts_without_a_bf16x3_pack(amd):
    """c_in = 48 (not a multiple of 32): no bf16x3 pack -> IAF_ERR_UNSUPPORTED, WNConv2d.deconv runs the zero-inserted form"""
    B, ci, co, H, W = 2, 48, 32, 4, 4
    rng = np.random.RandomState(9)
    p = gi.deconv_params(rng, ci, co)
    x = rng.standard_normal((B, ci, H, W))
    conv = amd.WNConv2d(ci, co)
    conv.prepare_deconv(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    out = torch.empty((B, co, 2 * H, 2 * W), device="cuda")
    P = amd.layers._ptr
    xd = dev(x)
    rc = amd._capi.lib().iaf_conv3x3_forward_deconv(conv._h, P(xd), None, 0, 0, None, P(out), B, H, W, amd.layers._stream())
    assert rc == amd._capi.IAF_ERR_UNSUPPORTED
    np.testing.assert_allclose(host(conv.deconv(dev(x))), ref_deconv(x, p, False), rtol=0, atol=ATOL)
