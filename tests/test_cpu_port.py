"""The torch-CPU port timed as bench.py's cpu_baseline must agree with the oracle. CPU only."""
import numpy as np
import torch

import golden_inputs as gi
from oracle import iaf_cpu_port as Pt
from oracle import iaf_oracle as O


def test_cpu_port_matches_oracle():
    rng = np.random.RandomState(4)
    n_z, n_h, B, H, W = 32, [64, 64], 2, 6, 5
    params = gi.ar_multiconv2d_params(rng, n_z, n_h, [n_z, n_z])
    z = rng.standard_normal((B, n_z, H, W)).astype(np.float32)
    ctx = rng.standard_normal((B, n_h[0], H, W)).astype(np.float32)
    tp = Pt.as_torch(params)
    w = Pt.prepare_weights(tp, n_z, n_h)
    zn, s = Pt.iaf_step(torch.from_numpy(z), torch.from_numpy(ctx), w, len(n_h))
    p32 = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in params.items()}
    ez, es = O.iaf_step(z.astype(np.float64), ctx.astype(np.float64), p32, n_h)
    np.testing.assert_allclose(zn.numpy(), ez, atol=1e-4)
    np.testing.assert_allclose(s.numpy(), es, atol=1e-4)


# ---------------------------------------------------------------- the conv PRIMITIVE, against an implementation that is not ours
# the golden fixtures pin the reference's control flow on top of a conv leaf written for the shim,
# and the oracle's conv is the same algorithm.  torch's convs (oneDNN) are an independent implementation of the same
# primitives; the TF "SAME" rule (out = ceil(n/s), pad_total = max((out-1)s + k - n, 0), extra padding at the END) is
# applied here explicitly, as torch has no such mode for strides > 1.
def _tf_same_torch(x, w_hwio, stride):
    import torch.nn.functional as F
    kh, kw = w_hwio.shape[:2]
    n, c, hh, ww = x.shape
    pads = []
    for size, k, s in ((ww, kw, stride[1]), (hh, kh, stride[0])):          # F.pad order: last dim first
        out = -(-size // s)
        tot = max((out - 1) * s + k - size, 0)
        pads += [tot // 2, tot - tot // 2]
    xp = F.pad(torch.from_numpy(x), pads)
    return F.conv2d(xp, torch.from_numpy(w_hwio).permute(3, 2, 0, 1).contiguous(), stride=stride).numpy()


import pytest  # noqa: E402


@pytest.mark.parametrize("shape", [(2, 5, 7, 6, 5, 3, (1, 1)), (1, 3, 4, 1, 1, 3, (1, 1)), (2, 4, 6, 8, 8, 3, (2, 2)), (1, 3, 5, 7, 5, 3, (2, 2)),
                                   (2, 3, 4, 6, 6, 5, (2, 2)), (1, 2, 3, 2, 9, 3, (1, 1)), (1, 8, 16, 16, 16, 3, (2, 2))],
                         ids=lambda s: "B%d_%d_%d_%dx%d_k%d_s%d" % (s[0], s[1], s[2], s[3], s[4], s[5], s[6][0]))
def test_conv_primitive_oracle_and_shim_vs_torch(shape):
    """SAME-padded NCHW cross-correlation, strides 1 and 2, odd and even sizes, 1x1 images, 5x5 filters (x_enc's shape)"""
    import tf_shim
    B, ci, co, H, W, k, stride = shape
    rng = np.random.RandomState(21)
    x, w = rng.standard_normal((B, ci, H, W)), rng.standard_normal((k, k, ci, co))
    ref = _tf_same_torch(x, w, stride)
    np.testing.assert_allclose(O.conv2d_same_nchw(x, w, stride), ref, rtol=1e-11, atol=1e-11)
    y = tf_shim.nn_conv2d(x, w, [1, 1, stride[0], stride[1]], "SAME", data_format="NCHW")
    np.testing.assert_allclose(np.asarray(y), ref, rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("shape", [(2, 6, 4, 5, 5), (1, 12, 8, 4, 4), (2, 3, 5, 1, 3), (1, 16, 16, 8, 8)], ids=lambda s: "B%d_%d_%d_%dx%d" % s)
def test_deconv_primitive_oracle_and_shim_vs_torch(shape):
    """conv2d_transpose(SAME, stride 2, 3x3): torch's conv_transpose2d produces the full (2H+1) result; TF's SAME
    output is its first 2H rows / columns (the forward conv pads at the end only)"""
    import torch.nn.functional as F
    import tf_shim
    B, ci, co, H, W = shape
    rng = np.random.RandomState(22)
    x, V = rng.standard_normal((B, ci, H, W)), rng.standard_normal((3, 3, co, ci))
    g, b = 0.1 * rng.standard_normal(co), 0.1 * rng.standard_normal(co)
    wn = np.exp(g).reshape(1, 1, co, 1) * V / np.sqrt(np.maximum((V ** 2).sum(axis=(0, 1, 2), keepdims=True), 1e-12))   # layers.py:104
    full = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(wn).permute(3, 2, 0, 1).contiguous(), stride=2).numpy()
    ref = full[:, :, :2 * H, :2 * W] + b.reshape(1, -1, 1, 1)
    np.testing.assert_allclose(O.deconv2d(x, V, g, b), ref, rtol=1e-10, atol=1e-10)
    y = tf_shim.nn_conv2d_transpose(np.transpose(x, (0, 2, 3, 1)), wn, [B, 2 * H, 2 * W, co], [1, 2, 2, 1], "SAME")
    np.testing.assert_allclose(np.transpose(np.asarray(y), (0, 3, 1, 2)) + b.reshape(1, -1, 1, 1), ref, rtol=1e-10, atol=1e-10)


def test_theano_conv_primitive_vs_torch():
    """the Theano leaf: dnn_conv conv_mode='conv' ('valid' true convolution, OIHW) == torch cross-correlation with the
    kernel rotated by 180 degrees"""
    import torch.nn.functional as F
    import theano_shim
    rng = np.random.RandomState(23)
    x, k = rng.standard_normal((2, 5, 7, 6)), rng.standard_normal((4, 5, 3, 3))
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(k[:, :, ::-1, ::-1].copy())).numpy()
    y = theano_shim.dnn_conv(theano_shim.TT(x), theano_shim.TT(k), border_mode="valid")
    np.testing.assert_allclose(y.a, ref, rtol=1e-11, atol=1e-11)
