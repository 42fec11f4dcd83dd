"""Structural properties the reference's math implies checked on the oracle:
autoregressive Jacobian, log-det identity, analytic KL, direct-loop conv.  CPU only."""
import numpy as np
import pytest

import golden_inputs as gi
from oracle import iaf_oracle as O


def _jacobian(f, z, eps=1e-6):
    z = z.copy()
    n = z.size
    J = np.zeros((n, n))
    for i in range(n):
        zp = z.copy().reshape(-1)
        zm = z.copy().reshape(-1)
        zp[i] += eps
        zm[i] -= eps
        J[:, i] = (f(zp.reshape(z.shape)).reshape(-1) - f(zm.reshape(z.shape)).reshape(-1)) / (2 * eps)
    return J


@pytest.mark.parametrize("n_z,n_h", [(4, [8, 8]), (4, [8]), (8, [4, 4])])
def test_tf_variant_jacobian_triangular_and_logdet(n_z, n_h):
    rng = np.random.RandomState(11)
    H = W = 4
    params = gi.ar_multiconv2d_params(rng, n_z, n_h, [n_z, n_z])
    z = rng.standard_normal((1, n_z, H, W))
    ctx = rng.standard_normal((1, n_h[0], H, W))
    J = _jacobian(lambda zz: O.iaf_step(zz, ctx, params, n_h)[0], z)
    # ordering under which the TF statement is autoregressive: reverse-raster pixel, ascending channel
    # (cross-correlation + mask => output depends on pixels right/below and on lower channels at the centre)
    order = []
    for pix in reversed(range(H * W)):
        for c in range(n_z):
            order.append(c * H * W + pix)
    Jo = J[np.ix_(order, order)]
    assert np.abs(np.triu(Jo, 1)).max() < 1e-8
    _, s = O.iaf_step(z, ctx, params, n_h)
    sign, logdet = np.linalg.slogdet(J)
    assert sign > 0
    assert abs(logdet - (-s.sum())) < 1e-6       # log|dz'/dz| = -sum s  <=>  logqs += s (tf_train.py:72)


def test_theano_variant_jacobian_triangular_and_logdet():
    rng = np.random.RandomState(12)
    n_z, n_h, H, W = 4, [8, 8], 4, 4
    w = {}
    sizes = [n_z] + n_h
    for i in range(len(n_h)):
        w["p_%d_w" % i] = 0.05 * rng.standard_normal((sizes[i + 1], sizes[i] + 1, 3, 3))
        w["p_%d_b" % i] = 0.1 * rng.standard_normal(sizes[i + 1])
        w["p_%d_s" % i] = 0.1 * rng.standard_normal(sizes[i + 1])
    for i in range(2):
        w["p_out_%d_w" % i] = 0.05 * rng.standard_normal((n_z, n_h[-1] + 1, 3, 3))
        w["p_out_%d_b" % i] = 0.1 * rng.standard_normal(n_z)
        w["p_out_%d_s" % i] = 0.1 * rng.standard_normal(n_z)
    z = rng.standard_normal((1, n_z, H, W))
    ctx = rng.standard_normal((1, n_h[0], H, W))
    J = _jacobian(lambda zz: O.theano_iaf2_nl(zz, ctx, w, "p", n_z, n_h)[0], z)
    # flipped kernel => depends on pixels left/above: raster pixel order, ascending channel
    order = [c * H * W + pix for pix in range(H * W) for c in range(n_z)]
    Jo = J[np.ix_(order, order)]
    assert np.abs(np.triu(Jo, 1)).max() < 1e-8
    _, s = O.theano_iaf2_nl(z, ctx, w, "p", n_z, n_h)
    sign, logdet = np.linalg.slogdet(J)
    assert sign > 0 and abs(logdet + s.sum()) < 1e-6


def test_kl_is_analytic_gaussian_kl_when_ar_outputs_are_zero():
    """With the output convs producing 0 (V=0 -> w=0, b=0) the flow is the identity, so
    E_q[logqs - logps] must approach the closed-form diagonal-Gaussian KL; checked per element in
    expectation over many eps via the exact identity for a single eps: logq(z) - logp(z)."""
    rng = np.random.RandomState(13)
    B, Z, Hh, H, W = 2, 4, 8, 3, 3
    params = gi.ar_multiconv2d_params(rng, Z, [Hh, Hh], [Z, Z])
    for i in range(2):
        params["layer_out_%d/V" % i] = np.zeros_like(params["layer_out_%d/V" % i])
        params["layer_out_%d/b" % i] = np.zeros_like(params["layer_out_%d/b" % i])
    f = lambda *s: rng.standard_normal(s)
    qm, ql, rm, rl, pm, pl = f(B, Z, H, W), 0.2 * f(B, Z, H, W), f(B, Z, H, W), 0.2 * f(B, Z, H, W), f(B, Z, H, W), 0.2 * f(B, Z, H, W)
    uc, dc, eps = f(B, Hh, H, W), f(B, Hh, H, W), f(B, Z, H, W)
    blk = O.posterior_block(qm, ql, rm, rl, pm, pl, uc, dc, eps, params, [Hh, Hh], 0.0)
    np.testing.assert_allclose(blk["z"], blk["z0"], atol=1e-14)
    mu, lv = qm + rm, 2 * (ql + rl)
    expect = (-0.5 * (O.LOG2PI + lv + eps ** 2)) - O.gaussian_diag_logps(pm, 2 * pl, blk["z0"])
    np.testing.assert_allclose(blk["logqs"] - blk["logps"], expect, atol=1e-12)
    np.testing.assert_allclose(blk["kl_cost"], expect.sum(axis=(1, 2, 3)), atol=1e-10)
    np.testing.assert_allclose(blk["kl_obj"], blk["kl_cost"], atol=1e-12)        # kl_min == 0


def test_free_bits_semantics():
    rng = np.random.RandomState(14)
    B, Z, Hh, H, W = 3, 4, 8, 3, 3
    params = gi.ar_multiconv2d_params(rng, Z, [Hh, Hh], [Z, Z])
    f = lambda *s: rng.standard_normal(s)
    args = [f(B, Z, H, W), 0.2 * f(B, Z, H, W), f(B, Z, H, W), 0.2 * f(B, Z, H, W), f(B, Z, H, W), 0.2 * f(B, Z, H, W),
            f(B, Hh, H, W), f(B, Hh, H, W), f(B, Z, H, W)]
    big = O.posterior_block(*args, params, [Hh, Hh], 1e9)
    np.testing.assert_allclose(big["kl_obj"], np.full(B, Z * 1e9))         # every channel clamped (tf_train.py:79-82)
    none = O.posterior_block(*args, params, [Hh, Hh], 0.0)
    np.testing.assert_allclose(none["kl_obj"], none["kl_cost"])
    mid = O.posterior_block(*args, params, [Hh, Hh], 0.25)
    kl = mid["logqs"] - mid["logps"]
    per_c = kl.sum(axis=(2, 3)).mean(axis=0)
    np.testing.assert_allclose(mid["kl_obj"], np.full(B, np.maximum(per_c, 0.25).sum()))


def test_masked_conv_equals_direct_loops():
    """ar_conv2d vs an O(n^2) direct-loop statement of 'SAME cross-correlation with masked,
    weight-normed filter' on a tiny shape."""
    rng = np.random.RandomState(15)
    B, ci, co, H, W = 2, 4, 8, 3, 4
    p = gi.conv_params(rng, ci, co)
    x = rng.standard_normal((B, ci, H, W))
    y = O.ar_conv2d(x, p["V"], p["g"], p["b"], zerodiagonal=False)
    mask = O.get_conv_ar_mask(3, 3, ci, co, False)
    v = mask * p["V"]
    ref = np.zeros((B, co, H, W))
    for o in range(co):
        nrm = np.sqrt(max((v[:, :, :, o] ** 2).sum(), 1e-12))
        for n in range(B):
            for i in range(H):
                for j in range(W):
                    acc = 0.0
                    for a in range(3):
                        for b in range(3):
                            ii, jj = i + a - 1, j + b - 1
                            if 0 <= ii < H and 0 <= jj < W:
                                for c in range(ci):
                                    acc += x[n, c, ii, jj] * np.exp(p["g"][o]) * v[a, b, c, o] / nrm
                    ref[n, o, i, j] = acc + p["b"][o]
    np.testing.assert_allclose(y, ref, atol=1e-12)
