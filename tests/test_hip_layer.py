"""GPU parity tests for the widened path: the plain weight-normed 3x3 convs around the IAF step
(tf_utils/layers.py:31-64) and the whole non-downsampling IAFLayer.up / .down (tf_train.py:29-95) on the GPU, against
the committed reference outputs (tests/golden/iaf_layer.npz) and the CPU oracle.  Tolerance as in test_hip_parity."""
import os

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


# ---------------------------------------------------------------- whole IAFLayer vs the reference's outputs
@pytest.mark.parametrize("name", sorted(gi.LAYER_CASES))
def test_iaf_layer_up_down_vs_reference_golden(amd, golden_dir, name):
    """IAFLayer.up then .down (tf_train.py:29-95, mode "train") with every op on the GPU; all seven tensors the
    reference run produced.  The tiny fixtures (z 4, h 8) run on the direct-conv fallback, cfg2 on the MFMA kernels."""
    g = np.load(os.path.join(golden_dir, "iaf_layer.npz"))
    c = gi.layer_case_inputs(name)
    layer = amd.IAFLayer(c["z_size"], c["h_size"], depth_ar=2, kl_min=c["kl_min"])
    layer.load({k: dev(v) for k, v in c["params"].items()})
    up_out = layer.up(dev(c["up_input"]))
    np.testing.assert_allclose(host(up_out), g[name + "/up_out"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(layer.posterior.qz_mean), g[name + "/qz_mean"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(layer.posterior.qz_logsd), g[name + "/qz_logsd"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(layer.posterior.up_context), g[name + "/up_context"], atol=ATOL, rtol=0)
    out, kl_obj, kl_cost = layer.down(dev(c["down_input"]), dev(c["eps_post"]))
    np.testing.assert_allclose(host(out), g[name + "/output"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(kl_obj), g[name + "/kl_obj"], atol=2e-3, rtol=1e-4)
    np.testing.assert_allclose(host(kl_cost), g[name + "/kl_cost"], atol=2e-3, rtol=1e-4)


# ---------------------------------------------------------------- the conv operator vs the oracle
CONV_SHAPES = [
    # (B, n_in, n_out, H, W)
    (2, 160, 384, 8, 8),      # up_conv1 at cfg2
    (3, 160, 448, 16, 16),    # down_conv1
    (2, 192, 160, 8, 8),      # down_conv2
    (1, 16, 16, 1, 1),        # a single pixel: every non-centre tap falls outside
    (2, 32, 48, 5, 7),        # ragged: tiles straddle rows and images
    (1, 64, 64, 3, 40),       # wide rows
    (2, 16, 32, 32, 32),
    (3, 6, 10, 4, 5),         # fallback path
]


@pytest.mark.parametrize("shape", CONV_SHAPES, ids=lambda s: "B%d_%dto%d_%dx%d" % s)
def test_wnconv2d_vs_oracle(amd, shape):
    B, n_in, n_out, H, W = shape
    rng = np.random.RandomState(1234 + n_in + n_out)
    p = gi.conv_params(rng, n_in, n_out)
    x = rng.standard_normal((B, n_in, H, W))
    conv = amd.WNConv2d(n_in, n_out)
    conv.prepare(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    y = conv(dev(x))[0]
    e = O.conv2d(f32(x), f32(p["V"]), f32(p["g"]), f32(p["b"]))
    np.testing.assert_allclose(host(y), e, atol=ATOL, rtol=0)
    # fused ELU on the input + residual
    res = rng.standard_normal((B, n_out, H, W))
    y2 = conv(dev(x), elu_input=True, residual=dev(res))[0]
    e2 = f32(res) + 0.1 * O.conv2d(O.elu(f32(x)), f32(p["V"]), f32(p["g"]), f32(p["b"]))
    np.testing.assert_allclose(host(y2), e2, atol=ATOL, rtol=0)


@pytest.mark.parametrize("shape", [(2, 32, 160, 448, 8, 8), (3, 4, 8, 24, 6, 6), (2, 64, 64, 320, 5, 3)],
                         ids=lambda s: "B%d_z%d_h%d_out%d_%dx%d" % s)
def test_wnconv2d_concat_and_split(amd, shape):
    """input = elu(concat(z, h_det)) (tf_train.py:87-88); output split six ways (tf_train.py:54)"""
    B, zs, hs, n_out, H, W = shape
    rng = np.random.RandomState(99)
    p = gi.conv_params(rng, zs + hs, n_out)
    a, b = rng.standard_normal((B, zs, H, W)), rng.standard_normal((B, hs, H, W))
    split = [zs] * 4 + [(n_out - 4 * zs) // 2] * 2
    conv = amd.WNConv2d(zs + hs, n_out)
    conv.prepare(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    outs = conv(dev(a), x2=dev(b), elu_input=True, split=split)
    e = O.conv2d(O.elu(np.concatenate([f32(a), f32(b)], axis=1)), f32(p["V"]), f32(p["g"]), f32(p["b"]))
    for got, want in zip(outs, O.split_channels(e, split)):
        np.testing.assert_allclose(host(got), want, atol=ATOL, rtol=0)


TUNES = [(5, 4, 1, 1), (2, 4, 1, 1), (1, 4, 1, 1), (5, 2, 2, 1), (5, 4, 1, 2), (5, 2, 2, 2), (2, 2, 1, 2), (1, 1, 2, 2),
         (2, 2, 1, 4), (1, 1, 1, 4), (4, 4, 1, 1), (3, 4, 1, 1)]


@pytest.mark.parametrize("tune", TUNES, ids=lambda t: "nt%d_px%d_wco%d_ks%d" % t)
def test_wnconv2d_every_launch_shape(amd, tune):
    B, n_in, H, W = 3, 160, 8, 8
    nt, pxt, wco, ks = tune
    n_out = 16 * nt * wco * 2
    rng = np.random.RandomState(5)
    p = gi.conv_params(rng, n_in, n_out)
    x = rng.standard_normal((B, n_in, H, W))
    conv = amd.WNConv2d(n_in, n_out)
    conv.prepare(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    try:
        conv.set_tuning(nt, pxt, wco, ks)
        y = conv(dev(x))[0]
    except amd.UnsupportedError as e:              # ONLY "not covered" (e.g. more than 160 KiB of LDS) skips; any other error fails
        pytest.skip(str(e))
    e = O.conv2d(f32(x), f32(p["V"]), f32(p["g"]), f32(p["b"]))
    np.testing.assert_allclose(host(y), e, atol=ATOL, rtol=0)


BF3P = [(nt, ppw, wco, 4) for (ppw, wco) in ((2, 1), (4, 1), (2, 2)) for nt in (5, 4, 2)] + [(4, 2, 3, 4), (2, 2, 3, 4)]


@pytest.mark.parametrize("shp", BF3P, ids=lambda t: "nt%d_ppw%d_wco%d_ks%d" % t)
@pytest.mark.parametrize("case", [(3, 160, 320, 8, 8), (32, 160, 160, 16, 16), (2, 32, 40 * 16, 5, 7), (8, 160, 384, 16, 16), (8, 160, 448, 16, 16)], ids=lambda s: "B%d_%dto%d_%dx%d" % s)
def test_plain_conv_on_the_bf16_matrix_cores_every_shape(amd, shp, case):
    """the 9-tap plain conv as bf16x3 split products (iaf_conv_bf3.hpp, NTP = 9, halo on both sides of the pixel tile): every
    compiled launch shape, with the fused ELU / residual, against the oracle -- and fp32-grade, not just inside the tolerance"""
    B, n_in, n_out, H, W = case
    nt, ppw, wco, ks = shp
    tiles, per_wg = n_out // 16, nt * wco
    covered = -(-tiles // per_wg) * per_wg
    if (covered - tiles) * 4 > tiles:          # (ragged co groups, round 6: the last workgroup's surplus tiles are computed and dropped -- up to a quarter)
        pytest.skip("this co tiling wastes more than a quarter of %d output tiles" % tiles)
    rng = np.random.RandomState(31 + nt + ppw)
    p = gi.conv_params(rng, n_in, n_out)
    x, res = rng.standard_normal((B, n_in, H, W)), rng.standard_normal((B, n_out, H, W))
    conv = amd.WNConv2d(n_in, n_out)
    conv.prepare(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    try:
        conv.set_tuning(-nt, ppw, wco, ks)
        assert conv.runs_bf16x3(B, H, W)
        y = conv(dev(x))[0]
        y2 = conv(dev(x), elu_input=True, residual=dev(res))[0]
    except amd.UnsupportedError as e:
        pytest.skip(str(e))
    e = O.conv2d(f32(x), f32(p["V"]), f32(p["g"]), f32(p["b"]))
    np.testing.assert_allclose(host(y), e, atol=ATOL, rtol=0)
    assert np.abs(host(y) - e).max() < 2e-5 * max(1.0, np.abs(e).max())
    e2 = f32(res) + 0.1 * O.conv2d(O.elu(f32(x)), f32(p["V"]), f32(p["g"]), f32(p["b"]))
    np.testing.assert_allclose(host(y2), e2, atol=ATOL, rtol=0)
    conv.set_precision("f32")
    assert not conv.runs_bf16x3(B, H, W)
    yf = conv(dev(x))[0]
    assert float((yf - y).abs().max()) < 3e-5 * max(1.0, np.abs(e).max())


def test_plain_conv_bf16x3_concat_split_and_size_rule(amd):
    """down_conv2 / down_conv1 shapes through the size rule (bf16x3 from 4096 pixels on): concat of two inputs + ELU, output
    split six ways; below the threshold the exact-fp32 kernel runs"""
    rng = np.random.RandomState(77)
    B, zs, hs, H, W = 32, 32, 160, 16, 16
    p = gi.conv_params(rng, zs + hs, hs)
    a, b = rng.standard_normal((B, zs, H, W)), rng.standard_normal((B, hs, H, W))
    conv = amd.WNConv2d(zs + hs, hs)
    conv.prepare(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    assert conv.runs_bf16x3(B, H, W) and not conv.runs_bf16x3(2, 8, 8)
    y = conv(dev(a), x2=dev(b), elu_input=True)[0]
    chunk = 8
    for b0 in range(0, B, chunk):
        e = O.conv2d(O.elu(np.concatenate([f32(a[b0:b0 + chunk]), f32(b[b0:b0 + chunk])], axis=1)), f32(p["V"]), f32(p["g"]), f32(p["b"]))
        np.testing.assert_allclose(host(y[b0:b0 + chunk]), e, atol=ATOL, rtol=0)
    n_out = 4 * zs + 2 * hs
    p = gi.conv_params(rng, hs, n_out)
    x = rng.standard_normal((B, hs, H, W))
    conv = amd.WNConv2d(hs, n_out)
    conv.prepare(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    assert conv.runs_bf16x3(B, H, W)
    split = [zs] * 4 + [hs] * 2
    outs = conv(dev(x), elu_input=True, split=split)
    e = O.conv2d(O.elu(f32(x[:4])), f32(p["V"]), f32(p["g"]), f32(p["b"]))
    for got, want in zip(outs, O.split_channels(e, split)):
        np.testing.assert_allclose(host(got[:4]), want, atol=ATOL, rtol=0)


def test_conv2d_function_api_under_tf_names(amd, golden_dir):
    """the reference call site tf_train.py:36 under its variable scope names"""
    c = gi.layer_case_inputs("layer_cfg2_8x8")
    g = np.load(os.path.join(golden_dir, "iaf_layer.npz"))
    store = amd.VariableStore()
    for k, v in c["params"].items():
        store.set("model/IAF_0_0/" + k, dev(v))
    zs, hs = c["z_size"], c["h_size"]
    with amd.variable_scope("model", store), amd.variable_scope("IAF_0_0", store):
        x = amd.conv2d("up_conv1", torch.nn.functional.elu(dev(c["up_input"])), 2 * zs + 2 * hs, store=store)
    qz_mean, qz_logsd, up_context, _ = torch.split(x, [zs, zs, hs, hs], dim=1)
    np.testing.assert_allclose(host(qz_mean), g["layer_cfg2_8x8/qz_mean"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(up_context), g["layer_cfg2_8x8/up_context"], atol=ATOL, rtol=0)


def test_full_size_layer_properties(amd):
    """BASELINE configs[1] sizes (B 16, z 32, h 160, 16x16): the layer is a residual map, so zeroing g's exp (g -> -inf is
    not representable; use V = 0 on down_conv2) must return the input exactly, and batch entries are independent when
    free bits are off."""
    B, zs, hs, H, W = 16, 32, 160, 16, 16
    c = gi.layer_case_inputs("layer_cfg2_8x8")
    rng = np.random.RandomState(3)
    params = {k: dev(v) for k, v in c["params"].items()}
    layer = amd.IAFLayer(zs, hs, depth_ar=2, kl_min=0.0)
    layer.load(params)
    up_in, down_in = rng.standard_normal((B, hs, H, W)), rng.standard_normal((B, hs, H, W))
    eps = rng.standard_normal((B, zs, H, W))
    layer.up(dev(up_in))
    out, kl_obj, kl_cost = layer.down(dev(down_in), dev(eps))
    # With these weights |z| behind the IAF step reaches 1e7: beyond fp16's range, which down_conv2's default arithmetic (two fp16
    # planes, round 6) says out loud -- its outputs are not finite, its range word is up, the conv's next call raises RangeError ONCE and
    # it computes on bf16 planes (fp32's exponent range) from then on (include/iaf_hip.h, iaf_conv3x3_range_errors); the LAYER repeats
    # that call behind a RuntimeWarning
    assert float(layer.last_block["z"].abs().max()) > 65504.0
    assert layer.down_conv2.range_errors() & 1 and not torch.isfinite(out).all()
    with pytest.warns(RuntimeWarning, match="repeated on bf16 planes"):
        out, kl_obj, kl_cost = layer.down(dev(down_in), dev(eps))
    assert torch.isfinite(out).all() and not layer.down_conv2.runs_f16x2(B, H, W)
    # batch independence: the first 4 entries alone give the same values
    layer.up(dev(up_in[:4]))
    out4, kl_obj4, kl_cost4 = layer.down(dev(down_in[:4]), dev(eps[:4]))
    np.testing.assert_allclose(host(out4), host(out)[:4], atol=2e-5, rtol=0)
    np.testing.assert_allclose(host(kl_cost4), host(kl_cost)[:4], atol=2e-3, rtol=1e-5)
    np.testing.assert_allclose(host(kl_obj), host(kl_cost), atol=2e-3, rtol=1e-5)       # kl_min = 0 (tf_train.py:84-85)
    # b = 0 and V = 0 on the last conv: l2_normalize(0) = 0 (layers.py:60), so output == input bit for bit
    params["down_conv2/V"] = torch.zeros_like(params["down_conv2/V"])
    params["down_conv2/b"] = torch.zeros_like(params["down_conv2/b"])
    layer.load(params)
    layer.up(dev(up_in))
    out0, _, _ = layer.down(dev(down_in), dev(eps))
    assert torch.equal(out0, dev(down_in))


def test_conv3x3_argument_errors(amd):
    conv = amd.WNConv2d(32, 32)
    x = torch.zeros((1, 32, 4, 4), device="cuda")
    with pytest.raises(amd._capi.IafHipError):      # not prepared
        conv(x)
    V, g, b = torch.zeros((3, 3, 32, 32), device="cuda"), torch.zeros(32, device="cuda"), torch.zeros(32, device="cuda")
    conv.prepare(V, g, b)
    with pytest.raises(ValueError):
        conv(torch.zeros((1, 16, 4, 4), device="cuda"))
    with pytest.raises(ValueError):
        conv(x, split=[16, 8])
    with pytest.raises(ValueError):                 # MFMA path: split points must be multiples of 4
        conv(x, split=[6, 26])
    with pytest.raises(ValueError):
        conv.prepare(torch.zeros((3, 3, 32, 16), device="cuda"), g, b)


def test_autotuned_layer_matches_reference_golden(amd, golden_dir):
    """the launch-shape search must not change results: cfg2 fixture again with autotune on, then a plain call"""
    name = "layer_cfg2_8x8"
    g = np.load(os.path.join(golden_dir, "iaf_layer.npz"))
    c = gi.layer_case_inputs(name)
    layer = amd.IAFLayer(c["z_size"], c["h_size"], depth_ar=2, kl_min=c["kl_min"])
    layer.load({k: dev(v) for k, v in c["params"].items()})
    for tune in (True, False):
        up_out = layer.up(dev(c["up_input"]), autotune=tune)
        out, kl_obj, kl_cost = layer.down(dev(c["down_input"]), dev(c["eps_post"]), autotune=tune)
        np.testing.assert_allclose(host(up_out), g[name + "/up_out"], atol=ATOL, rtol=0)
        np.testing.assert_allclose(host(out), g[name + "/output"], atol=ATOL, rtol=0)
        np.testing.assert_allclose(host(kl_cost), g[name + "/kl_cost"], atol=2e-3, rtol=1e-4)
    assert layer.up_conv1._tuned and layer.down_conv2._tuned


# ---------------------------------------------------------------- ar_conv2d on its own, data-dependent init, likelihood
@pytest.mark.parametrize("shape", [(2, 32, 160, False, 8, 8), (2, 160, 32, True, 8, 8), (3, 64, 64, True, 5, 7),
                                   (3, 64, 64, False, 5, 7), (2, 4, 8, False, 6, 6), (2, 8, 4, True, 6, 6)],
                         ids=lambda s: "B%d_%dto%d_zd%d_%dx%d" % s)
def test_ar_conv2d_single_vs_oracle(amd, shape):
    """one masked conv through the operator API (layers.py:144-154), both mask variants, MFMA and fallback paths"""
    B, n_in, n_out, zd, H, W = shape
    rng = np.random.RandomState(77)
    p = gi.conv_params(rng, n_in, n_out)
    x = rng.standard_normal((B, n_in, H, W))
    store = amd.VariableStore()
    for k, v in p.items():
        store.set("m/c/" + k, dev(v))
    with amd.variable_scope("m", store):
        y = amd.ar_conv2d("c", dev(x), n_out, zerodiagonal=zd, store=store)
    e = O.ar_conv2d(f32(x), f32(p["V"]), f32(p["g"]), f32(p["b"]), zerodiagonal=zd)
    np.testing.assert_allclose(host(y), e, atol=ATOL, rtol=0)


def test_data_dependent_init_vs_reference_golden(amd, golden_dir):
    """the init=True branch (layers.py:38-51) on the reference's own output: ar_conv2d("c", x, 16, zerodiagonal=False,
    init_scale=0.7) under arg_scope(init=True)"""
    g = np.load(os.path.join(golden_dir, "init_ar_conv.npz"))
    store = amd.VariableStore()
    store.set("c/V", dev(g["V0"]))
    y = amd.ar_conv2d("c", dev(g["x"]), 16, zerodiagonal=False, init_scale=0.7, init=True, store=store)
    np.testing.assert_allclose(host(y), g["y"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(store.vars["c/g"]), g["g"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(host(store.vars["c/b"]), g["b"], atol=1e-5, rtol=0)
    # the non-init branch with the variables just created (the reference's quirk: g = log(scale)/3 but w = exp(g)*...)
    y2 = amd.ar_conv2d("c", dev(g["x"]), 16, zerodiagonal=False, store=store)
    e2 = O.ar_conv2d(f32(g["x"]), f32(g["V0"]), g["g"], g["b"], zerodiagonal=False)
    np.testing.assert_allclose(host(y2), e2, atol=ATOL, rtol=0)


@pytest.mark.parametrize("shape", [(4, 32, 160, 8, 8), (3, 4, 8, 6, 6)], ids=lambda s: "B%d_z%d_h%d_%dx%d" % s)
def test_ar_multiconv2d_init_vs_oracle(amd, shape):
    """ar_multiconv2d with init=True arriving via arg_scope (tf_train.py:175, layers.py:160): layer-by-layer
    data-dependent init; the oracle composes layers.py:38-51 the way layers.py:161-166 chains it"""
    B, zs, hs, H, W = shape
    rng = np.random.RandomState(31)
    z, ctx = rng.standard_normal((B, zs, H, W)), rng.standard_normal((B, hs, H, W))
    V = {"layer_0": 0.05 * rng.standard_normal((3, 3, zs, hs)), "layer_1": 0.05 * rng.standard_normal((3, 3, hs, hs)),
         "layer_out_0": 0.05 * rng.standard_normal((3, 3, hs, zs)), "layer_out_1": 0.05 * rng.standard_normal((3, 3, hs, zs))}
    store = amd.VariableStore()
    for k, v in V.items():
        store.set("s/" + k + "/V", dev(v))
    outs = amd.ar_multiconv2d("s", dev(z), dev(ctx), [hs, hs], [zs, zs], store=store, init=True)
    h = f32(z)
    exp_gb = {}
    for i, nm in enumerate(["layer_0", "layer_1"]):
        mask = O.get_conv_ar_mask(3, 3, h.shape[1], hs, False)
        y, g_, b_ = O.conv2d_init(h, f32(V[nm]), init_scale=1.0, mask=mask)
        exp_gb[nm] = (g_, b_)
        if i == 0:
            y = y + f32(ctx)
        h = O.elu(y)
    for i, nm in enumerate(["layer_out_0", "layer_out_1"]):
        mask = O.get_conv_ar_mask(3, 3, hs, zs, True)
        y, g_, b_ = O.conv2d_init(h, f32(V[nm]), init_scale=1.0, mask=mask)
        exp_gb[nm] = (g_, b_)
        np.testing.assert_allclose(host(outs[i]), y, atol=2e-4, rtol=0)      # unit-variance outputs, 3 normalisations deep
    for nm, (g_, b_) in exp_gb.items():
        np.testing.assert_allclose(host(store.vars["s/" + nm + "/g"]), g_, atol=2e-5, rtol=0)
        np.testing.assert_allclose(host(store.vars["s/" + nm + "/b"]), b_, atol=2e-4, rtol=0)
    # the initialised stack then runs through the fused engine path and reproduces the init-mode outputs
    again = amd.ar_multiconv2d("s", dev(z), dev(ctx), [hs, hs], [zs, zs], store=store)
    exp = O.ar_multiconv2d(f32(z), f32(ctx), {k[2:]: host(v) for k, v in store.vars.items()}, [hs, hs], [zs, zs])
    np.testing.assert_allclose(host(again[0]), exp[0], atol=ATOL, rtol=0)


def test_discretized_logistic_vs_reference_golden(amd, golden_dir):
    g = np.load(os.path.join(golden_dir, "distributions.npz"))
    out = amd.discretized_logistic(dev(g["dl_mean"]), -1.3, sample=dev(g["dl_sample"]))
    np.testing.assert_allclose(host(out), g["dl_logp"], rtol=2e-5, atol=1e-3)


def test_discretized_logistic_cifar_shape_vs_oracle(amd):
    """tf_train.py:210 shape: x, orig_x [B,3,32,32], dec_log_stdv a scalar variable"""
    rng = np.random.RandomState(8)
    B = 32
    mean = rng.uniform(-0.6, 0.6, (B, 3, 32, 32))
    sample = np.floor(rng.uniform(0, 256, (B, 3, 32, 32))) / 256.0 - 0.5
    ls = torch.tensor([-2.0], device="cuda")
    out = amd.discretized_logistic(dev(mean), ls, sample=dev(sample))
    e = O.discretized_logistic(f32(mean), -2.0, f32(sample))
    np.testing.assert_allclose(host(out), e, rtol=2e-5, atol=1e-2)
    # per-element logscale tensor
    lst = rng.uniform(-2.5, -1.0, mean.shape)
    out = amd.discretized_logistic(dev(mean), dev(lst), sample=dev(sample))
    np.testing.assert_allclose(host(out), O.discretized_logistic(f32(mean), f32(lst), f32(sample)), rtol=2e-5, atol=1e-2)


def test_conv_prep_batch_equals_individual_prepare(amd, golden_dir):
    """one batched weight-prep launch for all plain convs of a layer == per-conv prepare: cfg2 fixture end to end"""
    name = "layer_cfg2_8x8"
    g = np.load(os.path.join(golden_dir, "iaf_layer.npz"))
    c = gi.layer_case_inputs(name)
    params = {k: dev(v) for k, v in c["params"].items()}
    layer = amd.IAFLayer(c["z_size"], c["h_size"], depth_ar=2, kl_min=c["kl_min"])
    junk = {k: torch.randn_like(v) for k, v in params.items()}
    layer.load(junk)                                         # make sure stale packs cannot pass
    amd.ConvPrepBatch(layer.convs()).run(amd.IAFLayer.conv_params(params))
    amd.PrepBatch([layer.posterior.stack]).run([amd.IAFLayer.stack_params(params)])
    up_out = layer.up(dev(c["up_input"]))
    out, kl_obj, kl_cost = layer.down(dev(c["down_input"]), dev(c["eps_post"]))
    np.testing.assert_allclose(host(up_out), g[name + "/up_out"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(out), g[name + "/output"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(kl_obj), g[name + "/kl_obj"], atol=2e-3, rtol=1e-4)


# ---------------------------------------------------------------- backward of the plain convs and of the whole layer
def _relerr(got, want):
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)


@pytest.mark.parametrize("shape", [(3, 160, 384, 8, 8), (2, 192, 160, 16, 16), (2, 32, 48, 5, 7), (4, 160, 448, 8, 8)],
                         ids=lambda s: "B%d_%dto%d_%dx%d" % s)
def test_wnconv2d_backward_vs_autograd_oracle(amd, shape):
    """L = <dy, res + 0.1*conv(elu(x))>: dx, dV, dg, db against torch-fp64 autograd of layers.py:52-64"""
    from oracle import iaf_grad_oracle as G
    B, n_in, n_out, H, W = shape
    rng = np.random.RandomState(71)
    p = gi.conv_params(rng, n_in, n_out)
    x, dy = rng.standard_normal((B, n_in, H, W)), rng.standard_normal((B, n_out, H, W))
    xt = G._t(f32(x), True)
    pt = {k: G._t(f32(v), True) for k, v in p.items()}
    y = 0.1 * G.conv2d(torch.nn.functional.elu(xt), pt["V"], pt["g"], pt["b"])
    (y * G._t(f32(dy))).sum().backward()
    conv = amd.WNConv2d(n_in, n_out)
    conv.set_training(True)
    V, g, b = dev(p["V"]), dev(p["g"]), dev(p["b"])
    conv.prepare(V, g, b)
    half = n_out // 2 // 4 * 4
    dyd = dev(dy)
    dys = [dyd[:, :half].contiguous(), dyd[:, half:].contiguous()]             # gradient arrives as split tensors
    res = dev(rng.standard_normal(x.shape))
    (dx,), dV, dg, db = conv.backward(dev(x), dys, V, g, elu_input=True, dy_scale=0.1, dx_residual=res)
    assert _relerr(host(dx) - host(res), xt.grad.numpy()) < 1e-4
    assert _relerr(host(dV), pt["V"].grad.numpy()) < 1e-4
    assert _relerr(host(dg), pt["g"].grad.numpy()) < 1e-4
    assert _relerr(host(db), pt["b"].grad.numpy()) < 1e-4


@pytest.mark.parametrize("shape", [(16, 160, 384, 16, 16), (16, 192, 160, 16, 16), (32, 160, 160, 16, 16), (16, 160, 448, 16, 16)],
                         ids=lambda s: "B%d_%dto%d_%dx%d" % s)
def test_wnconv2d_data_gradient_on_the_bf16_matrix_cores(amd, shape):
    """from 4096 pixels on the data gradient dX = W^T dY of a plain conv runs the bf16x3 kernel on the TRANSPOSED bf16x3 pack
    (iaf_pack_t3_kernel; K = n_out up to 384 fits the LDS tile, 448 stays on the exact-fp32 kernel): against torch-fp64
    autograd of layers.py:52-64, and against the exact-fp32 kernels on the same inputs (fp32-grade: not just inside 1e-4)"""
    from oracle import iaf_grad_oracle as G
    B, n_in, n_out, H, W = shape
    rng = np.random.RandomState(72)
    p = gi.conv_params(rng, n_in, n_out)
    x, dy = rng.standard_normal((B, n_in, H, W)), rng.standard_normal((B, n_out, H, W))
    V, g, b = dev(p["V"]), dev(p["g"]), dev(p["b"])
    res = dev(rng.standard_normal(x.shape))
    got = {}
    for prec in ("bf16x3", "f32", "f16x2"):
        conv = amd.WNConv2d(n_in, n_out)
        conv.set_precision(prec)
        conv.set_training(True)
        conv.prepare(V, g, b)
        (dx,), dV, dg, db = conv.backward(dev(x), [dev(dy)], V, g, elu_input=True, dy_scale=0.1, dx_residual=res)
        got[prec] = [host(dx) - host(res), host(dV), host(dg), host(db)]
    xt = G._t(f32(x), True)
    pt = {k: G._t(f32(v), True) for k, v in p.items()}
    chunk = 4                                        # (fp64 autograd of a 16 x 160 x 16 x 16 conv in slices: memory)
    gx = []
    for b0 in range(0, B, chunk):
        xs = G._t(f32(x[b0:b0 + chunk]), True)
        y = 0.1 * G.conv2d(torch.nn.functional.elu(xs), pt["V"], pt["g"], pt["b"])
        (y * G._t(f32(dy[b0:b0 + chunk]))).sum().backward()
        gx.append(xs.grad.numpy())
    gx = np.concatenate(gx)
    for prec in ("bf16x3", "f32", "f16x2"):
        assert _relerr(got[prec][0], gx) < 1e-4
        assert _relerr(got[prec][1], pt["V"].grad.numpy()) < 1e-4
        assert _relerr(got[prec][2], pt["g"].grad.numpy()) < 1e-4
        assert _relerr(got[prec][3], pt["b"].grad.numpy()) < 1e-4
    e3, e32 = _relerr(got["bf16x3"][0], gx), _relerr(got["f32"][0], gx)
    print("dX rel err vs fp64 autograd: bf16x3 data gradient %.3g, exact fp32 %.3g" % (e3, e32))
    assert e3 <= 2.0 * e32 + 1e-6
    # round 6: under "f16x2" the data gradient runs on two fp16 planes with a power-of-two scale per staged tile (iaf_conv_bf3.hpp DG16) --
    # also for K = n_out = 448, whose two-plane tile fits the LDS
    e16 = _relerr(got["f16x2"][0], gx)
    print("   two fp16 planes: %.3g" % e16)
    assert e16 <= 2.0 * e32 + 1e-6


@pytest.mark.parametrize("dy_scale", [1e-12, 1.0, 1e10, "mixed"], ids=lambda v: "dy_x%s" % v)
def test_fp16_plane_data_gradient_has_no_exponent_range_of_its_own(amd, dy_scale):
    """a gradient tensor's magnitudes have nothing to do with fp16's range: the workgroup scales the tile it stages by a power of two taken
    from the tile's own largest element and its sums back -- gradients of 1e-12 and of 1e10 come out with the relative error of the unit case, a batch whose
    images differ by up to 1e6 within 3e-5 of each IMAGE's largest gradient, and nothing raises a range word"""
    from oracle import iaf_grad_oracle as G
    B, n_in, n_out, H, W = 16, 160, 160, 16, 16
    rng = np.random.RandomState(75)
    p = gi.conv_params(rng, n_in, n_out)
    x, dy = rng.standard_normal((B, n_in, H, W)), rng.standard_normal((B, n_out, H, W))
    if dy_scale == "mixed":
        dy *= (10.0 ** rng.randint(-3, 4, size=(B, 1, 1, 1)))
    else:
        dy *= dy_scale
    V, g, b = dev(p["V"]), dev(p["g"]), dev(p["b"])
    conv = amd.WNConv2d(n_in, n_out)
    conv.set_precision("f16x2")
    conv.set_training(True)
    conv.prepare(V, g, b)
    (dx,), dV, dg, db = conv.backward(dev(x), [dev(dy)], V, g, elu_input=True)
    pt = {k: G._t(f32(v), True) for k, v in p.items()}
    worst = 0.0
    for b0 in range(0, B, 4):
        xs = G._t(f32(x[b0:b0 + 4]), True)
        y = G.conv2d(torch.nn.functional.elu(xs), pt["V"], pt["g"], pt["b"])
        (y * G._t(f32(dy[b0:b0 + 4]))).sum().backward()
        gx = xs.grad.numpy()
        for i in range(4):
            worst = max(worst, _relerr(host(dx)[b0 + i], gx[i]))
    print("dX rel err per image vs fp64 autograd: %.3g" % worst)
    # (a tile's halo reaches 17 pixels into the neighbouring image: where that image's gradients are 1e6 larger they set the tile's scale)
    assert worst < (3e-5 if dy_scale == "mixed" else 3e-6)
    assert conv.range_errors() == 0




@pytest.mark.parametrize("scale", [(1.0, 1.0), (3e4, 1e-5), (1e-6, 2e3)], ids=["unit", "x3e4_dy1e-5", "x1e-6_dy2e3"])
@pytest.mark.parametrize("shape", [(3, 32, 64, 12, 12), (5, 64, 160, 8, 8), (2, 160, 224, 16, 16), (1, 96, 192, 24, 8), (8, 192, 160, 16, 16)],
                         ids=lambda s: "B%d_%dto%d_%dx%d" % s)
def test_wnconv2d_weight_gradient_on_the_bf16_matrix_cores(amd, shape, scale):
    """dV, dg, db of a plain conv with the weight gradient dW[tap] = X_shifted^T dY on the bf16 matrix cores (iaf_wgrad_bf3.hip:
    tap-row workgroups, transposing LDS reads, border masks on the fragments) against torch-fp64 autograd of layers.py:52-64,
    and against the exact-fp32 MFMA kernel on the same inputs.  Shapes: pixel counts that leave the last K block of a range
    partial (432 = 13.5 blocks, 320 = 10, 192 = 6) and put image borders inside K blocks (12- and 24-pixel rows against
    32-pixel blocks); scales: operands whose bf16x3 planes sit far from 1 (the split is by value, not by exponent)."""
    from oracle import iaf_grad_oracle as G
    B, n_in, n_out, H, W = shape
    sx, sy = scale
    rng = np.random.RandomState(73)
    p = gi.conv_params(rng, n_in, n_out)
    x, dy = sx * rng.standard_normal((B, n_in, H, W)), sy * rng.standard_normal((B, n_out, H, W))
    V, g, b = dev(p["V"]), dev(p["g"]), dev(p["b"])
    got = {}
    for prec in ("bf16x3", "f32", "f16x2"):
        conv = amd.WNConv2d(n_in, n_out)
        conv.set_precision(prec)
        conv.set_training(True)
        conv.prepare(V, g, b)
        (dx,), dV, dg, db = conv.backward(dev(x), [dev(dy)], V, g)
        got[prec] = [host(dV), host(dg), host(db)]
    xt = G._t(f32(x))
    pt = {k: G._t(f32(v), True) for k, v in p.items()}
    (G.conv2d(xt, pt["V"], pt["g"], pt["b"]) * G._t(f32(dy))).sum().backward()
    want = [pt[k].grad.numpy() for k in ("V", "g", "b")]
    for prec in ("bf16x3", "f32", "f16x2"):
        for a, w_ in zip(got[prec], want):
            assert _relerr(a, w_) < 1e-4
    e3, e32 = _relerr(got["bf16x3"][0], want[0]), _relerr(got["f32"][0], want[0])
    print("dV rel err vs fp64 autograd: bf16x3 weight gradient %.3g, exact fp32 %.3g" % (e3, e32))
    assert e3 <= 2.0 * e32 + 1e-6
    # a conv whose arithmetic is "f16x2" (round 6): its weight gradient stays on the bf16 planes (a two-plane form was built and measured
    # slower: profiles/r06/experiments/wgrad_fp16_planes_not_kept.txt) -- the same bar at every operand scale
    e16 = _relerr(got["f16x2"][0], want[0])
    print("   two fp16 planes: %.3g" % e16)
    assert e16 <= 2.0 * e32 + 1e-6


@pytest.mark.parametrize("size", [None, (8, 16, 16)], ids=["fixture_B2_8x8", "B8_16x16"])
@pytest.mark.parametrize("kl_min", [0.25, 0.0])
def test_iaf_layer_backward_vs_autograd_oracle(amd, kl_min, size):
    """whole IAFLayer (up + down) at z 32 / h 160, 8x8: gradients of L = <dU, up_out> + <dD, output> + <dK, kl_obj>
    w.r.t. both inputs and all 28 variables vs torch-fp64 autograd of the restated layer (itself pinned to the reference
    fixtures and finite differences in tests/test_grad_oracle.py).  Bar: 1e-4 of the tensor's max |reference|."""
    from oracle import iaf_grad_oracle as G
    c = gi.layer_case_inputs("layer_cfg2_8x8")
    zs, hs = c["z_size"], c["h_size"]
    rng = np.random.RandomState(9)
    up_in, down_in, eps = c["up_input"], c["down_input"], c["eps_post"]
    if size is not None:        # a larger problem: 11 pixel ranges in the weight gradient, the last one partial
        B, H, W = size
        # N(0,1) inputs through these random convs give posterior log-stds of several units on a few of the 65k latent
        # positions (|z| reaches 1e4, where fp32 itself is only good to 1e-3): keep the larger case in a sane range
        up_in, down_in, eps = (0.3 * rng.standard_normal((B, ch, H, W)) for ch in (hs, hs, zs))
    dU, dD, dK = rng.standard_normal(up_in.shape), rng.standard_normal(down_in.shape), rng.standard_normal(up_in.shape[0])
    p32 = {k: f32(v) for k, v in c["params"].items()}
    want, fw = G.iaf_layer_grads(f32(up_in), f32(down_in), f32(eps), p32, zs, hs, kl_min, f32(dU), f32(dD), f32(dK))
    params = {k: dev(v) for k, v in c["params"].items()}
    layer = amd.IAFLayer(zs, hs, depth_ar=2, kl_min=kl_min)
    layer.set_training(True)
    layer.load(params)
    up_out = layer.up_train(dev(up_in))
    out, kl_obj, kl_cost = layer.down_train(dev(down_in), dev(eps))
    np.testing.assert_allclose(host(up_out), fw["up_out"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(out), fw["output"], atol=ATOL * max(1.0, np.abs(fw["output"]).max()), rtol=0)
    np.testing.assert_allclose(host(kl_obj), fw["kl_obj"], atol=2e-3, rtol=1e-4)
    grads = {}
    d_down_in = layer.down_backward(dev(dD), dev(dK), params, grads)
    d_up_in = layer.up_backward(dev(dU), params, grads)
    assert _relerr(host(d_down_in), want["down_inp"]) < 1e-4
    assert _relerr(host(d_up_in), want["up_inp"]) < 1e-4
    assert sorted(grads) == sorted(want["params"])
    worst = max((_relerr(host(grads[k]), want["params"][k]), k) for k in grads)
    assert worst[0] < 1e-4, worst


@pytest.mark.parametrize("kl_min", [0.25, 0.0])
@pytest.mark.parametrize("size", [(2, 16, 16), (3, 8, 8)], ids=lambda s: "B%d_%dx%d" % s)
def test_downsampling_iaf_layer_backward_vs_autograd_oracle(amd, kl_min, size):
    """the first layer of a coarser level (tf_train.py:33,42-43,89-91: stride-2 up_conv1, resize 0.5, down_deconv2, resize 2)
    trains: gradients w.r.t. both inputs (up input at full resolution, down input at half) and all variables -- incl.
    down_deconv2 with its per-input-channel weight norm -- vs torch-fp64 autograd of the restated layer, itself pinned to the
    reference's own IAFLayer(downsample=True) outputs and to finite differences (tests/test_grad_oracle.py)"""
    from oracle import iaf_grad_oracle as G
    c = gi.layer_ds_case_inputs("layer_ds_cfg2")
    zs, hs = c["z_size"], c["h_size"]
    B, H, W = size
    rng = np.random.RandomState(19)
    up_in = 0.3 * rng.standard_normal((B, hs, H, W))
    down_in, eps = 0.3 * rng.standard_normal((B, hs, H // 2, W // 2)), 0.3 * rng.standard_normal((B, zs, H // 2, W // 2))
    dU, dD, dK = rng.standard_normal((B, hs, H // 2, W // 2)), rng.standard_normal((B, hs, H, W)), rng.standard_normal(B)
    p32 = {k: f32(v) for k, v in c["params"].items()}
    want, fw = G.iaf_layer_grads(f32(up_in), f32(down_in), f32(eps), p32, zs, hs, kl_min, f32(dU), f32(dD), f32(dK), downsample=True)
    params = {k: dev(v) for k, v in c["params"].items()}
    layer = amd.IAFLayer(zs, hs, depth_ar=2, kl_min=kl_min, downsample=True)
    layer.set_training(True)
    layer.load(params)
    up_out = layer.up_train(dev(up_in))
    out, kl_obj, kl_cost = layer.down_train(dev(down_in), dev(eps))
    np.testing.assert_allclose(host(up_out), fw["up_out"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(out), fw["output"], atol=ATOL * max(1.0, np.abs(fw["output"]).max()), rtol=0)
    np.testing.assert_allclose(host(kl_obj), fw["kl_obj"], atol=2e-3, rtol=1e-4)
    grads = {}
    d_down_in = layer.down_backward(dev(dD), dev(dK), params, grads)
    d_up_in = layer.up_backward(dev(dU), params, grads)
    assert tuple(d_down_in.shape) == down_in.shape and tuple(d_up_in.shape) == up_in.shape
    assert _relerr(host(d_down_in), want["down_inp"]) < 1e-4
    assert _relerr(host(d_up_in), want["up_inp"]) < 1e-4
    assert sorted(grads) == sorted(want["params"])
    worst = max((_relerr(host(grads[k]), want["params"][k]), k) for k in grads)
    assert worst[0] < 1e-4, worst
    # inference forward of the same layer object equals the training forward
    assert torch.equal(layer.up(dev(up_in)), up_out)


def test_two_level_model_trains_end_to_end_vs_autograd_oracle(amd):
    """BASELINE configs[1] in miniature as ONE connected model (tf_train.py:186-206): a 16x16 layer, then the downsampling
    layer that opens the 8x8 level, then an 8x8 layer; up pass bottom-up, down pass top-down from h_top, the loss the sum of
    the kl_obj terms plus a linear read-out of the final output.  Every gradient -- the h_top parameter's, the input's and all
    variables of the three layers -- against torch-fp64 autograd of the chained restated layers."""
    from oracle import iaf_grad_oracle as G
    zs, hs, B, kl_min = 32, 160, 2, 0.25
    rng = np.random.RandomState(23)
    cA, cB, cC = (gi.layer_case_inputs("layer_cfg2_8x8"), gi.layer_ds_case_inputs("layer_ds_cfg2"), gi.layer_case_inputs("layer_cfg2_8x8"))
    pA = {k: v for k, v in cA["params"].items()}
    pB = {k: v for k, v in cB["params"].items()}
    pC = {k: v + 0.01 * rng.standard_normal(v.shape) for k, v in cC["params"].items()}
    x = 0.3 * rng.standard_normal((B, hs, 16, 16))
    h_top = 0.3 * rng.standard_normal((1, hs, 8, 8))
    epsA, epsB, epsC = (0.3 * rng.standard_normal((B, zs, s, s)) for s in (16, 8, 8))
    d_out = rng.standard_normal((B, hs, 16, 16))

    # ---- oracle: chained torch fp64 layers (G.iaf_layer runs up then down of ONE layer: compose by hand)
    import torch.nn.functional as F
    t = lambda a, g=False: G._t(f32(a), g)
    xt, ht = t(x, True), t(h_top, True)
    PA, PB, PC = ({k: t(v, True) for k, v in p.items()} for p in (pA, pB, pC))

    def up(inp, P, ds):
        c1 = G.conv2d_stride2 if ds else G.conv2d
        q = c1(F.elu(inp), P["up_conv1/V"], P["up_conv1/g"], P["up_conv1/b"])
        qm, ql, uc, h = torch.split(q, [zs, zs, hs, hs], dim=1)
        h = G.conv2d(F.elu(h), P["up_conv3/V"], P["up_conv3/g"], P["up_conv3/b"])
        return (inp[:, :, ::2, ::2] if ds else inp) + 0.1 * h, (qm, ql, uc)

    def down(inp, P, st, eps, ds):
        q = G.conv2d(F.elu(inp), P["down_conv1/V"], P["down_conv1/g"], P["down_conv1/b"])
        pm, pl, rm, rl, dc, hd = torch.split(q, [zs] * 4 + [hs] * 2, dim=1)
        sp = {k[len("ar_multiconv2d/"):]: v for k, v in P.items() if k.startswith("ar_multiconv2d/")}
        z, kl_obj, _ = G.posterior_block(st[0], st[1], rm, rl, pm, pl, st[2], dc, t(eps), sp, [hs, hs], kl_min)
        cat = F.elu(torch.cat([z, hd], dim=1))
        if ds:
            h = G.deconv2d(cat, P["down_deconv2/V"], P["down_deconv2/g"], P["down_deconv2/b"])
            return inp.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3) + 0.1 * h, kl_obj
        return inp + 0.1 * G.conv2d(cat, P["down_conv2/V"], P["down_conv2/g"], P["down_conv2/b"]), kl_obj

    hA, sA = up(xt, PA, False)
    hB, sB = up(hA, PB, True)
    hC, sC = up(hB, PC, False)
    hcur = ht.repeat(B, 1, 1, 1)                                      # tf_train.py:186-188 (h_top tiled over the batch)
    hcur, klC = down(hcur, PC, sC, epsC, False)
    hcur, klB = down(hcur, PB, sB, epsB, True)
    hcur, klA = down(hcur, PA, sA, epsA, False)
    loss = (klA + klB + klC).sum() + (hcur * t(d_out)).sum()
    loss.backward()

    # ---- the engine
    layers = [amd.IAFLayer(zs, hs, 2, kl_min), amd.IAFLayer(zs, hs, 2, kl_min, downsample=True), amd.IAFLayer(zs, hs, 2, kl_min)]
    dps = [{k: dev(v) for k, v in p.items()} for p in (pA, pB, pC)]
    for L, dp in zip(layers, dps):
        L.set_training(True)
        L.load(dp)
    h = dev(x)
    for L in layers:
        h = L.up_train(h)
    h = dev(h_top).repeat(B, 1, 1, 1).contiguous()
    kls = []
    for L, e in zip(reversed(layers), (epsC, epsB, epsA)):
        h, kl_obj, _ = L.down_train(h, dev(e))
        kls.append(kl_obj)
    np.testing.assert_allclose(host(h), hcur.detach().numpy(), atol=2e-4, rtol=0)
    np.testing.assert_allclose(sum(float(k.sum()) for k in kls), float((klA + klB + klC).sum().detach()), rtol=2e-5)
    grads = [{}, {}, {}]
    ones = torch.ones(B, device="cuda")
    d = dev(d_out)
    for i in (0, 1, 2):                                               # backward of the down pass, bottom layer first
        d = layers[i].down_backward(d, ones, dps[i], grads[i])
    d_h_top = d.sum(dim=0, keepdim=True)
    d = torch.zeros(B, hs, 8, 8, device="cuda")                       # the top of the up pass feeds nothing else
    for i in (2, 1, 0):
        d = layers[i].up_backward(d, dps[i], grads[i])
    assert _relerr(host(d_h_top), ht.grad.numpy()) < 1e-4
    assert _relerr(host(d), xt.grad.numpy()) < 1e-4
    for gi_, P, nm in ((grads[0], PA, "A"), (grads[1], PB, "B"), (grads[2], PC, "C")):
        assert sorted(gi_) == sorted(P)
        # (the top layer's up-pass output feeds nothing, tf_train.py:186: autograd leaves its up_conv3 gradients unset, the
        # engine writes the zeros they are)
        ref = {k: (P[k].grad.numpy() if P[k].grad is not None else np.zeros(tuple(P[k].shape))) for k in gi_}
        for k in gi_:
            if not np.any(ref[k]):
                assert not np.any(host(gi_[k])), (nm, k)
        worst = max((_relerr(host(gi_[k]), ref[k]), k) for k in gi_ if np.any(ref[k]))
        assert worst[0] < 2e-4, (nm,) + worst


def test_deferred_weightnorm_backward_equals_immediate(amd):
    """two layers of different spatial size: backward with the weight-norm pass deferred to ONE batched launch per kind
    must give exactly the gradients of the per-layer launches"""
    c = gi.layer_case_inputs("layer_cfg2_8x8")
    zs, hs = c["z_size"], c["h_size"]
    rng = np.random.RandomState(4)
    sizes = [(2, 8, 8), (3, 4, 4)]
    layers, data = [], []
    for (B, H, W) in sizes:
        params = {k: dev(v + 0.01 * rng.standard_normal(v.shape)) for k, v in c["params"].items()}
        layer = amd.IAFLayer(zs, hs, depth_ar=2, kl_min=0.25)
        layer.set_training(True)
        layer.load(params)
        f = lambda ch: dev(rng.standard_normal((B, ch, H, W)))
        data.append(dict(params=params, up=f(hs), down=f(hs), eps=f(zs), dU=f(hs), dD=f(hs), dK=dev(rng.standard_normal(B))))
        layers.append(layer)

    def run(defer):
        grads = [dict() for _ in layers]
        batch = None
        if defer:
            batch = amd.WnBwdBatch(stacks=[L.posterior.stack for L in layers], convs=[cv for L in layers for cv in L.convs()])
            for g_, d in zip(grads, data):          # deferred mode writes through the batch: allocate the slots up front
                for k, v in d["params"].items():
                    g_[k] = torch.zeros_like(v)
        outs = []
        for L, d, g_ in zip(layers, data, grads):
            L.up_train(d["up"])
            L.down_train(d["down"], d["eps"])
            dd = L.down_backward(d["dD"], d["dK"], d["params"], g_)
            du = L.up_backward(d["dU"], d["params"], g_)
            outs.append((dd, du))
        if defer:
            batch.run(stack_params=[amd.IAFLayer.stack_params(d["params"]) for d in data],
                      stack_grads=[amd.IAFLayer.stack_params(g_) for g_ in grads],
                      conv_params=[t for d in data for t in amd.IAFLayer.conv_params(d["params"])],
                      conv_grads=[t for g_ in grads for t in amd.IAFLayer.conv_params(g_)])
        torch.cuda.synchronize()
        return grads, outs

    g0, o0 = run(False)
    g0 = [{k: v.clone() for k, v in g_.items()} for g_ in g0]
    g1, o1 = run(True)
    for a, b in zip(o0, o1):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for ga, gb in zip(g0, g1):
        assert sorted(ga) == sorted(gb)
        for k in ga:
            assert torch.equal(ga[k], gb[k]), k


def test_batch_objects_reject_misuse(amd):
    """the batched prep / deferred weight-norm entry points report misuse through status codes, not crashes"""
    masked = amd.WNConv2d(32, 32, ar_mask=True)
    with pytest.raises(ValueError):                       # batched prep is for plain convs (IAF_ERR_UNSUPPORTED)
        amd.ConvPrepBatch([masked])
    tiny = amd.WNConv2d(6, 10)                            # fallback-path conv: trains since round 5 (tests/test_hip_generic_backward.py),
    tiny.set_training(True)                               # but not through the batched weight-norm backward (IAF_ERR_UNSUPPORTED)
    with pytest.raises(ValueError):
        amd.WnBwdBatch(convs=[tiny])
    with pytest.raises(ValueError):                       # a masked single conv still has no backward
        masked.set_training(True)
    conv = amd.WNConv2d(32, 32)
    with pytest.raises(amd._capi.IafHipError):            # deferral needs set_training first (IAF_ERR_NOT_PREPARED)
        amd.WnBwdBatch(convs=[conv])
    conv.set_training(True)
    V, g, b = torch.randn((3, 3, 32, 32), device="cuda") * 0.05, torch.zeros(32, device="cuda"), torch.zeros(32, device="cuda")
    conv.prepare(V, g, b)
    batch = amd.WnBwdBatch(convs=[conv])
    grads = (torch.zeros_like(V), torch.zeros_like(g), torch.zeros_like(b))
    with pytest.raises(amd._capi.IafHipError):            # nothing pending yet
        batch.run(conv_params=[(V, g, b)], conv_grads=[grads])
    x, dy = torch.randn((2, 32, 4, 4), device="cuda"), torch.randn((2, 32, 4, 4), device="cuda")
    conv(x)
    (dx,), dV, dg, db = conv.backward(x, [dy], V, g, grads_out=grads)
    batch.run(conv_params=[(V, g, b)], conv_grads=[grads])
    torch.cuda.synchronize()
    assert torch.isfinite(grads[0]).all() and float(grads[0].abs().max()) > 0
    stack = amd.ARStack(32, [64])
    rng = np.random.RandomState(0)
    sp = {k: dev(v) for k, v in gi.ar_multiconv2d_params(rng, 32, [64], [32, 32]).items()}
    sb = amd.WnBwdBatch(stacks=[stack])
    with pytest.raises(amd._capi.IafHipError):            # no deferred backward pending on the stack
        sb.run(stack_params=[sp], stack_grads=[{k: torch.zeros_like(v) for k, v in sp.items()}])


def test_backward_autotune_keeps_gradients(amd):
    """the data-gradient launch-shape search may pick a split-K shape (different summation order): gradients must stay
    within fp32 round-off of the untuned call"""
    rng = np.random.RandomState(17)
    n_in, n_out, B, H, W = 160, 160, 4, 8, 8
    p = gi.conv_params(rng, n_in, n_out)
    V, g, b = dev(p["V"]), dev(p["g"]), dev(p["b"])
    x, dy = dev(rng.standard_normal((B, n_in, H, W))), dev(rng.standard_normal((B, n_out, H, W)))
    conv = amd.WNConv2d(n_in, n_out)
    conv.set_training(True)
    conv.prepare(V, g, b)
    conv(x, elu_input=True)
    (dx0,), dV0, dg0, db0 = conv.backward(x, [dy], V, g, elu_input=True)
    (dx1,), dV1, dg1, db1 = conv.backward(x, [dy], V, g, elu_input=True, autotune=True)
    (dx2,), dV2, dg2, db2 = conv.backward(x, [dy], V, g, elu_input=True)          # now with the tuned shape pinned
    assert ("bwd", B, H, W) in conv._tuned
    for a, b_ in ((dx0, dx1), (dx0, dx2), (dV0, dV1), (dV0, dV2), (dg0, dg2), (db0, db2)):
        np.testing.assert_allclose(host(a), host(b_), atol=1e-5 * float(a.abs().max()) + 1e-7, rtol=0)


# ---------------------------------------------------------------- downsampling IAFLayer, init / sample modes (tf_train.py:33,42-43,60-66,89-91)
def test_resample_and_deconv_vs_reference_golden(amd, golden_dir):
    """resize_nearest_neighbor (layers.py:169-175) and deconv2d (layers.py:67-112) against the outputs of the reference's
    own functions: deconv = zero-insert + the stride-1 conv kernel with the rotated, deconv-normalised filter"""
    g = np.load(os.path.join(golden_dir, "iaf_layer_ds.npz"))
    np.testing.assert_array_equal(host(amd.resize_nearest_neighbor(dev(g["resize/x"]), 0.5)), f32(g["resize/half"]))
    np.testing.assert_array_equal(host(amd.resize_nearest_neighbor(dev(g["resize/x"]), 2)), f32(g["resize/double"]))
    x = g["deconv/x"]
    conv = amd.WNConv2d(x.shape[1], g["deconv/V"].shape[2])
    conv.prepare_deconv(dev(g["deconv/V"]), dev(g["deconv/g"]), dev(g["deconv/b"]))
    y = conv(amd.resample2(dev(x), "up_zero_odd"))[0]
    np.testing.assert_allclose(host(y), g["deconv/y"], atol=ATOL, rtol=0)


@pytest.mark.parametrize("shape", [(2, 48, 32, 8, 8), (3, 192, 160, 8, 8), (1, 16, 16, 3, 5)], ids=lambda s: "B%d_%d_%d_%dx%d" % s)
def test_deconv2d_vs_oracle(amd, shape):
    """MFMA path of the deconv (channels % 16 == 0), incl. the down_deconv2 shape 192 -> 160 at 8x8 -> 16x16"""
    B, n_in, n_out, H, W = shape
    rng = np.random.RandomState(12)
    p = gi.deconv_params(rng, n_in, n_out)
    x = rng.standard_normal((B, n_in, H, W))
    conv = amd.WNConv2d(n_in, n_out)
    conv.prepare_deconv(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    y = conv(amd.resample2(dev(x), "up_zero_odd"))[0]
    e = O.deconv2d(f32(x), f32(p["V"]), f32(p["g"]), f32(p["b"]))
    np.testing.assert_allclose(host(y), e, atol=ATOL, rtol=0)


@pytest.mark.parametrize("name", sorted(gi.LAYER_DS_CASES))
def test_iaf_layer_downsample_and_modes_vs_reference_golden(amd, golden_dir, name):
    """IAFLayer.up/.down with downsample=True and in modes "init" / "sample", every op on the GPU, against the outputs
    of the reference's own tf_train.IAFLayer (tests/golden/make_golden.py: gen_layers_ds)"""
    g = np.load(os.path.join(golden_dir, "iaf_layer_ds.npz"))
    c = gi.layer_ds_case_inputs(name)
    layer = amd.IAFLayer(c["z_size"], c["h_size"], depth_ar=2, kl_min=c["kl_min"], downsample=c["downsample"], mode=c["mode"])
    layer.load({k: dev(v) for k, v in c["params"].items()})
    up_out = layer.up(dev(c["up_input"]))
    np.testing.assert_allclose(host(up_out), g[name + "/up_out"], atol=ATOL, rtol=0)
    po = layer.posterior
    np.testing.assert_allclose(host(po.qz_mean), g[name + "/qz_mean"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(po.qz_logsd), g[name + "/qz_logsd"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(po.up_context), g[name + "/up_context"], atol=ATOL, rtol=0)
    out, kl_obj, kl_cost = layer.down(dev(c["down_input"]), dev(c["eps_post"]), eps_prior=dev(c["eps_prior"]))
    np.testing.assert_allclose(host(out), g[name + "/output"], atol=2 * ATOL, rtol=0)
    np.testing.assert_allclose(host(kl_obj), g[name + "/kl_obj"], atol=2e-3, rtol=2e-4)
    np.testing.assert_allclose(host(kl_cost), g[name + "/kl_cost"], atol=2e-3, rtol=2e-4)
    if c["downsample"]:
        assert tuple(up_out.shape[2:]) == (c["H"] // 2, c["W"] // 2) and tuple(out.shape[2:]) == (c["H"], c["W"])


def test_two_level_stack_chains_through_the_downsampling_layer(amd):
    """BASELINE configs[1] structure at reduced depth: level 0 at 16x16, level 1 at 8x8 whose first layer downsamples
    (tf_train.py:188-200): the up pass goes 16x16 -> 8x8, the down pass 8x8 -> 16x16, against the oracle's layers"""
    zs, hs, B = 32, 160, 2
    rng = np.random.RandomState(77)
    spec = [(0, False), (0, False), (1, True), (1, False)]         # (level, downsample) in up-pass order
    layers, params = [], []
    for lvl, ds in spec:
        p = {}
        for nm, (ci, co) in (("up_conv1", (hs, 2 * zs + 2 * hs)), ("up_conv3", (hs, hs)), ("down_conv1", (hs, 4 * zs + 2 * hs))):
            for k, v in gi.conv_params(rng, ci, co).items():
                p[nm + "/" + k] = v
        for k, v in gi.ar_multiconv2d_params(rng, zs, [hs, hs], [zs, zs]).items():
            p["ar_multiconv2d/" + k] = v
        last = gi.deconv_params(rng, hs + zs, hs) if ds else gi.conv_params(rng, hs + zs, hs)
        for k, v in last.items():
            p[("down_deconv2/" if ds else "down_conv2/") + k] = v
        L = amd.IAFLayer(zs, hs, depth_ar=2, kl_min=0.25, downsample=ds)
        L.load({k: dev(v) for k, v in p.items()})
        layers.append(L); params.append(p)
    x = 0.5 * rng.standard_normal((B, hs, 16, 16))
    h_top = 0.5 * rng.standard_normal((B, hs, 8, 8))
    eps = [0.5 * rng.standard_normal((B, zs, 16 >> lvl, 16 >> lvl)) for lvl, _ in spec]
    # GPU
    h = dev(x)
    for L in layers:
        h = L.up(h)
    assert tuple(h.shape) == (B, hs, 8, 8)
    d = dev(h_top)
    kls = []
    for L, e in zip(reversed(layers), reversed(eps)):
        d, kl_obj, kl_cost = L.down(d, dev(e))
        kls.append(kl_cost)
    assert tuple(d.shape) == (B, hs, 16, 16)
    # oracle
    eh, st = f32(x), []
    for (lvl, ds), p in zip(spec, params):
        p32 = {k: f32(v) for k, v in p.items()}
        eh, qm, ql, uc = O.iaf_layer_up(eh, p32, zs, hs, downsample=ds)
        st.append((qm, ql, uc))
    ed, ekl = f32(h_top), []
    for (lvl, ds), p, (qm, ql, uc), e in zip(reversed(spec), reversed(params), reversed(st), reversed(eps)):
        p32 = {k: f32(v) for k, v in p.items()}
        ed, _, klc, _ = O.iaf_layer_down(ed, p32, qm, ql, uc, f32(e), zs, hs, 0.25, downsample=ds)
        ekl.append(klc)
    np.testing.assert_allclose(host(h), eh, atol=2 * ATOL, rtol=0)
    np.testing.assert_allclose(host(d), ed, atol=5 * ATOL, rtol=0)
    for a, b in zip(kls, ekl):
        np.testing.assert_allclose(host(a), b, atol=5e-3, rtol=3e-4)
