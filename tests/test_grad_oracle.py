"""The gradient oracle (torch fp64 autograd of the restated forward) must (1) reproduce the NumPy oracle's forward and
(2) agree with central finite differences.  CPU only."""
import numpy as np

import golden_inputs as gi
from oracle import iaf_grad_oracle as G
from oracle import iaf_oracle as O


def _case(seed, B=2, n_z=4, n_h=(8, 8), H=4, W=3):
    rng = np.random.RandomState(seed)
    params = gi.ar_multiconv2d_params(rng, n_z, list(n_h), [n_z, n_z])
    z = rng.standard_normal((B, n_z, H, W))
    ctx = rng.standard_normal((B, n_h[0], H, W))
    dzn = rng.standard_normal(z.shape)
    dls = rng.standard_normal(z.shape)
    return params, z, ctx, dzn, dls


def test_forward_matches_numpy_oracle():
    params, z, ctx, dzn, dls = _case(1)
    _, zn, ls = G.iaf_step_grads(z, ctx, params, [8, 8], dzn, dls)
    ez, es = O.iaf_step(z, ctx, params, [8, 8])
    np.testing.assert_allclose(zn, ez, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(ls, es, rtol=1e-11, atol=1e-12)


def test_gradients_match_finite_differences():
    params, z, ctx, dzn, dls = _case(2)
    grads, _, _ = G.iaf_step_grads(z, ctx, params, [8, 8], dzn, dls)

    def loss(zz, cc, pp):
        a, b = O.iaf_step(zz, cc, pp, [8, 8])
        return (a * dzn).sum() + (b * dls).sum()

    rng = np.random.RandomState(0)
    eps = 1e-6
    for name, base in [("z", z), ("context", ctx), ("layer_0/V", params["layer_0/V"]), ("layer_1/g", params["layer_1/g"]),
                       ("layer_out_0/V", params["layer_out_0/V"]), ("layer_out_1/b", params["layer_out_1/b"]),
                       ("layer_1/V", params["layer_1/V"])]:
        for _ in range(6):
            idx = tuple(rng.randint(0, s) for s in base.shape)

            def at(delta):
                arr = base.copy()
                arr[idx] += delta
                pp = dict(params)
                zz, cc = z, ctx
                if name == "z":
                    zz = arr
                elif name == "context":
                    cc = arr
                else:
                    pp[name] = arr
                return loss(zz, cc, pp)

            fd = (at(eps) - at(-eps)) / (2 * eps)
            assert abs(fd - grads[name][idx]) < 1e-6 * max(1.0, abs(fd)), (name, idx, fd, grads[name][idx])


def test_masked_weights_get_zero_gradient():
    """the mask multiplies V inside the graph (layers.py:57) -> dV is zero wherever the mask is"""
    params, z, ctx, dzn, dls = _case(3)
    grads, _, _ = G.iaf_step_grads(z, ctx, params, [8, 8], dzn, dls)
    for nm, zd in (("layer_0", False), ("layer_1", False), ("layer_out_0", True), ("layer_out_1", True)):
        V = params[nm + "/V"]
        mask = O.get_conv_ar_mask(3, 3, V.shape[2], V.shape[3], zd)
        assert np.abs(grads[nm + "/V"][mask == 0]).max() == 0.0


def test_posterior_block_forward_matches_numpy_oracle():
    rng = np.random.RandomState(5)
    B, Z, Hh, H, W = 3, 4, 8, 3, 3
    params = gi.ar_multiconv2d_params(rng, Z, [Hh, Hh], [Z, Z])
    f = lambda c: rng.standard_normal((B, c, H, W))
    inp = dict(qm=f(Z), ql=0.2 * f(Z), rm=f(Z), rl=0.2 * f(Z), pm=f(Z), pl=0.2 * f(Z), uc=f(Hh), dc=f(Hh), eps=f(Z))
    for kl_min in (0.0, 0.25):
        _, z, kl_obj, kl_cost = G.posterior_block_grads(inp, params, [Hh, Hh], kl_min, np.ones((B, Z, H, W)), np.ones(B))
        e = O.posterior_block(inp["qm"], inp["ql"], inp["rm"], inp["rl"], inp["pm"], inp["pl"], inp["uc"], inp["dc"],
                              inp["eps"], params, [Hh, Hh], kl_min)
        np.testing.assert_allclose(z, e["z"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(kl_obj, e["kl_obj"], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(kl_cost, e["kl_cost"], rtol=1e-11, atol=1e-11)


# ---------------------------------------------------------------- whole IAFLayer (plain convs + posterior block)
def test_layer_forward_matches_reference_golden_and_numpy_oracle(golden_dir):
    """pins the torch restatement of IAFLayer.up/.down that the layer-gradient oracle differentiates"""
    import os
    g = np.load(os.path.join(golden_dir, "iaf_layer.npz"))
    for name in ("layer_tiny_fb", "layer_tiny_nofb", "layer_k2"):
        c = gi.layer_case_inputs(name)
        zero = np.zeros_like
        _, fw = G.iaf_layer_grads(c["up_input"], c["down_input"], c["eps_post"], c["params"], c["z_size"], c["h_size"],
                                  c["kl_min"], zero(c["up_input"]), zero(c["down_input"]), np.zeros(c["up_input"].shape[0]))
        np.testing.assert_allclose(fw["up_out"], g[name + "/up_out"], rtol=1e-5, atol=1e-5)      # fixtures are fp32 runs
        np.testing.assert_allclose(fw["output"], g[name + "/output"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(fw["kl_obj"], g[name + "/kl_obj"], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(fw["kl_cost"], g[name + "/kl_cost"], rtol=1e-5, atol=1e-4)
        up_out, qm, ql, uc = O.iaf_layer_up(c["up_input"], c["params"], c["z_size"], c["h_size"])
        out, kl_obj, kl_cost, _ = O.iaf_layer_down(c["down_input"], c["params"], qm, ql, uc, c["eps_post"], c["z_size"],
                                                   c["h_size"], c["kl_min"])
        np.testing.assert_allclose(fw["up_out"], up_out, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(fw["output"], out, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(fw["kl_obj"], kl_obj, rtol=1e-11, atol=1e-10)


def test_layer_gradients_match_finite_differences():
    c = gi.layer_case_inputs("layer_tiny_fb")
    rng = np.random.RandomState(11)
    dU, dD = rng.standard_normal(c["up_input"].shape), rng.standard_normal(c["down_input"].shape)
    dK = rng.standard_normal(c["up_input"].shape[0])
    args = (c["z_size"], c["h_size"], c["kl_min"])
    grads, _ = G.iaf_layer_grads(c["up_input"], c["down_input"], c["eps_post"], c["params"], *args, dU, dD, dK)

    def loss(u, d, pp):
        up_out, qm, ql, uc = O.iaf_layer_up(u, pp, c["z_size"], c["h_size"])
        out, kl_obj, _, _ = O.iaf_layer_down(d, pp, qm, ql, uc, c["eps_post"], *args)
        return (up_out * dU).sum() + (out * dD).sum() + (kl_obj * dK).sum()

    eps = 1e-6
    for name in ["up_inp", "down_inp", "up_conv1/V", "up_conv1/g", "up_conv3/V", "down_conv1/V", "down_conv1/b",
                 "down_conv2/V", "down_conv2/g", "ar_multiconv2d/layer_0/V"]:
        base = {"up_inp": c["up_input"], "down_inp": c["down_input"]}.get(name, c["params"].get(name))
        got = grads[name] if name in grads else grads["params"][name]
        for _ in range(4):
            idx = tuple(rng.randint(0, s) for s in base.shape)

            def at(delta):
                arr = base.copy()
                arr[idx] += delta
                pp = dict(c["params"])
                u, d = c["up_input"], c["down_input"]
                if name == "up_inp":
                    u = arr
                elif name == "down_inp":
                    d = arr
                else:
                    pp[name] = arr
                return loss(u, d, pp)

            fd = (at(eps) - at(-eps)) / (2 * eps)
            assert abs(fd - got[idx]) < 2e-6 * max(1.0, abs(fd)), (name, idx, fd, got[idx])


# ---------------------------------------------------------------- the Theano statement (graphy/nodes/ar.py) of the same step
def _theano_case(seed, B=2, n_z=4, n_h=(8, 8), H=4, W=3):
    rng = np.random.RandomState(seed)
    nm = "c"
    w, sizes = {}, [n_z] + list(n_h)
    for i in range(len(n_h)):
        w["%s_%d_w" % (nm, i)] = 0.3 * rng.standard_normal((sizes[i + 1], sizes[i] + 1, 3, 3))
        w["%s_%d_b" % (nm, i)] = 0.1 * rng.standard_normal(sizes[i + 1])
        w["%s_%d_s" % (nm, i)] = 0.1 * rng.standard_normal(sizes[i + 1])
    for i in range(2):
        w["%s_out_%d_w" % (nm, i)] = 0.3 * rng.standard_normal((n_z, sizes[-1] + 1, 3, 3))
        w["%s_out_%d_b" % (nm, i)] = 0.1 * rng.standard_normal(n_z)
        w["%s_out_%d_s" % (nm, i)] = 0.1 * rng.standard_normal(n_z)
    z, ctx = rng.standard_normal((B, n_z, H, W)), rng.standard_normal((B, n_h[0], H, W))
    return nm, w, z, ctx, rng.standard_normal(z.shape), rng.standard_normal(z.shape)


import pytest  # noqa: E402


@pytest.mark.parametrize("flip", [False, True], ids=["plain", "flipmask"])
def test_theano_forward_matches_numpy_oracle(flip):
    nm, w, z, ctx, dzn, dls = _theano_case(11)
    _, zn, ls = G.theano_iaf2_nl_grads(z, ctx, w, nm, 4, [8, 8], dzn, dls, flipmask=flip)
    ez, es = O.theano_iaf2_nl(z, ctx, w, nm, 4, [8, 8], flipmask=flip)
    np.testing.assert_allclose(zn, ez, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(ls, es, rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("flip", [False, True], ids=["plain", "flipmask"])
def test_theano_gradients_match_finite_differences(flip):
    nm, w, z, ctx, dzn, dls = _theano_case(12)
    grads, _, _ = G.theano_iaf2_nl_grads(z, ctx, w, nm, 4, [8, 8], dzn, dls, flipmask=flip)

    def loss(zz, cc, ww):
        a, b = O.theano_iaf2_nl(zz, cc, ww, nm, 4, [8, 8], flipmask=flip)
        return (a * dzn).sum() + (b * dls).sum()

    rng = np.random.RandomState(0)
    eps = 1e-6
    for name, base in [("z", z), ("context", ctx)] + [(k, w[k]) for k in ("c_0_w", "c_1_w", "c_1_s", "c_out_0_w", "c_out_1_w",
                                                                           "c_out_1_b", "c_out_0_s")]:
        for _ in range(8):
            idx = tuple(rng.randint(0, s) for s in base.shape)

            def at(delta):
                arr = base.copy()
                arr[idx] += delta
                ww = dict(w)
                zz, cc = z, ctx
                if name == "z":
                    zz = arr
                elif name == "context":
                    cc = arr
                else:
                    ww[name] = arr
                return loss(zz, cc, ww)

            fd = (at(eps) - at(-eps)) / (2 * eps)
            assert abs(fd - grads[name][idx]) < 1e-6 * max(1.0, abs(fd)), (name, idx, fd, grads[name][idx])


def test_theano_masked_and_border_weight_gradients():
    """masked entries of w get an exact zero; the border channel's live taps do get a gradient (they are weights like any
    other, ar.py:288-296 + conv.py:71-83)"""
    nm, w, z, ctx, dzn, dls = _theano_case(13)
    for flip in (False, True):
        grads, _, _ = G.theano_iaf2_nl_grads(z, ctx, w, nm, 4, [8, 8], dzn, dls, flipmask=flip)
        for key, n_in, n_out, zd in (("c_0_w", 4, 8, False), ("c_1_w", 8, 8, False), ("c_out_0_w", 8, 4, True)):
            mask = O.theano_ar_mask(n_in, n_out, 3, zd, flip, True)
            assert (grads[key][mask == 0] == 0).all()
            live_border = grads[key][:, n_in][mask[:, n_in] == 1]
            assert np.abs(live_border).max() > 0


@pytest.mark.parametrize("cname", sorted(gi.CVAE_CASES))
def test_theano_cvae_iaf_forward_matches_reference_golden(golden_dir, cname):
    """the autograd oracle's statement of the layer's IAF part, fed the conv outputs of the NumPy oracle, reproduces what the
    reference's own models.py produced (kl, obj_kl) -- so its gradients are T.grad's"""
    import os
    posterior, B, n_h, n_z, depth_ar, H, W, kl_min = gi.CVAE_CASES[cname]
    g = np.load(os.path.join(golden_dir, "theano_cvae_layer.npz"))
    pre = cname + "/w_shape/"
    c = gi.cvae_case_inputs(cname, {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)})
    ref = O.theano_cvae_layer("1", posterior, c["w"], n_h, n_z, depth_ar, c["up_input"], c["down_input"], c["eps_up"], c["eps_down"])
    eps = c["eps_up"] if posterior == "up_iaf2_nl" else c["eps_down"]
    sw = {k: v for k, v in c["w"].items() if "_posterior_conv1_" in k}
    zero = lambda a: np.zeros_like(a)
    n_up = n_h + n_z if posterior == "up_iaf2_nl" else n_h
    _, fw = G.theano_cvae_iaf_grads(posterior, ref["up_conv1"], ref["down_conv1"], eps, sw, "1", n_h, n_z, depth_ar, kl_min,
                                    np.zeros((B, n_up, H, W)), np.zeros((B, n_h + n_z, H, W)),
                                    np.zeros(()) if kl_min > 0 else np.zeros(B))
    np.testing.assert_allclose(fw["kl"], g[cname + "/kl"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(fw["obj_kl"], g[cname + "/obj_kl"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(fw["h"][:, n_h:], ref["z"], rtol=1e-10, atol=1e-10)


# ---------------------------------------------------------------- the downsampling layer (tf_train.py:33,42-43,89-91)
def test_downsampling_layer_forward_matches_reference_golden(golden_dir):
    """the torch restatement the gradient oracle differentiates (stride-2 conv, resize 0.5 / 2, deconv2d) reproduces what the
    reference's own tf_train.IAFLayer(downsample=True) produced"""
    import os
    g = np.load(os.path.join(golden_dir, "iaf_layer_ds.npz"))
    for name in ("layer_ds_tiny", "layer_ds_cfg2"):
        c = gi.layer_ds_case_inputs(name)
        zero = np.zeros_like
        _, fw = G.iaf_layer_grads(c["up_input"], c["down_input"], c["eps_post"], c["params"], c["z_size"], c["h_size"], c["kl_min"],
                                  np.zeros_like(g[name + "/up_out"]), np.zeros_like(g[name + "/output"]),
                                  np.zeros(c["up_input"].shape[0]), downsample=True)
        np.testing.assert_allclose(fw["up_out"], g[name + "/up_out"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(fw["output"], g[name + "/output"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(fw["kl_obj"], g[name + "/kl_obj"], rtol=1e-9, atol=1e-8)


def test_downsampling_layer_gradients_match_finite_differences():
    c = gi.layer_ds_case_inputs("layer_ds_tiny")
    zs, hs, kl_min = c["z_size"], c["h_size"], c["kl_min"]
    rng = np.random.RandomState(12)
    B, _, H, W = c["up_input"].shape
    dU, dD = rng.standard_normal((B, hs, H // 2, W // 2)), rng.standard_normal((B, hs, H, W))
    dK = rng.standard_normal(B)
    grads, _ = G.iaf_layer_grads(c["up_input"], c["down_input"], c["eps_post"], c["params"], zs, hs, kl_min, dU, dD, dK,
                                 downsample=True)

    def loss(u, d, pp):
        up_out, qm, ql, uc = O.iaf_layer_up(u, pp, zs, hs, downsample=True)
        out, kl_obj, _, _ = O.iaf_layer_down(d, pp, qm, ql, uc, c["eps_post"], zs, hs, kl_min, downsample=True)
        return (up_out * dU).sum() + (out * dD).sum() + (kl_obj * dK).sum()

    eps = 1e-6
    for name in ["up_inp", "down_inp", "up_conv1/V", "up_conv1/g", "up_conv3/V", "down_conv1/V", "down_deconv2/V",
                 "down_deconv2/g", "down_deconv2/b", "ar_multiconv2d/layer_1/V"]:
        base = {"up_inp": c["up_input"], "down_inp": c["down_input"]}.get(name, c["params"].get(name))
        got = grads[name] if name in grads else grads["params"][name]
        for _ in range(5):
            idx = tuple(rng.randint(0, s) for s in base.shape)

            def at(delta):
                arr = base.copy()
                arr[idx] += delta
                pp = dict(c["params"])
                u, d = c["up_input"], c["down_input"]
                if name == "up_inp":
                    u = arr
                elif name == "down_inp":
                    d = arr
                else:
                    pp[name] = arr
                return loss(u, d, pp)

            fd = (at(eps) - at(-eps)) / (2 * eps)
            assert abs(fd - got[idx]) < 2e-6 * max(1.0, abs(fd)), (name, idx, fd, got[idx])


def test_model_gradient_oracle_forward_equals_reference_and_gradients_equal_finite_differences(golden_dir):
    """oracle/iaf_grad_oracle.py:cvae1_grads (torch fp64 autograd of the whole model's objective, tf_train.py:150-211): its forward
    reproduces the reference's own _forward outputs (tests/golden/cvae1_forward.npz) and its gradients equal central finite differences
    of the NumPy oracle's obj for entries of every kind of variable (x_enc, x_dec with its per-input-channel norm, h_top, dec_log_stdv,
    a deconv layer, a masked stack)."""
    import os
    import golden_inputs as gi
    g = np.load(os.path.join(golden_dir, "cvae1_forward.npz"))
    c = gi.model_case_inputs("model_tiny")
    args = (c["z_size"], c["h_size"], c["depth"], c["num_blocks"], c["kl_min"])
    grads, xo, obj = G.cvae1_grads(c["x"], c["params"], *args, c["noise"])
    np.testing.assert_allclose(xo, g["model_tiny/x_out"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(obj, g["model_tiny/obj"], rtol=1e-12)
    f = lambda p: O.cvae1_forward(c["x"], p, *args, 1, c["noise"])[1]
    rng = np.random.RandomState(0)
    for k in ("x_enc/V", "x_enc/g", "x_dec/V", "x_dec/g", "x_dec/b", "h_top", "dec_log_stdv", "IAF_1_0/down_deconv2/V", "IAF_0_1/up_conv1/g",
              "IAF_1_1/ar_multiconv2d/layer_1/V", "IAF_0_0/down_conv1/b"):
        v = np.array(c["params"][k], dtype=np.float64)
        idx = tuple(rng.randint(0, s) for s in v.shape)
        h = 1e-5
        vp, vm = v.copy(), v.copy()
        vp[idx] += h
        vm[idx] -= h
        fd = (f(dict(c["params"], **{k: vp})) - f(dict(c["params"], **{k: vm}))) / (2 * h)
        np.testing.assert_allclose(grads[k][idx], fd, rtol=2e-5, atol=1e-6, err_msg=k)
