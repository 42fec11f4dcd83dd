"""Pin the CPU oracle against reference outputs (tests/golden/*.npz, produced by executing the
reference's own Python through tests/golden/tf_shim.py) and against the reference's
known-answer tests (tf_utils/distributions_test.py:7-38).  CPU only."""
import os

import numpy as np
import pytest

import golden_inputs as gi
from oracle import iaf_oracle as O

TOL = dict(rtol=1e-10, atol=1e-11)   # fp64 restatement vs fp64 reference control flow


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


# ---------------------------------------------------------------- a1 / a2
def test_masks_match_reference(golden_dir):
    g = _load(golden_dir, "masks")
    for n_in, n_out, zd in gi.MASK_CASES:
        key = "%d_%d_%d" % (n_in, n_out, int(zd))
        np.testing.assert_array_equal(O.get_linear_ar_mask(n_in, n_out, zd), g["lin_" + key].astype(np.float32))
        np.testing.assert_array_equal(O.get_conv_ar_mask(3, 3, n_in, n_out, zd), g["conv_" + key].astype(np.float32))
    np.testing.assert_array_equal(O.get_conv_ar_mask(5, 5, 8, 16, False), g["conv5x5_8_16_0"].astype(np.float32))


def test_mask_live_mac_count_config2():
    # live MAC/pixel 23,120 + 115,280 + 2*22,960 = 184,320 of 368,640 dense (exactly 50 %)
    live = (O.get_conv_ar_mask(3, 3, 32, 160, False).sum() + O.get_conv_ar_mask(3, 3, 160, 160, False).sum()
            + 2 * O.get_conv_ar_mask(3, 3, 160, 32, True).sum())
    assert int(live) == 184320
    assert int(O.get_conv_ar_mask(3, 3, 32, 160, False).sum()) == 23120
    assert int(O.get_conv_ar_mask(3, 3, 160, 160, False).sum()) == 115280
    assert int(O.get_conv_ar_mask(3, 3, 160, 32, True).sum()) == 22960


def test_mask_rejects_non_multiple():
    with pytest.raises(AssertionError):
        O.get_linear_ar_mask(64, 160)        # n_h=160 invalid with n_z=64


# ---------------------------------------------------------------- a3 - a5, IAF step
@pytest.mark.parametrize("name", sorted(gi.AR_CASES))
def test_ar_multiconv2d_matches_reference(golden_dir, name):
    g = _load(golden_dir, "ar_multiconv2d")
    c = gi.ar_case_inputs(name)
    m_raw, s_raw = O.ar_multiconv2d(c["z"], c["context"], c["params"], c["n_h"], [c["n_z"]] * 2)
    np.testing.assert_allclose(m_raw, g[name + "/m_raw"], **TOL)
    np.testing.assert_allclose(s_raw, g[name + "/s_raw"], **TOL)
    z_new, logsd = O.iaf_step(c["z"], c["context"], c["params"], c["n_h"])
    np.testing.assert_allclose(z_new, g[name + "/z_new"], **TOL)
    np.testing.assert_allclose(logsd, g[name + "/logsd"], **TOL)


def test_data_dependent_init_matches_reference(golden_dir):
    g = _load(golden_dir, "init_ar_conv")
    y, gg, b = O.ar_conv2d_init(g["x"], g["V0"], init_scale=0.7, zerodiagonal=False)
    np.testing.assert_allclose(y, g["y"], **TOL)
    np.testing.assert_allclose(gg, g["g"], **TOL)
    np.testing.assert_allclose(b, g["b"], **TOL)


# ---------------------------------------------------------------- a6 - a8 IAFLayer
@pytest.mark.parametrize("name", sorted(gi.LAYER_CASES))
def test_iaf_layer_matches_reference(golden_dir, name):
    g = _load(golden_dir, "iaf_layer")
    c = gi.layer_case_inputs(name)
    up_out, qz_mean, qz_logsd, up_context = O.iaf_layer_up(c["up_input"], c["params"], c["z_size"], c["h_size"])
    np.testing.assert_allclose(up_out, g[name + "/up_out"], **TOL)
    np.testing.assert_allclose(qz_mean, g[name + "/qz_mean"], **TOL)
    np.testing.assert_allclose(qz_logsd, g[name + "/qz_logsd"], **TOL)
    np.testing.assert_allclose(up_context, g[name + "/up_context"], **TOL)
    output, kl_obj, kl_cost, _ = O.iaf_layer_down(c["down_input"], c["params"], qz_mean, qz_logsd, up_context,
                                                  c["eps_post"], c["z_size"], c["h_size"], c["kl_min"])
    np.testing.assert_allclose(output, g[name + "/output"], **TOL)
    np.testing.assert_allclose(kl_obj, g[name + "/kl_obj"], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(kl_cost, g[name + "/kl_cost"], rtol=1e-10, atol=1e-9)


# ---------------------------------------------------------------- a6 / a9 distributions
def test_distributions_match_reference(golden_dir):
    g = _load(golden_dir, "distributions")
    s = O.gaussian_diag_sample(g["mean"], g["logvar"], g["eps"])
    np.testing.assert_allclose(s, g["sample"], **TOL)
    np.testing.assert_allclose(O.gaussian_diag_logps(g["mean"], g["logvar"], s), g["logps_self"], **TOL)
    np.testing.assert_allclose(O.gaussian_diag_logps(g["mean"], g["logvar"], g["other"]), g["logps_other"], **TOL)
    np.testing.assert_allclose(O.gaussian_diag_logps(g["mean"], g["logvar"], g["other"]), g["logps_fn"], **TOL)
    np.testing.assert_allclose(O.logsumexp(g["lse_x"]), g["lse"], **TOL)
    for k in (1, 4, 12):
        np.testing.assert_allclose(O.compute_lowerbound(g["lb_log_pxz"], g["lb_kl"], k), g["lb_k%d" % k], **TOL)
    np.testing.assert_array_equal(O.repeat(g["rep_x"], 3), g["rep_3"])
    np.testing.assert_allclose(O.discretized_logistic(g["dl_mean"], -1.3, g["dl_sample"]), g["dl_logp"], **TOL)


def test_reference_known_answer_tests(golden_dir):
    """tf_utils/distributions_test.py:7-38, expected values recomputed the way the test does."""
    g = _load(golden_dir, "distributions")
    a10 = np.arange(10.0)
    res = np.log(np.sum(np.exp(a10)))                                         # :9
    assert abs(O.logsumexp(a10.reshape([1, -1]))[0] - res) < 1e-12             # :12-13
    assert abs(g["kat_lse_arange10"][0] - res) < 1e-12
    a = np.log(np.array([0.3, 0.3, 0.3, 0.3])).reshape([1, -1])
    b = np.log(np.array([0.1, 0.5, 0.9, 0.6])).reshape([1, -1])
    res = -(-np.log(4) + np.log(np.sum(np.exp(a - b))))                       # :19
    assert abs(np.sum(O.compute_lowerbound(a, b, 4)) - res) < 1e-4             # :21-22 (places=4)
    assert abs(np.sum(g["kat_lb_k4"]) - res) < 1e-4
    res = (b - a).sum()                                                       # :28
    assert abs(np.sum(O.compute_lowerbound(a.reshape([-1, 1]), b.reshape([-1, 1]), 1)) - res) < 1e-4   # :30-31
    assert abs(np.sum(g["kat_lb_k1"]) - res) < 1e-4
    x = np.random.RandomState(0).randn(10, 5, 2)
    np.testing.assert_allclose(O.repeat(x, 2), np.repeat(x, 2, axis=0))       # :33-38


def test_streaming_lowerbound_equals_reference_formula():
    rng = np.random.RandomState(3)
    n, k = 7, 1000
    w = -40 + 6 * rng.standard_normal((n, k))
    ref = O.compute_lowerbound(w.reshape(-1), np.zeros(n * k), k)
    chunks = [w[:, i:i + 64] for i in range(0, k, 64)]
    np.testing.assert_allclose(O.streaming_lowerbound(chunks, k), ref, rtol=1e-12, atol=1e-12)


# ---------------------------------------------------------------- a13
def test_split_and_average_grads_match_reference(golden_dir):
    g = _load(golden_dir, "common")
    parts = O.split_channels(g["split_x"], [2, 2, 4, 4])
    for i, p in enumerate(parts):
        np.testing.assert_array_equal(p, g["split_%d" % i])
    towers = [[g["tower%d_g%d" % (t, i)] for i in range(3)] for t in range(4)]
    avg = O.average_grads(towers)
    for i in range(3):
        np.testing.assert_allclose(avg[i], g["avg_g%d" % i], rtol=1e-13, atol=1e-15)


def test_adamax_slot_semantics():
    # adamax.py:49-55: "v" first moment, "m" infinity norm
    var, grad = np.array([1.0, -2.0]), np.array([0.5, -0.25])
    v1, m1, vv1 = O.adamax_step(var, grad, np.zeros(2), np.zeros(2), lr=0.01)
    np.testing.assert_allclose(vv1, 0.1 * grad)
    np.testing.assert_allclose(m1, np.abs(grad))
    np.testing.assert_allclose(v1, var - 0.01 * 0.1 * np.sign(grad))


# ---------------------------------------------------------------- a10 / a11: the Theano statement, pinned
# tests/golden/theano_ar.npz holds outputs of the reference's OWN Theano-side source (graphy/nodes/ar.py, conv.py,
# __init__.py, rand.py) executed eagerly on tests/golden/theano_shim.py after an in-memory lib2to3 pass
# (tests/golden/make_golden_theano.py).
@pytest.mark.parametrize("cname", sorted(gi.THEANO_CASES))
def test_theano_multiconv2d_matches_reference(golden_dir, cname):
    g = _load(golden_dir, "theano_ar")
    B, n_z, n_h, H, W, flip = gi.THEANO_CASES[cname]
    w, z, ctx = gi.theano_case_inputs(cname)
    m_raw, s_raw = O.theano_multiconv2d(z, ctx, w, gi.THEANO_NAME, n_z, n_h, [n_z, n_z], flipmask=flip)
    np.testing.assert_allclose(m_raw, g[cname + "/m_raw"], **TOL)
    np.testing.assert_allclose(s_raw, g[cname + "/s_raw"], **TOL)


def test_theano_single_conv_pad_channel_and_gaussian_match_reference(golden_dir):
    g = _load(golden_dir, "theano_ar")
    for zd in (False, True):
        k = "conv_zd%d" % int(zd)
        y = O.theano_ar_conv2d(g[k + "/x"], g[k + "/w"], g[k + "/b"], g[k + "/s"], 8, 16, zerodiagonal=zd)
        np.testing.assert_allclose(y, g[k + "/y"], **TOL)                                      # ar.py:200-375
    np.testing.assert_array_equal(O.theano_pad2dwithchannel(g["pad/x"]), g["pad/y"])         # conv.py:71-83
    np.testing.assert_allclose(O.gaussian_diag_logps(g["gauss/mean"], g["gauss/logvar"], g["gauss/sample"]),
                               g["gauss/logps"], **TOL)                                       # rand.py:78-87


# ---------------------------------------------------------------- a13 / 8f-2: Adamax, pinned
def test_adamax_matches_reference(golden_dir):
    """tests/golden/adamax.npz: six steps of the reference's own tf_utils/adamax.py (AdamaxOptimizer._apply_dense,
    executed unmodified on NumPy-backed variable stubs, tests/golden/make_golden_adamax.py)"""
    g = _load(golden_dir, "adamax")
    var, m, v = g["var0"].copy(), np.zeros_like(g["var0"]), np.zeros_like(g["var0"])
    for t in range(g["grads"].shape[0]):
        var, m, v = O.adamax_step(var, g["grads"][t], m, v, float(g["lr"]))
        np.testing.assert_array_equal(var, g["var_%d" % t])
        np.testing.assert_array_equal(m, g["m_%d" % t])
        np.testing.assert_array_equal(v, g["v_%d" % t])


# ---------------------------------------------------------------- downsampling IAFLayer, init / sample modes
def test_deconv2d_and_resize_vs_reference_golden(golden_dir):
    """tf_utils/layers.py:67-112 (deconv2d incl. its over-(kh,kw,out) weight norm) and 169-175 (resize_nearest_neighbor)"""
    g = np.load(os.path.join(golden_dir, "iaf_layer_ds.npz"))
    y = O.deconv2d(g["deconv/x"], g["deconv/V"], g["deconv/g"], g["deconv/b"])
    np.testing.assert_allclose(y, g["deconv/y"], rtol=1e-10, atol=1e-12)
    np.testing.assert_array_equal(O.resize_nearest_neighbor(g["resize/x"], 0.5), g["resize/half"])
    np.testing.assert_array_equal(O.resize_nearest_neighbor(g["resize/x"], 2), g["resize/double"])


@pytest.mark.parametrize("name", sorted(gi.LAYER_DS_CASES))
def test_iaf_layer_downsample_and_modes_vs_reference_golden(golden_dir, name):
    """tf_train.IAFLayer.up/.down with downsample=True (stride-2 up_conv1, resize 0.5 / 2, down_deconv2) and modes
    "init" / "sample" (tf_train.py:33,42-43,60-66,89-91), executed from the reference's own source"""
    g = np.load(os.path.join(golden_dir, "iaf_layer_ds.npz"))
    c = gi.layer_ds_case_inputs(name)
    zs, hs = c["z_size"], c["h_size"]
    up_out, qz_mean, qz_logsd, up_context = O.iaf_layer_up(c["up_input"], c["params"], zs, hs, downsample=c["downsample"])
    for k, v in (("up_out", up_out), ("qz_mean", qz_mean), ("qz_logsd", qz_logsd), ("up_context", up_context)):
        np.testing.assert_allclose(v, g[name + "/" + k], rtol=1e-10, atol=1e-12)
    out, kl_obj, kl_cost, _ = O.iaf_layer_down(c["down_input"], c["params"], qz_mean, qz_logsd, up_context, c["eps_post"], zs, hs,
                                               c["kl_min"], mode=c["mode"], downsample=c["downsample"], eps_prior=c["eps_prior"])
    np.testing.assert_allclose(out, g[name + "/output"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(kl_obj, g[name + "/kl_obj"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(kl_cost, g[name + "/kl_cost"], rtol=1e-9, atol=1e-9)


# ---------------------------------------------------------------- Theano: flipmask and the whole cvae_layer
@pytest.mark.parametrize("key,zd,flip,n_in,n_out", [("conv_zd0_flip1_16_32", False, True, 16, 32), ("conv_zd1_flip1_16_32", True, True, 16, 32),
                                                    ("conv_zd1_flip1_32_16", True, True, 32, 16), ("conv_zd0_flip0_32_16", False, False, 32, 16)])
def test_theano_single_conv_flipmask_vs_reference_golden(golden_dir, key, zd, flip, n_in, n_out):
    g = np.load(os.path.join(golden_dir, "theano_ar.npz"))
    y = O.theano_ar_conv2d(g[key + "/x"], g[key + "/w"], g[key + "/b"], g[key + "/s"], n_in, n_out, zd, flip)
    np.testing.assert_allclose(y, g[key + "/y"], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("cname", sorted(k for k, v in gi.THEANO_CASES.items() if v[5]))
def test_theano_multiconv2d_flipmask_vs_reference_golden(golden_dir, cname):
    g = np.load(os.path.join(golden_dir, "theano_ar.npz"))
    B, n_z, n_h, H, W, flip = gi.THEANO_CASES[cname]
    w, z, ctx = gi.theano_case_inputs(cname)
    m, s = O.theano_multiconv2d(z, ctx, w, gi.THEANO_NAME, n_z, n_h, [n_z, n_z], flipmask=True)
    np.testing.assert_allclose(m, g[cname + "/m_raw"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(s, g[cname + "/s_raw"], rtol=1e-10, atol=1e-12)


def _cvae_case(golden_dir, cname):
    g = np.load(os.path.join(golden_dir, "theano_cvae_layer.npz"))
    pre = cname + "/w_shape/"
    shapes = {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}
    return g, gi.cvae_case_inputs(cname, shapes)


@pytest.mark.parametrize("cname", sorted(gi.CVAE_CASES))
def test_theano_cvae_layer_vs_reference_golden(golden_dir, cname):
    """models.cvae_layer.up / .down_q executed from the reference's own models.py (make_golden_theano.py: gen_cvae_layers)"""
    posterior, B, n_h, n_z, depth_ar, H, W, kl_min = gi.CVAE_CASES[cname]
    g, c = _cvae_case(golden_dir, cname)
    r = O.theano_cvae_layer("1", posterior, c["w"], n_h, n_z, depth_ar, c["up_input"], c["down_input"], c["eps_up"], c["eps_down"])
    np.testing.assert_allclose(r["up_out"], g[cname + "/up_out"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(r["down_out"], g[cname + "/down_out"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(r["kl"], g[cname + "/kl"], rtol=1e-9, atol=1e-9)
    obj, kl_sum = O.theano_free_bits(r["kl"], kl_min)
    np.testing.assert_allclose(kl_sum, g[cname + "/kl_sum"], rtol=1e-10)
    np.testing.assert_allclose(obj, g[cname + "/obj_kl"], rtol=1e-10)


@pytest.mark.parametrize("name", sorted(gi.MODEL_CASES))
def test_cvae1_forward_vs_the_references_own_forward(golden_dir, name):
    """the whole model forward, oracle.cvae1_forward, against CVAE1._forward of the reference executed on the TF shim
    (tests/golden/make_golden_model.py -> cvae1_forward.npz): x_enc, the layer stack with its downsampling layers, h_top,
    x_dec + clip, discretized_logistic, obj and the k-sample loss (tf_train.py:150-218)"""
    g = np.load(os.path.join(golden_dir, "cvae1_forward.npz"))
    c = gi.model_case_inputs(name)
    xo, obj, loss = O.cvae1_forward(c["x"], c["params"], c["z_size"], c["h_size"], c["depth"], c["num_blocks"], c["kl_min"], c["k"],
                                    c["noise"], mode=c["mode"])
    np.testing.assert_allclose(xo, g[name + "/x_out"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(obj, g[name + "/obj"], rtol=1e-12)
    np.testing.assert_allclose(loss, g[name + "/loss"], rtol=1e-12)
    bpd = loss / (np.log(2.) * 3 * c["image_size"] ** 2 * c["B"])                    # tf_train.py:133, one tower
    np.testing.assert_allclose(bpd, g[name + "/bits_per_dim"], rtol=1e-12)
