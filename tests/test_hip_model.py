"""GPU parity test for the caller of the hot path: iaf_amd.CVAE1.forward -- the reference's whole model forward,
CVAE1._forward (tf_train.py:150-218): image scaling, x_enc (5x5 stride-2 conv), the IAFLayer stack with its downsampling layers,
h_top, x_dec (5x5 deconv) + clip, discretized_logistic, obj and the k-sample loss -- against the outputs of the reference's OWN
_forward executed on the TF shim (tests/golden/cvae1_forward.npz, tests/golden/make_golden_model.py) and against the CPU oracle's
restatement of it (oracle.cvae1_forward).  Also the edge entry points on their own against the oracle's leaf functions."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import iaf_oracle as O

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("name", sorted(gi.MODEL_CASES))
def test_cvae1_forward_vs_reference_golden(amd, golden_dir, name):
    g = np.load(os.path.join(golden_dir, "cvae1_forward.npz"))
    c = gi.model_case_inputs(name)
    model = amd.CVAE1(z_size=c["z_size"], h_size=c["h_size"], kl_min=c["kl_min"], depth=c["depth"], num_blocks=c["num_blocks"],
                      k=c["k"], image_size=c["image_size"], mode=c["mode"])
    model.load({k: dev(v) for k, v in c["params"].items()})
    x = torch.from_numpy(c["x"]).cuda()
    x_out, obj, loss = model.forward(x, [dev(e) for e in c["noise"]])
    # the reference's outputs were computed in fp64 from fp64 variables; the GPU holds fp32 copies of the same numbers
    np.testing.assert_allclose(host(x_out), g[name + "/x_out"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(host(obj)[0], g[name + "/obj"], rtol=2e-5)
    np.testing.assert_allclose(host(loss)[0], g[name + "/loss"], rtol=2e-5)
    np.testing.assert_allclose(model.bits_per_dim(host(loss)[0], c["B"]), g[name + "/bits_per_dim"], rtol=2e-5)


def test_edge_convs_vs_oracle(amd):
    """conv2d 5x5 stride 2 from 3 channels, deconv2d 5x5 stride 2 to 3 channels (+ elu, clip), odd sizes and a 3x3 stride-1 case"""
    lib, P, st = amd._capi.lib(), amd.layers._ptr, amd.layers._stream
    rng = np.random.RandomState(3)
    for (B, ci, co, H, W, k, s) in ((3, 3, 24, 10, 14, 5, 2), (2, 5, 7, 9, 7, 3, 1), (2, 3, 160, 32, 32, 5, 2)):
        p = gi.conv_params(rng, ci, co, ksize=k)
        x = rng.standard_normal((B, ci, H, W))
        w = torch.empty((k, k, ci, co), device="cuda")
        dV, dg, db, dx = dev(p["V"]), dev(p["g"]), dev(p["b"]), dev(x)          # (kept alive across the launches)
        amd._capi.check(lib.iaf_convk_weightnorm(P(dV), P(dg), P(w), k, k, ci, co, 0, st()))
        OH, OW = -(-H // s), -(-W // s)
        y = torch.empty((B, co, OH, OW), device="cuda")
        amd._capi.check(lib.iaf_convk_forward(P(dx), P(w), P(db), P(y), B, ci, H, W, co, k, k, s, 1, st()))
        want = O.conv2d(O.elu(f32(x)), f32(p["V"]), f32(p["g"]), f32(p["b"]), stride=(s, s))
        np.testing.assert_allclose(host(y), want, rtol=0, atol=1e-4)
    for (B, ci, co, H, W, k) in ((3, 24, 3, 5, 7, 5), (2, 160, 3, 16, 16, 5), (2, 6, 4, 4, 4, 3)):
        p = gi.deconv_params(rng, ci, co, k=k)
        x = rng.standard_normal((B, ci, H, W))
        w = torch.empty((k, k, co, ci), device="cuda")
        dV, dg, db, dx = dev(p["V"]), dev(p["g"]), dev(p["b"]), dev(x)
        amd._capi.check(lib.iaf_convk_weightnorm(P(dV), P(dg), P(w), k, k, ci, co, 1, st()))
        y = torch.empty((B, co, 2 * H, 2 * W), device="cuda")
        amd._capi.check(lib.iaf_deconvk_forward(P(dx), P(w), P(db), P(y), B, ci, H, W, co, k, k, 2, 1, -0.3, 0.4, st()))
        want = np.clip(O.deconv2d(O.elu(f32(x)), f32(p["V"]), f32(p["g"]), f32(p["b"])), np.float32(-0.3), np.float32(0.4))
        np.testing.assert_allclose(host(y), want, rtol=0, atol=1e-4)


def test_image_scaling_tile_and_sums(amd):
    lib, P, st = amd._capi.lib(), amd.layers._ptr, amd.layers._stream
    rng = np.random.RandomState(4)
    x = rng.randint(0, 256, size=(3, 3, 4, 4)).astype(np.uint8)
    out = torch.empty((6, 3, 4, 4), device="cuda")
    xd = torch.from_numpy(x).cuda()
    amd._capi.check(lib.iaf_image_to_float(xd.data_ptr(), P(out), 3, 48, 2, st()))
    want = O.repeat(np.clip((x.astype(np.float64) + 0.5) / 256.0, 0.0, 1.0) - 0.5, 2)        # tf_train.py:153-159
    np.testing.assert_allclose(host(out), want, rtol=0, atol=1e-7)
    v = rng.standard_normal(5)
    t = torch.empty((2, 5, 3, 3), device="cuda")
    vd = dev(v)
    amd._capi.check(lib.iaf_tile_channels(P(vd), P(t), 2, 5, 9, st()))
    np.testing.assert_array_equal(host(t), np.tile(f32(v).reshape(1, 5, 1, 1), [2, 1, 3, 3]))
    a, b = rng.standard_normal(1000), rng.standard_normal(1000)
    s = torch.empty(1, device="cuda")
    ad, bd = dev(a), dev(b)
    amd._capi.check(lib.iaf_sum_axpy(P(ad), P(bd), -1.0, P(s), 1000, st()))
    np.testing.assert_allclose(host(s)[0], np.sum(f32(a) - f32(b)), rtol=0, atol=1e-3)


def test_cvae1_forward_backward_vs_autograd_oracle(amd, golden_dir):
    """d obj / d EVERY variable of the model (what opt.compute_gradients(obj) computes, tf_train.py:128,211) against torch-fp64 autograd
    of the restated forward (oracle/iaf_grad_oracle.py:cvae1_grads -- its forward equals the reference's own _forward output to 1e-15
    and its gradients equal central finite differences of the NumPy oracle, tests/test_grad_oracle.py), at the BASELINE channel counts."""
    from oracle import iaf_grad_oracle as G
    name = "model_cfg"
    g = np.load(os.path.join(golden_dir, "cvae1_forward.npz"))
    c = gi.model_case_inputs(name)
    p32 = {k: f32(v) for k, v in c["params"].items()}
    noise32 = [f32(e) for e in c["noise"]]
    want, _, want_obj = G.cvae1_grads(c["x"], p32, c["z_size"], c["h_size"], c["depth"], c["num_blocks"], c["kl_min"], noise32)
    model = amd.CVAE1(z_size=c["z_size"], h_size=c["h_size"], kl_min=c["kl_min"], depth=c["depth"], num_blocks=c["num_blocks"], k=1,
                      image_size=c["image_size"])
    model.set_training(True)
    params = {k: dev(v) for k, v in c["params"].items()}
    model.load(params)
    x_out, obj, grads = model.forward_backward(torch.from_numpy(c["x"]).cuda(), [dev(e) for e in c["noise"]])
    np.testing.assert_allclose(host(x_out), g[name + "/x_out"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(host(obj)[0], want_obj, rtol=2e-5)
    assert set(grads) == set(want), sorted(set(want) ^ set(grads))
    worst = (0.0, None)
    for k in sorted(want):
        got, w = host(grads[k]), want[k]
        err = float(np.abs(got - w).max() / (np.abs(w).max() + 1e-3))
        worst = max(worst, (err, k))
        assert err < 2e-3, (k, err, float(np.abs(w).max()))
    print("worst relative gradient error %.2e (%s)" % worst)


@pytest.mark.parametrize("n_buckets", [2, 3, 8])
def test_cvae1_gradient_buckets_are_complete_at_their_segment_end_and_equal_one_bucket(amd, n_buckets):
    """bench.py --train --model runs the backward in 4 gradient buckets and issues each bucket's all-reduce the moment
    on_bucket(i) fires -- so (a) every variable of bucket i must already hold its FINAL gradient at that moment and (b) the bucketed
    backward must compute what the single-bucket one does (tf_train.py:128: one compute_gradients).  n_buckets = 2 * layers = 8: one
    bucket per layer and pass."""
    import iaf_amd.parallel as par
    c = gi.model_case_inputs("model_cfg")
    x = torch.from_numpy(c["x"]).cuda()
    noise = [dev(e) for e in c["noise"]]

    def run(nb):
        model = amd.CVAE1(z_size=c["z_size"], h_size=c["h_size"], kl_min=c["kl_min"], depth=c["depth"], num_blocks=c["num_blocks"], k=1,
                          image_size=c["image_size"])
        model.set_training(True)
        host_p = {k: dev(v) for k, v in c["params"].items()}
        model.load(host_p)
        flat = par.FlatParams({k: host_p[k] for k in model.completion_order()})
        model.load(flat.p)
        names = model.set_grad_buckets(nb)
        flat.grads.fill_(float("nan"))                          # whatever is read before it is written shows
        snaps = []

        def on_bucket(i):
            torch.cuda.synchronize()
            snaps.append({k: flat.g[k].clone() for k in names[i]})
        _, obj, _ = model.forward_backward(x, noise, grads=flat.g, on_bucket=on_bucket)
        torch.cuda.synchronize()
        assert len(snaps) == len(names)
        for i, snap in enumerate(snaps):                        # (a) complete when reported
            for k, v in snap.items():
                assert torch.isfinite(v).all(), (nb, i, k)
                assert torch.equal(v, flat.g[k]), (nb, i, k)
        assert sorted(k for g in names for k in g) == sorted(flat.g)       # every variable in exactly one bucket
        return {k: flat.g[k].clone() for k in flat.g}, float(obj.item())

    one, obj1 = run(1)
    many, objn = run(n_buckets)
    assert obj1 == objn
    for k in one:                                               # (b) the same numbers (same kernels, same order inside a layer)
        assert torch.equal(one[k], many[k]), k


def test_cvae1_init_pass_vs_the_references_own_init_branches(amd, golden_dir):
    """CVAE1.init_pass: the data-dependent initialisation of every conv of the model (x_enc, plain / strided / masked convs, deconvs,
    x_dec) against the reference's own _forward executed in mode "init" on the TF shim (tests/golden/cvae1_init.npz): all 68 g / b
    variables and the pass's output.  Moments over (N,H,W) in fp32 vs the reference's fp64."""
    g = np.load(os.path.join(golden_dir, "cvae1_init.npz"))
    name = "model_init"
    c = gi.model_case_inputs(name)
    model = amd.CVAE1(z_size=c["z_size"], h_size=c["h_size"], kl_min=c["kl_min"], depth=c["depth"], num_blocks=c["num_blocks"],
                      k=c["k"], image_size=c["image_size"], mode="init")
    params = {k: dev(v) for k, v in c["params"].items() if not (k.endswith("/g") or k.endswith("/b"))}
    x_out, pout = model.init_pass(torch.from_numpy(c["x"]).cuda(), params, [dev(e) for e in c["noise"]])
    want = {k[len(name + "/var/"):]: g[k] for k in g.files if k.startswith(name + "/var/")}
    assert set(k for k in pout if k.endswith("/g") or k.endswith("/b")) == set(want)
    worst = (0.0, None)
    for k in sorted(want):
        err = float(np.abs(host(pout[k]) - want[k]).max())
        worst = max(worst, (err, k))
        assert err < 2e-4, (k, err)
    np.testing.assert_allclose(host(x_out), g[name + "/x_out"], rtol=0, atol=2e-4)
    print("init pass: worst |g, b difference| %.2e (%s)" % worst)
    # the model is left loaded with the initialised variables: an ordinary forward runs
    model2 = amd.CVAE1(z_size=c["z_size"], h_size=c["h_size"], kl_min=c["kl_min"], depth=c["depth"], num_blocks=c["num_blocks"],
                       k=c["k"], image_size=c["image_size"])
    model2.load(pout)
    xo, obj, loss = model2.forward(torch.from_numpy(c["x"]).cuda(), [dev(e) for e in c["noise"]])
    assert np.isfinite(host(obj)).all() and np.isfinite(host(xo)).all()


def test_cvae1_init_then_training_steps_lower_the_objective(amd):
    """the reference's start of training end to end on one fixed batch: data-dependent init pass (CVAE1(hps, "init")), then steps of
    prepare_weights -> forward_backward (gradients into the flat buffer) -> fused Adamax + EMA (tf_train.py:128,146-159): the objective falls"""
    import iaf_amd.parallel as par
    c = gi.model_case_inputs("model_cfg")
    x = torch.from_numpy(c["x"]).cuda()
    noise = [dev(e) for e in c["noise"]]
    model = amd.CVAE1(z_size=c["z_size"], h_size=c["h_size"], kl_min=c["kl_min"], depth=c["depth"], num_blocks=c["num_blocks"], k=1,
                      image_size=c["image_size"])
    _, p0 = model.init_pass(x, {k: dev(v) for k, v in c["params"].items() if not (k.endswith("/g") or k.endswith("/b"))}, noise)
    flat = par.FlatParams({k: p0[k] for k in sorted(p0)})
    model.set_training(True)
    model.load(flat.p)
    objs = []
    for _ in range(12):
        model.prepare_weights()
        _, obj, _ = model.forward_backward(x, noise, grads=flat.g)
        objs.append(float(obj.item()))
        flat.adamax_ema_step(2e-3, world=1)
    assert all(np.isfinite(objs)), objs
    assert objs[-1] < objs[0] - 0.02 * abs(objs[0]), objs
    assert sum(1 for a, b in zip(objs, objs[1:]) if b < a) >= 9, objs


def test_cvae1_iw_eval_streamed_equals_the_k_sample_forward(amd):
    """CVAE1.iw_eval (k passes with k = 1, per-image terms streamed into the running log-sum-exp) == the loss of ONE forward with hps.k = k
    on the same noise (tf_train.py:159,218: images repeated k times, compute_lowerbound over the [n, k] matrix) -- BASELINE config 5's
    evaluation at model level; and == the CPU oracle"""
    c = gi.model_case_inputs("model_k2")
    k = 3
    rng = np.random.RandomState(5)
    B = c["B"]
    passes = [[rng.standard_normal((B,) + e.shape[1:]) for e in c["noise"]] for _ in range(k)]
    # the k-sample forward wants [B k, ...] noise, image-major / sample-minor (repeat(x, k), distributions.py:40-52)
    merged = [np.stack([passes[s][i] for s in range(k)], axis=1).reshape((B * k,) + passes[0][i].shape[1:]) for i in range(len(c["noise"]))]
    kw = dict(z_size=c["z_size"], h_size=c["h_size"], kl_min=c["kl_min"], depth=c["depth"], num_blocks=c["num_blocks"], image_size=c["image_size"])
    params = {n: dev(v) for n, v in c["params"].items()}
    x = torch.from_numpy(c["x"]).cuda()
    mk = amd.CVAE1(k=k, **kw)
    mk.load(params)
    _, _, loss_k = mk.forward(x, [dev(e) for e in merged])
    m1 = amd.CVAE1(k=1, **kw)
    m1.load(params)
    loss_s = m1.iw_eval(x, [[dev(e) for e in p] for p in passes])
    want = O.cvae1_forward(c["x"], {n: f32(v) for n, v in c["params"].items()}, c["z_size"], c["h_size"], c["depth"], c["num_blocks"],
                           c["kl_min"], k, [f32(e) for e in merged])[2]
    np.testing.assert_allclose(host(loss_k)[0], want, rtol=2e-5)
    np.testing.assert_allclose(host(loss_s)[0], want, rtol=2e-5)
