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
