"""The bf16x3 forward kernels (iaf_conv_bf3.hpp: every fp32 operand split into three bf16 parts, six part-products
accumulated in fp32 on the bf16 matrix cores) against the exact-fp32 MFMA kernels and the fp64 oracle.

The default precision of an ARStack is "bf16x3", so every other GPU parity test already runs it; this file pins down
(a) that the bf16x3 kernels are really the ones running for the BASELINE shapes, (b) that their error against the fp64
oracle is fp32-grade -- not larger than the exact-fp32 MFMA chain's, far inside north_star's 1e-4 -- (c) that every
compiled bf16x3 launch shape computes the same conv, and (d) keeps the exact-fp32 kernels covered now that they are no
longer the default."""
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
    iaf_amd._capi.lib()
    return iaf_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy().astype(np.float64)


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def _case(seed, B, n_z, n_h, d, H, W):
    rng = np.random.RandomState(seed)
    params = gi.ar_multiconv2d_params(rng, n_z, [n_h] * d, [n_z, n_z])
    return params, rng.standard_normal((B, n_z, H, W)), rng.standard_normal((B, n_h, H, W))


def _pin_bf3(st, n_z, n_h, d, ppw, pxt, ks):
    nt_h = [n for n in (5, 4, 2) if (n_h // 16) % n == 0][0]
    nt_o = [n for n in (4, 2) if (2 * n_z // 16) % n == 0][0]
    for layer in range(d):
        st.set_tuning_bf3(layer, nt_h, ppw, pxt, ks)
    st.set_tuning_bf3(d, nt_o, ppw, pxt, ks)


SHAPES = [
    (32, 32, 160, 2, 16, 16),    # BASELINE config 2, both levels
    (32, 32, 160, 2, 8, 8),
    (16, 32, 64, 1, 16, 16),     # config 1
    (32, 64, 64, 4, 8, 8),       # config 4
    (32, 64, 128, 4, 16, 16),
    (8, 64, 192, 4, 16, 16),
    (64, 32, 160, 2, 16, 16),    # towards config 5: two rounds of workgroups
    (3, 32, 160, 2, 5, 7),       # ragged pixel count
    (2, 32, 96, 2, 4, 4),        # 6 co tiles (nt = 2)
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "B%d_z%d_h%d_d%d_%dx%d" % s)
def test_bf16x3_error_is_fp32_grade(amd, shape):
    B, n_z, n_h, d, H, W = shape
    params, z, ctx = _case(300 + H + n_h, *shape)
    dp = {k: dev(v) for k, v in params.items()}
    p32 = {k: f32(v) for k, v in params.items()}
    em, es = [], []
    for b0 in range(0, B, 16):
        m_, s_ = O.ar_multiconv2d(f32(z[b0:b0 + 16]), f32(ctx[b0:b0 + 16]), p32, [n_h] * d, [n_z, n_z])
        em.append(m_); es.append(s_)
    em, es = np.concatenate(em), np.concatenate(es)
    err = {}
    for prec in ("bf16x3", "f32"):
        st = amd.ARStack(n_z, [n_h] * d)
        st.set_precision(prec)
        st.prepare(dp)
        if prec == "bf16x3":       # pin a bf16x3 shape: small launches would otherwise stay on the fp32 kernel
            _pin_bf3(st, n_z, n_h, d, 2, 1, 4)
        for layer in range(d + 1):
            assert st.layer_precision(layer, B, H, W) == prec, "layer %d runs %s" % (layer, st.layer_precision(layer, B, H, W))
        m_raw, s_raw = st.ar_multiconv2d(dev(z), dev(ctx))
        err[prec] = max(np.abs(host(m_raw) - em).max(), np.abs(host(s_raw) - es).max())
        z_new, logsd = st.iaf_step(dev(z), dev(ctx))
        ez = (f32(z) - 0.1 * em) / np.exp(0.1 * es)
        np.testing.assert_allclose(host(z_new), ez, atol=ATOL, rtol=0)
        np.testing.assert_allclose(host(logsd), 0.1 * es, atol=ATOL, rtol=0)
    print("max |raw conv output - fp64 oracle|: bf16x3 %.3g, exact-fp32 MFMA %.3g" % (err["bf16x3"], err["f32"]))
    assert err["f32"] < 2e-5 and err["bf16x3"] < 2e-5                    # both two orders inside the 1e-4 bar ...
    assert err["bf16x3"] <= 2.0 * err["f32"] + 1e-6                      # ... and the split products are no worse


BF3_SHAPES = [(4, 1, 4, 1), (2, 1, 4, 1), (1, 1, 4, 1), (1, 4, 1, 1), (2, 1, 4, 2), (1, 1, 4, 2)]


@pytest.mark.parametrize("shp", BF3_SHAPES, ids=lambda s: "ppw%d_pxt%d_ks%d_wco%d" % s)
@pytest.mark.parametrize("cfg", [(5, 32, 160, 2, 8, 8), (3, 64, 128, 4, 5, 7), (32, 32, 160, 2, 16, 16)],
                         ids=lambda s: "B%d_z%d_h%d_d%d_%dx%d" % s)
def test_every_bf16x3_launch_shape_agrees(amd, shp, cfg):
    B, n_z, n_h, d, H, W = cfg
    ppw, pxt, ks, wco = shp
    params, z, ctx = _case(41, *cfg)
    st = amd.ARStack(n_z, [n_h] * d)
    st.prepare({k: dev(v) for k, v in params.items()})
    hid_tiles, out_tiles = n_h // 16, 2 * n_z // 16
    nt_h = [n for n in (5, 4, 2) if hid_tiles % (n * wco) == 0][0]
    nt_o = [n for n in (4, 2) if out_tiles % (n * wco) == 0][0]
    try:
        for layer in range(d):
            st.set_tuning_bf3(layer, nt_h, ppw, pxt, ks, wco)
        st.set_tuning_bf3(d, nt_o, ppw, pxt, ks, wco)
        z_new, logsd = st.iaf_step(dev(z), dev(ctx))
    except amd.UnsupportedError as e:
        pytest.skip(str(e))
    p32 = {k: f32(v) for k, v in params.items()}
    ez, es = O.iaf_step(f32(z), f32(ctx), p32, [n_h] * d)
    np.testing.assert_allclose(host(logsd), es, atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(z_new), ez, atol=ATOL, rtol=0)
    assert np.abs(host(logsd) - es).max() < 1e-5


@pytest.mark.parametrize("H", [16, 8])
def test_bf16x3_batch_independence_and_ar_structure_bit_exact(amd, H):
    """same launch shape -> running samples one at a time reproduces the batched result bit for bit; and a perturbation
    of z at (pixel q, channel c) leaves every position that precedes it in the IAF ordering bit-identical (the masked
    weights are exactly zero in all three bf16 planes)"""
    B, n_z, n_h, d = 32, 32, 160, 2
    params, z, ctx = _case(2024 + H, B, n_z, n_h, d, H, H)
    st = amd.ARStack(n_z, [n_h] * d)
    st.prepare({k: dev(v) for k, v in params.items()})
    st.set_tuning_bf3(0, 5, 1, 4, 1)
    st.set_tuning_bf3(1, 5, 1, 1, 4)
    st.set_tuning_bf3(2, 2, 1, 1, 4)
    zd, cd = dev(z), dev(ctx)
    zf, sf = st.iaf_step(zd, cd)
    for b in (0, 13, 31):
        zb, sb = st.iaf_step(zd[b:b + 1].contiguous(), cd[b:b + 1].contiguous())
        assert torch.equal(zb, zf[b:b + 1]) and torch.equal(sb, sf[b:b + 1])
    qh, qw, c = H // 2, H // 2 - 1, 11
    z2 = zd.clone()
    z2[:, c, qh, qw] += 0.5
    allowed = torch.zeros(zd.shape, dtype=torch.bool, device="cuda")
    allowed[:, :, :qh, :] = True
    allowed[:, :, qh, :qw] = True
    allowed_s = allowed.clone()
    allowed_s[:, c + 1:, qh, qw] = True
    allowed_z = allowed_s.clone()
    allowed_z[:, c, qh, qw] = True
    z1, s1 = st.iaf_step(z2, cd)
    assert int(((z1 != zf) & ~allowed_z).sum()) == 0 and int(((s1 != sf) & ~allowed_s).sum()) == 0
    assert bool((s1 != sf).any())


@pytest.mark.parametrize("kl_min", [0.0, 0.25])
def test_posterior_block_both_precisions_vs_oracle(amd, kl_min):
    """the extended unit (IN_POSTERIOR staging + posterior epilogue) through both kernel families"""
    B, n_z, n_h, d, H, W = 8, 32, 160, 2, 8, 8
    rng = np.random.RandomState(55)
    params = gi.ar_multiconv2d_params(rng, n_z, [n_h] * d, [n_z, n_z])
    f = lambda c: rng.standard_normal((B, c, H, W))
    qm, ql, rm, rl, pm, pl = f(n_z), 0.25 * f(n_z), f(n_z), 0.25 * f(n_z), f(n_z), 0.25 * f(n_z)
    uc, dc, eps = f(n_h), f(n_h), f(n_z)
    e = O.posterior_block(f32(qm), f32(ql), f32(rm), f32(rl), f32(pm), f32(pl), f32(uc), f32(dc), f32(eps),
                          {k: f32(v) for k, v in params.items()}, [n_h] * d, kl_min)
    for prec in ("bf16x3", "f32"):
        st = amd.ARStack(n_z, [n_h] * d)
        st.set_precision(prec)
        st.prepare({k: dev(v) for k, v in params.items()})
        if prec == "bf16x3":
            _pin_bf3(st, n_z, n_h, d, 1, 1, 4)
        out = st.posterior_block(dev(qm), dev(ql), dev(rm), dev(rl), dev(pm), dev(pl), dev(uc), dev(dc), dev(eps),
                                 kl_min, want_kl_elem=True)
        np.testing.assert_allclose(host(out["z"]), e["z"], atol=ATOL, rtol=0)
        np.testing.assert_allclose(host(out["kl_elem"]), e["logqs"] - e["logps"], atol=ATOL, rtol=1e-5)
        np.testing.assert_allclose(host(out["kl_cost"]), e["kl_cost"], atol=2e-3, rtol=1e-4)
        np.testing.assert_allclose(host(out["kl_obj"]), e["kl_obj"], atol=2e-3, rtol=1e-4)


def test_theano_statement_both_precisions(amd):
    """flipped taps (halo before the tile) + border-indicator epilogue through the bf16x3 kernels"""
    B, n_z, n_h, d, H, W = 4, 32, 160, 2, 8, 8
    rng = np.random.RandomState(404)
    nm = "1_posterior_conv1"
    w = {}
    sizes = [n_z] + [n_h] * d
    for i in range(d):
        w["%s_%d_w" % (nm, i)] = 0.05 * rng.standard_normal((sizes[i + 1], sizes[i] + 1, 3, 3))
        w["%s_%d_b" % (nm, i)] = 0.1 * rng.standard_normal(sizes[i + 1])
        w["%s_%d_s" % (nm, i)] = 0.1 * rng.standard_normal(sizes[i + 1])
    for i in range(2):
        w["%s_out_%d_w" % (nm, i)] = 0.05 * rng.standard_normal((n_z, sizes[-1] + 1, 3, 3))
        w["%s_out_%d_b" % (nm, i)] = 0.1 * rng.standard_normal(n_z)
        w["%s_out_%d_s" % (nm, i)] = 0.1 * rng.standard_normal(n_z)
    z, ctx = rng.standard_normal((B, n_z, H, W)), rng.standard_normal((B, n_h, H, W))
    w32 = {k: f32(v) for k, v in w.items()}
    em, es = O.theano_multiconv2d(f32(z), f32(ctx), w32, nm, n_z, [n_h] * d, [n_z, n_z])
    for prec in ("bf16x3", "f32"):
        conv = amd.multiconv2d(nm, n_z, [n_h] * d, [n_z, n_z], (3, 3), False, nl="elu", w=None)
        conv.stack.set_precision(prec)
        if prec == "bf16x3":
            _pin_bf3(conv.stack, n_z, n_h, d, 1, 1, 4)
        m_raw, s_raw = conv(dev(z), dev(ctx), {k: dev(v) for k, v in w.items()})
        assert conv.stack.layer_precision(1, B, H, W) == prec
        np.testing.assert_allclose(host(m_raw), em, atol=ATOL, rtol=0)
        np.testing.assert_allclose(host(s_raw), es, atol=ATOL, rtol=0)


@pytest.mark.parametrize("shape", [(4, 32, 160, 2, 16, 16), (4, 32, 160, 2, 8, 8), (3, 32, 64, 1, 16, 16), (2, 64, 64, 4, 8, 8),
                                   (3, 32, 160, 2, 5, 7), (32, 32, 160, 2, 16, 16)], ids=lambda s: "B%d_z%d_h%d_d%d_%dx%d" % s)
def test_exact_f32_kernels_vs_oracle(amd, shape):
    """the exact-fp32 MFMA path, selected explicitly (it is what the data gradients and the plain convs run)"""
    B, n_z, n_h, d, H, W = shape
    params, z, ctx = _case(100 + B + H, *shape)
    st = amd.ARStack(n_z, [n_h] * d)
    st.set_precision("f32")
    st.prepare({k: dev(v) for k, v in params.items()})
    z_new, logsd = st.iaf_step(dev(z), dev(ctx))
    ez, es = O.iaf_step(f32(z), f32(ctx), {k: f32(v) for k, v in params.items()}, [n_h] * d)
    np.testing.assert_allclose(host(logsd), es, atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(z_new), ez, atol=ATOL, rtol=0)


def test_autotune_picks_a_kernel_per_layer_and_keeps_parity(amd):
    """iaf_stack_autotune times the fp32 kernel and every bf16x3 shape and pins the fastest for this size; the result of
    the tuned stack still meets the parity bar and the choice is reported"""
    B, n_z, n_h, d, H, W = 32, 32, 160, 2, 16, 16
    params, z, ctx = _case(11, B, n_z, n_h, d, H, W)
    st = amd.ARStack(n_z, [n_h] * d)
    st.prepare({k: dev(v) for k, v in params.items()})
    picks = st.autotune(dev(z), dev(ctx), reps=10)
    assert len(picks) == d + 1 and all(us > 0 or c in ("fused into next", "one-launch step") for c, us in picks)
    print("autotune:", picks)
    if picks[0][0] == "one-launch step":            # the whole step measured faster as ONE launch (iaf_step_fused.hpp)
        assert all(c == "one-launch step" for c, _ in picks) and picks[-1][1] > 0 and st.step_is_fused(B, H, W) > 0
    else:
        assert st.step_is_fused(B, H, W) == 0
    for layer, (choice, _) in enumerate(picks):
        if choice not in ("fused into next", "one-launch step") and not choice.endswith("+layer0"):
            assert st.layer_precision(layer, B, H, W) == ("f32" if choice == "f32" else "bf16x3")
    z_new, logsd = st.iaf_step(dev(z), dev(ctx))
    p32 = {k: f32(v) for k, v in params.items()}
    ez, es = [], []
    for b0 in range(0, B, 16):
        a, b = O.iaf_step(f32(z[b0:b0 + 16]), f32(ctx[b0:b0 + 16]), p32, [n_h] * d)
        ez.append(a); es.append(b)
    np.testing.assert_allclose(host(z_new), np.concatenate(ez), atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(logsd), np.concatenate(es), atol=ATOL, rtol=0)


# ---------------------------------------------------------------- first masked conv fused into the second one's kernel (IN_FUSED0)
@pytest.mark.parametrize("shp", [(5, 2, 1, 4, 1), (5, 1, 1, 4, 1), (5, 2, 1, 4, 2), (2, 2, 1, 4, 1), (2, 1, 1, 4, 1)],
                         ids=lambda s: "nt%d_ppw%d_pxt%d_ks%d_wco%d" % s)
@pytest.mark.parametrize("cfg", [(32, 32, 160, 2, 16, 16), (32, 32, 160, 2, 8, 8), (3, 32, 160, 2, 5, 7), (16, 32, 64, 1, 8, 8),
                                 (2, 32, 128, 3, 4, 4)], ids=lambda s: "B%d_z%d_h%d_d%d_%dx%d" % s)
def test_fused_first_layer_vs_oracle_and_unfused(amd, shp, cfg):
    """IN_FUSED0: layer 0 (32 -> n_h) computed in the prologue of layer 1's kernel (layer 1 = a hidden layer, or the output
    pair when depth_ar = 1), straight into its LDS tile.  Checked against the oracle, against the unfused path, and bit for
    bit against the unfused path when that runs layer 0 without K-slicing (same products in the same order)."""
    B, n_z, n_h, d, H, W = cfg
    nt, ppw, pxt, ks, wco = shp
    params, z, ctx = _case(77, *cfg)
    dp = {k: dev(v) for k, v in params.items()}
    zd, cd = dev(z), dev(ctx)
    tiles1 = (n_h if d > 1 else 2 * n_z) // 16
    if tiles1 % (nt * wco) or (d == 1 and nt % 2):
        pytest.skip("layer 1 has %d co tiles" % tiles1)
    ref = amd.ARStack(n_z, [n_h] * d)
    ref.set_fuse_first("never")
    ref.prepare(dp)
    nt0 = [n for n in (5, 4, 2) if (n_h // 16) % n == 0][0]
    ref.set_tuning_bf3(0, nt0, 1, 4, 1)                      # layer 0 without K-slicing
    ref.set_tuning_bf3(1, nt, ppw, pxt, ks, wco)
    st = amd.ARStack(n_z, [n_h] * d)
    st.set_fuse_first("always")
    st.prepare(dp)
    st.set_tuning_bf3(1, nt, ppw, pxt, ks, wco)
    try:
        fused_us = 1e3 * st.time_layer(-1, zd, cd, reps=3)  # raises UnsupportedError if this shape cannot carry the fused layer
    except amd.UnsupportedError as e:
        pytest.skip(str(e))
    assert fused_us > 0
    z_ref, s_ref = ref.iaf_step(zd, cd)
    z_new, logsd = st.iaf_step(zd, cd)
    assert torch.equal(z_new, z_ref) and torch.equal(logsd, s_ref)
    p32 = {k: f32(v) for k, v in params.items()}
    ez, es = [], []
    for b0 in range(0, B, 16):
        a, b = O.iaf_step(f32(z[b0:b0 + 16]), f32(ctx[b0:b0 + 16]), p32, [n_h] * d)
        ez.append(a); es.append(b)
    np.testing.assert_allclose(host(z_new), np.concatenate(ez), atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(logsd), np.concatenate(es), atol=ATOL, rtol=0)


@pytest.mark.parametrize("kl_min", [0.0, 0.25])
def test_fused_first_layer_posterior_block(amd, kl_min):
    """the fused prologue with the posterior sample as its input (z0 from qm / rm / ql / rl / eps) and two contexts"""
    B, n_z, n_h, d, H, W = 8, 32, 160, 2, 16, 16
    rng = np.random.RandomState(56)
    params = gi.ar_multiconv2d_params(rng, n_z, [n_h] * d, [n_z, n_z])
    f = lambda c: rng.standard_normal((B, c, H, W))
    qm, ql, rm, rl, pm, pl = f(n_z), 0.25 * f(n_z), f(n_z), 0.25 * f(n_z), f(n_z), 0.25 * f(n_z)
    uc, dc, eps = f(n_h), f(n_h), f(n_z)
    e = O.posterior_block(f32(qm), f32(ql), f32(rm), f32(rl), f32(pm), f32(pl), f32(uc), f32(dc), f32(eps),
                          {k: f32(v) for k, v in params.items()}, [n_h] * d, kl_min)
    outs = {}
    for mode in ("always", "never"):
        st = amd.ARStack(n_z, [n_h] * d)
        st.set_fuse_first(mode)
        st.prepare({k: dev(v) for k, v in params.items()})
        st.set_tuning_bf3(0, 5, 1, 4, 1)
        st.set_tuning_bf3(1, 5, 2, 1, 4)
        st.set_tuning_bf3(2, 2, 2, 1, 4)
        out = st.posterior_block(dev(qm), dev(ql), dev(rm), dev(rl), dev(pm), dev(pl), dev(uc), dev(dc), dev(eps),
                                 kl_min, want_kl_elem=True)
        np.testing.assert_allclose(host(out["z"]), e["z"], atol=ATOL, rtol=0)
        np.testing.assert_allclose(host(out["kl_elem"]), e["logqs"] - e["logps"], atol=ATOL, rtol=1e-5)
        np.testing.assert_allclose(host(out["kl_obj"]), e["kl_obj"], atol=2e-3, rtol=1e-4)
        outs[mode] = out
    assert torch.equal(outs["always"]["z"], outs["never"]["z"]) and torch.equal(outs["always"]["kl_elem"], outs["never"]["kl_elem"])
