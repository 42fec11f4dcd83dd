"""The two-plane fp16 step kernels ("f16x2", round 6: iaf_step_fused.hpp F16, include/iaf_hip.h IAF_PRECISION_F16X2).

VERDICT r05 "next" #1 set the gate under which this arithmetic may carry the headline number: tests/test_hip_dynamic_range.py passes
unmodified (it builds its "one-launch" stack with the DEFAULT precision, which is f16x2 where the kernels are compiled -- asserted
here) AND, at every case of tests/test_hip_baseline_configs.py the kernels cover, the largest error against the fp64 oracle is at most
1.5x that of the exact-fp32 MFMA kernels (`--precision f32`).  This file holds the second half, and the behaviour at the edge fp16 has
and fp32 does not: operands beyond 65504.  Further down: the same arithmetic in the plain convs around the step (forward: the same gate and
range protocol; one weight pack per conv) and the whole-row prep of stacks that keep only split packs.  (The plain convs' data gradient on
two fp16 planes, which has no exponent range of its own: tests/test_hip_layer.py.)

Reference operator: tf_utils/layers.py:31-64,158-166; tf_train.py:29-95."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import iaf_oracle as O

pytestmark = pytest.mark.gpu
N_Z, N_H, D = 32, 160, 2


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


def _oracle_step(z, ctx, params, chunk=16):
    p32 = {k: f32(v) for k, v in params.items()}
    ez, es = [], []
    for b0 in range(0, z.shape[0], chunk):
        a, b = O.iaf_step(f32(z[b0:b0 + chunk]), f32(ctx[b0:b0 + chunk]), p32, [N_H] * D)
        ez.append(a); es.append(b)
    return np.concatenate(ez), np.concatenate(es)


def _stack(amd, params, precision=None, **packs):
    st = amd.ARStack(N_Z, [N_H] * D)
    if precision:
        st.set_precision(precision)
    dp = {k: dev(v) for k, v in params.items()}
    st.prepare(dp)
    if packs:
        st.set_packs(**packs)
        st.prepare(dp)
    return st, dp


def test_the_default_precision_runs_the_fp16_kernels_where_they_are_compiled(amd):
    """what 'tests/test_hip_dynamic_range.py passes unmodified' rests on: a stack built without a precision runs the step of the two
    BASELINE geometries on the two-plane fp16 kernels -- and nothing else does"""
    rng = np.random.RandomState(1)
    st, _ = _stack(amd, gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z]))
    assert st.step_is_f16(32, 16, 16) and st.step_is_f16(32, 8, 8)
    assert not st.step_is_f16(256, 8, 8)                      # (layer-by-layer / R = 2 at this size)
    st.set_precision("bf16x3")
    assert not st.step_is_f16(32, 16, 16)
    other = amd.ARStack(32, [64])                             # config 1: no fp16 kernel compiled
    assert not other.step_is_f16(16, 16, 16)
    th = amd.ARStack(N_Z, [N_H] * D, variant=amd._capi.IAF_VARIANT_THEANO)
    assert not th.step_is_f16(32, 16, 16)


@pytest.mark.parametrize("B,H", [(32, 16), (32, 8), (256, 16), (5, 16), (7, 8)], ids=["config2_16", "config2_8", "config5_16", "B5_16", "B7_8"])
def test_error_vs_fp64_is_within_1p5x_of_the_exact_fp32_kernels(amd, B, H):
    """VERDICT r05's gate: max |f16x2 - fp64 oracle| <= 1.5 x max |exact fp32 MFMA - fp64 oracle|, IAF step, whole batch"""
    rng = np.random.RandomState(100 + B + H)
    params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
    z, ctx = rng.standard_normal((B, N_Z, H, H)), rng.standard_normal((B, N_H, H, H))
    ez, es = _oracle_step(z, ctx, params)
    errs = {}
    for prec in ("f32", "bf16x3", "f16x2"):
        st, _ = _stack(amd, params, prec)
        if prec == "f16x2":
            assert st.step_is_f16(B, H, H)
        zn, ls = st.iaf_step(dev(z), dev(ctx))
        errs[prec] = max(np.abs(host(zn) - ez).max(), np.abs(host(ls) - es).max())
        if prec == "f16x2":
            assert st.range_errors() == 0
    print("B=%d %dx%d: max |. - fp64 oracle|: %s" % (B, H, H, ", ".join("%s %.3g" % kv for kv in errs.items())))
    assert errs["f16x2"] <= 1.5 * errs["f32"], errs
    assert errs["f16x2"] < 1e-4


@pytest.mark.parametrize("H", [16, 8])
def test_posterior_block_on_fp16_planes_vs_oracle(amd, H):
    """the extended unit (tf_train.py:56-85) through the F16 kernels: sample in front, KL elements and free-bits reductions behind"""
    B, kl_min = 32, 0.25
    rng = np.random.RandomState(300 + H)
    params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
    f = lambda c: rng.standard_normal((B, c, H, H))
    qm, ql, rm, rl, pm, pl = f(N_Z), 0.25 * f(N_Z), f(N_Z), 0.25 * f(N_Z), f(N_Z), 0.25 * f(N_Z)
    uc, dc, eps = f(N_H), f(N_H), f(N_Z)
    st, _ = _stack(amd, params, "f16x2")
    assert st.step_is_f16(B, H, H)
    out = st.posterior_block(dev(qm), dev(ql), dev(rm), dev(rl), dev(pm), dev(pl), dev(uc), dev(dc), dev(eps), kl_min, want_kl_elem=True)
    p32 = {k: f32(v) for k, v in params.items()}
    e = O.posterior_block(f32(qm), f32(ql), f32(rm), f32(rl), f32(pm), f32(pl), f32(uc), f32(dc), f32(eps), p32, [N_H] * D, kl_min)
    np.testing.assert_allclose(host(out["z"]), e["z"], atol=1e-4, rtol=0)
    ekl = e["logqs"] - e["logps"]
    np.testing.assert_allclose(host(out["kl_elem"]), ekl, atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(host(out["kl_cost"]), ekl.sum(axis=(1, 2, 3)), atol=2e-3, rtol=1e-4)
    kl_obj = np.tile(np.maximum(ekl.sum(axis=(2, 3)).mean(axis=0, keepdims=True), kl_min), (B, 1)).sum(axis=1)
    np.testing.assert_allclose(host(out["kl_obj"]), kl_obj, atol=2e-3, rtol=1e-4)
    assert st.range_errors() == 0 and st.exchange_errors() == 0


@pytest.mark.parametrize("H", [16, 8])
def test_an_operand_beyond_fp16_is_loud_and_the_stack_goes_back_to_bf16x3(amd, H):
    """activations of 1e5 and more (fp32 has them, fp16 does not): the F16 launch's outputs are not finite, the range word is up, the NEXT
    call says so once (RangeError) and from then on the stack computes the same inputs on the bf16x3 kernels -- to the oracle"""
    B = 8
    rng = np.random.RandomState(900 + H)
    params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
    z, ctx = 1e5 * rng.standard_normal((B, N_Z, H, H)), 1e5 * rng.standard_normal((B, N_H, H, H))
    st, dp = _stack(amd, params, "f16x2")
    m, s = st.ar_multiconv2d(dev(z), dev(ctx))
    assert not (np.isfinite(host(m)).all() and np.isfinite(host(s)).all())
    assert st.range_errors() & 1
    with pytest.raises(amd._capi.RangeError):
        st.ar_multiconv2d(dev(z), dev(ctx))
    assert not st.step_is_f16(B, H, H)
    m, s = st.ar_multiconv2d(dev(z), dev(ctx))                # bf16x3 now: fp32's exponent range
    p32 = {k: f32(v) for k, v in params.items()}
    em, es = O.ar_multiconv2d(f32(z), f32(ctx), p32, [N_H] * D, [N_Z, N_Z])
    scale = max(np.abs(em).max(), np.abs(es).max())
    assert np.abs(host(m) - em).max() < 1e-5 * scale and np.abs(host(s) - es).max() < 1e-5 * scale
    # re-armed: ordinary inputs run on fp16 planes again
    st.set_precision("f16x2")
    st.prepare(dp, force=True)
    assert st.step_is_f16(B, H, H) and st.range_errors() == 0
    z1, c1 = rng.standard_normal((B, N_Z, H, H)), rng.standard_normal((B, N_H, H, H))
    m, s = st.ar_multiconv2d(dev(z1), dev(c1))
    em, es = O.ar_multiconv2d(f32(z1), f32(c1), p32, [N_H] * D, [N_Z, N_Z])
    assert np.abs(host(m) - em).max() < 1e-4 and np.abs(host(s) - es).max() < 1e-4
    assert st.range_errors() == 0


def test_a_weight_beyond_fp16_is_found_by_the_prep_launch(amd):
    """exp(g) = e^15 on one conv: its weights pass 65504 -- the prep of the fp16 pack raises the word (bit 1)"""
    rng = np.random.RandomState(3)
    params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
    params["layer_1/g"] = params["layer_1/g"] + 15.0
    st, _ = _stack(amd, params, "f16x2")
    assert st.range_errors() & 2
    z, ctx = rng.standard_normal((4, N_Z, 16, 16)), rng.standard_normal((4, N_H, 16, 16))
    with pytest.raises(amd._capi.RangeError):
        st.ar_multiconv2d(dev(z), dev(ctx))
    m, s = st.ar_multiconv2d(dev(z), dev(ctx))
    assert np.isfinite(host(m)).all() and np.isfinite(host(s)).all()


def test_only_the_fp16_pack_kept_up_to_date(amd):
    """iaf_stack_set_packs(IAF_PACK_F16X2): the prep writes 4 bytes per weight; same results; after a range failure the stack asks for
    another prepare (its bf16x3 pack was never written) instead of reading stale weights"""
    B, H = 32, 16
    rng = np.random.RandomState(11)
    params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
    z, ctx = rng.standard_normal((B, N_Z, H, H)), rng.standard_normal((B, N_H, H, H))
    full, _ = _stack(amd, params, "f16x2")
    only, dp = _stack(amd, params, "f16x2", f32=False, bf16x3=False, f16x2=True)
    a, b = full.iaf_step(dev(z), dev(ctx))
    c, d = only.iaf_step(dev(z), dev(ctx))
    # (equal up to the last bit of the weight norm: a prep launch that writes only split packs sums the squares in another order)
    assert (a - c).abs().max().item() < 2e-6 and (b - d).abs().max().item() < 2e-6
    with pytest.raises(ValueError):
        bf = amd.ARStack(N_Z, [N_H] * D)
        bf.set_precision("bf16x3")
        bf.set_packs(f32=False, bf16x3=False, f16x2=True)     # (the fp16 pack exists for f16x2 stacks only)
    big = 1e6 * z
    only.iaf_step(dev(big), dev(ctx))
    torch.cuda.synchronize()
    with pytest.raises(amd._capi.RangeError):
        only.iaf_step(dev(z), dev(ctx))
    with pytest.raises(amd._capi.IafHipError):                # IAF_ERR_NOT_PREPARED: no bf16x3 pack yet
        only.iaf_step(dev(z), dev(ctx))
    only.prepare(dp, force=True)
    e, f = only.iaf_step(dev(z), dev(ctx))
    ez, es = _oracle_step(z, ctx, params)
    assert np.abs(host(e) - ez).max() < 1e-4 and np.abs(host(f) - es).max() < 1e-4


@pytest.mark.parametrize("H", [16, 8])
def test_nan_and_inf_inputs(amd, H):
    """a NaN in the inputs is the caller's NaN (tf_train.py:283-285), not a range failure: NaN out where it reaches, the word stays down;
    an inf is beyond fp16 -- loud, like any operand past 65504"""
    B = 8
    rng = np.random.RandomState(40 + H)
    params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
    z, ctx = rng.standard_normal((B, N_Z, H, H)).astype(np.float32), rng.standard_normal((B, N_H, H, H)).astype(np.float32)
    st, _ = _stack(amd, params, "f16x2")
    zn = z.copy()
    zn[0, 3, H - 1, H - 1] = np.nan
    zn.view(np.uint32)[1, 0, 0, 0] = 0xffffffff               # an all-ones NaN
    a, b = st.iaf_step(dev(zn), dev(ctx))
    ha = host(a)
    assert np.isnan(ha[0]).any() and np.isfinite(ha[3:]).all()
    assert st.range_errors() == 0 and st.exchange_errors() == 0
    a2, b2 = st.iaf_step(dev(z), dev(ctx))                    # the buffers are intact: clean inputs, clean outputs
    ez, es = _oracle_step(z, ctx, params)
    assert np.abs(host(a2) - ez).max() < 1e-4
    zi = z.copy()
    zi[0, 0, 0, 0] = np.inf
    st.iaf_step(dev(zi), dev(ctx))
    assert st.range_errors() & 1


def test_graph_replays_two_streams_and_scrambled_ticket_order(amd):
    """the exchange form on fp16 planes has rows, counters and a 'not there yet' pattern of its own (IAF_XSENT_F16): graphs replayed with
    fresh inputs, a second stream, work lists picked by hash and tickets drawn late (knobs 1 + 2) -- all equal to the eager result"""
    B, H = 32, 16
    rng = np.random.RandomState(77)
    params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
    st, _ = _stack(amd, params, "f16x2")
    assert st.step_is_f16(B, H, H) and st.step_exchanges(B, H, H)
    zs = [dev(rng.standard_normal((B, N_Z, H, H))) for _ in range(3)]
    cs = [dev(rng.standard_normal((B, N_H, H, H))) for _ in range(3)]
    want = [tuple(t.clone() for t in st.iaf_step(z, c)) for z, c in zip(zs, cs)]
    torch.cuda.synchronize()
    st.set_halo_exchange_debug(3)
    for (wa, wb), z, c in zip(want, zs, cs):
        a, b = st.iaf_step(z, c)
        assert torch.equal(a, wa) and torch.equal(b, wb)
    st.set_halo_exchange_debug(0)
    zin, cin = zs[0].clone(), cs[0].clone()
    out = (torch.empty_like(zin), torch.empty_like(zin))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        st.iaf_step(zin, cin, out=out)                        # warm-up on the capture stream: its own exchange set
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            st.iaf_step(zin, cin, out=out)
    for (wa, wb), z, c in zip(want, zs, cs):
        zin.copy_(z); cin.copy_(c)
        torch.cuda.synchronize()
        g.replay()
        a, b = st.iaf_step(z, c)                              # eager on the default stream, beside the replay
        torch.cuda.synchronize()
        assert torch.equal(out[0], wa) and torch.equal(out[1], wb)
        assert torch.equal(a, wa) and torch.equal(b, wb)
    assert st.exchange_errors() == 0 and st.range_errors() == 0


@pytest.mark.parametrize("B", [4, 32], ids=["B4", "B32"])
def test_training_forward_on_fp16_planes_feeds_the_same_backward(amd, B):
    """iaf_step_train stores the hidden activations in fp32 whatever planes the convs read: gradients against fp64 autograd of the restated
    forward, as for bf16x3 (tests/test_hip_parity.py).  B = 32 (8192 pixels): the masked convs' data gradients run the split-product
    kernels (bf16 planes: the two-plane form with a tile-local scale that the plain convs' data gradient uses was measured slower on the
    5-tap K loops, profiles/r06/experiments/wgrad_fp16_planes_not_kept.txt)"""
    from oracle import iaf_grad_oracle as G
    H = 16
    rng = np.random.RandomState(21)
    params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
    z, ctx = rng.standard_normal((B, N_Z, H, H)), rng.standard_normal((B, N_H, H, H))
    dzn, dls = rng.standard_normal((B, N_Z, H, H)), rng.standard_normal((B, N_Z, H, H))
    st = amd.ARStack(N_Z, [N_H] * D)
    st.set_precision("f16x2")
    st.set_training(True)
    dp = {k: dev(v) for k, v in params.items()}
    st.prepare(dp)
    zd, cd = dev(z), dev(ctx)
    zn, ls = st.iaf_step_train(zd, cd)
    zi, li = st.iaf_step(zd, cd)
    assert torch.equal(zn, zi) and torch.equal(ls, li)           # training forward == inference forward, bit for bit
    dz, dctx, grads = st.iaf_step_backward(zd, cd, zn, ls, dev(dzn), dev(dls), dp)
    ref, _, _ = G.iaf_step_grads(f32(z), f32(ctx), {k: f32(v) for k, v in params.items()}, [N_H] * D, f32(dzn), f32(dls))

    def close(a, r, tol, what):
        assert np.abs(a - r).max() <= tol * max(1e-30, np.abs(r).max()), what
    close(host(dz), ref["z"], 1e-4, "dz")
    close(host(dctx), ref["context"], 1e-4, "dcontext")
    for k in sorted(params):
        close(host(grads[k]), ref[k], 2e-4, k)


# ---------------------------------------------------------------- the plain 9-tap conv on two fp16 planes (iaf_conv_bf3.hpp F16)
PLAIN = [(32, 160, 160, 16, 16), (32, 160, 448, 16, 16), (32, 192, 160, 16, 16), (64, 160, 384, 8, 8), (5, 160, 320, 30, 30)]


def _conv(amd, p, n_in, n_out, precision=None):
    conv = amd.WNConv2d(n_in, n_out)
    if precision:
        conv.set_precision(precision)
    conv.prepare(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    return conv


def test_a_plain_conv_runs_the_fp16_planes_by_default_where_the_split_kernels_cover_it(amd):
    rng = np.random.RandomState(2)
    conv = _conv(amd, gi.conv_params(rng, 160, 160), 160, 160)
    assert conv.runs_bf16x3(32, 16, 16) and conv.runs_f16x2(32, 16, 16)
    assert not conv.runs_f16x2(2, 8, 8)                       # below the size rule: the exact-fp32 kernel
    conv.set_precision("bf16x3")
    assert conv.runs_bf16x3(32, 16, 16) and not conv.runs_f16x2(32, 16, 16)
    conv.set_precision("f32")
    assert not conv.runs_bf16x3(32, 16, 16) and not conv.runs_f16x2(32, 16, 16)
    odd = _conv(amd, gi.conv_params(rng, 48, 64), 48, 64)     # c_in % 32 != 0: no split pack at all
    assert not odd.runs_f16x2(32, 16, 16)
    odd.set_precision("f16x2")                                # (bf16x3 semantics: accepted, nothing to run it on)
    assert not odd.runs_f16x2(32, 16, 16)


@pytest.mark.parametrize("case", PLAIN, ids=lambda s: "B%d_%dto%d_%dx%d" % s)
def test_plain_conv_error_vs_fp64_is_within_1p5x_of_the_exact_fp32_kernel(amd, case):
    """the gate of the step kernels, held for the plain convs of IAFLayer.up / .down (tf_train.py:33-36,87-94) too:
    ELU'd concat input, residual -- max |f16x2 - fp64| <= 1.5 x max |exact fp32 - fp64|"""
    B, n_in, n_out, H, W = case
    rng = np.random.RandomState(500 + n_in + n_out + H)
    p = gi.conv_params(rng, n_in, n_out)
    x, res = rng.standard_normal((B, n_in, H, W)), rng.standard_normal((B, n_out, H, W))
    nb = min(B, 4)                                            # the oracle's share of the batch (the first and the last images)
    sel = np.r_[0:nb // 2, B - (nb - nb // 2):B]
    e = f32(res[sel]) + 0.1 * O.conv2d(O.elu(f32(x[sel])), f32(p["V"]), f32(p["g"]), f32(p["b"]))
    errs = {}
    for prec in ("f32", "bf16x3", "f16x2"):
        conv = _conv(amd, p, n_in, n_out, prec)
        assert conv.runs_f16x2(B, H, W) == (prec == "f16x2")
        y = conv(dev(x), elu_input=True, residual=dev(res))[0]
        errs[prec] = np.abs(host(y)[sel] - e).max()
        if prec == "f16x2":
            assert conv.range_errors() == 0
    print("%s: max |. - fp64 oracle|: %s" % (case, ", ".join("%s %.3g" % kv for kv in errs.items())))
    assert errs["f16x2"] <= 1.5 * errs["f32"], errs
    assert errs["f16x2"] < 2e-5


def test_plain_conv_operand_beyond_fp16_is_loud_and_the_conv_goes_back_to_bf16x3(amd):
    B, n_in, n_out, H = 16, 160, 160, 16
    rng = np.random.RandomState(61)
    p = gi.conv_params(rng, n_in, n_out)
    conv = _conv(amd, p, n_in, n_out, "f16x2")
    x = 1e5 * rng.standard_normal((B, n_in, H, H))
    y = conv(dev(x))[0]
    assert not np.isfinite(host(y)).all()
    assert conv.range_errors() & 1
    with pytest.raises(amd._capi.RangeError):
        conv(dev(x))
    assert not conv.runs_f16x2(B, H, H) and conv.runs_bf16x3(B, H, H)
    y = conv(dev(x))[0]                                       # bf16x3 now: fp32's exponent range
    e = O.conv2d(f32(x[:2]), f32(p["V"]), f32(p["g"]), f32(p["b"]))
    assert np.abs(host(y)[:2] - e).max() < 1e-5 * np.abs(e).max()
    conv.set_precision("f16x2")                               # re-armed
    conv.prepare(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    assert conv.runs_f16x2(B, H, H) and conv.range_errors() == 0
    x1 = rng.standard_normal((B, n_in, H, H))
    y = conv(dev(x1))[0]
    e = O.conv2d(f32(x1[:2]), f32(p["V"]), f32(p["g"]), f32(p["b"]))
    assert np.abs(host(y)[:2] - e).max() < 2e-5 and conv.range_errors() == 0
    # a caller's NaN is data, not a range failure
    xn = x1.astype(np.float32).copy()
    xn[3, 5, 7, 7] = np.nan
    y = host(conv(dev(xn))[0])
    assert np.isnan(y[3]).any() and np.isfinite(y[:3]).all() and conv.range_errors() == 0


def test_plain_conv_weight_beyond_fp16_is_found_by_its_prep(amd):
    rng = np.random.RandomState(62)
    p = gi.conv_params(rng, 160, 160)
    p["g"] = p["g"] + 15.0
    conv = _conv(amd, p, 160, 160, "f16x2")
    assert conv.range_errors() & 2
    x = rng.standard_normal((16, 160, 16, 16))
    with pytest.raises(amd._capi.RangeError):
        conv(dev(x))
    y = conv(dev(x))[0]
    e = O.conv2d(f32(x[:1]), f32(p["V"]), f32(p["g"]), f32(p["b"]))
    assert np.abs(host(y)[:1] - e).max() < 1e-5 * np.abs(e).max()


# ---------------------------------------------------------------- one pack per conv (iaf_conv3x3_set_packs) and the whole-row prep of split-pack-only stacks
def test_plain_conv_keeps_only_the_pack_its_launch_reads(amd):
    """WNConv2d.trim_packs: 4 (fp16 planes) / 6 (bf16 planes) / 4 (fp32) instead of 14 bytes written per weight by every prep launch; same
    results; a launch that needs a pack that is not kept fails loudly (IAF_ERR_NOT_PREPARED) instead of reading stale weights"""
    rng = np.random.RandomState(71)
    n_in, n_out = 160, 160
    p = gi.conv_params(rng, n_in, n_out)
    x_big, x_small = rng.standard_normal((32, n_in, 16, 16)), rng.standard_normal((2, n_in, 8, 8))
    full = _conv(amd, p, n_in, n_out)
    want_big, want_small = full(dev(x_big))[0], full(dev(x_small))[0]
    dp = [dev(p[k]) for k in ("V", "g", "b")]
    for (B, H), kept, other in (((32, 16), "f16x2", x_small), ((2, 8), "f32", x_big)):
        conv = _conv(amd, p, n_in, n_out)
        assert conv.trim_packs(B, H, H) == kept
        # new weights: only the kept pack follows them
        p2 = gi.conv_params(rng, n_in, n_out)
        conv.prepare(*[dev(p2[k]) for k in ("V", "g", "b")])
        conv.prepare(*dp)
        x = x_big if kept == "f16x2" else x_small
        got, want = conv(dev(x))[0], want_big if kept == "f16x2" else want_small
        # (equal up to the last bit of the weight norm: a prep launch that writes only split packs sums the squares in another order)
        assert (got - want).abs().max().item() < 4e-6 * max(1.0, float(want.abs().max()))
        with pytest.raises(amd._capi.IafHipError):            # the other size reads another pack: not kept
            conv(dev(other))
        conv.set_packs()                                      # every pack again: the next prepare fills them
        with pytest.raises(amd._capi.IafHipError):
            conv(dev(other))
        conv.prepare(*dp, force=True)
        want = want_small if kept == "f16x2" else want_big
        assert (conv(dev(other))[0] - want).abs().max().item() < 4e-6 * max(1.0, float(want.abs().max()))
    bf = _conv(amd, p, n_in, n_out, "bf16x3")
    assert bf.trim_packs(32, 16, 16) == "bf16x3" and bf.trim_packs(32, 16, 16, strided=True) == "bf16x3"
    bf.prepare(*dp, force=True)
    assert (bf(dev(x_big))[0] - want_big).abs().max().item() < 2e-5
    tr = _conv(amd, p, n_in, n_out)
    tr.set_training(True)
    with pytest.raises(amd.UnsupportedError):
        tr.set_packs(f32=False)


def test_fp16_only_plain_conv_after_a_range_failure_asks_for_a_prepare(amd):
    rng = np.random.RandomState(72)
    p = gi.conv_params(rng, 160, 160)
    dp = [dev(p[k]) for k in ("V", "g", "b")]
    conv = _conv(amd, p, 160, 160)
    assert conv.trim_packs(32, 16, 16) == "f16x2"
    conv.prepare(*dp, force=True)
    x = rng.standard_normal((32, 160, 16, 16))
    conv(dev(1e6 * x))
    torch.cuda.synchronize()
    with pytest.raises(amd._capi.RangeError):
        conv(dev(x))
    with pytest.raises(amd._capi.IafHipError):                # IAF_ERR_NOT_PREPARED: its bf16x3 pack was never written
        conv(dev(x))
    conv.prepare(*dp, force=True)
    y = conv(dev(x))[0]
    e = O.conv2d(f32(x[:2]), f32(p["V"]), f32(p["g"]), f32(p["b"]))
    assert np.abs(host(y)[:2] - e).max() < 2e-5 and not conv.runs_f16x2(32, 16, 16)


@pytest.mark.parametrize("geom", [(32, 160, 2), (32, 64, 1), (64, 64, 4), (64, 128, 4), (64, 192, 4), (32, 128, 2), (32, 160, 3)],
                         ids=lambda g: "nz%d_nh%d_d%d" % g)
@pytest.mark.parametrize("packs", ["bf16x3", "f16x2"])
def test_split_packs_only_stacks_prep_through_whole_rows_equals_the_default_prep(amd, geom, packs):
    """a stack that keeps only split packs re-derives them in prep_tile_fast (iaf_kernels_prep.hpp: 16-byte row loads, tile through LDS, tiles
    paired per XCD): the same weights as the default prep up to the order of the l2 norm's sum -- every n_in of the compiled geometries, through
    the batched launch (odd tile counts: the padded grid) and the per-stack one"""
    n_z, n_h, d = geom
    B = 4
    rng = np.random.RandomState(400 + n_h + d)
    params = gi.ar_multiconv2d_params(rng, n_z, [n_h] * d, [n_z, n_z])
    dp = {k: dev(v) for k, v in params.items()}
    ref = amd.ARStack(n_z, [n_h] * d)
    ref.set_precision(packs)
    ref.prepare(dp)
    H = 8
    if packs == "f16x2":                                      # a size whose step runs on the fp16 planes (the only pack kept)
        sizes = [h for h in (16, 8, 4) if ref.step_is_f16(B, h, h)]
        if not sizes:
            pytest.skip("no two-plane fp16 step kernel for this geometry")
        H = sizes[0]
    z, ctx = dev(rng.standard_normal((B, n_z, H, H))), dev(rng.standard_normal((B, n_h, H, H)))
    only = [amd.ARStack(n_z, [n_h] * d) for _ in range(3)]
    for st in only:
        st.set_precision(packs)
        st.prepare(dp)
        try:
            st.set_packs(f32=False, bf16x3=packs == "bf16x3", f16x2=packs == "f16x2")
        except (amd.UnsupportedError, ValueError) as e:       # (no step kernel of this arithmetic for the geometry: nothing to compare)
            pytest.skip(str(e))
    only[0].prepare(dp, force=True)
    amd.PrepBatch(only[1:]).run([dp, dp])
    wm, ws = ref.ar_multiconv2d(z, ctx)
    scale = max(1.0, float(wm.abs().max()), float(ws.abs().max()))
    for st in only:
        m, s = st.ar_multiconv2d(z, ctx)
        assert (m - wm).abs().max().item() < 4e-6 * scale and (s - ws).abs().max().item() < 4e-6 * scale
        assert st.range_errors() == 0
