"""The PAIR form of the one-launch IAF step at 8-pixel rows (iaf_amd/csrc/iaf_step_fused.hpp, "PAIR"; the step itself is
tf_train.py:69-72 over layers.py:158-166 at [B, n_z, 8, 8]): two workgroups share two image rows, each computes half of the last hidden
layer's channels and half of the output pair from half of the weights, and they swap their halves of the last hidden layer through
device memory.  Same protocol as the halo exchange of the 16-pixel rows (tests/test_hip_halo_exchange.py), so the same questions:

 * numbers: against the fp64 oracle and against the recomputing one-row kernel, whole images and ragged heights;
 * order: partners hold adjacent tickets of a work list; knobs 1 (lists ignore the placement) and 2 (tickets out of dispatch order) must
   not change a bit of the outcome, on grids of less than one and of many rounds of the chip, launch after launch with fresh inputs
   (a half left over from an earlier launch would be an O(1) error: the same runs prove the re-arming);
 * failure: knob 8 makes one workgroup keep its half to itself -- its partner's bounded wait gives up: NaN in that partner's outputs,
   ExchangeError from the next call, the recomputing kernel from then on, the pair form back after set_halo_exchange(True);
 * the form is OPT-IN (knob 32): measured, it is slower than the one-row kernel at the BASELINE batch size (18.9 vs 17.0 us: the K
   loops are bound by what one wave per SIMD can issue, not by the weight port; profiles/r05/experiments/pair_form.txt);
 * NaN inputs are data, not an exchange failure, whatever their payload: "not there yet" is a pair of SIGNALLING
   bf16 NaNs, which no arithmetic result can be; a NaN of all ones (round 4's pattern) and the pattern itself go in as inputs."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import iaf_oracle as O

pytestmark = pytest.mark.gpu
PAIR = 32            # knob of iaf_stack_set_halo_exchange_debug: the pair form is opt-in (slower than the one-row kernel at B = 32 on MI355X)


@pytest.fixture(scope="module")
def amd():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    import iaf_amd
    iaf_amd._capi.lib()
    return iaf_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def _stacks(amd, seed, variant="tf"):
    """(pair-form stack, recomputing stack) on the same weights, BASELINE geometry n_z = 32, n_h = 160, depth_ar = 2"""
    rng = np.random.RandomState(seed)
    hp = gi.ar_multiconv2d_params(rng, 32, [160, 160], [32, 32])
    params = {k: dev(v) for k, v in hp.items()}
    ps, rc = amd.ARStack(32, [160, 160], variant=variant), amd.ARStack(32, [160, 160], variant=variant)
    ps.set_halo_exchange_debug(PAIR)
    rc.set_halo_exchange(False)
    ps.prepare(params)
    rc.prepare(params)
    return ps, rc, hp


def _close(a, r, what, tol=2e-6):
    a, r = a.double(), r.double()
    assert torch.isfinite(a).all(), what
    err = float((a - r).abs().max())
    assert err <= tol * max(1.0, float(r.abs().max())), (what, err)


@pytest.mark.parametrize("cfg", [(32, 8), (5, 8), (3, 5), (2, 7), (1, 1), (7, 2)], ids=lambda c: "B%d_H%d" % c)
def test_pair_step_vs_oracle(amd, cfg):
    B, H = cfg
    ps, rc, hp = _stacks(amd, 50 + B + H)
    assert ps.step_pairs(B, H, 8) and ps.step_is_fused(B, H, 8) == 2 and not rc.step_pairs(B, H, 8) and rc.step_is_fused(B, H, 8) == 1
    rng = np.random.RandomState(7 + B)
    z, ctx = rng.standard_normal((B, 32, H, 8)), rng.standard_normal((B, 160, H, 8))
    z_new, logsd = ps.iaf_step(dev(z), dev(ctx))
    m_raw, s_raw = ps.ar_multiconv2d(dev(z), dev(ctx))
    p32 = {k: f32(v) for k, v in hp.items()}
    ez, es = O.iaf_step(f32(z), f32(ctx), p32, [160, 160])
    em, esr = O.ar_multiconv2d(f32(z), f32(ctx), p32, [160, 160], [32, 32])
    assert np.abs(z_new.cpu().numpy() - ez).max() < 1e-4 and np.abs(logsd.cpu().numpy() - es).max() < 1e-5
    assert np.abs(m_raw.cpu().numpy() - em).max() < 1e-4 and np.abs(s_raw.cpu().numpy() - esr).max() < 1e-4
    assert ps.exchange_errors() == 0


@pytest.mark.parametrize("variant", ["theano", "theano_flipmask"])
def test_pair_step_theano_statements(amd, variant):
    """graphy/nodes/ar.py:378-423 at 8-pixel rows through the pair form (image rotated by 180 degrees, border channel as an epilogue
    term): equal to the recomputing kernel of the same statement; tests/test_hip_fused_step.py holds both to the oracle"""
    rng = np.random.RandomState(3)
    w, sizes = {}, [32, 160, 160]
    for i in range(2):
        w["%d_w" % i] = 0.05 * rng.standard_normal((sizes[i + 1], sizes[i] + 1, 3, 3))
        w["%d_b" % i] = 0.1 * rng.standard_normal(sizes[i + 1])
        w["%d_s" % i] = 0.1 * rng.standard_normal(sizes[i + 1])
        w["out_%d_w" % i] = 0.05 * rng.standard_normal((32, 161, 3, 3))
        w["out_%d_b" % i] = 0.1 * rng.standard_normal(32)
        w["out_%d_s" % i] = 0.1 * rng.standard_normal(32)
    ps, rc = amd.ARStack(32, [160, 160], variant=variant), amd.ARStack(32, [160, 160], variant=variant)
    ps.set_halo_exchange_debug(PAIR)
    rc.set_halo_exchange(False)
    params = {k: dev(v) for k, v in w.items()}
    ps.prepare(params)
    rc.prepare(params)
    assert ps.step_pairs(8, 8, 8) and not rc.step_pairs(8, 8, 8)
    g = torch.Generator(device="cuda").manual_seed(4)
    for H in (8, 5):
        z, ctx = torch.randn(8, 32, H, 8, device="cuda", generator=g), torch.randn(8, 160, H, 8, device="cuda", generator=g)
        zp, sp = ps.iaf_step(z, ctx)
        zr, sr = rc.iaf_step(z, ctx)
        _close(sp, sr, "logsd")
        _close(zp, zr, "z")
    assert ps.exchange_errors() == 0


@pytest.mark.parametrize("knob", [0, 1, 2, 3], ids=lambda k: "knob%d" % k)
@pytest.mark.parametrize("B", [32, 5, 64, 300], ids=lambda b: "B%d" % b)
def test_results_do_not_depend_on_order_or_placement(amd, B, knob):
    ps, rc, _ = _stacks(amd, 11)
    ps.set_fuse_step("always")                                    # (beyond the size rule too: 300 images = nine rounds of the chip)
    rc.set_fuse_step("always")
    assert ps.step_pairs(B, 8, 8)
    ps.set_halo_exchange_debug(PAIR | knob)
    g = torch.Generator(device="cuda").manual_seed(1000 + B + knob)
    for rep in range(4 if B <= 64 else 2):
        z = torch.randn(B, 32, 8, 8, device="cuda", generator=g)
        ctx = torch.randn(B, 160, 8, 8, device="cuda", generator=g)
        zx, sx = ps.iaf_step(z, ctx)
        zr, sr = rc.iaf_step(z, ctx)
        _close(sx, sr, "logsd rep %d" % rep)
        _close(zx, zr, "z rep %d" % rep)
    assert ps.exchange_errors() == 0


def test_posterior_block_and_training_forward_in_pair_form(amd):
    """tf_train.py:56-85 at 8x8 (sample in front, KL sums and free bits behind -- the last workgroup's helper waves) and the training
    forward, which also stores the hidden activations the backward reads: each half writes its own channels"""
    ps, rc, hp = _stacks(amd, 13)
    ps.set_halo_exchange_debug(PAIR | 3)
    g = torch.Generator(device="cuda").manual_seed(5)
    B = 32
    for rep in range(3):
        t = lambda c, s=1.0: s * torch.randn(B, c, 8, 8, device="cuda", generator=g)
        args = dict(qz_mean=t(32), qz_logsd=t(32, .25), rz_mean=t(32), rz_logsd=t(32, .25), pz_mean=t(32), pz_logsd=t(32, .25),
                    eps=t(32), up_context=t(160), down_context=t(160))
        ox = ps.posterior_block(kl_min=0.25, **args)
        orr = rc.posterior_block(kl_min=0.25, **args)
        for k in ("z", "kl_obj", "kl_cost"):
            a, r = ox[k].double(), orr[k].double()
            assert torch.isfinite(a).all()
            assert float((a - r).abs().max()) <= 3e-6 * max(1.0, float(r.abs().max())), k
    assert ps.exchange_errors() == 0
    # training: forward + backward through both stacks, every gradient equal
    pt, rt = amd.ARStack(32, [160, 160]), amd.ARStack(32, [160, 160])
    pt.set_halo_exchange_debug(PAIR)
    rt.set_halo_exchange(False)
    params = {k: dev(v) for k, v in hp.items()}
    for st in (pt, rt):
        st.set_training(True)
        st.prepare(params)
    z, ctx = torch.randn(B, 32, 8, 8, device="cuda", generator=g), torch.randn(B, 160, 8, 8, device="cuda", generator=g)
    dz, ds = torch.randn(B, 32, 8, 8, device="cuda", generator=g), torch.randn(B, 32, 8, 8, device="cuda", generator=g)
    outs = []
    for st in (pt, rt):
        zn, ls = st.iaf_step_train(z, ctx)
        outs.append((zn, ls, st.iaf_step_backward(z, ctx, zn, ls, dz, ds, params)))
    _close(outs[0][0], outs[1][0], "z (training forward)")
    ga, gb = outs[0][2], outs[1][2]
    flat = lambda o: [o] if torch.is_tensor(o) else [v for k in sorted(o) for v in flat(o[k])] if isinstance(o, dict) else [v for e in o for v in flat(e)]
    for a, r in zip(flat(ga), flat(gb)):
        assert float((a.double() - r.double()).abs().max()) <= 1e-5 * max(1.0, float(r.double().abs().max()))


@pytest.mark.parametrize("launches", [1, 3], ids=lambda n: "graph_of_%d" % n)
def test_replayed_graphs(amd, launches):
    ps, rc, _ = _stacks(amd, 15)
    g = torch.Generator(device="cuda").manual_seed(21)
    z = torch.randn(32, 32, 8, 8, device="cuda", generator=g)
    ctx = torch.randn(32, 160, 8, 8, device="cuda", generator=g)
    ps.iaf_step(z, ctx)                                         # (warm-up: the exchange set is allocated outside the capture)
    torch.cuda.synchronize()
    outs = [(torch.empty_like(z), torch.empty_like(z)) for _ in range(launches)]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cur = z
        for o in outs:
            ps.iaf_step(cur, ctx, out=o)
            cur = o[0]
    for rep in range(5):
        z.copy_(torch.randn(32, 32, 8, 8, device="cuda", generator=g))
        ctx.copy_(torch.randn(32, 160, 8, 8, device="cuda", generator=g))
        graph.replay()
        cur = z
        for o in outs:
            zr, sr = rc.iaf_step(cur, ctx)
            _close(o[1], sr, "logsd")
            _close(o[0], zr, "z")
            cur = zr
    assert ps.exchange_errors() == 0


def test_a_partner_that_never_hands_over_is_loud_and_the_stack_recovers(amd):
    ps, rc, _ = _stacks(amd, 16)
    g = torch.Generator(device="cuda").manual_seed(33)
    B = 8
    z = torch.randn(B, 32, 8, 8, device="cuda", generator=g)
    ctx = torch.randn(B, 160, 8, 8, device="cuda", generator=g)
    zr, sr = rc.iaf_step(z, ctx)
    zx, sx = ps.iaf_step(z, ctx)
    _close(zx, zr, "before the fault")
    ps.set_halo_exchange_debug(PAIR | 8)                         # the second half of (image 0, rows 0-1) keeps its channels to itself
    zf, sf = ps.iaf_step(z, ctx)
    torch.cuda.synchronize()
    # its partner -- z channels 0..15 of those two rows -- waited, gave up and says so in its numbers; nobody else is touched
    assert torch.isnan(zf[0, :16, 0:2]).all() and torch.isnan(sf[0, :16, 0:2]).all()
    mask = torch.ones_like(zf, dtype=torch.bool)
    mask[0, :16, 0:2] = False
    assert torch.isfinite(zf[mask]).all()
    assert float((zf[mask].double() - zr[mask].double()).abs().max()) <= 2e-6 * max(1.0, float(zr.abs().max()))
    assert ps.exchange_errors() != 0
    ps.set_halo_exchange_debug(PAIR)
    with pytest.raises(amd.ExchangeError):                       # said once, as the next call's status ...
        ps.iaf_step(z, ctx)
    assert not ps.step_pairs(B, 8, 8) and ps.step_is_fused(B, 8, 8) == 1
    z2, s2 = ps.iaf_step(z, ctx)                                 # ... and the stack carries on with the one-row kernel
    assert torch.equal(z2, zr) and torch.equal(s2, sr)
    ps.set_halo_exchange(True)
    assert ps.step_pairs(B, 8, 8) and ps.exchange_errors() == 0
    z3, s3 = ps.iaf_step(z, ctx)
    _close(z3, zr, "re-armed")
    _close(s3, sr, "re-armed")
    assert ps.exchange_errors() == 0


@pytest.mark.parametrize("hw", [8, 16])
def test_nan_inputs_with_the_all_ones_pattern_are_data_not_an_exchange_failure(amd, hw):
    """Until round 5 the hand-overs treated a dword of all ones as "not there yet"; a caller's NaN may carry exactly that pattern
    (uninitialised memory) and reaches the activations of a channel pair unchanged (x + NaN keeps the payload): the consumer stalled until
    its bounded wait gave up.  Now the pattern is one no arithmetic produces: NaN in the outputs that depend on the inputs' NaNs, the rest
    untouched, no ExchangeError."""
    ps, rc, _ = _stacks(amd, 19)
    B = 8
    assert ps.step_pairs(B, hw, hw) if hw == 8 else ps.step_exchanges(B, hw, hw)
    assert not amd.ARStack(32, [160, 160]).step_pairs(B, 8, 8)      # (opt-in)
    g = torch.Generator(device="cuda").manual_seed(44)
    z = torch.randn(B, 32, hw, hw, device="cuda", generator=g)
    ctx = torch.randn(B, 160, hw, hw, device="cuda", generator=g)
    zr, sr = rc.iaf_step(z, ctx)
    allones = torch.tensor([-1, -1], dtype=torch.int32, device="cuda").view(torch.float32)       # 0xffffffff: a NaN
    assert torch.isnan(allones).all()
    for img, chans, row in ((0, slice(0, 2), 2), (1, slice(96, 98), 4), (2, slice(158, 160), hw - 2)):
        ctx[img, chans, row, 3] = allones                        # an adjacent channel pair = one dword of a bf16 plane
    z[3, 4:6, 2, 5] = allones
    # ... and the pattern the buffers really hold between launches (IAF_XSENT, a pair of SIGNALLING bf16 NaNs) as an fp32 input: arithmetic
    # quiets it on the way, so it cannot reach an exported activation either
    snan = torch.tensor([0xffbfffbf - (1 << 32)] * 2, dtype=torch.int32, device="cuda").view(torch.float32)
    assert torch.isnan(snan).all()
    ctx[0, 8:10, 2, 7] = snan
    z[3, 8:10, 2, 9 % hw] = snan
    for rep in range(2):
        zx, sx = ps.iaf_step(z, ctx)
        torch.cuda.synchronize()
        assert ps.exchange_errors() == 0
        for img in range(4):
            assert bool(torch.isnan(zx[img]).any())
        _close(zx[4:], zr[4:], "images without NaN inputs")
        _close(sx[4:], sr[4:], "images without NaN inputs")
    ctx2 = torch.randn(B, 160, hw, hw, device="cuda", generator=g)
    z2 = torch.randn(B, 32, hw, hw, device="cuda", generator=g)
    za, sa = ps.iaf_step(z2, ctx2)                               # the buffers are clean afterwards
    zb, sb = rc.iaf_step(z2, ctx2)
    _close(za, zb, "after the NaN launches")
    assert ps.exchange_errors() == 0
