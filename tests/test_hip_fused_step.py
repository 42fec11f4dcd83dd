"""The one-launch IAF step (iaf_amd/csrc/iaf_step_fused.hpp: every masked conv of the stack + the affine transform /
log-det term / KL elements in ONE kernel, a workgroup per R full-width image rows, hidden activations in LDS, halo rows
recomputed -- or, at 16-pixel rows of the TF statement, handed from row block to row block through device memory) against the
fp64 oracle and against the layer-by-layer kernels it replaces at the BASELINE sizes.

Every other GPU parity test that builds an ARStack(32, [160, 160]) or (32, [64]) on 16- or 8-pixel-wide images already
runs this kernel (it is the default there); this file pins down (a) WHERE it runs and where it steps aside, (b) all
four epilogue modes against the oracle incl. image heights that are not a multiple of the rows per workgroup and the
posterior-sample input, (c) agreement with the layer-by-layer path to fp32 round-off, (d) that a sample's result does not
depend on its batch and that the autoregressive structure is bit-exact (tf_train.py:69-72 is only a valid flow if
z_new[i] depends on z[<i] alone)."""
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


def test_where_the_step_runs_as_one_launch(amd):
    st = amd.ARStack(32, [160, 160])
    assert st.step_is_fused(32, 16, 16) == 2 and st.step_is_fused(32, 8, 8) == 1 and st.step_is_fused(256, 16, 16) == 2
    assert not st.step_pairs(32, 8, 8)                           # (the pair form at 8-pixel rows is opt-in: tests/test_hip_pair_step.py)
    assert st.step_is_fused(256, 8, 8) == 0                      # large batch of small images: layer by layer (weight stream)
    assert st.step_is_fused(3, 5, 16) == 2                       # any height, any batch
    assert st.step_is_fused(32, 4, 4) == 0                       # 4-pixel rows, one workgroup per image walking 1.2 MB of weights: no
    assert amd.ARStack(32, [64]).step_is_fused(16, 4, 4) == 4     # ... a whole 4x4 image per workgroup where the weight set is small
    assert amd.ARStack(64, [64] * 4).step_is_fused(32, 4, 4) == 4 and amd.ARStack(64, [128] * 4).step_is_fused(32, 4, 4) == 0
    assert st.step_is_fused(32, 32, 32) == 0                     # no compiled geometry for 32-pixel rows
    assert amd.ARStack(32, [64]).step_is_fused(16, 16, 16) == 2   # BASELINE configs[0]
    deep = amd.ARStack(64, [192] * 4)                            # BASELINE configs[3]: five LDS regions
    # 16-pixel rows at n_h = 192: five regions of R + depth_ar .. R + 1 rows do not fit 160 KiB -- in the halo-exchange form every
    # region holds R + 1 rows and the step is one launch (150 KiB)
    assert deep.step_is_fused(32, 16, 16) == 2 and deep.step_exchanges(32, 16, 16) and deep.step_is_fused(32, 8, 8) == 1
    assert st.step_exchanges(32, 16, 16) and not st.step_exchanges(32, 8, 8)
    assert amd.ARStack(32, [160, 160], variant="theano").step_exchanges(32, 16, 16)            # ... in all three statements
    assert amd.ARStack(64, [64] * 4).step_is_fused(32, 16, 16) == 2 and amd.ARStack(64, [128] * 4).step_is_fused(32, 8, 8) == 1
    assert amd.ARStack(64, [64] * 3).step_is_fused(32, 16, 16) == 0                       # no compiled geometry
    assert amd.ARStack(32, [160, 160], variant="theano").step_is_fused(32, 16, 16) == 2           # all three statements
    assert amd.ARStack(32, [160, 160], variant="theano_flipmask").step_is_fused(32, 8, 8) == 1
    st.set_fuse_step("never")
    assert st.step_is_fused(32, 16, 16) == 0
    st.set_fuse_step("auto")
    st.set_precision("f32")                                       # the one-launch step is bf16x3 arithmetic
    assert st.step_is_fused(32, 16, 16) == 0
    st.set_precision("bf16x3")
    st.set_tuning_bf3(1, 5, 2, 1, 4)                              # a pinned per-layer shape means the layer-by-layer path
    assert st.step_is_fused(32, 16, 16) == 0
    params, z, ctx = _case(1, 2, 32, 160, 2, 4, 32)
    st2 = amd.ARStack(32, [160, 160])
    st2.prepare({k: dev(v) for k, v in params.items()})
    with pytest.raises(amd.UnsupportedError):
        st2.time_layer(-2, dev(z), dev(ctx), reps=2)


@pytest.mark.parametrize("cfg", [(32, 32, 160, 2, 16, 16), (32, 32, 160, 2, 8, 8), (5, 32, 160, 2, 5, 16), (3, 32, 160, 2, 3, 8),
                                 (1, 32, 160, 2, 1, 16), (2, 32, 160, 2, 7, 8), (64, 32, 160, 2, 16, 16), (128, 32, 160, 2, 8, 8), (16, 32, 64, 1, 16, 16),
                                 (16, 32, 64, 1, 8, 8), (4, 32, 64, 1, 3, 16), (128, 32, 64, 1, 8, 8),
                                 (16, 32, 64, 1, 4, 4), (32, 32, 160, 2, 4, 4), (5, 32, 160, 2, 3, 4), (3, 32, 64, 1, 9, 4),
                                 (32, 64, 64, 4, 16, 16), (4, 64, 64, 4, 8, 8), (3, 64, 128, 4, 8, 8), (32, 64, 192, 4, 8, 8),
                                 (3, 64, 192, 4, 4, 4), (2, 64, 64, 4, 5, 16), (2, 64, 128, 4, 3, 4),
                                 (8, 32, 160, 3, 16, 16), (5, 32, 160, 3, 8, 8), (3, 32, 160, 3, 7, 16), (4, 32, 64, 3, 8, 8),
                                 (6, 32, 64, 3, 16, 16), (4, 32, 64, 3, 4, 4),
                                 (32, 32, 128, 2, 16, 16), (5, 32, 128, 2, 8, 8), (3, 32, 128, 2, 5, 4), (2, 32, 128, 2, 3, 16),
                                 (32, 32, 64, 2, 16, 16), (5, 32, 64, 2, 8, 8), (3, 32, 64, 2, 4, 4), (2, 32, 64, 2, 7, 16)],
                         ids=lambda s: "B%d_z%d_h%d_d%d_%dx%d" % s)
def test_iaf_step_and_raw_outputs_vs_oracle(amd, cfg):
    """tf_train.py:69-72 and layers.py:158-166 through the one-launch step; heights that are not a multiple of the rows
    per workgroup, single rows, one sample"""
    B, n_z, n_h, d, H, W = cfg
    params, z, ctx = _case(300 + H + W + d, *cfg)
    st = amd.ARStack(n_z, [n_h] * d)
    st.set_fuse_step("always")                                    # also beyond the size rule: wherever a compiled geometry exists
    assert st.step_is_fused(B, H, W) == (4 if W == 4 else 2 if (W == 16 or B * H >= 1024) else 1)
    st.prepare({k: dev(v) for k, v in params.items()})
    zd, cd = dev(z), dev(ctx)
    z_new, logsd = st.iaf_step(zd, cd)
    m_raw, s_raw = st.ar_multiconv2d(zd, cd)
    p32 = {k: f32(v) for k, v in params.items()}
    chunk = 32
    for b0 in range(0, B, chunk):
        sl = slice(b0, min(B, b0 + chunk))
        ez, es = O.iaf_step(f32(z[sl]), f32(ctx[sl]), p32, [n_h] * d)
        np.testing.assert_allclose(host(logsd[sl]), es, atol=ATOL, rtol=0)
        np.testing.assert_allclose(host(z_new[sl]), ez, atol=ATOL, rtol=0)
        assert np.abs(host(logsd[sl]) - es).max() < 1e-5          # fp32-grade, not just inside the tolerance
        em, esr = O.ar_multiconv2d(f32(z[sl]), f32(ctx[sl]), p32, [n_h] * d, [n_z, n_z])
        np.testing.assert_allclose(host(m_raw[sl]), em, atol=ATOL, rtol=0)
        np.testing.assert_allclose(host(s_raw[sl]), esr, atol=ATOL, rtol=0)
    assert st.time_layer(-2, zd, cd, reps=2) > 0


@pytest.mark.parametrize("kl_min", [0.0, 0.25])
@pytest.mark.parametrize("cfg", [(8, 32, 160, 2, 16, 16), (5, 32, 160, 2, 8, 8), (3, 32, 160, 2, 5, 16), (4, 32, 64, 1, 8, 8),
                                 (6, 32, 160, 2, 4, 4), (8, 32, 128, 2, 16, 16), (5, 32, 64, 2, 8, 8)],
                         ids=lambda s: "B%d_z%d_h%d_d%d_%dx%d" % s)
def test_posterior_block_vs_oracle(amd, cfg, kl_min):
    """the extended unit (tf_train.py:56-85): posterior sample computed in the staging, two contexts, KL elements and free
    bits behind the same launch"""
    B, n_z, n_h, d, H, W = cfg
    rng = np.random.RandomState(77 + H)
    params = gi.ar_multiconv2d_params(rng, n_z, [n_h] * d, [n_z, n_z])
    f = lambda c: rng.standard_normal((B, c, H, W))
    qm, ql, rm, rl, pm, pl = f(n_z), 0.25 * f(n_z), f(n_z), 0.25 * f(n_z), f(n_z), 0.25 * f(n_z)
    uc, dc, eps = f(n_h), f(n_h), f(n_z)
    st = amd.ARStack(n_z, [n_h] * d)
    st.set_fuse_step("always")
    assert st.step_is_fused(B, H, W) > 0
    st.prepare({k: dev(v) for k, v in params.items()})
    out = st.posterior_block(dev(qm), dev(ql), dev(rm), dev(rl), dev(pm), dev(pl), dev(uc), dev(dc), dev(eps), kl_min,
                             want_kl_elem=True)
    e = O.posterior_block(f32(qm), f32(ql), f32(rm), f32(rl), f32(pm), f32(pl), f32(uc), f32(dc), f32(eps),
                          {k: f32(v) for k, v in params.items()}, [n_h] * d, kl_min)
    np.testing.assert_allclose(host(out["z"]), e["z"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(out["kl_elem"]), e["logqs"] - e["logps"], atol=ATOL, rtol=1e-5)
    np.testing.assert_allclose(host(out["kl_cost"]), e["kl_cost"], atol=2e-3, rtol=1e-4)
    np.testing.assert_allclose(host(out["kl_obj"]), e["kl_obj"], atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("cfg", [(32, 32, 160, 2, 16, 16), (32, 32, 160, 2, 8, 8), (6, 32, 64, 1, 16, 16)],
                         ids=lambda s: "B%d_z%d_h%d_d%d_%dx%d" % s)
def test_agrees_with_the_layer_by_layer_path(amd, cfg):
    """same arithmetic (bf16x3 split products, fp32 accumulate), different summation order: fp32 round-off apart"""
    B, n_z, n_h, d, H, W = cfg
    params, z, ctx = _case(500 + H, *cfg)
    dp = {k: dev(v) for k, v in params.items()}
    one, lbl = amd.ARStack(n_z, [n_h] * d), amd.ARStack(n_z, [n_h] * d)
    lbl.set_fuse_step("never")
    one.prepare(dp)
    lbl.prepare(dp)
    assert one.step_is_fused(B, H, W) > 0 and lbl.step_is_fused(B, H, W) == 0
    za, sa = one.iaf_step(dev(z), dev(ctx))
    zb, sb = lbl.iaf_step(dev(z), dev(ctx))
    assert float((sa - sb).abs().max()) < 2e-6 and float((za - zb).abs().max()) < 3e-5
    # the inverse flow runs its Jacobi sweeps through the same launch
    z0a = one.iaf_step_inverse(za, dev(ctx))[0]
    np.testing.assert_allclose(host(z0a), f32(z), atol=2e-4, rtol=0)


@pytest.mark.parametrize("H", [16, 8])
def test_batch_independence_and_ar_structure_bit_exact(amd, H):
    """a workgroup only ever sees one image: samples run one at a time reproduce the batched result bit for bit; and a
    perturbation of z at (pixel q, channel c) leaves every position that precedes it in the IAF ordering bit-identical
    (masked weights are exact zeros in all three bf16 planes; rows a workgroup recomputes for its neighbour's benefit never
    leave it), also across the row-block boundaries between workgroups"""
    B, n_z, n_h, d = 32, 32, 160, 2
    params, z, ctx = _case(2030 + H, B, n_z, n_h, d, H, H)
    st = amd.ARStack(n_z, [n_h] * d)
    st.prepare({k: dev(v) for k, v in params.items()})
    assert st.step_is_fused(B, H, H) > 0 and st.step_is_fused(1, H, H) == st.step_is_fused(B, H, H)
    zd, cd = dev(z), dev(ctx)
    zf, sf = st.iaf_step(zd, cd)
    for b in (0, 13, 31):
        zb, sb = st.iaf_step(zd[b:b + 1].contiguous(), cd[b:b + 1].contiguous())
        assert torch.equal(zb, zf[b:b + 1]) and torch.equal(sb, sf[b:b + 1])
    # the masked convs look right and below (layers.py:133-141): a change of z[c] at pixel (qh, qw) may reach the pixels
    # above and to the left of it, and channels > c of the same pixel -- everything else must not move by a single bit
    for (qh, qw, c) in ((H // 2, H // 2 - 1, 11), (H - 1, H - 1, 31), (1, 0, 0), (2, H - 1, 5)):
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
        assert bool((z1 != zf).any())


def test_repeated_launches_are_deterministic(amd):
    B, n_z, n_h, d, H = 32, 32, 160, 2, 16
    params, z, ctx = _case(9, B, n_z, n_h, d, H, H)
    st = amd.ARStack(n_z, [n_h] * d)
    st.prepare({k: dev(v) for k, v in params.items()})
    zd, cd = dev(z), dev(ctx)
    z0, s0 = st.iaf_step(zd, cd)
    for _ in range(200):
        z1, s1 = st.iaf_step(zd, cd)
        assert torch.equal(z1, z0) and torch.equal(s1, s0)
    assert st.exchange_errors() == 0


def test_halo_exchange_can_be_switched_off_per_stack(amd):
    """iaf_stack_set_halo_exchange: the same stack with and without the exchange agrees to fp32 round-off (the layer that reads
    the imported row sums its taps in two parts); a geometry that only exists in the exchange form steps aside to the
    layer-by-layer kernels when it is off"""
    B, n_z, n_h, d, H = 32, 32, 160, 2, 16
    params, z, ctx = _case(41, B, n_z, n_h, d, H, H)
    st = amd.ARStack(n_z, [n_h] * d)
    st.prepare({k: dev(v) for k, v in params.items()})
    zd, cd = dev(z), dev(ctx)
    assert st.step_exchanges(B, H, H)
    z1, s1 = st.iaf_step(zd, cd)
    st.set_halo_exchange(False)
    assert not st.step_exchanges(B, H, H) and st.step_is_fused(B, H, H) == 2
    z0, s0 = st.iaf_step(zd, cd)
    for a, r in ((z1, z0), (s1, s0)):
        assert float((a - r).abs().max()) <= 2e-6 * max(1.0, float(r.abs().max()))
    st.set_halo_exchange(True)
    z2, s2 = st.iaf_step(zd, cd)
    assert torch.equal(z2, z1) and torch.equal(s2, s1) and st.exchange_errors() == 0
    deep = amd.ARStack(64, [192] * 4)
    assert deep.step_is_fused(32, 16, 16) == 2
    deep.set_halo_exchange(False)
    assert deep.step_is_fused(32, 16, 16) == 0 and deep.step_is_fused(32, 8, 8) == 1


_XCH_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests/golden")
import golden_inputs as gi, iaf_amd
B, H = int(sys.argv[3]), 16
rng = np.random.RandomState(77)
params = gi.ar_multiconv2d_params(rng, 32, [160, 160], [32, 32])
z, ctx = rng.standard_normal((B, 32, H, H)), rng.standard_normal((B, 160, H, H))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
st = iaf_amd.ARStack(32, [160, 160])
st.prepare({k: dev(v) for k, v in params.items()})
zd, cd = dev(z), dev(ctx)
outs = [st.iaf_step(zd, cd) for _ in range(3)][-1]
torch.cuda.synchronize()
assert st.exchange_errors() == 0
np.savez(sys.argv[2], z=outs[0].cpu().numpy(), s=outs[1].cpu().numpy())
"""


@pytest.mark.parametrize("B", [32, 64, 5], ids=["B32_one_round", "B64_two_rounds_of_workgroups", "B5"])
def test_halo_exchange_agrees_with_halo_recompute(amd, B, tmp_path):
    """16-pixel rows of the TF statement: the row blocks of an image hand each other their first hidden rows through
    device memory instead of recomputing them (iaf_step_fused.hpp, XCH).  The imported row is bit for bit what the
    neighbour computed for itself; the layer that reads it sums its taps in two parts (own rows, then the row below), so
    against the recomputing kernel (IAF_FUSE_XCH=0, read once per process: a child process each) the outputs move by fp32
    round-off and no more -- also when the grid does not fit the chip in one round (B = 64: 512 workgroups; a workgroup
    only waits for one with a lower index) -- and no bounded wait may have given up (asserted in the child)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for mode in ("1", "0"):
        out = str(tmp_path / ("xch%s.npz" % mode))
        env = dict(os.environ, IAF_FUSE_XCH=mode)
        subprocess.run([sys.executable, "-c", _XCH_CHILD, root, out, str(B)], check=True, env=env, timeout=600)
        got[mode] = np.load(out)
    for k in ("z", "s"):
        a, r = got["1"][k].astype(np.float64), got["0"][k].astype(np.float64)
        assert np.isfinite(a).all()
        assert np.abs(a - r).max() <= 2e-6 * max(1.0, np.abs(r).max()), (k, np.abs(a - r).max())


def _theano_params(rng, name, n_z, n_h_list):
    w, sizes = {}, [n_z] + n_h_list
    for i in range(len(n_h_list)):
        w["%s_%d_w" % (name, i)] = 0.05 * rng.standard_normal((sizes[i + 1], sizes[i] + 1, 3, 3))
        w["%s_%d_b" % (name, i)] = 0.1 * rng.standard_normal(sizes[i + 1])
        w["%s_%d_s" % (name, i)] = 0.1 * rng.standard_normal(sizes[i + 1])
    for i in range(2):
        w["%s_out_%d_w" % (name, i)] = 0.05 * rng.standard_normal((n_z, sizes[-1] + 1, 3, 3))
        w["%s_out_%d_b" % (name, i)] = 0.1 * rng.standard_normal(n_z)
        w["%s_out_%d_s" % (name, i)] = 0.1 * rng.standard_normal(n_z)
    return w


@pytest.mark.parametrize("flip", [False, True], ids=["plain", "flipmask"])
@pytest.mark.parametrize("cfg", [(32, 32, 160, 2, 16, 16), (5, 32, 160, 2, 8, 8), (3, 32, 160, 2, 5, 16), (16, 32, 64, 1, 16, 16),
                                 (4, 32, 64, 1, 7, 8), (6, 32, 64, 1, 4, 4)], ids=lambda s: "B%d_z%d_h%d_d%d_%dx%d" % s)
def test_theano_statement_through_the_one_launch_step(amd, cfg, flip):
    """graphy/nodes/ar.py's statement (flipped kernel: taps look left / above; border-indicator channel; exp(3s), +1e-8)
    runs the same launch on the image rotated by 180 degrees, the border channel as an epilogue term; flipmask=True keeps the
    TF geometry.  Raw outputs and the up/down_iaf2_nl step (models.py:168-175) vs the oracle, and vs the layer-by-layer path."""
    B, n_z, n_h, d, H, W = cfg
    rng = np.random.RandomState(640 + H + W + (3 if flip else 0))
    w = _theano_params(rng, "q", n_z, [n_h] * d)
    z, ctx = rng.standard_normal((B, n_z, H, W)), rng.standard_normal((B, n_h, H, W))
    variant = "theano_flipmask" if flip else "theano"
    one, lbl = amd.ARStack(n_z, [n_h] * d, variant=variant), amd.ARStack(n_z, [n_h] * d, variant=variant)
    lbl.set_fuse_step("never")
    rel = {k[2:]: dev(v) for k, v in w.items()}
    one.prepare(rel)
    lbl.prepare(rel)
    assert one.step_is_fused(B, H, W) > 0 and lbl.step_is_fused(B, H, W) == 0
    w32 = {k: f32(v) for k, v in w.items()}
    m_raw, s_raw = one.ar_multiconv2d(dev(z), dev(ctx))
    em, es = O.theano_multiconv2d(f32(z), f32(ctx), w32, "q", n_z, [n_h] * d, [n_z, n_z], flipmask=flip)
    np.testing.assert_allclose(host(m_raw), em, atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(s_raw), es, atol=ATOL, rtol=0)
    z_new, logsd = one.iaf_step(dev(z), dev(ctx))
    ez, el = O.theano_iaf2_nl(f32(z), f32(ctx), w32, "q", n_z, [n_h] * d, flipmask=flip)
    np.testing.assert_allclose(host(z_new), ez, atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(logsd), el, atol=ATOL, rtol=0)
    zb, sb = lbl.iaf_step(dev(z), dev(ctx))
    assert float((logsd - sb).abs().max()) < 2e-6 and float((z_new - zb).abs().max()) < 3e-5


@pytest.mark.parametrize("kl_min", [0.0, 0.25])
@pytest.mark.parametrize("cfg", [(32, 16), (5, 16), (32, 8), (3, 8), (64, 8)], ids=lambda c: "B%d_%dx%d" % (c[0], c[1], c[1]))
def test_free_bits_reductions_inside_the_launch_equal_the_finish_launch(amd, cfg, kl_min):
    """tf_train.py:77-85 behind the one-launch step: in the kernels with helper waves (16-pixel rows in the exchange form, the
    BASELINE 8-pixel geometry) the LAST workgroup to arrive sums the per-row-block KL sums and applies batch mean / max(., kl_min) /
    channel sum itself (StepP::fin_*), in iaf_kl_finish_kernel's summation order -- so kl_obj and kl_cost must be BIT-identical to
    what the separate finish launch gives (debug knob 16 selects it); the training forward takes the same path and also gets its
    gate (1 where the batch mean of a channel's KL exceeds kl_min) from there."""
    B, HW = cfg
    rng = np.random.RandomState(90 + B + HW)
    params = {k: dev(v) for k, v in gi.ar_multiconv2d_params(rng, 32, [160, 160], [32, 32]).items()}
    one, two, three = amd.ARStack(32, [160, 160]), amd.ARStack(32, [160, 160]), amd.ARStack(32, [160, 160])
    two.set_halo_exchange_debug(16)                               # ... by the finish launch
    three.set_training(True)
    for st_ in (one, two, three):
        st_.prepare(params)
    assert "last workgroup" in one.posterior_block_launches(B, HW, HW) or B * (HW // (2 if HW == 16 else 1)) * 32 > 16384
    g = torch.Generator(device="cuda").manual_seed(B)
    for rep in range(3):
        t = lambda c, s=1.0: s * torch.randn(B, c, HW, HW, device="cuda", generator=g)
        args = [t(32), t(32, .25), t(32), t(32, .25), t(32), t(32, .25), t(160), t(160), t(32)]
        a = one.posterior_block(*args, kl_min)
        b = two.posterior_block(*args, kl_min)
        c = three.posterior_block_train(*args, kl_min)
        for o in (b, c):
            assert torch.equal(a["z"], o["z"])
            assert torch.equal(a["kl_cost"], o["kl_cost"]) and torch.equal(a["kl_obj"], o["kl_obj"])
        want = a["kl_cost"].double()
        if kl_min > 0:
            assert float((a["kl_obj"] - a["kl_obj"][0]).abs().max()) == 0.0
        else:
            assert torch.equal(a["kl_obj"], a["kl_cost"])
        assert torch.isfinite(want).all()
