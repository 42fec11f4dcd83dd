"""Randomised shape sweep on the GPU: small problems with awkward geometry (single pixels, one-row / one-column images,
widths that are not multiples of anything, pixel counts that do not fill a 16-pixel tile, batch 1, every channel count
class) through the masked stack and the plain convs, against the oracle.  Seeds are fixed: failures reproduce."""
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


def _stack_cases():
    rng = np.random.RandomState(2024)
    chans = [(16, 16), (16, 32), (32, 16), (32, 64), (64, 64), (32, 160), (48, 48), (16, 80), (4, 8), (6, 6), (3, 9)]
    cases = []
    for i in range(28):
        n_z, n_h = chans[rng.randint(len(chans))]
        d = int(rng.randint(0, 4))
        B = int(rng.choice([1, 1, 2, 3, 5]))
        H, W = [(1, 1), (1, 7), (9, 1), (2, 2), (3, 5), (4, 4), (5, 3), (7, 7), (1, 33), (6, 11), (8, 8), (2, 19)][rng.randint(12)]
        cases.append((i, n_z, n_h, d, B, H, W))
    return cases


@pytest.mark.parametrize("case", _stack_cases(), ids=lambda c: "s%d_z%d_h%d_d%d_B%d_%dx%d" % c)
def test_stack_random_geometry(amd, case):
    seed, n_z, n_h, d, B, H, W = case
    rng = np.random.RandomState(1000 + seed)
    params = gi.ar_multiconv2d_params(rng, n_z, [n_h] * d, [n_z, n_z])
    z, ctx = rng.standard_normal((B, n_z, H, W)), rng.standard_normal((B, n_h, H, W))
    stack = amd.ARStack(n_z, [n_h] * d)
    stack.prepare({k: dev(v) for k, v in params.items()})
    p32 = {k: f32(v) for k, v in params.items()}
    z_new, logsd = stack.iaf_step(dev(z), dev(ctx) if d > 0 else None)
    ez, es = O.iaf_step(f32(z), f32(ctx), p32, [n_h] * d)
    np.testing.assert_allclose(host(z_new), ez, atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(logsd), es, atol=ATOL, rtol=0)
    if d > 0:
        f = lambda c: rng.standard_normal((B, c, H, W))
        qm, ql, rm, rl, pm, pl, dc, eps = f(n_z), 0.2 * f(n_z), f(n_z), 0.2 * f(n_z), f(n_z), 0.2 * f(n_z), f(n_h), f(n_z)
        out = stack.posterior_block(dev(qm), dev(ql), dev(rm), dev(rl), dev(pm), dev(pl), dev(ctx), dev(dc), dev(eps), 0.25)
        e = O.posterior_block(f32(qm), f32(ql), f32(rm), f32(rl), f32(pm), f32(pl), f32(ctx), f32(dc), f32(eps), p32,
                              [n_h] * d, 0.25)
        np.testing.assert_allclose(host(out["z"]), e["z"], atol=ATOL, rtol=0)
        np.testing.assert_allclose(host(out["kl_cost"]), e["kl_cost"], atol=2e-3, rtol=1e-4)
        np.testing.assert_allclose(host(out["kl_obj"]), e["kl_obj"], atol=2e-3, rtol=1e-4)
        back, _, _, res = stack.iaf_step_inverse(z_new, dev(ctx), max_sweeps=H * W * n_z + 2, tol=1e-6, check_every=1)
        np.testing.assert_allclose(host(back), f32(z), atol=5e-5, rtol=0)


def _conv_cases():
    rng = np.random.RandomState(77)
    chans = [(16, 16), (16, 48), (48, 16), (32, 96), (160, 64), (64, 160), (80, 80), (5, 7), (12, 20), (8, 8)]
    cases = []
    for i in range(24):
        n_in, n_out = chans[rng.randint(len(chans))]
        B = int(rng.choice([1, 2, 3]))
        H, W = [(1, 1), (1, 9), (10, 1), (2, 3), (5, 5), (3, 17), (8, 8), (4, 6), (1, 40), (7, 2)][rng.randint(10)]
        cases.append((i, n_in, n_out, B, H, W, bool(rng.randint(2)), bool(rng.randint(2)), int(rng.randint(3))))
    return cases


@pytest.mark.parametrize("case", _conv_cases(), ids=lambda c: "c%d_%dto%d_B%d_%dx%d_elu%d_res%d_m%d" % c)
def test_conv3x3_random_geometry(amd, case):
    """plain (m=0) and single masked (m=1: zerodiagonal False, m=2: True) convs, optional fused ELU / residual"""
    seed, n_in, n_out, B, H, W, elu, res, m = case
    if m and not (n_in % n_out == 0 or n_out % n_in == 0):
        m = 0
    rng = np.random.RandomState(500 + seed)
    p = gi.conv_params(rng, n_in, n_out)
    x, r = rng.standard_normal((B, n_in, H, W)), rng.standard_normal((B, n_out, H, W))
    conv = amd.WNConv2d(n_in, n_out, ar_mask=None if m == 0 else (m == 2))
    conv.prepare(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    y = conv(dev(x), elu_input=elu, residual=dev(r) if res else None)[0]
    xin = O.elu(f32(x)) if elu else f32(x)
    if m == 0:
        e = O.conv2d(xin, f32(p["V"]), f32(p["g"]), f32(p["b"]))
    else:
        e = O.ar_conv2d(xin, f32(p["V"]), f32(p["g"]), f32(p["b"]), zerodiagonal=(m == 2))
    if res:
        e = f32(r) + 0.1 * e
    np.testing.assert_allclose(host(y), e, atol=ATOL, rtol=0)


def test_too_wide_image_is_a_clean_error(amd):
    """a 9-tap conv with 160 input channels keeps TM + 2(W+1) pixel slots in LDS: W = 128 does not fit 160 KiB.  The
    engine must say so (IAF_ERR_UNSUPPORTED -> ValueError), not crash or compute garbage."""
    conv = amd.WNConv2d(160, 32)
    rng = np.random.RandomState(0)
    p = gi.conv_params(rng, 160, 32)
    conv.prepare(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    with pytest.raises(ValueError):
        conv(torch.zeros((1, 160, 2, 128), device="cuda"))
    y = conv(torch.zeros((1, 160, 2, 64), device="cuda"))[0]            # 64 wide still fits
    assert torch.isfinite(y).all()
