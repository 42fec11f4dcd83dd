"""iaf_step_inverse without the host in its loop (round 6, VERDICT r05 "next" #8 / row f4b): the Jacobi sweeps of the inverse IAF step
are queued at once, the residual test runs on the device and raises the word the remaining sweep launches read (the one-launch step kernel
in MODE_INVERSE returns immediately), a last launch leaves the result in z0.  The reference never inverts the flow (tf_train.py:60-66,
models.py:330-359): checked by round trips against the forward step (tf_train.py:69-72), against the synchronising call, and as a captured
hipGraph replayed on fresh inputs."""
import numpy as np
import pytest
import torch

import golden_inputs as gi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    import iaf_amd
    iaf_amd._capi.lib()
    return iaf_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _case(amd, n_z, n_h, d, B, H, seed, precision=None):
    rng = np.random.RandomState(seed)
    params = gi.ar_multiconv2d_params(rng, n_z, [n_h] * d, [n_z, n_z])
    st = amd.ARStack(n_z, [n_h] * d)
    if precision:
        st.set_precision(precision)
    st.prepare({k: dev(v) for k, v in params.items()})
    z0, ctx = dev(rng.standard_normal((B, n_z, H, H))), dev(rng.standard_normal((B, n_h, H, H)))
    return st, z0, ctx, rng


@pytest.mark.parametrize("cfg", [(32, 160, 2, 32, 16, None), (32, 160, 2, 32, 8, None), (32, 160, 2, 32, 16, "bf16x3"), (32, 64, 1, 16, 4, None),
                                 (64, 128, 4, 8, 16, None), (24, 72, 2, 3, 8, None)],
                         ids=["config2_16_f16x2", "config2_8_f16x2", "config2_16_bf16x3", "config1_4x4", "config4_nh128_16", "generic_24_72"])
def test_queued_inverse_round_trip_and_equal_to_the_synchronising_call(amd, cfg):
    n_z, n_h, d, B, H, prec = cfg
    st, z0, ctx, _ = _case(amd, n_z, n_h, d, B, H, 7 + H + n_h, prec)
    z, logsd = st.iaf_step(z0, ctx)
    back, lb, sweeps, res = st.iaf_step_inverse(z, ctx, max_sweeps=60, tol=1e-6, check_every=2)
    assert 2 <= sweeps < 60 and 0 <= res <= 1e-6
    assert (back - z0).abs().max().item() < 2e-5 and (lb - logsd).abs().max().item() < 2e-5
    stats = torch.zeros(2, dtype=torch.int32, device="cuda")
    q, lq = st.iaf_step_inverse_queued(z, ctx, max_sweeps=60, tol=1e-6, check_every=2, stats=stats)
    torch.cuda.synchronize()
    assert torch.equal(q, back) and torch.equal(lq, lb)                        # the same sweeps, the same early stop
    assert int(stats[0].item()) == sweeps and stats[1:].view(torch.float32).item() == pytest.approx(res, abs=0)
    # an odd and an even number of sweeps until convergence land in z0 alike (the ping-pong's parity is the finish launch's business)
    for ms in (sweeps, sweeps + 1, sweeps + 7):
        q2, _ = st.iaf_step_inverse_queued(z, ctx, max_sweeps=ms, tol=1e-6, check_every=1)
        assert (q2 - z0).abs().max().item() < 2e-5, ms


def test_queued_inverse_as_a_graph_replayed_on_fresh_inputs(amd):
    """no host synchronisation anywhere in the call: it captures, and a replay converges for inputs it has never seen"""
    st, z0, ctx, rng = _case(amd, 32, 160, 2, 32, 16, 99)
    zin, cin = torch.empty_like(z0), torch.empty_like(ctx)
    out = (torch.empty_like(z0), torch.empty_like(z0))
    stats = torch.zeros(2, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    z, _ = st.iaf_step(z0, ctx)
    zin.copy_(z); cin.copy_(ctx)
    with torch.cuda.stream(side):
        side.wait_stream(torch.cuda.current_stream())
        st.iaf_step_inverse_queued(zin, cin, max_sweeps=24, tol=1e-6, check_every=2, out=out, stats=stats)      # warm-up: workspace, exchange set
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            st.iaf_step_inverse_queued(zin, cin, max_sweeps=24, tol=1e-6, check_every=2, out=out, stats=stats)
    for i in range(3):
        a = dev(rng.standard_normal(tuple(z0.shape)))
        c = dev(rng.standard_normal(tuple(ctx.shape)))
        za, _ = st.iaf_step(a, c)
        zin.copy_(za); cin.copy_(c)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert (out[0] - a).abs().max().item() < 2e-5
        assert 2 <= int(stats[0].item()) < 24
    assert st.exchange_errors() == 0


def test_never_converging_inputs_run_every_sweep_and_say_so(amd):
    """a NaN in z: the residual is +inf at every test, max_sweeps sweeps run, sweeps_done = max_sweeps"""
    st, z0, ctx, _ = _case(amd, 32, 160, 2, 4, 8, 5)
    z, _ = st.iaf_step(z0, ctx)
    z[0, 0, 0, 0] = float("nan")
    _, _, sweeps, res = st.iaf_step_inverse(z, ctx, max_sweeps=6, tol=1e-6, check_every=1)
    assert sweeps == 6 and res == float("inf")
