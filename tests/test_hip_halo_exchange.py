"""The hand-over of halo rows between the row blocks of an image inside the one-launch IAF step (iaf_amd/csrc/iaf_step_fused.hpp,
"XCH"; the step itself is tf_train.py:69-72 over layers.py:158-166): that its results do not depend on dispatch order, timing or
workgroup -> XCD placement (MI355X_MICROARCH.md: "placement-independent protocols only"), and that it cannot fail silently.

 * order: a workgroup takes a ticket from a work list and only waits for the holder of a lower ticket.  The debug knobs
   (include/iaf_hip.h, iaf_stack_set_halo_exchange_debug) scramble what the kernel could otherwise have relied on: 1 = the
   list is chosen by a hash of the workgroup index (neighbouring row blocks land on arbitrary XCDs, lists run dry and
   workgroups take from other lists), 2 = every workgroup delays its ticket by a pseudo-random time (tickets out of dispatch
   order).  Every combination must give the recomputing kernel's numbers, launch after launch with fresh inputs (a stale
   row of an earlier launch would be an O(1) error), on grids of less than one and of many rounds of the chip.
 * hand-over: the data is the flag -- a consumer leaves the "not there yet" pattern behind in every piece it has taken, so the
   same tests also prove that re-arming is complete (a piece left armed with old data would be taken for new).
 * failure: knob 8 makes one producer skip a row; its consumer's bounded wait gives up, and then NaN must come out, the error
   word must be set, the next call must raise ExchangeError, the stack must carry on with the recomputing kernels and the
   exchange must come back after set_halo_exchange(True)."""
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


def _stacks(amd, n_z, n_hs, seed, variant="tf"):
    """(exchanging stack, recomputing stack) on the same weights"""
    rng = np.random.RandomState(seed)
    params = {k: dev(v) for k, v in gi.ar_multiconv2d_params(rng, n_z, n_hs, [n_z, n_z]).items()}
    xs, rc = amd.ARStack(n_z, n_hs), amd.ARStack(n_z, n_hs)
    rc.set_halo_exchange(False)
    xs.prepare(params)
    rc.prepare(params)
    return xs, rc


def _close(a, r, what):
    a, r = a.double(), r.double()
    assert torch.isfinite(a).all(), what
    err = float((a - r).abs().max())
    assert err <= 2e-6 * max(1.0, float(r.abs().max())), (what, err)


@pytest.mark.parametrize("knob", [0, 1, 2, 3], ids=lambda k: "knob%d" % k)
@pytest.mark.parametrize("B", [32, 5, 64, 300], ids=lambda b: "B%d" % b)
def test_results_do_not_depend_on_order_or_placement(amd, B, knob):
    xs, rc = _stacks(amd, 32, [160, 160], 11)
    assert xs.step_exchanges(B, 16, 16) and not rc.step_exchanges(B, 16, 16) and rc.step_is_fused(B, 16, 16) == 2
    xs.set_halo_exchange_debug(knob)
    g = torch.Generator(device="cuda").manual_seed(1000 + B + knob)
    for rep in range(4 if B <= 64 else 2):
        z = torch.randn(B, 32, 16, 16, device="cuda", generator=g)
        ctx = torch.randn(B, 160, 16, 16, device="cuda", generator=g)
        zx, sx = xs.iaf_step(z, ctx)
        zr, sr = rc.iaf_step(z, ctx)
        _close(sx, sr, "logsd rep %d" % rep)
        _close(zx, zr, "z rep %d" % rep)
    assert xs.exchange_errors() == 0


@pytest.mark.parametrize("knob", [0, 3], ids=lambda k: "knob%d" % k)
@pytest.mark.parametrize("n_h", [64, 192])
def test_deep_stack_exchange_under_scrambled_order(amd, n_h, knob):
    """config 3 (n_z = 64, depth_ar = 4): four exported rows per block; n_h = 192 exists ONLY in the exchange form at 16-pixel rows,
    so its reference is the layer-by-layer path"""
    xs, rc = _stacks(amd, 64, [n_h] * 4, 12)
    xs.set_halo_exchange_debug(knob)
    assert xs.step_exchanges(8, 16, 16)
    if n_h == 192:
        assert rc.step_is_fused(8, 16, 16) == 0
    g = torch.Generator(device="cuda").manual_seed(77 + n_h)
    for rep in range(3):
        z = torch.randn(8, 64, 16, 16, device="cuda", generator=g)
        ctx = torch.randn(8, n_h, 16, 16, device="cuda", generator=g)
        zx, sx = xs.iaf_step(z, ctx)
        zr, sr = rc.iaf_step(z, ctx)
        a, r = sx.double(), sr.double()
        assert float((a - r).abs().max()) < 5e-6 and float((zx.double() - zr.double()).abs().max()) < 1e-4
    assert xs.exchange_errors() == 0


@pytest.mark.parametrize("knob", [0, 3], ids=lambda k: "knob%d" % k)
def test_depth_3_exchanges_three_rows_per_block(amd, knob):
    """depth_ar = 3 (models.py:92 allows any depth): an odd number of hidden layers through the same hand-over"""
    xs, rc = _stacks(amd, 32, [160] * 3, 18)
    xs.set_halo_exchange_debug(knob)
    # (three hidden regions of R + 3 ... R + 1 rows do not fit the LDS at n_h = 160: without the exchange this size runs layer by layer)
    assert xs.step_exchanges(16, 16, 16) and not rc.step_exchanges(16, 16, 16) and rc.step_is_fused(16, 16, 16) == 0
    g = torch.Generator(device="cuda").manual_seed(3)
    for rep in range(3):
        z = torch.randn(16, 32, 16, 16, device="cuda", generator=g)
        ctx = torch.randn(16, 160, 16, 16, device="cuda", generator=g)
        zx, sx = xs.iaf_step(z, ctx)
        zr, sr = rc.iaf_step(z, ctx)
        assert float((sx.double() - sr.double()).abs().max()) < 5e-6 and float((zx.double() - zr.double()).abs().max()) < 1e-4
    assert xs.exchange_errors() == 0


def test_posterior_block_exchange_under_scrambled_order(amd):
    """tf_train.py:56-85 in the one-launch form (sample in front, KL sums behind) through the same exchange"""
    xs, rc = _stacks(amd, 32, [160, 160], 13)
    xs.set_halo_exchange_debug(3)
    g = torch.Generator(device="cuda").manual_seed(5)
    B = 32
    for rep in range(3):
        t = lambda c, s=1.0: s * torch.randn(B, c, 16, 16, device="cuda", generator=g)
        args = dict(qz_mean=t(32), qz_logsd=t(32, .25), rz_mean=t(32), rz_logsd=t(32, .25), pz_mean=t(32), pz_logsd=t(32, .25),
                    eps=t(32), up_context=t(160), down_context=t(160))
        ox = xs.posterior_block(kl_min=0.25, **args)
        orr = rc.posterior_block(kl_min=0.25, **args)
        for k in ("z", "kl_obj", "kl_cost"):
            a, r = ox[k].double(), orr[k].double()
            assert torch.isfinite(a).all()
            assert float((a - r).abs().max()) <= 3e-6 * max(1.0, float(r.abs().max())), k
    assert xs.exchange_errors() == 0


def test_two_streams_share_a_stack(amd):
    """SURVEY 8b Threading: re-entrant per stream -- every stream has its own exchange set"""
    xs, rc = _stacks(amd, 32, [160, 160], 14)
    g = torch.Generator(device="cuda").manual_seed(9)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ins = [(torch.randn(32, 32, 16, 16, device="cuda", generator=g), torch.randn(32, 160, 16, 16, device="cuda", generator=g))
           for _ in range(2)]
    refs = [rc.iaf_step(z, c) for z, c in ins]
    torch.cuda.synchronize()
    outs = [[], []]
    for rep in range(20):
        for i, st in enumerate((s1, s2)):
            with torch.cuda.stream(st):
                outs[i].append(xs.iaf_step(*ins[i]))
    torch.cuda.synchronize()
    for i in range(2):
        for zx, sx in outs[i]:
            _close(sx, refs[i][1], "stream %d logsd" % i)
            _close(zx, refs[i][0], "stream %d z" % i)
    assert xs.exchange_errors() == 0


@pytest.mark.parametrize("launches", [1, 3], ids=lambda n: "graph_of_%d" % n)
def test_replayed_graphs(amd, launches):
    """a captured graph freezes the kernel arguments, so nothing per launch may come from the host: the launch epoch is counted
    on the device.  Graphs of an odd and of an even number of launches, replayed with fresh inputs."""
    xs, rc = _stacks(amd, 32, [160, 160], 15)
    g = torch.Generator(device="cuda").manual_seed(21)
    z = torch.randn(32, 32, 16, 16, device="cuda", generator=g)
    ctx = torch.randn(32, 160, 16, 16, device="cuda", generator=g)
    xs.iaf_step(z, ctx)                                         # (warm-up: the exchange set is allocated outside the capture)
    torch.cuda.synchronize()
    outs = [(torch.empty_like(z), torch.empty_like(z)) for _ in range(launches)]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cur = z
        for o in outs:
            xs.iaf_step(cur, ctx, out=o)
            cur = o[0]
    for rep in range(5):
        z.copy_(torch.randn(32, 32, 16, 16, device="cuda", generator=g))
        ctx.copy_(torch.randn(32, 160, 16, 16, device="cuda", generator=g))
        graph.replay()
        cur = z
        for o in outs:
            zr, sr = rc.iaf_step(cur, ctx)
            _close(o[1], sr, "logsd")
            _close(o[0], zr, "z")
            cur = zr
    assert xs.exchange_errors() == 0


def test_a_wait_that_gives_up_is_loud_and_the_stack_recovers(amd):
    xs, rc = _stacks(amd, 32, [160, 160], 16)
    g = torch.Generator(device="cuda").manual_seed(33)
    B = 8
    z = torch.randn(B, 32, 16, 16, device="cuda", generator=g)
    ctx = torch.randn(B, 160, 16, 16, device="cuda", generator=g)
    zr, sr = rc.iaf_step(z, ctx)
    zx, sx = xs.iaf_step(z, ctx)
    _close(zx, zr, "before the fault")
    xs.set_halo_exchange_debug(8)                                # image 0's bottom block never publishes its first hidden row
    zf, sf = xs.iaf_step(z, ctx)
    torch.cuda.synchronize()
    # the block above it (rows 12, 13 of image 0) waited, gave up and says so in its numbers.  The blocks further up in that image
    # may have given up as well (their rows come later the longer the chain below them waits, and every wait has the same bound)
    # or not (a block's exported first row never depends on an imported one at two rows per block); the faulty block itself
    # imports nothing, and no other image is touched
    assert torch.isnan(zf[0, :, 12:14]).all() and torch.isnan(sf[0, :, 12:14]).all()
    assert torch.isfinite(zf[1:]).all() and torch.isfinite(zf[0, :, 14:]).all()
    up, ref = zf[0, :, :12], zr[0, :, :12]                       # every number up there is NaN or right (waves give up one by one)
    assert bool((torch.isnan(up) | ((up - ref).abs() <= 1e-5)).all())
    _close(zf[1:], zr[1:], "the other images of the faulty launch")
    assert xs.exchange_errors() != 0
    xs.set_halo_exchange_debug(0)
    with pytest.raises(amd.ExchangeError):                       # said once, as the next call's status ...
        xs.iaf_step(z, ctx)
    assert not xs.step_exchanges(B, 16, 16) and xs.step_is_fused(B, 16, 16) == 2
    z2, s2 = xs.iaf_step(z, ctx)                                 # ... and the stack carries on, recomputing its halo rows
    assert torch.equal(z2, zr) and torch.equal(s2, sr)
    assert xs.exchange_errors() != 0                             # (sticky until re-armed)
    xs.set_halo_exchange(True)
    assert xs.step_exchanges(B, 16, 16) and xs.exchange_errors() == 0
    z3, s3 = xs.iaf_step(z, ctx)
    _close(z3, zr, "re-armed")
    _close(s3, sr, "re-armed")
    assert xs.exchange_errors() == 0


def test_launches_queued_behind_a_give_up_do_not_wait(amd):
    """the sticky word: a launch that was already queued when an earlier one gave up (a graph replay, an eager launch the host
    issued before the error word reached it) imports NaN at once instead of trusting the buffers"""
    xs, rc = _stacks(amd, 32, [160, 160], 17)
    g = torch.Generator(device="cuda").manual_seed(34)
    z = torch.randn(4, 32, 16, 16, device="cuda", generator=g)
    ctx = torch.randn(4, 160, 16, 16, device="cuda", generator=g)
    s1 = torch.cuda.Stream()
    s1.wait_stream(torch.cuda.current_stream())
    out = (torch.empty_like(z), torch.empty_like(z))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s1):
        xs.iaf_step(z, ctx)                                      # (warm-up on the stream of the capture: the graph and the
        torch.cuda.synchronize()                                 #  eager launches below share s1's exchange set)
        with torch.cuda.graph(graph, stream=s1):
            xs.iaf_step(z, ctx, out=out)
        xs.set_halo_exchange_debug(8)
        xs.iaf_step(z, ctx)                                      # gives up
        graph.replay()                                           # captured without the fault knob
        torch.cuda.synchronize()
        assert xs.exchange_errors() != 0
        # every image has row blocks that import: NaN in all of them; the bottom blocks import nothing
        assert all(bool(torch.isnan(out[0][b]).any()) for b in range(4))
        xs.set_halo_exchange_debug(0)
        xs.set_halo_exchange(True)
        graph.replay()
        torch.cuda.synchronize()
    zr, sr = rc.iaf_step(z, ctx)
    _close(out[0], zr, "replay after re-arming")
    assert xs.exchange_errors() == 0


def test_a_capture_takes_over_the_set_it_was_not_warmed_up_on(amd):
    """a capture on a stream without an exchange set of its own (torch.cuda.graph's side stream after a warm-up on the current
    stream) takes the warmed-up set over, counters included: later eager launches on the warm-up stream get a fresh set, so the
    graph's replays and those launches share nothing -- run concurrently both stay right, and a give-up of an eager launch (its
    set is dead from then on) does not reach the graph's replays (include/iaf_hip.h, State)"""
    xs, rc = _stacks(amd, 32, [160, 160], 19)
    g = torch.Generator(device="cuda").manual_seed(35)
    z = torch.randn(32, 32, 16, 16, device="cuda", generator=g)
    ctx = torch.randn(32, 160, 16, 16, device="cuda", generator=g)
    z2 = torch.randn(32, 32, 16, 16, device="cuda", generator=g)
    xs.iaf_step(z, ctx)                                          # warm-up on the current stream
    torch.cuda.synchronize()
    out = (torch.empty_like(z), torch.empty_like(z))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):                                # (captures on a side stream of torch's own)
        xs.iaf_step(z, ctx, out=out)
    zr, sr = rc.iaf_step(z, ctx)
    zr2, sr2 = rc.iaf_step(z2, ctx)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    eager = []
    for rep in range(30):                                        # replays on one stream, eager launches on the other, overlapping
        with torch.cuda.stream(side):
            graph.replay()
        eager.append(xs.iaf_step(z2, ctx))
    torch.cuda.synchronize()
    _close(out[0], zr, "replayed z")
    _close(out[1], sr, "replayed logsd")
    for ze, se in eager:
        _close(ze, zr2, "eager z")
        _close(se, sr2, "eager logsd")
    assert xs.exchange_errors() == 0
    xs.set_halo_exchange_debug(8)
    xs.iaf_step(z2, ctx)                                         # gives up: the eager stream's set is dead
    torch.cuda.synchronize()
    assert xs.exchange_errors() != 0
    out[0].zero_()
    graph.replay()                                               # the graph's set is not
    torch.cuda.synchronize()
    _close(out[0], zr, "replay beside a dead set")
    xs.set_halo_exchange_debug(0)
    xs.set_halo_exchange(True)
    assert xs.exchange_errors() == 0
