"""world_size-2 gloo tests (CPU) of the data-parallel layer: batch sharding, bucketed gradient averaging
(== the reference's average_grads on towers, tf_utils/common.py:78-115), bits_per_dim reduction, and replica
consistency of Adamax after averaged gradients."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import iaf_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from iaf_amd import parallel as par
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "common.npz"))
    # every rank plays one of the reference's towers (towers 0 and 1 of the golden fixture)
    grads = [torch.from_numpy(g["tower%d_g%d" % (rank, i)].copy()).float() for i in range(3)]
    par.GradBucket(grads, bucket_bytes=512).all_reduce_mean()          # small bucket size -> several buckets
    x = torch.arange(8 * 3, dtype=torch.float32).reshape(8, 3)
    shard = par.shard_batch(x)
    bpd = par.bits_per_dim(torch.tensor(100.0 * (rank + 1)), batch_size_per_rank=4)
    # one Adamax + EMA step from identical state with the averaged gradients: replicas must stay bit-identical
    var = torch.ones_like(grads[0])
    m, v = torch.zeros_like(var), torch.zeros_like(var)
    par.adamax_step_(var, grads[0], m, v, lr=0.002)
    shadow = par.ema_step_(torch.zeros_like(var), var)
    out[rank] = dict(grads=[t.numpy() for t in grads], shard=shard.numpy(), bpd=bpd, var=var.numpy(), m=m.numpy(),
                     v=v.numpy(), shadow=shadow.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_dp_two_ranks_gloo(golden_dir):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    g = np.load(os.path.join(golden_dir, "common.npz"))
    towers = [[g["tower%d_g%d" % (t, i)].astype(np.float32).astype(np.float64) for i in range(3)] for t in range(2)]
    ref = O.average_grads(towers)
    for r in range(world):
        for i in range(3):
            np.testing.assert_allclose(out[r]["grads"][i], ref[i], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(out[0]["shard"], np.arange(24, dtype=np.float32).reshape(8, 3)[:4])
    np.testing.assert_array_equal(out[1]["shard"], np.arange(24, dtype=np.float32).reshape(8, 3)[4:])
    expect_bpd = 300.0 / (np.log(2.0) * 3072 * 4 * 2)                     # tf_train.py:142
    assert abs(out[0]["bpd"] - expect_bpd) < 1e-12 and abs(out[1]["bpd"] - expect_bpd) < 1e-12
    for k in ("var", "m", "v", "shadow"):
        np.testing.assert_array_equal(out[0][k], out[1][k])               # replicas stay in sync
    ev, em, evv = O.adamax_step(np.ones_like(ref[0]), ref[0], np.zeros_like(ref[0]), np.zeros_like(ref[0]), lr=0.002)
    np.testing.assert_allclose(out[0]["var"], ev, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out[0]["m"], em, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out[0]["v"], evv, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out[0]["shadow"], O.ema_step(np.zeros_like(ev), ev), rtol=1e-5, atol=1e-7)


def _flat_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from iaf_amd import parallel as par
    rng = np.random.RandomState(3)                      # identical initial parameters on every rank
    shapes = {"IAF_0_0/ar_multiconv2d/layer_0/V": (3, 3, 4, 8), "IAF_0_0/ar_multiconv2d/layer_0/g": (8,), "x/odd": (5,)}
    fp = par.FlatParams({k: torch.from_numpy(rng.standard_normal(s)).float() for k, s in shapes.items()})
    grng = np.random.RandomState(100 + rank)            # rank-specific gradients (each rank saw its own batch shard)
    for step in range(3):
        for k, s in shapes.items():
            fp.g[k].copy_(torch.from_numpy(grng.standard_normal(s)).float())
        fp.all_reduce_grads()                           # ONE collective for the whole model
        fp.adamax_ema_step(0.01, world=world)
    out[rank] = dict(params=fp.params.numpy().copy(), ema=fp.ema.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_training_step_two_ranks_gloo():
    """the DP update path of bench.py --train on host tensors: flat gradient bucket -> all-reduce(sum) -> Adamax on grad/N
    -> EMA; replicas must end bit-identical and equal to the oracle fed with the tower-averaged gradients"""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_flat_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    np.testing.assert_array_equal(out[0]["params"], out[1]["params"])
    np.testing.assert_array_equal(out[0]["ema"], out[1]["ema"])
    rng = np.random.RandomState(3)
    shapes = {"IAF_0_0/ar_multiconv2d/layer_0/V": (3, 3, 4, 8), "IAF_0_0/ar_multiconv2d/layer_0/g": (8,), "x/odd": (5,)}
    var = {k: rng.standard_normal(s).astype(np.float32).astype(np.float64) for k, s in shapes.items()}
    m = {k: np.zeros_like(v) for k, v in var.items()}
    vv = {k: np.zeros_like(v) for k, v in var.items()}
    ema = {k: v.copy() for k, v in var.items()}
    grngs = [np.random.RandomState(100 + r) for r in range(world)]
    for step in range(3):
        for k, s in shapes.items():
            g = O.average_grads([[gr.standard_normal(s).astype(np.float32).astype(np.float64)] for gr in grngs])[0]
            var[k], m[k], vv[k] = O.adamax_step(var[k], g, m[k], vv[k], 0.01)
            ema[k] = O.ema_step(ema[k], var[k])
    off = 0
    for k, s in shapes.items():
        n = int(np.prod(s))
        np.testing.assert_allclose(out[0]["params"][off:off + n].reshape(s), var[k], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(out[0]["ema"][off:off + n].reshape(s), ema[k], rtol=1e-5, atol=1e-6)
        off += ((n + 3) // 4) * 4


def test_average_grads_against_reference_golden_four_towers(golden_dir):
    """single process, the reference's 4-tower fixture: sum then /N"""
    g = np.load(os.path.join(golden_dir, "common.npz"))
    towers = [[g["tower%d_g%d" % (t, i)] for i in range(3)] for t in range(4)]
    avg = O.average_grads(towers)
    for i in range(3):
        np.testing.assert_allclose(avg[i], g["avg_g%d" % i], rtol=1e-13, atol=1e-15)


def _overlap_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from iaf_amd import parallel as par
    rng = np.random.RandomState(3)
    # completion order of a two-layer model: the down-pass parameters of layers 0, 1, then the up-pass parameters of 1, 0
    names = ["IAF_0_0/down_conv1/V", "IAF_0_0/ar_multiconv2d/layer_0/g", "IAF_0_1/down_conv1/V", "IAF_0_1/odd",
             "IAF_0_1/up_conv1/V", "IAF_0_0/up_conv1/V"]
    shapes = [(3, 3, 4, 6), (7,), (3, 3, 4, 6), (5,), (3, 3, 2, 4), (3, 3, 2, 4)]
    fp = par.FlatParams({k: torch.from_numpy(rng.standard_normal(s)).float() for k, s in zip(names, shapes)})
    groups = [names[0:2], names[2:4], names[4:6]]
    bounds = par.OverlappedGradReduce.bounds_from_groups(fp, groups)
    red = par.OverlappedGradReduce(fp, bounds)
    grng = np.random.RandomState(100 + rank)
    for bi, grp in enumerate(groups):                  # "backward" fills bucket bi, then its all-reduce is issued
        for k in grp:
            fp.g[k].copy_(torch.from_numpy(grng.standard_normal(fp.g[k].shape)).float())
        red.reduce(bi)
    red.wait()
    fp.adamax_ema_step(0.01, world=world)
    out[rank] = dict(grads=fp.grads.numpy().copy(), params=fp.params.numpy().copy(), bounds=bounds,
                     offs=[fp.offset_of(k) for k in names])
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_bucketed_grad_reduce_two_ranks_gloo():
    """the bucketed all-reduce issued bucket by bucket as backward completes them == one all-reduce(sum) of the
    whole gradient buffer (tf_utils/common.py:83-86 before the 1/N), on every rank"""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_overlap_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    np.testing.assert_array_equal(out[0]["grads"], out[1]["grads"])
    np.testing.assert_array_equal(out[0]["params"], out[1]["params"])
    bounds = out[0]["bounds"]
    assert bounds[0][0] == 0 and all(a[1] == b[0] for a, b in zip(bounds[:-1], bounds[1:])) and all(b % 4 == 0 for _, b in bounds)
    shapes = [(3, 3, 4, 6), (7,), (3, 3, 4, 6), (5,), (3, 3, 2, 4), (3, 3, 2, 4)]
    grngs = [np.random.RandomState(100 + r) for r in range(world)]
    for (lo, hi), s in zip(out[0]["offs"], shapes):
        tot = sum(gr.standard_normal(s).astype(np.float32) for gr in grngs)
        np.testing.assert_allclose(out[0]["grads"][lo:hi].reshape(s), tot, rtol=1e-6, atol=1e-6)


def test_bucket_bounds_reject_misordered_or_incomplete_groups():
    """bounds_from_groups trusted its docstring contract; a misordered / non-contiguous group would put
    gradients of layers whose backward has not run yet into a bucket that reduce(i) already sends."""
    import pytest
    import iaf_amd.parallel as par
    named = {"a": torch.zeros(5), "b": torch.zeros(8), "c": torch.zeros(3), "d": torch.zeros(16)}
    fp = par.FlatParams(named, device="cpu")
    ok = par.OverlappedGradReduce.bounds_from_groups(fp, [["a", "b"], ["c"], ["d"]])
    assert ok[0][0] == 0 and ok[-1][1] == fp.grads.numel() and all(x[1] == y[0] for x, y in zip(ok[:-1], ok[1:]))
    assert ok[0][1] == fp.offset_of("c")[0] and ok[1][1] == fp.offset_of("d")[0]
    for bad in ([["b", "a"], ["d"], ["c"]],        # completion order differs from the flat order
                [["a", "c"], ["b"], ["d"]],        # a group that is not one contiguous run
                [["a", "b"], ["d"]],               # a parameter no bucket covers
                [["a", "b"], ["b", "c"], ["d"]],   # listed twice
                [["a", "b"], [], ["c", "d"]]):     # empty bucket
        with pytest.raises(ValueError):
            par.OverlappedGradReduce.bounds_from_groups(fp, bad)


def _model_worker(rank, world, port, out, nb):
    """One tower of the reference's data-parallel step on the WHOLE model (tf_train.py:124-147): this rank's images through the model's
    objective and gradient (the fp64 autograd oracle stands in for CVAE1.forward_backward, which needs the GPU), the gradients written
    into a flat buffer laid out by CVAE1.grad_bucket_names, every bucket all-reduced as the backward would complete it, then the fused
    Adamax(1/N) + EMA arithmetic."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import golden_inputs as gi
    from iaf_amd import parallel as par
    from iaf_amd.model import CVAE1
    from oracle import iaf_grad_oracle as G
    c = gi.model_case_inputs("model_tiny")
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    p32 = {k: f32(v) for k, v in c["params"].items()}
    buckets = CVAE1.grad_bucket_names(list(p32), c["depth"], c["num_blocks"], nb)
    order = [k for b in buckets for k in b]
    fp = par.FlatParams({k: torch.from_numpy(np.asarray(c["params"][k], np.float32)) for k in order})
    red = par.OverlappedGradReduce(fp, par.OverlappedGradReduce.bounds_from_groups(fp, buckets))
    per = c["x"].shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)                              # tf.split(0, num_gpus, x)[rank]  (tf_train.py:126)
    grads, _, obj = G.cvae1_grads(c["x"][sl], p32, c["z_size"], c["h_size"], c["depth"], c["num_blocks"], c["kl_min"], [f32(e[sl]) for e in c["noise"]])
    for bi, names in enumerate(buckets):                                  # the backward completes bucket bi, its all-reduce is issued
        for k in names:
            fp.g[k].copy_(torch.from_numpy(np.asarray(grads[k], np.float32)).reshape(fp.g[k].shape))
        red.reduce(bi)
    red.wait()
    fp.adamax_ema_step(0.002, world=world)
    out[rank] = dict(params=fp.params.numpy().copy(), ema=fp.ema.numpy().copy(), grads={k: np.asarray(grads[k], np.float64) for k in order},
                     offs={k: fp.offset_of(k) for k in order}, obj=float(obj))
    dist.barrier()
    dist.destroy_process_group()


def test_model_level_training_step_two_towers_gloo():
    """the model-level DP step at world size 2 -- two ranks, different images, the model's gradient buckets reduced in
    completion order: identical parameters on both ranks afterwards, equal to the reference's two-tower step (average_grads over the
    towers' gradients, tf_utils/common.py:83-86, then adamax.py:40-56 and the EMA of tf_train.py:157-158) computed in one process"""
    import pytest
    world = 2
    res = {}
    for nb in (1, 4):
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_model_worker, args=(world, _free_port(), out, nb), nprocs=world, join=True)
        np.testing.assert_array_equal(out[0]["params"], out[1]["params"])     # replicas stay bit-identical without a broadcast
        np.testing.assert_array_equal(out[0]["ema"], out[1]["ema"])
        assert out[0]["obj"] != out[1]["obj"]                                 # (the towers did see different images)
        res[nb] = out[0]
        # the reference's arithmetic, one process, fp64: per-variable mean over the towers, one Adamax step from zero slots, EMA
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
        import golden_inputs as gi
        c = gi.model_case_inputs("model_tiny")
        for k, (lo, hi) in out[0]["offs"].items():
            avg = O.average_grads([[out[r]["grads"][k]] for r in range(world)])[0]
            var = np.asarray(c["params"][k], np.float32).astype(np.float64)
            new, _, _ = O.adamax_step(var.copy(), avg, np.zeros_like(var), np.zeros_like(var), 0.002)
            np.testing.assert_allclose(out[0]["params"][lo:hi].reshape(var.shape), new, rtol=2e-6, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(res[1]["params"], res[4]["params"], rtol=0, atol=0)      # bucketing does not change a bit
