"""The gradient exchange behind the C ABI (include/iaf_hip.h: iaf_comm_*, csrc/iaf_comm.cpp): RCCL all-reduce(sum) in place
on segments of a device buffer -- what tf_utils/common.py:83-86 (average_grads) does with per-variable add_n over the
towers of tf_train.py:124-147; the 1/N rides in iaf_adamax_ema_step.  VERDICT r02 item 8: the collective of the training
loop no longer goes through torch.distributed on the GPU."""
import ctypes
import os
import tempfile

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def capi():
    import iaf_amd._capi as c
    c.lib()
    return c


def test_comm_argument_validation(capi):
    """no GPU needed: bad arguments come back as IAF_ERR_*, never abort"""
    lib = capi.lib()
    h = ctypes.c_void_p()
    idb = ctypes.create_string_buffer(capi.IAF_COMM_ID_BYTES)
    assert lib.iaf_comm_unique_id(None) == capi.IAF_ERR_NULL
    assert lib.iaf_comm_create(None, idb, 0, 1, 0) == capi.IAF_ERR_NULL
    assert lib.iaf_comm_create(ctypes.byref(h), None, 0, 1, 0) == capi.IAF_ERR_NULL
    assert lib.iaf_comm_create(ctypes.byref(h), idb, 0, 0, 0) == capi.IAF_ERR_SHAPE
    assert lib.iaf_comm_create(ctypes.byref(h), idb, 2, 2, 0) == capi.IAF_ERR_SHAPE
    assert lib.iaf_comm_create(ctypes.byref(h), idb, 0, 1, -1) == capi.IAF_ERR_SHAPE
    assert lib.iaf_allreduce_sum_f32(None, None, 4, None) == capi.IAF_ERR_NULL
    assert lib.iaf_comm_destroy(None) == capi.IAF_ERR_NULL
    assert lib.iaf_comm_size(None, None, None) == capi.IAF_ERR_NULL
    assert b"rccl" in lib.iaf_comm_library().lower()          # an RCCL could be bound (the process's own, else /opt/rocm's)


@pytest.mark.gpu
def test_one_rank_communicator_executes_the_allreduce(capi):
    """the only exchange a one-GPU box can run: a 1-rank communicator; sum over one rank = identity, executed by RCCL on
    the caller's stream (bit-exact), also on an unaligned segment of a larger buffer (the buckets of parallel.py)"""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    lib = capi.lib()
    idb = ctypes.create_string_buffer(capi.IAF_COMM_ID_BYTES)
    capi.check(lib.iaf_comm_unique_id(idb))
    assert any(idb.raw)
    h = ctypes.c_void_p()
    capi.check(lib.iaf_comm_create(ctypes.byref(h), idb, 0, 1, torch.cuda.current_device()))
    r, w = ctypes.c_int(-1), ctypes.c_int(-1)
    capi.check(lib.iaf_comm_size(h, ctypes.byref(r), ctypes.byref(w)))
    assert (r.value, w.value) == (0, 1)
    x = torch.randn(1 << 20, device="cuda")
    ref = x.clone()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    capi.check(lib.iaf_allreduce_sum_f32(h, ctypes.c_void_p(x.data_ptr()), x.numel(), ctypes.c_void_p(st.cuda_stream)))
    seg = x[12345:12345 + 77777]
    capi.check(lib.iaf_allreduce_sum_f32(h, ctypes.c_void_p(seg.data_ptr()), seg.numel(), ctypes.c_void_p(st.cuda_stream)))
    capi.check(lib.iaf_allreduce_sum_f32(h, ctypes.c_void_p(x.data_ptr()), 0, ctypes.c_void_p(st.cuda_stream)))   # n = 0: a no-op
    st.synchronize()
    assert torch.equal(x, ref)
    capi.check(lib.iaf_comm_destroy(h))


@pytest.mark.gpu
def test_overlapped_reduce_runs_through_the_c_abi_on_device_buffers():
    """OverlappedGradReduce on CUDA buffers: RcclComm (iaf_allreduce_sum_f32 on its exchange stream), not torch.distributed;
    world size 1 forced, so the result is the identity and the Adamax step behind wait() sees the reduced gradients"""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    import torch.distributed as dist
    import iaf_amd.parallel as par
    own = not dist.is_initialized()
    if own:
        f = tempfile.NamedTemporaryFile(delete=False)
        f.close()
        dist.init_process_group("gloo", init_method="file://" + f.name, rank=0, world_size=1)
    try:
        named = {"a": torch.randn(1000, device="cuda"), "b": torch.randn(64, 64, device="cuda"), "c": torch.randn(7, device="cuda")}
        fp = par.FlatParams(named)
        fp.grads.copy_(torch.randn_like(fp.grads))
        g0 = fp.grads.clone()
        red = par.OverlappedGradReduce(fp, par.OverlappedGradReduce.bounds_from_groups(fp, [["a"], ["b", "c"]]), force=True)
        assert red.active and red.comm is not None and red.comm.size() == (0, 1)
        assert "rccl" in red.comm.library.lower()
        for i in range(2):
            red.reduce(i)
        red.wait()
        torch.cuda.synchronize()
        assert torch.equal(fp.grads, g0)
        p0 = fp.params.clone()
        fp.adamax_ema_step(1e-3, world=1)
        torch.cuda.synchronize()
        assert not torch.equal(fp.params, p0)
        red.comm.close()
    finally:
        if own:
            dist.destroy_process_group()
            if os.path.exists(f.name):
                os.unlink(f.name)


def _two_rank_worker(rank, idfile, out):
    import ctypes as ct
    import time
    import iaf_amd._capi as c
    lib = c.lib()
    torch.cuda.set_device(rank)
    idb = ct.create_string_buffer(c.IAF_COMM_ID_BYTES)
    if rank == 0:                       # the id travels through a FILE: no torch.distributed anywhere in this exchange
        c.check(lib.iaf_comm_unique_id(idb))
        with open(idfile + ".tmp", "wb") as fh:
            fh.write(idb.raw)
        os.rename(idfile + ".tmp", idfile)
    else:
        while not os.path.exists(idfile):
            time.sleep(0.05)
        idb = ct.create_string_buffer(open(idfile, "rb").read(), c.IAF_COMM_ID_BYTES)
    h = ct.c_void_p()
    c.check(lib.iaf_comm_create(ct.byref(h), idb, rank, 2, rank))
    x = torch.full((4096,), float(rank + 1), device="cuda") + torch.arange(4096, device="cuda")
    c.check(lib.iaf_allreduce_sum_f32(h, ct.c_void_p(x.data_ptr()), x.numel(), ct.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    np.save(out % rank, x.cpu().numpy())
    c.check(lib.iaf_comm_destroy(h))


@pytest.mark.gpu
def test_two_ranks_exchange_without_torch_distributed(tmp_path):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (gpurun boxes expose one; the driver's scaling run covers N > 1)")
    import torch.multiprocessing as mp
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    idfile, out = str(tmp_path / "id.bin"), str(tmp_path / "r%d.npy")
    mp.spawn(_two_rank_worker, args=(idfile, out), nprocs=2, join=True)
    want = 3.0 + 2.0 * np.arange(4096)
    for r in range(2):
        np.testing.assert_array_equal(np.load(out % r), want)
