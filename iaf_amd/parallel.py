"""Data-parallel structure of the reference (tf_train.py:124-147, tf_utils/common.py:78-115), one process per GPU.

The reference replicates towers inside one TF graph, keeps the variables on the CPU and sums gradients with
plain `+` ops.  Here every rank owns a full parameter replica in HBM; the only exchange per training step is one
all-reduce(sum) over a flat fp32 gradient bucket followed by 1/N (== average_dense, common.py:83-86), done by
RCCL over xGMI (torch.distributed backend "nccl") or gloo on CPU.  The forward IAF path needs no collective:
every sample is independent and the free-bits mean is over the LOCAL tower batch (tf_train.py:79)."""
import math

import torch
import torch.distributed as dist


class RcclComm(object):
    """The gradient exchange through the engine's C ABI (include/iaf_hip.h: iaf_comm_*, csrc/iaf_comm.cpp): an RCCL
    communicator over this process group's ranks, one GPU per rank, and all-reduce(sum) in place on segments of a device
    buffer (tf_utils/common.py:83-86; the 1/N rides in the fused Adamax launch).  torch.distributed is used ONCE, to carry
    the 128-byte communicator id from rank 0 to the others; no collective of the training loop goes through it."""

    def __init__(self, group=None, device=None):
        import ctypes
        from . import _capi
        self._capi, self._ct = _capi, ctypes
        lib = _capi.lib()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.cuda.current_device() if device is None else int(device)
        idbuf = ctypes.create_string_buffer(_capi.IAF_COMM_ID_BYTES)
        if self.rank == 0:
            _capi.check(lib.iaf_comm_unique_id(idbuf))
        if self.world > 1:
            box = [bytes(idbuf.raw)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            idbuf = ctypes.create_string_buffer(box[0], _capi.IAF_COMM_ID_BYTES)
        h = ctypes.c_void_p()
        _capi.check(lib.iaf_comm_create(ctypes.byref(h), idbuf, self.rank, self.world, self.device))
        self._h = h
        self.library = lib.iaf_comm_library().decode()

    def size(self):
        r, w = self._ct.c_int(), self._ct.c_int()
        self._capi.check(self._capi.lib().iaf_comm_size(self._h, self._ct.byref(r), self._ct.byref(w)))
        return r.value, w.value

    def all_reduce_sum_(self, t, stream=None):
        """in place on the contiguous fp32 device tensor `t`, asynchronously on `stream` (default: the current one)"""
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("all_reduce_sum_ takes a contiguous fp32 device tensor")
        st = torch.cuda.current_stream() if stream is None else stream
        self._capi.check(self._capi.lib().iaf_allreduce_sum_f32(self._h, self._ct.c_void_p(t.data_ptr()), t.numel(),
                                                                self._ct.c_void_p(st.cuda_stream)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._capi.lib().iaf_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FlatParams(object):
    """All trainable tensors of a model as views into ONE flat fp32 buffer, with matching flat gradient, Adamax slot and
    EMA buffers.  The engine's backward writes gradients straight into the views, so a training step needs exactly one
    all-reduce (the reference sums 488 tensors one by one) and one fused optimiser launch."""

    def __init__(self, named_tensors, device=None):
        items = list(named_tensors.items())
        device = device or items[0][1].device
        total = sum(((t.numel() + 3) // 4) * 4 for _, t in items)     # keep every view 16-byte aligned
        self.params = torch.zeros(total, dtype=torch.float32, device=device)     # (zeros: the alignment gaps between views stay defined)
        self.grads = torch.zeros_like(self.params)
        self.slot_m = torch.zeros_like(self.params)
        self.slot_v = torch.zeros_like(self.params)
        self.ema = torch.empty_like(self.params)
        self.p, self.g = {}, {}
        off = 0
        for k, t in items:
            n = t.numel()
            self.p[k] = self.params[off:off + n].view(t.shape)
            self.p[k].copy_(t)
            self.g[k] = self.grads[off:off + n].view(t.shape)
            off += ((n + 3) // 4) * 4
        self.ema.copy_(self.params)

    def all_reduce_grads(self, group=None, async_op=False, comm=None):
        """sum over ranks (the 1/N is folded into the optimiser kernel).  comm: an RcclComm -- the exchange then runs through
        the engine's C ABI on the current stream (stream-ordered: nothing to wait for on the host)."""
        if comm is not None:
            comm.all_reduce_sum_(self.grads)
            return None
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            return dist.all_reduce(self.grads, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return None

    def offset_of(self, name):
        """(first element, one past the last element) of parameter `name` inside the flat buffers"""
        v = self.p[name]
        lo = (v.data_ptr() - self.params.data_ptr()) // 4
        return lo, lo + v.numel()

    def adamax_ema_step(self, lr, world=1, beta1=0.9, beta2=0.999, eps=1e-8, ema_decay=0.999):
        if self.params.is_cuda:
            import ctypes
            from . import _capi
            ptr = lambda t: ctypes.c_void_p(t.data_ptr())
            _capi.check(_capi.lib().iaf_adamax_ema_step(ptr(self.params), ptr(self.grads), ptr(self.slot_m), ptr(self.slot_v),
                                                        ptr(self.ema), self.params.numel(), lr, beta1, beta2, eps, ema_decay,
                                                        1.0 / world,
                                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
            # the launch wrote through raw pointers: tell torch, so that (data_ptr, _version)-keyed caches downstream
            # (ARStack.prepare, WNConv2d.prepare, IAFLayer.load) see new weights.  Views share the base's counter.
            torch.autograd.graph.increment_version((self.params, self.ema, self.slot_m, self.slot_v))
        else:        # host replicas (gloo tests): the same arithmetic with torch ops
            g = self.grads / float(world)
            adamax_step_(self.params, g, self.slot_m, self.slot_v, lr, beta1, beta2, eps)
            ema_step_(self.ema, self.params, ema_decay)


class OverlappedGradReduce(object):
    """The gradient exchange of a training step (tf_utils/common.py:83-86: sum over towers, then 1/N) as a few large
    all-reduces that OVERLAP the backward pass.  The flat gradient buffer is laid out in the order backward finishes
    the gradients and cut into contiguous buckets [lo, hi); `reduce(i)` is called right after the kernels that
    complete bucket i were enqueued: torch.distributed makes the collective's stream wait for exactly that point of
    the compute stream, so bucket i travels over xGMI while the backward of the remaining layers runs, and `wait()`
    joins the streams before the optimiser.  xGMI rings are per-link bound, so the buckets stay large (tens of MB);
    the 1/N is folded into the fused Adamax kernel.  `force` runs the collective even at world size 1 (single-GPU
    boxes: the RCCL path still executes).

    Device buffers travel through the engine's C ABI (RcclComm: iaf_allreduce_sum_f32 = ncclAllReduce on a dedicated
    exchange stream that waits for the point of the compute stream where the bucket was completed); host buffers (the gloo
    tests of the DP arithmetic) through torch.distributed."""

    def __init__(self, flat, bounds, group=None, force=False, comm=None):
        self.flat, self.group = flat, group
        self.bounds = [(int(a), int(b)) for a, b in bounds]
        n = flat.grads.numel()
        assert self.bounds and self.bounds[0][0] == 0 and self.bounds[-1][1] == n, "buckets must tile the flat buffer"
        assert all(a[1] == b[0] for a, b in zip(self.bounds[:-1], self.bounds[1:])), "buckets must be contiguous"
        self.active = dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.works = []
        self.comm, self.xstream, self._pending = None, None, False
        if self.active and flat.grads.is_cuda:
            self.comm = comm if comm is not None else RcclComm(group)
            self.xstream = torch.cuda.Stream()

    @staticmethod
    def bounds_from_groups(flat, groups):
        """groups: list of lists of parameter names, in completion order and in the flat buffer's order.  Raises
        ValueError if a group is not one contiguous run of the flat buffer starting where the previous group ended, or if
        a parameter is missing / listed twice: a bucket [lo, hi) would then hold gradients of layers whose backward has
        not run when reduce(i) sends it (silently stale at world size > 1)."""
        out, lo, seen = [], 0, set()
        for i, names in enumerate(groups):
            spans = sorted(flat.offset_of(k) for k in names)
            if not spans:
                raise ValueError("bucket %d is empty" % i)
            dup = [k for k in names if k in seen]
            if dup or len(set(names)) != len(names):
                raise ValueError("bucket %d lists parameters twice: %s" % (i, dup or names))
            seen.update(names)
            first = spans[0][0]
            if first - lo < 0 or first - lo >= 4:          # (a segment may start on the next 16-byte boundary)
                raise ValueError("bucket %d starts at flat offset %d, the previous bucket ended at %d: the groups are not "
                                 "in the flat buffer's order" % (i, first, lo))
            for (a0, a1), (b0, b1) in zip(spans[:-1], spans[1:]):
                if b0 < a1 or b0 - a1 >= 4:
                    raise ValueError("bucket %d is not one contiguous run of the flat buffer (gap or overlap at %d..%d)" % (i, a1, b0))
            hi = spans[-1][1]
            hi = flat.grads.numel() if i == len(groups) - 1 else ((hi + 3) // 4) * 4
            out.append((lo, hi))
            lo = hi
        if set(flat.p) != seen:
            raise ValueError("parameters not covered by any bucket: %s" % sorted(set(flat.p) - seen))
        return out

    def reduce(self, i):
        if not self.active:
            return
        lo, hi = self.bounds[i]
        if self.comm is not None:
            self.xstream.wait_stream(torch.cuda.current_stream())       # the bucket's gradients are complete at this point
            self.comm.all_reduce_sum_(self.flat.grads[lo:hi], stream=self.xstream)
            self._pending = True
        else:
            self.works.append(dist.all_reduce(self.flat.grads[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        if self._pending:
            torch.cuda.current_stream().wait_stream(self.xstream)       # the optimiser runs behind the last bucket
            self._pending = False
        for w in self.works:
            w.wait()
        self.works = []


def shard_batch(x, rank=None, world=None):
    """tf.split(0, num_gpus, x)[rank] (tf_train.py:126): equal contiguous shards of the global batch."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    n = x.shape[0]
    if n % world != 0:
        raise ValueError("global batch %d is not divisible by %d ranks" % (n, world))   # tf.split would raise too
    per = n // world
    return x[rank * per:(rank + 1) * per]


class GradBucket(object):
    """Flat fp32 bucket over a list of gradient tensors.  The reference averages 488 tensors one by one
    (SURVEY 2.2); over xGMI the ring is per-link bound, so one large message (166 MB at the README config)
    beats 488 small ones.  Buckets are filled in REVERSE parameter order so that a caller overlapping with
    backward can launch the bucket holding the top-down layers first."""

    def __init__(self, tensors, bucket_bytes=64 << 20):
        self.tensors = list(tensors)
        self.buckets = []
        cur, cur_bytes = [], 0
        for t in reversed(self.tensors):
            nb = t.numel() * 4
            if cur and cur_bytes + nb > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(t)
            cur_bytes += nb
        if cur:
            self.buckets.append(cur)

    def all_reduce_mean(self, group=None, async_op=False):
        """grad = (sum over ranks) / N for every tensor, in place (common.py:83-86)."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        handles = []
        for b in self.buckets:
            flat = torch.cat([t.reshape(-1) for t in b])
            h = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True) if world > 1 else None
            handles.append((h, flat, b))
        if async_op:
            return lambda: self._finish(handles, world)
        self._finish(handles, world)

    @staticmethod
    def _finish(handles, world):
        for h, flat, b in handles:
            if h is not None:
                h.wait()
            flat /= float(world)
            off = 0
            for t in b:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n


def average_grads(grads, group=None):
    """Drop-in for tf_utils/common.py:78-115 (dense branch) across RANKS instead of in-graph towers:
    every rank passes its own list of gradient tensors; on return each holds sum/N."""
    GradBucket(grads).all_reduce_mean(group)
    return grads


def bits_per_dim(local_loss_sum, batch_size_per_rank, num_pixels=3 * 32 * 32, group=None):
    """tf_train.py:142: add_n(losses) / (ln2 * num_pixels * batch_size * num_gpus)."""
    t = local_loss_sum.detach().clone().reshape(1).to(torch.float64)
    world = 1
    if dist.is_initialized():
        world = dist.get_world_size(group)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.item() / (math.log(2.0) * num_pixels * batch_size_per_rank * world)


def adamax_step_(var, grad, slot_m, slot_v, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf_utils/adamax.py:40-56 in place (NB slot "v" = first moment, "m" = infinity norm).  Identical on
    every replica after average_grads, so parameters stay in sync without a broadcast."""
    slot_v.mul_(beta1).add_(grad, alpha=1.0 - beta1)
    torch.maximum(slot_m.mul_(beta2).add_(eps), grad.abs(), out=slot_m)
    var.addcdiv_(slot_v, slot_m, value=-lr)
    return var


def ema_step_(shadow, var, decay=0.999):
    """tf.train.ExponentialMovingAverage(0.999).apply (tf_train.py:157-158)."""
    shadow.sub_(shadow - var, alpha=1.0 - decay)
    return shadow
