"""Host-side mirror of the reference's operator interface for the hot path
(tf_utils/layers.py:115-166): same names, argument meaning and error behaviour; the arithmetic is
done by the HIP engine through the C ABI.  PyTorch tensors are device storage only.

Variable naming follows the reference's TF scopes so that a weight loader written for it works:
    <scope>/<name>/layer_{i}/{V,g,b}   and   <scope>/<name>/layer_out_{i}/{V,g,b}
with V in HWIO [3,3,n_in,n_out] fp32 (layers.py:35,53-55).
"""
import contextlib
import ctypes
import os

import numpy as np
import torch

from . import _capi


# ---------------------------------------------------------------------------------------------
# masks (host, build time).  Closed forms of the loops at layers.py:115-141.
# ---------------------------------------------------------------------------------------------
def get_linear_ar_mask(n_in, n_out, zerodiagonal=False):
    """Channel MADE mask [n_in, n_out] float32 (tf_utils/layers.py:115-131).
    n_out >= n_in (k = n_out//n_in): output o belongs to group o//k and sees inputs <= group
    (< group if zerodiagonal).  n_out < n_in (k = n_in//n_out): output o sees inputs < (o+1)*k
    (< o*k if zerodiagonal)."""
    assert n_in % n_out == 0 or n_out % n_in == 0, "%d - %d" % (n_in, n_out)
    i = np.arange(n_in).reshape(-1, 1)
    o = np.arange(n_out).reshape(1, -1)
    if n_out >= n_in:
        grp = o // (n_out // n_in)
        live = (i < grp) if zerodiagonal else (i <= grp)
    else:
        k = n_in // n_out
        live = (i < o * k) if zerodiagonal else (i < (o + 1) * k)
    return live.astype(np.float32)


def get_conv_ar_mask(h, w, n_in, n_out, zerodiagonal=False):
    """Conv MADE mask [h, w, n_in, n_out] float32 (tf_utils/layers.py:134-141): taps above the
    centre row and left of the centre in the centre row are dead, the centre tap is the channel
    mask, everything after is full."""
    l, m = (h - 1) // 2, (w - 1) // 2
    mask = np.zeros([h, w, n_in, n_out], dtype=np.float32)
    mask[l, m + 1:] = 1.0
    mask[l + 1:] = 1.0
    mask[l, m] = get_linear_ar_mask(n_in, n_out, zerodiagonal)
    return mask


# ---------------------------------------------------------------------------------------------
# variable store with TF-style scopes
# ---------------------------------------------------------------------------------------------
class VariableStore(object):
    """name -> torch tensor, addressed through nested variable_scope()s like tf.get_variable."""

    def __init__(self):
        self.vars = {}
        self._scope = []
        self._stacks = {}

    def full_name(self, name):
        return "/".join(self._scope + [name])

    def get(self, name):
        full = self.full_name(name)
        if full not in self.vars:
            raise KeyError("variable %r not in store" % full)
        return self.vars[full]

    def set(self, full_name, tensor):
        self.vars[full_name] = tensor


_DEFAULT_STORE = VariableStore()


def default_store():
    return _DEFAULT_STORE


@contextlib.contextmanager
def variable_scope(name, store=None):
    st = store or _DEFAULT_STORE
    st._scope.append(name)
    try:
        yield st
    finally:
        st._scope.pop()


# ---------------------------------------------------------------------------------------------
# the engine-backed operator
# ---------------------------------------------------------------------------------------------
def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_act(t, name, shape=None):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous float32 CUDA tensor (NCHW)" % name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))


class ARStack(object):
    """One ar_multiconv2d variable set bound to an engine handle (iaf_stack_t).

    n_h: list of hidden sizes (the reference passes [h_size, h_size], tf_train.py:69, or
    depth_ar*[n_h2], models.py:92) -- all equal; n_out must be [n_z, n_z]."""

    def __init__(self, n_z, n_h, n_out=None, variant=_capi.IAF_VARIANT_TF):
        if variant in ("tf", "theano", "theano_flipmask"):
            variant = {"tf": _capi.IAF_VARIANT_TF, "theano": _capi.IAF_VARIANT_THEANO,
                       "theano_flipmask": _capi.IAF_VARIANT_THEANO_FLIPMASK}[variant]
        self.variant = variant
        n_h = list(n_h)
        n_out = [n_z, n_z] if n_out is None else list(n_out)
        sizes = [n_z] + n_h
        for a, b in zip(sizes[:-1], sizes[1:]):
            assert a % b == 0 or b % a == 0, "%d - %d" % (a, b)          # layers.py:116
        for o in n_out:
            assert sizes[-1] % o == 0 or o % sizes[-1] == 0, "%d - %d" % (sizes[-1], o)
        if len(set(n_h)) > 1:
            raise ValueError("the gfx950 engine needs equal hidden sizes, got %r" % (n_h,))
        if n_out != [n_z, n_z]:
            raise ValueError("the gfx950 engine implements the (mean, logsd) output pair n_out=[n_z, n_z]")
        self.n_z, self.n_h_list, self.depth_ar = int(n_z), n_h, len(n_h)
        self.n_h = int(n_h[0]) if n_h else int(n_z)
        self._h = ctypes.c_void_p()
        _capi.check(_capi.lib().iaf_stack_create(ctypes.byref(self._h), self.n_z, self.n_h, self.depth_ar, variant))
        if os.environ.get("IAF_PRECISION"):          # process-wide override of the default (bf16x3): "f32" | "bf16x3"
            self.set_precision(os.environ["IAF_PRECISION"])
        self._ws = None
        self._prep_key = None
        self._keepalive = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _capi.lib().iaf_stack_destroy(h)
            except Exception:
                pass

    # -- weights -------------------------------------------------------------------------------
    def conv_names(self):
        return ["layer_%d" % i for i in range(self.depth_ar)] + ["layer_out_0", "layer_out_1"]

    def prepare(self, params, force=False):
        """params: {"layer_0/V": tensor, "layer_0/g": ..., "layer_out_1/b": ...} (device fp32).
        Re-derives masked weight-normed weights on the GPU (layers.py:56-60); cached until a tensor
        is replaced or modified in place."""
        names = self.conv_names()
        tens = self._param_tensors(params)
        key = tuple((t.data_ptr(), t._version) for t in tens)
        if not force and key == self._prep_key:
            return
        n = len(names)
        arr = ctypes.c_void_p * n
        Vp = arr(*[t.data_ptr() for t in tens[0::3]])
        gp = arr(*[t.data_ptr() for t in tens[1::3]])
        bp = arr(*[t.data_ptr() for t in tens[2::3]])
        _capi.check(_capi.lib().iaf_stack_prepare(self._h, Vp, gp, bp, _stream()))
        self._prep_key, self._keepalive = key, tens

    def _param_tensors(self, params):
        """[V, g, b] per conv in engine order.  TF variant: "<conv>/V|g|b", V HWIO [3,3,n_in,n_out] (layers.py:53-55).
        Theano variant: "<i>_w|_s|_b" and "out_<i>_w|_s|_b" relative to the multiconv2d name, w OIHW
        [n_out, n_in+1, 3, 3] (ar.py:288-296); the engine's (V, g, b) slots carry (w, s, b)."""
        sizes = [self.n_z] + self.n_h_list
        tens = []
        theano = self.variant in (_capi.IAF_VARIANT_THEANO, _capi.IAF_VARIANT_THEANO_FLIPMASK)
        names = self.conv_names()
        for ci, nm in enumerate(names):
            n_in = sizes[min(ci, self.depth_ar)]
            n_out = sizes[ci + 1] if ci < self.depth_ar else self.n_z
            if theano:
                base = ("%d" % ci) if ci < self.depth_ar else ("out_%d" % (ci - self.depth_ar))
                V, g, b = params[base + "_w"], params[base + "_s"], params[base + "_b"]
                _check_act(V, base + "_w", (n_out, n_in + 1, 3, 3))
            else:
                V, g, b = params[nm + "/V"], params[nm + "/g"], params[nm + "/b"]
                _check_act(V, nm + "/V", (3, 3, n_in, n_out))
            _check_act(g, nm + " scale", (n_out,))
            _check_act(b, nm + " bias", (n_out,))
            tens += [V, g, b]
        return tens

    def _grad_keys(self):
        """per conv, the keys its (V, g, b) slots go by in a params / grads dict (see _param_tensors)"""
        if self.variant in (_capi.IAF_VARIANT_THEANO, _capi.IAF_VARIANT_THEANO_FLIPMASK):
            bases = ["%d" % i for i in range(self.depth_ar)] + ["out_0", "out_1"]
            return [(b + "_w", b + "_s", b + "_b") for b in bases]
        return [(nm + "/V", nm + "/g", nm + "/b") for nm in self.conv_names()]

    def time_layer(self, layer, z, context, reps=50):
        """average duration (ms) of GEMM layer `layer` over `reps` back-to-back launches between one HIP event pair;
        layer = -1: the fused launch of layers 0+1 (UnsupportedError if the stack would not fuse at this size)"""
        B, H, W = self._dims(z, context)
        ws, need = self.workspace(B, H, W, z.device)
        zn, ls = torch.empty_like(z), torch.empty_like(z)
        ms = ctypes.c_float()
        _capi.check(_capi.lib().iaf_step_time_layer(self._h, layer, _ptr(z), _ptr(context), _ptr(zn), _ptr(ls), B, H, W,
                                                    _ptr(ws), need, int(reps), _stream(), ctypes.byref(ms)))
        return ms.value

    # -- workspace -----------------------------------------------------------------------------
    def workspace(self, B, H, W, device):
        need = int(_capi.lib().iaf_stack_workspace_bytes(self._h, B, H, W))
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(max(need, 256), dtype=torch.uint8, device=device)
        return self._ws, need

    def set_tuning(self, layer, nt, pxt, wco, ks):
        _capi.check(_capi.lib().iaf_stack_set_tuning(self._h, layer, nt, pxt, wco, ks))

    def set_precision(self, precision):
        """"bf16x3" (default): forward masked convs on the bf16 matrix cores as six split products, fp32-grade;
        "f32": the exact-fp32 MFMA (bit-equal to an fmaf chain); "f16x2": as bf16x3, with the one-launch step kernels of the
        BASELINE geometries on TWO fp16 planes and three products (operands up to 65504; beyond: RangeError at the next call and
        bf16x3 from then on).  See include/iaf_hip.h."""
        code = {"f32": _capi.IAF_PRECISION_F32, "bf16x3": _capi.IAF_PRECISION_BF16X3,
                "f16x2": _capi.IAF_PRECISION_F16X2}.get(precision, precision)
        _capi.check(_capi.lib().iaf_stack_set_precision(self._h, int(code)))
        self._prep_key = None                        # (the set of packs the next prepare writes may have changed)

    def layer_precision(self, layer, B, H, W):
        """what a forward launch of GEMM layer `layer` at this size will run: "bf16x3" or "f32" (layers the bf16x3 kernels
        do not cover, or small launches they do not speed up, stay on the fp32 kernel)"""
        code = _capi.lib().iaf_stack_get_precision(self._h, int(layer), int(B), int(H), int(W))
        if code < 0:
            _capi.check(code)
        return "bf16x3" if code == _capi.IAF_PRECISION_BF16X3 else "f32"

    def autotune(self, z, context, reps=20):
        """time every GEMM layer as the exact-fp32 kernel and as every compiled bf16x3 launch shape on these buffers and
        keep the fastest for this (B, H, W) -- what cuDNN's algorithm search does for the reference's convs.  Returns
        [(choice, us)] per layer, choice = "f32" or "bf16x3(nt,ppw,pxt,ks,wco)"; where the whole step runs faster as ONE
        launch (iaf_stack_set_fuse_step) every layer reads "one-launch step" and the last entry carries the step's time.
        Synchronises: call before graph capture."""
        B, H, W = self._dims(z, context)
        ws, need = self.workspace(B, H, W, z.device)
        zn, ls = torch.empty_like(z), torch.empty_like(z)
        n = self.depth_ar + 1
        chosen, us = (ctypes.c_int * n)(), (ctypes.c_float * n)()
        _capi.check(_capi.lib().iaf_stack_autotune(self._h, _ptr(z), _ptr(context), _ptr(zn), _ptr(ls), B, H, W, _ptr(ws), need,
                                                   int(reps), _stream(), chosen, us))
        out = []
        for i in range(n):
            c = chosen[i]
            name = "f32" if c == 0 else ("one-launch step" if c == -2 else "fused into next" if c < 0 else
                                         "bf16x3(%d,%d,%d,%d,%d)" % (c // 10000, c // 1000 % 10, c // 100 % 10, c // 10 % 10, c % 10))
            if c > 0 and i == 1 and chosen[0] < 0:
                name += "+layer0"
            out.append((name, us[i]))
        return out

    def set_fuse_first(self, mode):
        """first masked conv inside the second one's kernel: "never" | "always" (whenever the kernels allow) | "auto"
        (default: only where autotune measured it clearly faster).  See include/iaf_hip.h."""
        code = {"never": 0, "always": 1, "auto": 2}.get(mode, mode)
        _capi.check(_capi.lib().iaf_stack_set_fuse_first(self._h, int(code)))

    def set_fuse_step(self, mode):
        """the whole IAF step as ONE launch: "auto" (default: where a compiled geometry covers it and the size rule or
        autotune's measurement favours it) | "never" | "always" (wherever a geometry covers it).  See include/iaf_hip.h."""
        code = {"never": 0, "auto": 1, "always": 2}.get(mode, mode)
        _capi.check(_capi.lib().iaf_stack_set_fuse_step(self._h, int(code)))

    def posterior_block_launches(self, B, H, W):
        """what posterior_block launches at this size (a description for bench lines)"""
        R = self.step_is_fused(B, H, W)
        if R:
            nrb = -(-H // R)
            if self.step_exchanges(B, H, W) or self.step_pairs(B, H, W) or (W == 8 and self.n_h == 160 and self.n_z == 32 and self.depth_ar == 2):     # kernels with helper waves
                if B * nrb * self.n_z <= 16384 and B * self.n_z <= 8192:
                    return "1 launch: the one-launch IAF step, whose last workgroup also does the block's free-bits reductions"
            return ("1 one-launch IAF step (its final loop leaves per-row-block KL sums) + %d KL reduction launch(es)"
                    % (1 if B * nrb * self.n_z <= 16384 else 2))
        return "%d masked convs + 2 KL reductions" % (self.depth_ar + 1)

    def set_packs(self, f32=True, bf16x3=True, f16x2=False):
        """which weight packs the prep launches keep up to date: f32=False drops the fp32 fragment pack (a stack whose
        every launch runs on the bf16 matrix cores; a launch that would need it raises); f16x2=True with the other two False keeps
        only the two-plane fp16 pack ("f16x2" stacks whose every launch is an F16 step kernel).  See include/iaf_hip.h."""
        _capi.check(_capi.lib().iaf_stack_set_packs(self._h, (_capi.IAF_PACK_BF16X3 if bf16x3 else 0) | (_capi.IAF_PACK_F32 if f32 else 0) |
                                                    (_capi.IAF_PACK_F16X2 if f16x2 else 0)))
        self._prep_key = None

    def step_is_f16(self, B, H, W):
        """True if the one-launch step at this size runs a two-plane fp16 kernel ("f16x2" precision, a compiled geometry)"""
        return bool(_capi.lib().iaf_stack_step_is_f16(self._h, int(B), int(H), int(W)))

    def range_errors(self):
        """the range word of an "f16x2" stack (iaf_stack_range_errors): 0 = no operand beyond fp16's range so far"""
        e = ctypes.c_uint(0)
        _capi.check(_capi.lib().iaf_stack_range_errors(self._h, ctypes.byref(e)))
        return int(e.value)

    def exchange_errors(self):
        """bounded waits of the halo exchange between row blocks that gave up (iaf_stack_exchange_errors): 0 = never"""
        e = ctypes.c_uint(0)
        _capi.check(_capi.lib().iaf_stack_exchange_errors(self._h, ctypes.byref(e)))
        return int(e.value)

    def set_halo_exchange(self, on=True):
        """False: the one-launch step of this stack recomputes its halo rows instead of exchanging them (include/iaf_hip.h)"""
        _capi.check(_capi.lib().iaf_stack_set_halo_exchange(self._h, 1 if on else 0))

    def set_halo_exchange_debug(self, knobs=0):
        """test knobs of the exchange (include/iaf_hip.h: 1 lists ignore the placement, 2 tickets out of dispatch order,
        8 fault injection); 0 = production"""
        _capi.check(_capi.lib().iaf_stack_set_halo_exchange_debug(self._h, int(knobs)))

    def step_exchanges(self, B, H, W):
        """True if the one-launch step at this size hands halo rows between its row blocks instead of recomputing them"""
        return bool(_capi.lib().iaf_stack_step_exchanges(self._h, int(B), int(H), int(W)))

    def step_pairs(self, B, H, W):
        """True if the one-launch step at this size runs in the pair form (8-pixel rows: two workgroups per two image rows, each half of
        the last hidden layer's channels and of the output pair; include/iaf_hip.h)"""
        return bool(_capi.lib().iaf_stack_step_pairs(self._h, int(B), int(H), int(W)))

    def step_is_fused(self, B, H, W):
        """rows per workgroup of the one-launch step at this size, 0 if the step runs layer by layer"""
        return int(_capi.lib().iaf_stack_step_is_fused(self._h, int(B), int(H), int(W)))

    def set_tuning_bf3(self, layer, nt, ppw, pxt, ks, wco=1):
        _capi.check(_capi.lib().iaf_stack_set_tuning_bf3(self._h, layer, nt, ppw, pxt, ks, wco))

    def step_work(self, B, H, W):
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _capi.check(_capi.lib().iaf_step_work(self._h, B, H, W, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return dict(live_flops=a.value, dense_flops=b.value, bytes=c.value)

    def layer_work(self, layer, B, H, W):
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _capi.check(_capi.lib().iaf_layer_work(self._h, layer, B, H, W, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return dict(live_flops=a.value, dense_flops=b.value, bytes=c.value)

    def profile_enable(self, layer, max_samples=4096):
        """bracket every launch of GEMM layer `layer` with HIP events on the launch stream"""
        _capi.check(_capi.lib().iaf_stack_profile_enable(self._h, layer, max_samples))
        self._prof_cap = max_samples

    def profile_read(self):
        cap = getattr(self, "_prof_cap", 0)
        buf = (ctypes.c_float * max(cap, 1))()
        n = ctypes.c_int()
        _capi.check(_capi.lib().iaf_stack_profile_read(self._h, buf, cap, ctypes.byref(n)))
        return [buf[i] for i in range(n.value)]

    # -- forward -------------------------------------------------------------------------------
    def _dims(self, z, context):
        _check_act(z, "z")
        if z.dim() != 4 or z.shape[1] != self.n_z:
            raise ValueError("z must be [B, %d, H, W], got %s" % (self.n_z, tuple(z.shape)))
        B, _, H, W = z.shape
        if self.depth_ar > 0:
            _check_act(context, "context", (B, self.n_h, H, W))
        return int(B), int(H), int(W)

    def ar_multiconv2d(self, z, context, out=None):
        """layers.py:158-166 -> [m_raw, s_raw]."""
        B, H, W = self._dims(z, context)
        m_raw, s_raw = out if out is not None else (torch.empty_like(z), torch.empty_like(z))
        ws, need = self.workspace(B, H, W, z.device)
        _capi.check(_capi.lib().iaf_ar_multiconv2d_forward(self._h, _ptr(z), _ptr(context), _ptr(m_raw), _ptr(s_raw),
                                                           B, H, W, _ptr(ws), need, _stream()))
        return [m_raw, s_raw]

    def iaf_step(self, z, context, out=None):
        """tf_train.py:69-72 -> (z_new, logsd);  logqs += logsd is the log-det accumulation."""
        B, H, W = self._dims(z, context)
        z_new, logsd = out if out is not None else (torch.empty_like(z), torch.empty_like(z))
        ws, need = self.workspace(B, H, W, z.device)
        _capi.check(_capi.lib().iaf_step_forward(self._h, _ptr(z), _ptr(context), _ptr(z_new), _ptr(logsd),
                                                 B, H, W, _ptr(ws), need, _stream()))
        return z_new, logsd

    def iaf_step_inverse(self, z, context, max_sweeps=64, tol=1e-6, check_every=4, out=None):
        """The inverse of iaf_step: z0 with iaf_step(z0, context)[0] == z (density evaluation of an arbitrary z; not in
        the reference).  Jacobi sweeps z0 <- z*exp(s(z0)) + m(z0); returns (z0, logsd, sweeps, residual).
        tol=0 runs exactly max_sweeps sweeps without synchronising (H*W*n_z sweeps are exact for any weights)."""
        B, H, W = self._dims(z, context)
        z0, logsd = out if out is not None else (torch.empty_like(z), torch.empty_like(z))
        ws, need = self.workspace(B, H, W, z.device)
        n, res = ctypes.c_int(), ctypes.c_float()
        _capi.check(_capi.lib().iaf_step_inverse(self._h, _ptr(z), _ptr(context), _ptr(z0), _ptr(logsd), B, H, W,
                                                 _ptr(ws), need, int(max_sweeps), float(tol), int(check_every), _stream(),
                                                 ctypes.byref(n), ctypes.byref(res)))
        return z0, logsd, n.value, res.value

    def iaf_step_inverse_queued(self, z, context, max_sweeps=64, tol=1e-6, check_every=4, out=None, stats=None):
        """iaf_step_inverse without ANY host synchronisation (iaf_step_inverse_device): every sweep, the residual tests and the early-out
        are queued on the stream -- capturable into a hipGraph.  `stats` (optional): an int32 tensor of two words on the device that receives
        (sweeps run, the last residual's float bits).  Returns (z0, logsd)."""
        B, H, W = self._dims(z, context)
        z0, logsd = out if out is not None else (torch.empty_like(z), torch.empty_like(z))
        ws, need = self.workspace(B, H, W, z.device)
        _capi.check(_capi.lib().iaf_step_inverse_device(self._h, _ptr(z), _ptr(context), _ptr(z0), _ptr(logsd), B, H, W, _ptr(ws), need,
                                                        int(max_sweeps), float(tol), int(check_every), _stream(),
                                                        _ptr(stats) if stats is not None else None))
        return z0, logsd

    # -- training ---------------------------------------------------------------------
    def set_training(self, on=True):
        """allocate the transposed weight packs used by the data-gradient kernels; re-prepare afterwards"""
        _capi.check(_capi.lib().iaf_stack_set_training(self._h, 1 if on else 0))
        self._prep_key = None
        self._train_ws = None

    def _train_workspace(self, B, H, W, device):
        need = int(_capi.lib().iaf_stack_train_workspace_bytes(self._h, B, H, W))
        ws = getattr(self, "_train_ws", None)
        if ws is None or ws.numel() < need or ws.device != device:
            self._train_ws = torch.empty(max(need, 256), dtype=torch.uint8, device=device)
        return self._train_ws, need

    def iaf_step_train(self, z, context):
        """iaf_step that keeps the hidden activations for iaf_step_backward (same outputs as iaf_step)"""
        B, H, W = self._dims(z, context)
        z_new, logsd = torch.empty_like(z), torch.empty_like(z)
        ws, need = self._train_workspace(B, H, W, z.device)
        _capi.check(_capi.lib().iaf_step_forward_train(self._h, _ptr(z), _ptr(context), _ptr(z_new), _ptr(logsd), B, H, W,
                                                       _ptr(ws), need, _stream()))
        return z_new, logsd

    def iaf_step_backward(self, z, context, z_new, logsd, dz_new, dlogsd, params):
        """Gradients of L (given dL/dz_new, dL/dlogsd) w.r.t. z, context and every V/g/b of `params`
        (what opt.compute_gradients derives for these lines, tf_train.py:138).  Must follow iaf_step_train on the
        same inputs.  Returns (dz, dcontext, grads) with grads keyed like params."""
        B, H, W = self._dims(z, context)
        for nm, t in (("z_new", z_new), ("logsd", logsd), ("dz_new", dz_new), ("dlogsd", dlogsd)):
            _check_act(t, nm, z.shape)
        tens = self._param_tensors(params)
        keys = self._grad_keys()
        grads = {}
        for ci, ks in enumerate(keys):
            for j in range(3):
                grads[ks[j]] = torch.empty_like(tens[3 * ci + j])
        n = len(keys)
        arr = ctypes.c_void_p * n
        Vp = arr(*[t.data_ptr() for t in tens[0::3]])
        gp = arr(*[t.data_ptr() for t in tens[1::3]])
        dVp = arr(*[grads[ks[0]].data_ptr() for ks in keys])
        dgp = arr(*[grads[ks[1]].data_ptr() for ks in keys])
        dbp = arr(*[grads[ks[2]].data_ptr() for ks in keys])
        dz = torch.empty_like(z)
        dctx = torch.empty_like(context) if self.depth_ar > 0 else None
        ws, need = self._train_workspace(B, H, W, z.device)
        _capi.check(_capi.lib().iaf_step_backward(self._h, _ptr(z), _ptr(context), _ptr(z_new), _ptr(logsd), _ptr(dz_new),
                                                  _ptr(dlogsd), _ptr(dz), _ptr(dctx), Vp, gp, dVp, dgp, dbp, B, H, W,
                                                  _ptr(ws), need, _stream()))
        return dz, dctx, grads

    def posterior_block_train(self, qz_mean, qz_logsd, rz_mean, rz_logsd, pz_mean, pz_logsd, up_context, down_context, eps,
                              kl_min):
        """posterior_block that keeps what posterior_block_backward needs (same z, kl_obj, kl_cost)"""
        B, H, W = self._dims(qz_mean, up_context)
        self._check_posterior_inputs(qz_mean, qz_logsd, rz_mean, rz_logsd, pz_mean, pz_logsd, eps, up_context, down_context)
        z = torch.empty_like(qz_mean)
        kl_obj = torch.empty(B, dtype=torch.float32, device=z.device)
        kl_cost = torch.empty_like(kl_obj)
        ws, need = self._train_workspace(B, H, W, z.device)
        _capi.check(_capi.lib().iaf_posterior_block_forward_train(
            self._h, _ptr(qz_mean), _ptr(qz_logsd), _ptr(rz_mean), _ptr(rz_logsd), _ptr(pz_mean), _ptr(pz_logsd),
            _ptr(up_context), _ptr(down_context), _ptr(eps), float(kl_min), _ptr(z), _ptr(kl_obj), _ptr(kl_cost), B, H, W,
            _ptr(ws), need, _stream()))
        return dict(z=z, kl_obj=kl_obj, kl_cost=kl_cost)

    def posterior_block_backward(self, qz_mean, qz_logsd, rz_mean, rz_logsd, pz_mean, pz_logsd, eps, kl_min, z, dz, dkl_obj,
                                 params, grads_out=None):
        """Backward of tf_train.py:56-85.  Returns dict(dmean (= d qz_mean = d rz_mean), dlogsd (= d qz_logsd = d rz_logsd),
        dpz_mean, dpz_logsd, dcontext (= d up_context = d down_context), grads {conv/V|g|b})."""
        _check_act(qz_mean, "qz_mean")
        if qz_mean.dim() != 4 or qz_mean.shape[1] != self.n_z:
            raise ValueError("qz_mean must be [B, %d, H, W], got %s" % (self.n_z, tuple(qz_mean.shape)))
        B, _, H, W = qz_mean.shape
        B, H, W = int(B), int(H), int(W)
        self._check_posterior_inputs(qz_mean, qz_logsd, rz_mean, rz_logsd, pz_mean, pz_logsd, eps)
        _check_act(z, "z", qz_mean.shape)
        _check_act(dz, "dz", qz_mean.shape)
        if dkl_obj is None:
            raise ValueError("dkl_obj (gradient of kl_obj, [B]) is required")
        _check_act(dkl_obj, "dkl_obj", (B,))
        tens = self._param_tensors(params)
        keys = self._grad_keys()
        grads = {} if grads_out is None else grads_out     # grads_out: pre-allocated views (e.g. into a flat bucket)
        for ci, ks in enumerate(keys):
            for j in range(3):
                if grads_out is None:
                    grads[ks[j]] = torch.empty_like(tens[3 * ci + j])
                else:
                    _check_act(grads[ks[j]], "grad " + ks[j], tens[3 * ci + j].shape)
        n = len(keys)
        arr = ctypes.c_void_p * n
        Vp = arr(*[t.data_ptr() for t in tens[0::3]])
        gp = arr(*[t.data_ptr() for t in tens[1::3]])
        dVp = arr(*[grads[ks[0]].data_ptr() for ks in keys])
        dgp = arr(*[grads[ks[1]].data_ptr() for ks in keys])
        dbp = arr(*[grads[ks[2]].data_ptr() for ks in keys])
        out = dict(dmean=torch.empty_like(qz_mean), dlogsd=torch.empty_like(qz_mean), dpz_mean=torch.empty_like(qz_mean),
                   dpz_logsd=torch.empty_like(qz_mean), grads=grads)
        out["dcontext"] = (torch.empty(B, self.n_h, H, W, dtype=torch.float32, device=qz_mean.device)
                           if self.depth_ar > 0 else None)
        ws, need = self._train_workspace(B, H, W, qz_mean.device)
        _capi.check(_capi.lib().iaf_posterior_block_backward(
            self._h, _ptr(qz_mean), _ptr(qz_logsd), _ptr(rz_mean), _ptr(rz_logsd), _ptr(pz_mean), _ptr(pz_logsd), _ptr(eps),
            float(kl_min), _ptr(z), _ptr(dz), _ptr(dkl_obj), _ptr(out["dmean"]), _ptr(out["dlogsd"]), _ptr(out["dpz_mean"]),
            _ptr(out["dpz_logsd"]), _ptr(out["dcontext"]), Vp, gp, dVp, dgp, dbp, B, H, W, _ptr(ws), need, _stream()))
        return out

    def _check_posterior_inputs(self, qz_mean, qz_logsd, rz_mean, rz_logsd, pz_mean, pz_logsd, eps, up_context=None,
                                down_context=None):
        """every raw pointer handed to the engine is a contiguous fp32 CUDA tensor of the shape the kernels index with"""
        for nm, t in (("qz_logsd", qz_logsd), ("rz_mean", rz_mean), ("rz_logsd", rz_logsd), ("pz_mean", pz_mean),
                      ("pz_logsd", pz_logsd), ("eps", eps)):
            _check_act(t, nm, qz_mean.shape)
        if self.depth_ar > 0 and up_context is not None:
            _check_act(down_context, "down_context", up_context.shape)

    def posterior_block(self, qz_mean, qz_logsd, rz_mean, rz_logsd, pz_mean, pz_logsd, up_context, down_context, eps,
                        kl_min, want_kl_elem=False, out=None):
        """tf_train.py:56-85 (mode "train") -> dict(z, kl_obj[B], kl_cost[B] [, kl_elem]).
        out: optional dict of pre-allocated result tensors (z, kl_obj, kl_cost) -- e.g. rows of a [layers, B] matrix."""
        B, H, W = self._dims(qz_mean, up_context)
        for nm, t in (("qz_logsd", qz_logsd), ("rz_mean", rz_mean), ("rz_logsd", rz_logsd), ("pz_mean", pz_mean),
                      ("pz_logsd", pz_logsd), ("eps", eps)):
            _check_act(t, nm, qz_mean.shape)
        if self.depth_ar > 0:
            _check_act(down_context, "down_context", up_context.shape)
        out = out or {}
        z = out.get("z") if out.get("z") is not None else torch.empty_like(qz_mean)
        kl_obj = out.get("kl_obj") if out.get("kl_obj") is not None else torch.empty(B, dtype=torch.float32, device=z.device)
        kl_cost = out.get("kl_cost") if out.get("kl_cost") is not None else torch.empty(B, dtype=torch.float32, device=z.device)
        _check_act(z, "out z", qz_mean.shape)
        _check_act(kl_obj, "out kl_obj", (B,))
        _check_act(kl_cost, "out kl_cost", (B,))
        kl_elem = torch.empty_like(qz_mean) if want_kl_elem else None
        ws, need = self.workspace(B, H, W, z.device)
        _capi.check(_capi.lib().iaf_posterior_block_forward(
            self._h, _ptr(qz_mean), _ptr(qz_logsd), _ptr(rz_mean), _ptr(rz_logsd), _ptr(pz_mean), _ptr(pz_logsd),
            _ptr(up_context), _ptr(down_context), _ptr(eps), float(kl_min), _ptr(z), _ptr(kl_obj), _ptr(kl_cost),
            _ptr(kl_elem), B, H, W, _ptr(ws), need, _stream()))
        out = dict(z=z, kl_obj=kl_obj, kl_cost=kl_cost)
        if want_kl_elem:
            out["kl_elem"] = kl_elem
        return out


RESAMPLE_MODES = {"down_even": 0, "down_odd": 1, "up_nearest": 2, "up_zero_odd": 3, "up_zero_even": 4, "down_sum4": 5}


def resample2(x, mode):
    """2x resampling of an NCHW tensor on the GPU (include/iaf_hip.h, iaf_resample2):
    "down_even" == resize_nearest_neighbor(x, 0.5), "up_nearest" == resize_nearest_neighbor(x, 2) (layers.py:169-175);
    "down_odd" keeps what a stride-2 SAME 3x3 conv keeps of the stride-1 conv; "up_zero_odd" is the zero-inserted
    input of conv2d_transpose; "up_zero_even" / "down_sum4" are the adjoints of "down_even" / "up_nearest" (backward)."""
    _check_act(x, "x")
    B, C, H, W = (int(v) for v in x.shape)
    m = RESAMPLE_MODES[mode]
    if m in (0, 1, 5):
        if H % 2 or W % 2:
            raise ValueError("2x downsampling needs even H and W, got %dx%d" % (H, W))
        out = torch.empty((B, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
        hs, ws = H // 2, W // 2
    else:
        out = torch.empty((B, C, 2 * H, 2 * W), dtype=x.dtype, device=x.device)
        hs, ws = H, W
    _capi.check(_capi.lib().iaf_resample2(_ptr(x), _ptr(out), B, C, hs, ws, m, _stream()))
    return out


def resize_nearest_neighbor(x, scale):
    """tf_utils/layers.py:169-175 for the two scales the reference uses (tf_train.py:43,90)"""
    if scale == 0.5:
        return resample2(x, "down_even")
    if scale == 2:
        return resample2(x, "up_nearest")
    raise ValueError("the gfx950 engine resizes by 0.5 or 2 (the IAFLayer uses), got %r" % (scale,))


class WNConv2d(object):
    """One plain weight-normed 3x3 conv (tf_utils/layers.py:31-64, mask=None, stride (1,1), pad SAME) bound to an
    engine handle (iaf_conv3x3_t).  The elementwise work the reference wraps around it in IAFLayer
    (tf_train.py:35-44, 52-54, 87-94) is fused into the call: ELU on the input, channel concat of two inputs,
    channel split of the output, and the residual `res + 0.1*y`."""

    def __init__(self, n_in, n_out, ar_mask=None, theano=False, flipmask=False):
        """ar_mask: None = plain conv2d; False / True = ar_conv2d with zerodiagonal=False / True (layers.py:144-154).
        theano=True (masked only): the Theano statement, N.ar.conv2d (graphy/nodes/ar.py:200-375), optionally flipmask;
        prepare() then takes (w OIHW [n_out, n_in+1, 3, 3], s, b)."""
        self.n_in, self.n_out, self.ar_mask, self.theano = int(n_in), int(n_out), ar_mask, bool(theano)
        self._h = ctypes.c_void_p()
        if theano:
            if ar_mask is None:
                raise ValueError("theano=True is the masked conv N.ar.conv2d: pass ar_mask=False/True (zerodiagonal)")
            _capi.check(_capi.lib().iaf_conv3x3_create_masked_theano(ctypes.byref(self._h), self.n_in, self.n_out,
                                                                     1 if ar_mask else 0, 1 if flipmask else 0))
        elif ar_mask is None:
            _capi.check(_capi.lib().iaf_conv3x3_create(ctypes.byref(self._h), self.n_in, self.n_out))
        else:
            _capi.check(_capi.lib().iaf_conv3x3_create_masked(ctypes.byref(self._h), self.n_in, self.n_out,
                                                              1 if ar_mask else 0))
        self._tuned, self._cur_tune = {}, None      # (B,H,W) -> launch shape found by autotune
        self._prep_key = None
        self._keepalive = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _capi.lib().iaf_conv3x3_destroy(h)
            except Exception:
                pass

    def prepare(self, V, g, b, force=False):
        """V HWIO [3,3,n_in,n_out], g/b [n_out] (layers.py:53-55); cached until a tensor is replaced or modified.
        (theano=True: V = w OIHW [n_out, n_in+1, 3, 3], g = s, ar.py:288-296.)"""
        _check_act(V, "V", (self.n_out, self.n_in + 1, 3, 3) if self.theano else (3, 3, self.n_in, self.n_out))
        _check_act(g, "g", (self.n_out,))
        _check_act(b, "b", (self.n_out,))
        key = tuple((t.data_ptr(), t._version) for t in (V, g, b))
        if not force and key == self._prep_key:
            return
        _capi.check(_capi.lib().iaf_conv3x3_prepare(self._h, _ptr(V), _ptr(g), _ptr(b), _stream()))
        self._prep_key, self._keepalive = key, (V, g, b)

    def prepare_deconv(self, V, g, b, force=False):
        """deconv2d variables (layers.py:83-112): V [3,3,n_out,n_in].  Afterwards __call__ on the zero-inserted input
        (resample2(x, "up_zero_odd")) at the output resolution equals the reference's conv2d_transpose(SAME, stride 2) + b."""
        _check_act(V, "V", (3, 3, self.n_out, self.n_in))
        _check_act(g, "g", (self.n_out,))
        _check_act(b, "b", (self.n_out,))
        key = ("deconv",) + tuple((t.data_ptr(), t._version) for t in (V, g, b))
        if not force and key == self._prep_key:
            return
        _capi.check(_capi.lib().iaf_conv3x3_prepare_deconv(self._h, _ptr(V), _ptr(g), _ptr(b), _stream()))
        self._prep_key, self._keepalive = key, (V, g, b)

    def set_tuning(self, nt, pxt, wco, ks):
        _capi.check(_capi.lib().iaf_conv3x3_set_tuning(self._h, nt, pxt, wco, ks))

    def set_precision(self, precision):
        """"bf16x3" (default: the forward conv on the bf16 matrix cores with split products, fp32-grade, where a launch shape
        covers it) or "f32" (the exact-fp32 MFMA kernel always)"""
        code = {"f32": _capi.IAF_PRECISION_F32, "bf16x3": _capi.IAF_PRECISION_BF16X3, "f16x2": _capi.IAF_PRECISION_F16X2}.get(precision, precision)
        rc = _capi.lib().iaf_conv3x3_set_precision(self._h, int(code))
        if rc == _capi.IAF_ERR_UNSUPPORTED and code == _capi.IAF_PRECISION_F16X2:      # (no split pack for this conv: bf16x3 semantics)
            rc = _capi.lib().iaf_conv3x3_set_precision(self._h, _capi.IAF_PRECISION_BF16X3)
        _capi.check(rc)
        self._prep_key = None

    def range_errors(self):
        """the range word of an "f16x2" conv (iaf_conv3x3_range_errors): 0 = no operand beyond fp16's range so far"""
        e = ctypes.c_uint(0)
        _capi.check(_capi.lib().iaf_conv3x3_range_errors(self._h, ctypes.byref(e)))
        return int(e.value)

    def runs_bf16x3(self, B, H, W):
        return bool(_capi.lib().iaf_conv3x3_runs_bf16x3(self._h, int(B), int(H), int(W)))

    def runs_f16x2(self, B, H, W):
        return bool(_capi.lib().iaf_conv3x3_runs_f16x2(self._h, int(B), int(H), int(W)))

    def set_packs(self, f32=True, bf16x3=True, f16x2=True):
        """which weight packs the prep launches of this plain conv keep up to date (iaf_conv3x3_set_packs; default: all three, 14 bytes
        written per weight).  A launch whose pack is not kept raises IafHipError (IAF_ERR_NOT_PREPARED); training convs keep every pack."""
        _capi.check(_capi.lib().iaf_conv3x3_set_packs(self._h, (_capi.IAF_PACK_F32 if f32 else 0) | (_capi.IAF_PACK_BF16X3 if bf16x3 else 0) |
                                                      (_capi.IAF_PACK_F16X2 if f16x2 else 0)))
        self._prep_key = None

    def trim_packs(self, B, H, W, strided=False):
        """keep only the pack the forward launch at this size reads: the two-plane fp16 one, else the bf16x3 one (also what the stride-2
        form reads: strided=True), else the fp32 one.  Call it again after autotune / set_precision / set_tuning; returns the pack kept."""
        if strided or (self.runs_bf16x3(B, H, W) and not self.runs_f16x2(B, H, W)):
            self.set_packs(f32=False, bf16x3=True, f16x2=False)
            return "bf16x3"
        if self.runs_f16x2(B, H, W):
            self.set_packs(f32=False, bf16x3=False, f16x2=True)
            return "f16x2"
        self.set_packs(f32=True, bf16x3=False, f16x2=False)
        return "f32"

    # -- training --------------------------------------------------------------------------------
    _shared_ws = {}     # device -> one scratch buffer shared by all plain convs (they run one after another)

    def set_training(self, on=True):
        """allocate the transposed weight pack the data gradient uses; the next prepare fills it"""
        _capi.check(_capi.lib().iaf_conv3x3_set_training(self._h, 1 if on else 0))
        self._prep_key = None

    def backward(self, x, dys, V, g, x2=None, elu_input=False, dy_scale=1.0, want_dx=True, dx_residual=None,
                 grads_out=None, autotune=False):
        """Backward of __call__ (what TF autodiff derives for layers.py:52-64 and the elu/concat/split/residual around it).
        dys: gradients of the split outputs (same shapes as __call__ returned); dy_scale multiplies them (0.1 for the
        residual form).  Returns (dxs, dV, dg, db): dxs = gradients w.r.t. [x] or [x, x2] (None if not want_dx),
        = [dx_residual +] act'(input) * W^T dY."""
        _check_act(x, "x")
        B, c1, H, W = (int(v) for v in x.shape)
        c_split = 0
        if x2 is not None:
            _check_act(x2, "x2", (B, self.n_in - c1, H, W))
            c_split = c1
        for d in dys:
            _check_act(d, "dy")
        n = len(dys)
        dyp = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dys])
        dyc = (ctypes.c_int * n)(*[int(d.shape[1]) for d in dys])
        dxs = None
        ndx, dxp, dxc = 0, None, None
        if want_dx:
            dxs = [torch.empty_like(x)] + ([torch.empty_like(x2)] if x2 is not None else [])
            ndx = len(dxs)
            dxp = (ctypes.c_void_p * ndx)(*[d.data_ptr() for d in dxs])
            dxc = (ctypes.c_int * ndx)(*[int(d.shape[1]) for d in dxs])
            if dx_residual is not None:
                _check_act(dx_residual, "dx_residual", tuple(x.shape))
        if grads_out is None:
            dV, dg, db = torch.empty_like(V), torch.empty_like(g), torch.empty_like(g)
        else:
            dV, dg, db = grads_out
        need = _capi.lib().iaf_conv3x3_train_workspace_bytes(self._h, B, H, W)
        ws = WNConv2d._shared_ws.get(x.device)
        if ws is None or ws.numel() * 4 < need:
            ws = WNConv2d._shared_ws[x.device] = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
        key = ("bwd", B, H, W)
        if autotune and want_dx and key not in self._tuned:       # first call at this size: search the dgrad launch shape
            sh, us = (ctypes.c_int * 4)(), ctypes.c_float()
            _capi.check(_capi.lib().iaf_conv3x3_autotune_backward(
                self._h, _ptr(x), _ptr(x2), c_split, 1 if elu_input else 0, dyp, dyc, n, float(dy_scale), dxp, dxc, ndx,
                _ptr(dx_residual), _ptr(V), _ptr(g), _ptr(dV), _ptr(dg), _ptr(db), B, H, W, _ptr(ws), need, 10, _stream(),
                sh, ctypes.byref(us)))
            self._tuned[key] = tuple(sh)
            return dxs, dV, dg, db
        _capi.check(_capi.lib().iaf_conv3x3_backward(
            self._h, _ptr(x), _ptr(x2), c_split, 1 if elu_input else 0, dyp, dyc, n, float(dy_scale), dxp, dxc, ndx,
            _ptr(dx_residual), _ptr(V), _ptr(g), _ptr(dV), _ptr(dg), _ptr(db), B, H, W, _ptr(ws), need, _stream()))
        return dxs, dV, dg, db

    def init(self, x, V, init_scale=0.1, x2=None, elu_input=False, add=None):
        """Data-dependent initialisation, the init=True branch of conv2d (layers.py:38-51).  Returns (y, g, b):
        y = scale*(x_init - mean) [+ add], g = log(scale)/3, b = -mean*scale with the moments of
        x_init = conv(x, l2_normalize(mask*V)) over (N,H,W).  Leaves the conv prepared with (V, g, b)."""
        zeros = torch.zeros(self.n_out, device=V.device, dtype=torch.float32)
        self.prepare(V, zeros, zeros.clone(), force=True)      # exp(0) * l2_normalize(mask*V), bias 0  (:44-45)
        x_init = self(x, x2=x2, elu_input=elu_input)[0]
        B, _, H, W = x_init.shape
        if add is not None:
            _check_act(add, "add", tuple(x_init.shape))
        g, b = torch.empty_like(zeros), torch.empty_like(zeros)
        _capi.check(_capi.lib().iaf_datainit_normalize(_ptr(x_init), _ptr(add), _ptr(x_init), _ptr(g), _ptr(b), B,
                                                       self.n_out, H * W, float(init_scale), _stream()))
        self.prepare(V, g, b, force=True)
        return x_init, g, b

    def work(self, B, H, W):
        fl, by = ctypes.c_double(), ctypes.c_double()
        _capi.check(_capi.lib().iaf_conv3x3_work(self._h, B, H, W, ctypes.byref(fl), ctypes.byref(by)))
        return fl.value, by.value

    def __call__(self, x, x2=None, elu_input=False, split=None, residual=None, out=None, autotune=False):
        """x [B,c,H,W] (+ optional x2 [B,n_in-c,H,W], concatenated along channels).  Returns the list of split
        tensors (`split` = channel counts, default [n_out]) or, with `residual`, the single tensor
        residual + 0.1*y."""
        _check_act(x, "x")
        B, c1, H, W = x.shape
        c_split = 0
        if x2 is not None:
            _check_act(x2, "x2", (B, self.n_in - c1, H, W))
            c_split = c1
        elif c1 != self.n_in:
            raise ValueError("x has %d channels, expected %d" % (c1, self.n_in))
        split = [self.n_out] if split is None else [int(v) for v in split]
        if sum(split) != self.n_out:
            raise ValueError("split %r does not sum to %d" % (split, self.n_out))
        if residual is not None:
            _check_act(residual, "residual", (B, self.n_out, H, W))
        if out is None:
            out = [torch.empty((B, c, H, W), device=x.device, dtype=torch.float32) for c in split]
        for o, c in zip(out, split):
            _check_act(o, "out", (B, c, H, W))
        n = len(split)
        outs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in out])
        chans = (ctypes.c_int * n)(*split)
        key = (B, H, W)
        if autotune and key not in self._tuned:
            sh, us = (ctypes.c_int * 4)(), ctypes.c_float()
            _capi.check(_capi.lib().iaf_conv3x3_autotune(self._h, _ptr(x), _ptr(x2), c_split, 1 if elu_input else 0,
                                                         _ptr(residual), outs, chans, n, B, H, W, 20, _stream(), sh,
                                                         ctypes.byref(us)))
            self._tuned[key] = tuple(sh)
            self._cur_tune = key
            return out
        if key in self._tuned and self._cur_tune != key:
            self.set_tuning(*self._tuned[key])
            self._cur_tune = key
        elif key not in self._tuned and self._cur_tune is not None:
            self.set_tuning(0, 0, 0, 0)
            self._cur_tune = None
        _capi.check(_capi.lib().iaf_conv3x3_forward(self._h, _ptr(x), _ptr(x2), c_split, 1 if elu_input else 0,
                                                    _ptr(residual), outs, chans, n, B, H, W, _stream()))
        return out


    def stride2(self, x, elu_input=False, split=None):
        """conv2d(..., stride=[2,2]) (tf_train.py:33,36) at its minimal work: x [B,n_in,2H,2W] -> split tensors [B,.,H,W].
        Shapes the strided kernel does not cover (UnsupportedError from the engine) run as the stride-1 conv subsampled at the
        odd positions -- the same numbers, four times the multiplies."""
        _check_act(x, "x")
        B, c1, H2, W2 = (int(v) for v in x.shape)
        if c1 != self.n_in or H2 % 2 or W2 % 2:
            raise ValueError("stride2: x must be [B,%d,even,even], got %r" % (self.n_in, tuple(x.shape)))
        H, W = H2 // 2, W2 // 2
        split = [self.n_out] if split is None else [int(v) for v in split]
        if sum(split) != self.n_out:
            raise ValueError("split %r does not sum to %d" % (split, self.n_out))
        out = [torch.empty((B, c, H, W), device=x.device, dtype=torch.float32) for c in split]
        n = len(split)
        outs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in out])
        chans = (ctypes.c_int * n)(*split)
        rc = _capi.lib().iaf_conv3x3_forward_stride2(self._h, _ptr(x), 1 if elu_input else 0, outs, chans, n, B, H, W, _stream())
        if rc == _capi.IAF_ERR_UNSUPPORTED:
            return [resample2(t, "down_odd") for t in self(x, elu_input=elu_input, split=split)]
        _capi.check(rc)
        return out

    def deconv(self, x, x2=None, elu_input=False, residual=None):
        """[residual upsampled + 0.1 *] deconv2d(..., stride 2) (tf_train.py:87-94) of a conv prepared by prepare_deconv, at its
        minimal work: x (, x2) [B,.,H,W] -> [B,n_out,2H,2W]; residual [B,n_out,H,W] is resize_nearest_neighbor'ed inside.
        Shapes the phase kernel does not cover run as the stride-1 conv of the zero-inserted inputs (the same numbers)."""
        _check_act(x, "x")
        B, c1, H, W = (int(v) for v in x.shape)
        c_split = 0
        if x2 is not None:
            _check_act(x2, "x2", (B, self.n_in - c1, H, W))
            c_split = c1
        elif c1 != self.n_in:
            raise ValueError("x has %d channels, expected %d" % (c1, self.n_in))
        if residual is not None:
            _check_act(residual, "residual", (B, self.n_out, H, W))
        out = torch.empty((B, self.n_out, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
        rc = _capi.lib().iaf_conv3x3_forward_deconv(self._h, _ptr(x), _ptr(x2), c_split, 1 if elu_input else 0, _ptr(residual),
                                                    _ptr(out), B, H, W, _stream())
        if rc == _capi.IAF_ERR_UNSUPPORTED:
            return self(resample2(x, "up_zero_odd"), x2=None if x2 is None else resample2(x2, "up_zero_odd"), elu_input=elu_input,
                        residual=None if residual is None else resample2(residual, "up_nearest"))[0]
        _capi.check(rc)
        return out


def _conv_op(name, x, num_filters, ar_mask, init, init_scale, st):
    n_in = int(x.shape[1])
    with variable_scope(name, st):
        key = ("conv2d", st.full_name(""), n_in, int(num_filters), ar_mask)
        conv = st._stacks.get(key)
        if conv is None:
            conv = st._stacks[key] = WNConv2d(n_in, num_filters, ar_mask)
        if init:      # layers.py:38-51: V keeps its initial value, g and b are created from the data
            y, g, b = conv.init(x, st.get("V"), init_scale)
            st.set(st.full_name("g"), g)
            st.set(st.full_name("b"), b)
            return y
        conv.prepare(st.get("V"), st.get("g"), st.get("b"))
    return conv(x)[0]


def conv2d(name, x, num_filters, filter_size=(3, 3), stride=(1, 1), pad="SAME", init_scale=0.1, init=False, mask=None,
           store=None, **_):
    """Drop-in for tf_utils/layers.py:31-64 for the shape IAFLayer uses at its non-downsampling levels: 3x3, stride 1,
    SAME, no mask (masked: ar_conv2d below).  Variables <scope>/<name>/{V,g,b} come from `store`; with init=True V must
    hold its initial value and g, b are written to the store."""
    if tuple(filter_size) != (3, 3) or tuple(stride) != (1, 1) or pad != "SAME" or mask is not None:
        raise ValueError("the gfx950 engine implements conv2d for filter 3x3, stride 1, SAME, mask=None")
    return _conv_op(name, x, num_filters, None, init, init_scale, store or _DEFAULT_STORE)


def ar_conv2d(name, x, num_filters, filter_size=(3, 3), stride=(1, 1), pad="SAME", init_scale=1., zerodiagonal=True,
              init=False, store=None, **_):
    """Drop-in for tf_utils/layers.py:144-154: one MADE-masked weight-normed conv."""
    if tuple(filter_size) != (3, 3) or tuple(stride) != (1, 1) or pad != "SAME":
        raise ValueError("the gfx950 engine implements ar_conv2d for filter 3x3, stride 1, SAME")
    return _conv_op(name, x, num_filters, bool(zerodiagonal), init, init_scale, store or _DEFAULT_STORE)


def split(x, split_dim, split_sizes):
    """tf_utils/common.py:21-36: cut `x` along `split_dim` into pieces of the given sizes (the boundary between
    down_conv1 / up_conv1 and the IAF step, tf_train.py:37,54).  Same failure mode as the reference: sizes that do not
    add up fail its assert (common.py:24).  Device-side copies by torch (plumbing); WNConv2d(split=...) avoids them altogether by
    writing the pieces directly."""
    n = int(x.shape[split_dim])
    assert sum(split_sizes) == n, "split sizes %r do not add up to dimension %d" % (list(split_sizes), n)   # common.py:24
    out, begin = [], 0
    for size in split_sizes:
        out.append(x.narrow(split_dim, begin, size).contiguous())
        begin += size
    return out


def kl_free_bits(kl, kl_min, want_gate=False):
    """[B, C, H, W] KL elements -> (kl_cost [B], kl_obj [B]) (tf_train.py:77-85) with the engine's reduction kernels;
    want_gate: also the per-channel gate [C] of the free bits (1 where the batch mean of the channel's KL exceeds kl_min)"""
    _check_act(kl, "kl")
    B, C, H, W = (int(v) for v in kl.shape)
    kl_obj = torch.empty(B, dtype=torch.float32, device=kl.device)
    kl_cost = torch.empty_like(kl_obj)
    scratch = torch.empty(B * C, dtype=torch.float32, device=kl.device)
    if want_gate:
        gate = torch.empty(C, dtype=torch.float32, device=kl.device)
        _capi.check(_capi.lib().iaf_kl_free_bits_gate(_ptr(kl), _ptr(kl_obj), _ptr(kl_cost), _ptr(gate), B, C, H * W, float(kl_min),
                                                      _ptr(scratch), _stream()))
        return kl_cost, kl_obj, gate
    _capi.check(_capi.lib().iaf_kl_free_bits(_ptr(kl), _ptr(kl_obj), _ptr(kl_cost), B, C, H * W, float(kl_min), _ptr(scratch),
                                             _stream()))
    return kl_cost, kl_obj


class PrepBatch(object):
    """Weight prep (mask, l2-normalise, exp(g), repack) for MANY stacks in one launch -- what a model does once at
    the start of every step (the reference re-derives the normalised weights inside every conv2d call,
    layers.py:56-60)."""

    def __init__(self, stacks):
        self.stacks = list(stacks)
        arr = (ctypes.c_void_p * len(self.stacks))(*[s._h.value for s in self.stacks])
        self._h = ctypes.c_void_p()
        _capi.check(_capi.lib().iaf_prep_batch_create(ctypes.byref(self._h), arr, len(self.stacks)))
        self._keepalive = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _capi.lib().iaf_prep_batch_destroy(h)
            except Exception:
                pass

    def run(self, params_list):
        tens = []
        for st, params in zip(self.stacks, params_list):
            tens += st._param_tensors(params)
        n = len(tens) // 3
        arr = ctypes.c_void_p * n
        Vp = arr(*[t.data_ptr() for t in tens[0::3]])
        gp = arr(*[t.data_ptr() for t in tens[1::3]])
        bp = arr(*[t.data_ptr() for t in tens[2::3]])
        _capi.check(_capi.lib().iaf_prep_batch_run(self._h, Vp, gp, bp, _stream()))
        self._keepalive = tens
        for st in self.stacks:
            st._prep_key = None      # per-stack cache no longer describes what is on the device


class ConvPrepBatch(object):
    """PrepBatch for the plain convs: l2-normalise, exp(g), repack of MANY WNConv2d objects in one launch."""

    def __init__(self, convs):
        self.convs = list(convs)
        arr = (ctypes.c_void_p * len(self.convs))(*[c._h.value for c in self.convs])
        self._h = ctypes.c_void_p()
        _capi.check(_capi.lib().iaf_conv3x3_prep_batch_create(ctypes.byref(self._h), arr, len(self.convs)))
        self._keepalive = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _capi.lib().iaf_conv3x3_prep_batch_destroy(h)
            except Exception:
                pass

    def run(self, vgb_list):
        """vgb_list: one (V, g, b) tuple of device tensors per conv, in construction order."""
        tens = []
        for c, (V, g, b) in zip(self.convs, vgb_list):
            _check_act(V, "V", (3, 3, c.n_in, c.n_out))
            _check_act(g, "g", (c.n_out,))
            _check_act(b, "b", (c.n_out,))
            tens += [V, g, b]
        n = len(self.convs)
        arr = ctypes.c_void_p * n
        Vp = arr(*[t.data_ptr() for t in tens[0::3]])
        gp = arr(*[t.data_ptr() for t in tens[1::3]])
        bp = arr(*[t.data_ptr() for t in tens[2::3]])
        _capi.check(_capi.lib().iaf_conv3x3_prep_batch_run(self._h, Vp, gp, bp, _stream()))
        self._keepalive = tens
        for c in self.convs:
            c._prep_key = None


class WnBwdBatch(object):
    """Deferred mask + weight-norm backward of MANY stacks and / or plain convs: their backward calls stop after the
    weight-gradient reduction and `run` finishes all of them in one launch per kind (a stack's own pass is 24
    workgroups on a 256-CU chip).  Construction switches the objects to deferred mode."""

    def __init__(self, stacks=(), convs=()):
        self.stacks, self.convs = list(stacks), list(convs)
        self._hs = self._hc = None
        lib = _capi.lib()
        if self.stacks:
            for st in self.stacks:
                _capi.check(lib.iaf_stack_set_defer_weightnorm(st._h, 1))
            arr = (ctypes.c_void_p * len(self.stacks))(*[st._h.value for st in self.stacks])
            self._hs = ctypes.c_void_p()
            _capi.check(lib.iaf_wn_bwd_batch_create(ctypes.byref(self._hs), arr, len(self.stacks)))
        if self.convs:
            for c in self.convs:
                _capi.check(lib.iaf_conv3x3_set_defer_weightnorm(c._h, 1))
            arr = (ctypes.c_void_p * len(self.convs))(*[c._h.value for c in self.convs])
            self._hc = ctypes.c_void_p()
            _capi.check(lib.iaf_conv3x3_wn_bwd_batch_create(ctypes.byref(self._hc), arr, len(self.convs)))

    def __del__(self):
        try:
            lib = _capi.lib()
            if getattr(self, "_hs", None):
                lib.iaf_wn_bwd_batch_destroy(self._hs)
            if getattr(self, "_hc", None):
                lib.iaf_conv3x3_wn_bwd_batch_destroy(self._hc)
        except Exception:
            pass

    @staticmethod
    def _tables(tens):
        arr = ctypes.c_void_p * len(tens)
        return arr(*[t.data_ptr() for t in tens])

    def run(self, stack_params=(), stack_grads=(), conv_params=(), conv_grads=()):
        """stack_params / stack_grads: one {conv/V|g|b: tensor} dict per stack (as for PrepBatch.run and
        posterior_block_backward(grads_out=)); conv_params / conv_grads: one (V, g, b) / (dV, dg, db) tuple per conv."""
        lib = _capi.lib()
        if self._hs:
            V, g, dV, dg, db = [], [], [], [], []
            for st, p, gr in zip(self.stacks, stack_params, stack_grads):
                for kV, kg, kb in st._grad_keys():
                    V.append(p[kV]); g.append(p[kg])
                    dV.append(gr[kV]); dg.append(gr[kg]); db.append(gr[kb])
            _capi.check(lib.iaf_wn_bwd_batch_run(self._hs, self._tables(V), self._tables(g), self._tables(dV),
                                                 self._tables(dg), self._tables(db), _stream()))
        if self._hc:
            V = [p[0] for p in conv_params]; g = [p[1] for p in conv_params]
            dV = [q[0] for q in conv_grads]; dg = [q[1] for q in conv_grads]; db = [q[2] for q in conv_grads]
            _capi.check(lib.iaf_conv3x3_wn_bwd_batch_run(self._hc, self._tables(V), self._tables(g), self._tables(dV),
                                                         self._tables(dg), self._tables(db), _stream()))


class _Struct(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __call__(self, *a, **k):
        return self.f(*a, **k)


def multiconv2d(name, n_in, n_h, n_out, size_kernel=(3, 3), flipmask=False, nl="elu", w=None):
    """Mirror of graphy/nodes/ar.py:378-423 (the Theano statement of the operator): returns a callable struct
    `f(h, context, w)` -> [out_0, out_1] reading w[name+'_%d_w'|'_b'|'_s'] and w[name+'_out_%d_...'] (ar.py:288-296),
    as constructed at models.py:63,92 with nl='elu' (train.py:61); flipmask as in ar.py:263-264."""
    if isinstance(n_out, int):
        n_out = [n_out]
    if tuple(size_kernel) != (3, 3) or nl != "elu":
        raise ValueError("the gfx950 engine implements size_kernel=(3,3), nl='elu'")
    stack = ARStack(n_in, list(n_h), list(n_out),
                    variant=_capi.IAF_VARIANT_THEANO_FLIPMASK if flipmask else _capi.IAF_VARIANT_THEANO)

    def f(h, context, w, return_hiddens=False):
        if return_hiddens:
            raise ValueError("hidden activations stay on chip; return_hiddens is not available")
        rel = {k[len(name) + 1:]: v for k, v in w.items() if k.startswith(name + "_")}
        stack.prepare(rel)
        return stack.ar_multiconv2d(h, context)

    return _Struct(f=f, w=w, stack=stack, postup=lambda updates, w: updates)


def ar_conv2d_theano(name, n_in, n_out, size_kernel=(3, 3), zerodiagonal=True, flipmask=False, w=None):
    """Mirror of N.ar.conv2d (graphy/nodes/ar.py:200-375) with its defaults pad_channel=True, border_mode='valid',
    l2norm=True: returns a callable struct f(h, w) reading w[name+'_w'|'_b'|'_s']."""
    if tuple(size_kernel) != (3, 3):
        raise ValueError("the gfx950 engine implements size_kernel=(3,3)")
    assert n_out % n_in == 0 or n_in % n_out == 0                                  # ar.py:250,257
    conv = WNConv2d(n_in, n_out, ar_mask=bool(zerodiagonal), theano=True, flipmask=flipmask)

    def f(h, w):
        conv.prepare(w[name + "_w"], w[name + "_s"], w[name + "_b"])
        return conv(h)[0]

    return _Struct(f=f, w=w, conv=conv, postup=lambda updates, w: updates)


def _is_elu(nl):
    return nl in ("elu", None) or getattr(nl, "__name__", "") == "elu"


def _ar_multiconv2d_init(x, context, n_h, n_out, st):
    """layers.py:158-166 with init=True arriving through arg_scope (tf_train.py:175): every masked conv is initialised
    from the data flowing through the stack; the context is added to the first layer's NORMALISED output (:163-164)."""
    h, first = x, True
    for i, size in enumerate(n_h):
        with variable_scope("layer_%d" % i, st):
            conv = WNConv2d(int(h.shape[1]), size, ar_mask=False)
            h, g, b = conv.init(h, st.get("V"), 1.0, elu_input=not first, add=context if i == 0 else None)
            st.set(st.full_name("g"), g)
            st.set(st.full_name("b"), b)
        first = False
    outs = []
    for i, size in enumerate(n_out):
        with variable_scope("layer_out_%d" % i, st):
            conv = WNConv2d(int(h.shape[1]), size, ar_mask=True)
            y, g, b = conv.init(h, st.get("V"), 1.0, elu_input=not first)
            st.set(st.full_name("g"), g)
            st.set(st.full_name("b"), b)
            outs.append(y)
    return outs


def ar_multiconv2d(name, x, context, n_h, n_out, nl="elu", store=None, init=False, **_):
    """Drop-in for tf_utils/layers.py:158-166 (call site tf_train.py:69):
        x = ar_multiconv2d("ar_multiconv2d", z, context, [h_size, h_size], [z_size, z_size])
    Variables are looked up as <current scope>/<name>/layer_{i}/{V,g,b} etc. in `store`."""
    if not _is_elu(nl):
        raise ValueError("the gfx950 engine fuses the ELU non-linearity only (layers.py:159 default)")
    st = store or _DEFAULT_STORE
    n_z = int(x.shape[1])
    if init:
        with variable_scope(name, st):
            return _ar_multiconv2d_init(x, context, list(n_h), list(n_out), st)
    with variable_scope(name, st):
        prefix = st.full_name("")
        key = (prefix, n_z, tuple(n_h), tuple(n_out))
        stack = st._stacks.get(key)
        if stack is None:
            stack = st._stacks[key] = ARStack(n_z, n_h, n_out)
        params = {}
        for conv in stack.conv_names():
            for v in ("V", "g", "b"):
                params[conv + "/" + v] = st.get(conv + "/" + v)
    stack.prepare(params)
    return stack.ar_multiconv2d(x, context)
