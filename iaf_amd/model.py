"""CVAE1 -- the caller of the hot path: the reference's whole model forward, CVAE1._forward (tf_train.py:150-218), for one tower:
image scaling, conv2d("x_enc", 5x5, stride 2), the bottom-up pass through depth x num_blocks IAFLayers, the tiled h_top, the
top-down pass accumulating kl_obj / kl_cost, deconv2d("x_dec", 5x5) + clip, discretized_logistic, obj and the (k-sample) loss.
Every number is computed by HIP launches behind the C ABI (include/iaf_hip.h); torch tensors are storage.  The two edge convs
have 3 channels on one side and run as direct convolutions (csrc/iaf_model_edge.hpp); the layers are iaf_amd.IAFLayer.
forward() for every mode; forward_backward() = one tower's training objective and its gradient w.r.t. every variable (mode "train",
k = 1): the layers' backward (IAFLayer.down_backward / .up_backward) chained through the model + the backward of the two ends."""
import math

import torch

from . import _capi
from .distributions import StreamingLowerBound, compute_lowerbound, discretized_logistic
from .iaf_layer import IAFLayer
from .layers import (ConvPrepBatch, PrepBatch, VariableStore, WnBwdBatch, ar_multiconv2d, resample2, variable_scope, _check_act, _ptr,
                     _stream)


class CVAE1(object):
    """Same hyper-parameters as the reference's HParams (tf_train.py:98-112) that _forward reads: z_size, h_size, kl_min, depth
    (number of resolution levels), num_blocks (layers per level), k (importance samples), image_size.  `mode` as in the
    reference: "train" (posterior samples), "init" (prior samples through the posterior block), "sample"."""

    def __init__(self, z_size=32, h_size=160, kl_min=0.25, depth=2, num_blocks=2, k=1, image_size=32, depth_ar=2, mode="train"):
        self.z_size, self.h_size, self.kl_min = int(z_size), int(h_size), float(kl_min)
        self.depth, self.num_blocks, self.k, self.image_size, self.mode = int(depth), int(num_blocks), int(k), int(image_size), mode
        self.depth_ar = int(depth_ar)
        if self.image_size % (2 ** self.depth):
            raise ValueError("image_size must be divisible by 2**depth (tf_train.py:183,192)")
        # tf_train.py:176-181: the first layer of every level but the first downsamples
        self.layers = [[IAFLayer(z_size, h_size, depth_ar=depth_ar, kl_min=kl_min, downsample=(i > 0 and j == 0), mode=mode)
                        for j in range(self.num_blocks)] for i in range(self.depth)]
        self._w_enc = self._w_dec = None
        self.params = None

    def load(self, params):
        """params: device fp32 tensors under the reference's variable names (tf_train.py:175-215): x_enc/{V [5,5,3,h],g,b},
        IAF_<i>_<j>/<IAFLayer names>, h_top [h], x_dec/{V [5,5,3,h],g,b}, dec_log_stdv []."""
        lib, hs = _capi.lib(), self.h_size
        for nm, shape in (("x_enc/V", (5, 5, 3, hs)), ("x_enc/g", (hs,)), ("x_enc/b", (hs,)), ("x_dec/V", (5, 5, 3, hs)),
                          ("x_dec/g", (3,)), ("x_dec/b", (3,)), ("h_top", (hs,))):
            _check_act(params[nm], nm, shape)
        self._w_enc = torch.empty_like(params["x_enc/V"])
        self._w_dec = torch.empty_like(params["x_dec/V"])
        _capi.check(lib.iaf_convk_weightnorm(_ptr(params["x_enc/V"]), _ptr(params["x_enc/g"]), _ptr(self._w_enc), 5, 5, 3, hs, 0,
                                             _stream()))                                                    # layers.py:56-60
        _capi.check(lib.iaf_convk_weightnorm(_ptr(params["x_dec/V"]), _ptr(params["x_dec/g"]), _ptr(self._w_dec), 5, 5, hs, 3, 1,
                                             _stream()))                                                    # layers.py:104-106
        self._lparams = {}
        for i, level in enumerate(self.layers):
            for j, layer in enumerate(level):
                pre = "IAF_%d_%d/" % (i, j)
                self._lparams[(i, j)] = {k[len(pre):]: v for k, v in params.items() if k.startswith(pre)}
                layer.load(self._lparams[(i, j)])
        self.params = params

    def trim_packs(self, B):
        """inference with batch B: every plain conv of every layer keeps only the weight pack its launch reads (IAFLayer.trim_packs); the
        prep launches of prepare_weights() then write 4-6 instead of 14 bytes per weight.  After load(); not for training."""
        kept = {}
        for i, level in enumerate(self.layers):
            for j, layer in enumerate(level):
                # input resolution of layer (i, j): level i runs at image_size / 2^(i+1); its first layer (downsample) is fed the finer level
                H = self.image_size // (2 ** (i + 1))
                Hin = 2 * H if layer.downsample else H
                kept[(i, j)] = layer.trim_packs(B, Hin, Hin)
        self.prepare_weights()
        return kept

    def prepare_weights(self):
        """Re-derive every weight norm from the loaded variables (what the reference's graph does inside every step, layers.py:56-60) in
        batched launches: all masked stacks in one, all plain convs in one, a downsampling layer's deconv and the two ends on their own.
        For a training loop that updates the variables in place (parallel.FlatParams): load(params) once, then prepare_weights() per step."""
        if self.params is None:
            raise RuntimeError("CVAE1.load(params) first")
        lib, p, hs = _capi.lib(), self.params, self.h_size
        if getattr(self, "_prep", None) is None:
            order = [(i, j) for i in range(self.depth) for j in range(self.num_blocks)]
            convs = [(ij, nm) for ij in order
                     for nm in ("up_conv1", "up_conv3", "down_conv1") + (() if self.layers[ij[0]][ij[1]].downsample else ("down_conv2",))]
            self._prep = dict(order=order, convs=convs, s=PrepBatch([self.layers[i][j].posterior.stack for i, j in order]),
                              c=ConvPrepBatch([getattr(self.layers[ij[0]][ij[1]], nm) for ij, nm in convs]))
        P = self._prep
        P["s"].run([IAFLayer.stack_params(self._lparams[ij]) for ij in P["order"]])
        P["c"].run([(self._lparams[ij][nm + "/V"], self._lparams[ij][nm + "/g"], self._lparams[ij][nm + "/b"]) for ij, nm in P["convs"]])
        for i, j in P["order"]:
            layer = self.layers[i][j]
            if layer.downsample:
                lp = self._lparams[(i, j)]
                layer.down_conv2.prepare_deconv(lp["down_deconv2/V"], lp["down_deconv2/g"], lp["down_deconv2/b"], force=True)
        _capi.check(lib.iaf_convk_weightnorm(_ptr(p["x_enc/V"]), _ptr(p["x_enc/g"]), _ptr(self._w_enc), 5, 5, 3, hs, 0, _stream()))
        _capi.check(lib.iaf_convk_weightnorm(_ptr(p["x_dec/V"]), _ptr(p["x_dec/g"]), _ptr(self._w_dec), 5, 5, hs, 3, 1, _stream()))

    def init_pass(self, x, params, noise):
        """The data-dependent initialisation pass: the reference builds CVAE1(hps, "init") first, i.e. _forward under
        arg_scope([conv2d, deconv2d], init=True) (tf_train.py:175) with every IAFLayer in mode "init" (z from the prior, :60-61).  Every conv
        -- x_enc, the plain and strided convs, the masked convs of ar_multiconv2d (conv2d is in the scope), the deconvs, x_dec -- takes its g
        and b from the moments of its own un-gained output on this batch (layers.py:38-51, 87-100) and hands the NORMALISED output on.
        params: the V of every conv (g / b entries are ignored), h_top, dec_log_stdv; noise as in forward() (the priors' draws are used).
        Returns (x_out, params_out) with params_out = params + every g and b; the model is left loaded with params_out."""
        lib, hs, zs = _capi.lib(), self.h_size, self.z_size
        B, S, n = self._check_inputs(x, noise, None)
        dev, st = x.device, _stream
        f32 = dict(dtype=torch.float32, device=dev)
        out = {k: v for k, v in params.items() if not (k.endswith("/g") or k.endswith("/b"))}

        def norm(raw, init_scale):                        # layers.py:46-51: moments over (N,H,W), g = log(scale)/3, b = -mean scale
            B_, C_, H_, W_ = (int(v) for v in raw.shape)
            y, g, b = torch.empty_like(raw), torch.empty(C_, **f32), torch.empty(C_, **f32)
            _capi.check(lib.iaf_datainit_normalize(_ptr(raw), None, _ptr(y), _ptr(g), _ptr(b), B_, C_, H_ * W_, float(init_scale), st()))
            return y, g, b

        def axpby(a, sa, b, sb):
            o = torch.empty_like(a)
            _capi.check(lib.iaf_axpby(_ptr(a), float(sa), _ptr(b), float(sb), _ptr(o), a.numel(), st()))
            return o

        def conv_init(conv, name, inp, kind="plain", x2=None):
            V = params[name + "/V"]
            z0, z1 = torch.zeros(conv.n_out, **f32), torch.zeros(conv.n_out, **f32)
            if kind == "deconv":
                conv.prepare_deconv(V, z0, z1, force=True)
                raw = conv.deconv(inp, x2=x2, elu_input=True)
            elif kind == "stride2":
                conv.prepare(V, z0, z1, force=True)
                raw = conv.stride2(inp, elu_input=True)[0]
            else:
                conv.prepare(V, z0, z1, force=True)
                raw = conv(inp, x2=x2, elu_input=True)[0]
            y, out[name + "/g"], out[name + "/b"] = norm(raw, 0.1)
            return y

        def pieces(t, sizes):
            r, off = [], 0
            for c in sizes:
                r.append(t.narrow(1, off, c).contiguous())
                off += c
            return r

        xf = torch.empty((n, 3, S, S), **f32)
        _capi.check(lib.iaf_image_to_float(x.data_ptr(), _ptr(xf), B, 3 * S * S, self.k, st()))
        w = torch.empty_like(params["x_enc/V"])
        zg, zb = torch.zeros(hs, **f32), torch.zeros(hs, **f32)
        _capi.check(lib.iaf_convk_weightnorm(_ptr(params["x_enc/V"]), _ptr(zg), _ptr(w), 5, 5, 3, hs, 0, st()))
        raw = torch.empty((n, hs, S // 2, S // 2), **f32)
        _capi.check(lib.iaf_convk_forward(_ptr(xf), _ptr(w), _ptr(zb), _ptr(raw), n, 3, S, S, hs, 5, 5, 2, 0, st()))
        h, out["x_enc/g"], out["x_enc/b"] = norm(raw, 0.1)
        ups = {}
        for i in range(self.depth):                                                                        # tf_train.py:29-44
            for j in range(self.num_blocks):
                layer, pre = self.layers[i][j], "IAF_%d_%d/" % (i, j)
                y1 = conv_init(layer.up_conv1, pre + "up_conv1", h, "stride2" if layer.downsample else "plain")
                qm, ql, uc, hh = pieces(y1, [zs, zs, hs, hs])
                y3 = conv_init(layer.up_conv3, pre + "up_conv3", hh)
                h = axpby(resample2(h, "down_even") if layer.downsample else h, 1.0, y3, 0.1)
                ups[(i, j)] = (qm, ql, uc)
        St = S // 2 ** self.depth
        h = torch.empty((n, hs, St, St), **f32)
        _capi.check(lib.iaf_tile_channels(_ptr(params["h_top"]), _ptr(h), n, hs, St * St, st()))
        li = 0
        for i in reversed(range(self.depth)):                                                              # tf_train.py:46-95, mode "init"
            for j in reversed(range(self.num_blocks)):
                layer, pre = self.layers[i][j], "IAF_%d_%d/" % (i, j)
                y = conv_init(layer.down_conv1, pre + "down_conv1", h)
                pz_mean, pz_logsd, rz_mean, rz_logsd, down_context, h_det = pieces(y, [zs] * 4 + [hs] * 2)
                z = torch.empty_like(pz_mean)
                _capi.check(lib.iaf_gaussian_sample_logsd(_ptr(pz_mean), _ptr(pz_logsd), _ptr(noise[2 * li]), _ptr(z), z.numel(), st()))   # :60-61
                qm, ql, uc = ups[(i, j)]
                context = axpby(uc, 1.0, down_context, 1.0)                                                # :58
                store = VariableStore()
                ar = pre + "ar_multiconv2d/"
                for k_, v in params.items():
                    if k_.startswith(ar) and k_.endswith("/V"):
                        store.set(k_, v)
                with variable_scope("IAF_%d_%d" % (i, j), store):
                    m_raw, s_raw = ar_multiconv2d("ar_multiconv2d", z, context, [hs] * self.depth_ar, [zs, zs], store=store, init=True)   # :69
                for k_, v in store.vars.items():
                    if k_.endswith("/g") or k_.endswith("/b"):
                        out[k_] = v
                z2 = torch.empty_like(z)
                _capi.check(lib.iaf_affine_transform(_ptr(z), _ptr(m_raw), _ptr(s_raw), 0.1, _ptr(z2), z.numel(), st()))                # :70-71
                if layer.downsample:
                    y = conv_init(layer.down_conv2, pre + "down_deconv2", z2, "deconv", x2=h_det)
                    h = axpby(resample2(h, "up_nearest"), 1.0, y, 0.1)
                else:
                    y = conv_init(layer.down_conv2, pre + "down_conv2", z2, x2=h_det)
                    h = axpby(h, 1.0, y, 0.1)                                                              # :94
                li += 1
        w = torch.empty_like(params["x_dec/V"])
        zg3, zb3 = torch.zeros(3, **f32), torch.zeros(3, **f32)
        _capi.check(lib.iaf_convk_weightnorm(_ptr(params["x_dec/V"]), _ptr(zg3), _ptr(w), 5, 5, hs, 3, 1, st()))
        raw = torch.empty((n, 3, S, S), **f32)
        _capi.check(lib.iaf_deconvk_forward(_ptr(h), _ptr(w), _ptr(zb3), _ptr(raw), n, hs, S // 2, S // 2, 3, 5, 5, 2, 1, 0.0, 0.0, st()))
        y, out["x_dec/g"], out["x_dec/b"] = norm(raw, 0.1)
        x_out = torch.empty_like(y)
        _capi.check(lib.iaf_clip(_ptr(y), -0.5 + 1 / 512., 0.5 - 1 / 512., _ptr(x_out), y.numel(), st()))                               # :208
        self.load(out)
        return x_out, out

    def iw_eval(self, x, noise_passes):
        """The k-sample importance-weighted bound of the whole model without materialising k samples at once (tf_train.py:168-170,218 with
        hps.k = len(noise_passes); BASELINE config 5 evaluates k = 10^4): one top-down pass with k = 1 per sample, the per-image terms
        log_pxz and sum-of-KL streamed into the running log-sum-exp (StreamingLowerBound).  The bottom-up pass -- image scaling, x_enc,
        every layer's up_conv1 / up_conv3 (tf_train.py:183-187) -- depends on x only: it runs ONCE and its products (qz_mean, qz_logsd,
        up_context of every layer, tf_train.py:38) are kept across the k passes.  noise_passes: one noise list (as for forward) per sample.
        Returns the loss [1] = sum over images of -log (1/k) sum_s exp(log_pxz_s - kl_s); bits_per_dim(loss, B) as usual."""
        if self.k != 1:
            raise ValueError("iw_eval streams the samples: build the model with k = 1")
        B = int(x.shape[0])
        acc = StreamingLowerBound(B, x.device)
        xf = None
        for noise in noise_passes:
            self._check_inputs(x, noise, self.k)      # every pass's noise list (host-only checks: the kernels take raw pointers)
            if xf is None:
                xf = self._bottom_up(x, noise)
            log_pxz, kl_cost = self._top_down(xf, B, noise, terms=True)
            acc.update(log_pxz.reshape(B, 1), kl_cost.reshape(B, 1))
        lb = acc.result()
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        _capi.check(_capi.lib().iaf_sum_axpy(_ptr(lb), None, 0.0, _ptr(loss), B, _stream()))
        return loss

    def _check_inputs(self, x, noise, k):
        """what forward(), forward_backward() and init_pass() require of (x, noise): the kernels behind them take raw pointers"""
        if self.params is None and k is not None:
            raise RuntimeError("CVAE1.load(params) first")
        if not torch.is_tensor(x) or x.dtype != torch.uint8 or not x.is_cuda or not x.is_contiguous() or x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("x must be a contiguous uint8 [B,3,S,S] device tensor")
        B, _, S, S2 = (int(v) for v in x.shape)
        if S != self.image_size or S2 != S:
            raise ValueError("image size %r, model built for %d" % (tuple(x.shape), self.image_size))
        if len(noise) != 2 * self.depth * self.num_blocks:
            raise ValueError("noise: %d tensors expected (prior, posterior per layer, top-down)" % (2 * self.depth * self.num_blocks))
        n = B * (self.k if k is None else k)
        li = 0
        for i in reversed(range(self.depth)):
            for _ in range(self.num_blocks):
                Sl = S // 2 ** (i + 1)
                for e in noise[2 * li:2 * li + 2]:
                    if e is None:
                        continue
                    if (not torch.is_tensor(e) or e.dtype != torch.float32 or not e.is_cuda or not e.is_contiguous()
                            or tuple(e.shape) != (n, self.z_size, Sl, Sl)):
                        raise ValueError("noise[%d..%d]: contiguous fp32 device tensors [%d,%d,%d,%d] (batch x k rows)"
                                         % (2 * li, 2 * li + 1, n, self.z_size, Sl, Sl))
                li += 1
        return B, S, n

    def forward(self, x, noise, _terms=False):
        """x: uint8 [B,3,S,S] on the device.  noise: per layer in top-down order the pair (eps_prior, eps_post) the reference's two
        DiagonalGaussians draw (distributions.py:15-24), flattened into one list -- eps_post is used in mode "train", eps_prior in
        "init" / "sample".  Returns (x_out [B k,3,S,S], obj [1], loss [1]) as tf_train.py:218."""
        xf = self._bottom_up(x, noise)
        return self._top_down(xf, int(x.shape[0]), noise, terms=_terms)

    def _bottom_up(self, x, noise):
        """tf_train.py:153-187: image scaling (+ repeat k), x_enc, the up pass of every layer (which leaves qz_mean, qz_logsd, up_context in
        the layers, :38).  Returns the scaled image the likelihood compares with."""
        B, S, n = self._check_inputs(x, noise, self.k)
        lib, p, hs, k = _capi.lib(), self.params, self.h_size, self.k
        dev = x.device
        xf = torch.empty((n, 3, S, S), dtype=torch.float32, device=dev)
        _capi.check(lib.iaf_image_to_float(x.data_ptr(), _ptr(xf), B, 3 * S * S, k, _stream()))           # tf_train.py:153-159
        h = torch.empty((n, hs, S // 2, S // 2), dtype=torch.float32, device=dev)
        _capi.check(lib.iaf_convk_forward(_ptr(xf), _ptr(self._w_enc), _ptr(p["x_enc/b"]), _ptr(h), n, 3, S, S, hs, 5, 5, 2, 0,
                                          _stream()))                                                       # :183
        for level in self.layers:                                                                          # :184-187
            for layer in level:
                h = layer.up(h)
        return xf

    def _top_down(self, xf, B, noise, terms=False):
        """tf_train.py:189-218 on the state _bottom_up left: h_top, the down pass, x_dec, the likelihood, obj and loss"""
        lib, p, hs, k = _capi.lib(), self.params, self.h_size, self.k
        n, S, dev = int(xf.shape[0]), int(xf.shape[2]), xf.device
        St = S // 2 ** self.depth
        h = torch.empty((n, hs, St, St), dtype=torch.float32, device=dev)
        _capi.check(lib.iaf_tile_channels(_ptr(p["h_top"]), _ptr(h), n, hs, St * St, _stream()))           # :189-192
        nl = self.depth * self.num_blocks
        objs, costs = [], []
        li = 0
        for level in reversed(self.layers):                                                                # :195-200
            for layer in reversed(level):
                eps_prior, eps_post = noise[2 * li], noise[2 * li + 1]
                h, cur_obj, cur_cost = layer.down(h, eps_post, eps_prior=eps_prior)
                objs.append(cur_obj)
                costs.append(cur_cost)
                li += 1
        objs, costs = torch.stack(objs), torch.stack(costs)          # [nl, n] (one gathering copy each; the sums are iaf_colsum's)
        kl_obj = torch.empty(n, dtype=torch.float32, device=dev)
        kl_cost = torch.empty(n, dtype=torch.float32, device=dev)
        _capi.check(lib.iaf_colsum(_ptr(objs), _ptr(kl_obj), nl, n, _stream()))
        _capi.check(lib.iaf_colsum(_ptr(costs), _ptr(kl_cost), nl, n, _stream()))
        x_out = torch.empty((n, 3, S, S), dtype=torch.float32, device=dev)
        _capi.check(lib.iaf_deconvk_forward(_ptr(h), _ptr(self._w_dec), _ptr(p["x_dec/b"]), _ptr(x_out), n, hs, S // 2, S // 2, 3, 5, 5,
                                            2, 1, -0.5 + 1 / 512., 0.5 - 1 / 512., _stream()))             # :206-208
        log_pxz = discretized_logistic(x_out, p["dec_log_stdv"], sample=xf)                                # :210
        if terms:
            return log_pxz, kl_cost
        obj = torch.empty(1, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        _capi.check(lib.iaf_sum_axpy(_ptr(kl_obj), _ptr(log_pxz), -1.0, _ptr(obj), n, _stream()))          # :211
        lb = compute_lowerbound(log_pxz, kl_cost, k)                                                       # :218
        _capi.check(lib.iaf_sum_axpy(_ptr(lb), None, 0.0, _ptr(loss), B, _stream()))
        return x_out, obj, loss

    # -- training: d obj / d every variable, what opt.compute_gradients(obj) hands the optimizer (tf_train.py:128, 211) -----------------
    def set_training(self, on=True):
        """allocate what the backward needs in every layer (transposed packs); call load(params) afterwards"""
        for level in self.layers:
            for layer in level:
                layer.set_training(on)
        for sg in (getattr(self, "_segs", None) or []):              # back to per-op weight-norm backward for anyone using the layers directly
            lib = _capi.lib()
            for st in sg["wn"].stacks:
                _capi.check(lib.iaf_stack_set_defer_weightnorm(st._h, 0))
            for cv in sg["wn"].convs:
                _capi.check(lib.iaf_conv3x3_set_defer_weightnorm(cv._h, 0))
        self._training = bool(on)
        self.params = None
        self._prep = None
        # mask + weight-norm backward of the stacks and plain convs of a gradient bucket in ONE launch per kind at the end of the bucket's
        # backward segment (a conv's own pass is 10-28 workgroups on a 256-CU chip; set_grad_buckets, default: one bucket = the whole model);
        # a downsampling layer's deconv differentiates its own norm inside its backward
        self._segs = None

    # The gradient exchange of data-parallel training (tf_train.py:124-147, tf_utils/common.py:83-86) overlaps the backward when the
    # flat gradient buffer is laid out in the order the backward COMPLETES the gradients and cut into contiguous buckets
    # (parallel.OverlappedGradReduce).  That order, for this model: the top end (dec_log_stdv, x_dec), then the top-down pass backwards
    # = layer by layer in up-pass order (each layer's last conv, its ar_multiconv2d stack, down_conv1), h_top, then the bottom-up pass
    # backwards (up_conv3, up_conv1 of every layer in reverse), x_enc.
    _DOWN_KEYS = ("down_conv2/", "down_deconv2/", "ar_multiconv2d/", "down_conv1/")

    @staticmethod
    def _names_of_layer(all_names, ij, down):
        pre = "IAF_%d_%d/" % ij
        return [k for k in all_names if k.startswith(pre) and (k[len(pre):].startswith(CVAE1._DOWN_KEYS) == down)]

    def _layer_names(self, ij, down):
        return CVAE1._names_of_layer(self.params if self.params is not None else {}, ij, down)

    @staticmethod
    def grad_bucket_names(all_names, depth, num_blocks, n_buckets=1):
        """Pure bookkeeping (no device): the variable names of a depth x num_blocks model, bucket by bucket, in the order forward_backward
        completes their gradients -- ceil(n/2) groups of layers for the top-down pass's backward (bucket 0 also holds the top end), floor(n/2)
        for the bottom-up pass's (the first with h_top, the last with x_enc).  Concatenated = completion_order()."""
        order = [(i, j) for i in range(depth) for j in range(num_blocks)]
        nb = max(1, min(int(n_buckets), 2 * len(order)))
        chunks = lambda lst, n: [lst[q * len(lst) // n:(q + 1) * len(lst) // n] for q in range(n)]
        top, bottom = ["dec_log_stdv", "x_dec/V", "x_dec/g", "x_dec/b"], ["x_enc/V", "x_enc/g", "x_enc/b"]
        names_of = lambda g, down: [k for ij in g for k in CVAE1._names_of_layer(all_names, ij, down)]
        if nb < 2:
            out = [top + names_of(order, True) + ["h_top"] + names_of(list(reversed(order)), False) + bottom]
        else:
            dgroups, ugroups = chunks(order, (nb + 1) // 2), chunks(list(reversed(order)), nb // 2)
            out = [(top if gi == 0 else []) + names_of(g, True) for gi, g in enumerate(dgroups)]
            out += [(["h_top"] if gi == 0 else []) + names_of(g, False) + (bottom if gi == len(ugroups) - 1 else []) for gi, g in enumerate(ugroups)]
        flat = [k for b in out for k in b]
        if sorted(flat) != sorted(all_names):
            raise ValueError("not the variables of a %d x %d CVAE1: %s" % (depth, num_blocks, sorted(set(flat) ^ set(all_names))[:6]))
        return out

    def completion_order(self):
        """every variable name in the order forward_backward completes its gradient (lay parallel.FlatParams out in this order)"""
        if self.params is None:
            raise RuntimeError("CVAE1.load(params) first")
        return [k for b in CVAE1.grad_bucket_names(list(self.params), self.depth, self.num_blocks, 1) for k in b]

    def set_grad_buckets(self, n_buckets=1):
        """Cut the backward into `n_buckets` segments, each completing one contiguous run of completion_order(): ceil(n/2) groups of layers
        for the top-down pass's backward, floor(n/2) for the bottom-up pass's; every segment ends with the deferred mask + weight-norm
        backward of exactly its own convs (one launch per kind).  Returns the variable names per bucket (for
        OverlappedGradReduce.bounds_from_groups).  forward_backward(on_bucket=) then reports each bucket the moment it is complete."""
        if not getattr(self, "_training", False) or self.params is None:
            raise RuntimeError("CVAE1.set_training(True), then load(params), before set_grad_buckets")
        order = [(i, j) for i in range(self.depth) for j in range(self.num_blocks)]
        nb = max(1, min(int(n_buckets), 2 * len(order)))
        chunks = lambda lst, n: [lst[q * len(lst) // n:(q + 1) * len(lst) // n] for q in range(n)]
        dgroups = chunks(order, (nb + 1) // 2) if nb >= 2 else [order]
        ugroups = chunks(list(reversed(order)), nb // 2) if nb >= 2 else []
        L = lambda ij: self.layers[ij[0]][ij[1]]
        dconvs = lambda ij: [(ij, "down_conv1")] + ([] if L(ij).downsample else [(ij, "down_conv2")])
        uconvs = lambda ij: [(ij, "up_conv3"), (ij, "up_conv1")]
        segs = []
        for gi, g in enumerate(dgroups):
            convs = [c for ij in g for c in dconvs(ij)]
            up_too = nb < 2                                          # a single bucket: everything completes at the very end
            if up_too:
                convs += [c for ij in reversed(order) for c in uconvs(ij)]
            names = (["dec_log_stdv", "x_dec/V", "x_dec/g", "x_dec/b"] if gi == 0 else []) + [k for ij in g for k in self._layer_names(ij, True)]
            if up_too:
                names += ["h_top"] + [k for ij in reversed(order) for k in self._layer_names(ij, False)] + ["x_enc/V", "x_enc/g", "x_enc/b"]
            segs.append(dict(down=g, up=list(reversed(order)) if up_too else [], h_top=up_too, x_enc=up_too, names=names, conv_ids=convs,
                             stack_ids=list(g),
                             wn=WnBwdBatch(stacks=[L(ij).posterior.stack for ij in g], convs=[getattr(L(ij), nm) for ij, nm in convs])))
        for gi, g in enumerate(ugroups):
            convs = [c for ij in g for c in uconvs(ij)]
            last = gi == len(ugroups) - 1
            names = (["h_top"] if gi == 0 else []) + [k for ij in g for k in self._layer_names(ij, False)] + \
                    (["x_enc/V", "x_enc/g", "x_enc/b"] if last else [])
            segs.append(dict(down=[], up=g, h_top=(gi == 0), x_enc=last, names=names, conv_ids=convs, stack_ids=[],
                             wn=WnBwdBatch(stacks=[], convs=[getattr(L(ij), nm) for ij, nm in convs])))
        names = CVAE1.grad_bucket_names(list(self.params), self.depth, self.num_blocks, nb)
        assert len(names) == len(segs) and all(sorted(sg["names"]) == sorted(nm) for sg, nm in zip(segs, names))
        for sg, nm in zip(segs, names):
            sg["names"] = nm
        self._segs = segs
        return names

    def fb_begin(self, x, noise, grads=None, autotune=False):
        """forward_backward, first part: the forward pass (keeping what the backward reads) and the backward of the top end --
        obj = sum(kl_obj - log_pxz), likelihood, clip, x_dec (tf_train.py:206-211).  Then fb_segment(0 .. n_buckets-1)."""
        if not getattr(self, "_training", False) or self.params is None:
            raise RuntimeError("CVAE1.set_training(True), then load(params), before forward_backward")
        if self.k != 1 or self.mode != "train":
            raise ValueError("forward_backward: mode 'train', k = 1 (the training objective, tf_train.py:211)")
        if getattr(self, "_segs", None) is None:
            self.set_grad_buckets(1)
        B, S, n = self._check_inputs(x, noise, 1)
        lib, p, hs = _capi.lib(), self.params, self.h_size
        dev, st = x.device, _stream
        f32 = dict(dtype=torch.float32, device=dev)
        # ---- forward, keeping what the backward reads
        xf = torch.empty((n, 3, S, S), **f32)
        _capi.check(lib.iaf_image_to_float(x.data_ptr(), _ptr(xf), B, 3 * S * S, 1, st()))
        h = torch.empty((n, hs, S // 2, S // 2), **f32)
        _capi.check(lib.iaf_convk_forward(_ptr(xf), _ptr(self._w_enc), _ptr(p["x_enc/b"]), _ptr(h), n, 3, S, S, hs, 5, 5, 2, 0, st()))
        for level in self.layers:
            for layer in level:
                h = layer.up_train(h, autotune=autotune)
        St = S // 2 ** self.depth
        h = torch.empty((n, hs, St, St), **f32)
        _capi.check(lib.iaf_tile_channels(_ptr(p["h_top"]), _ptr(h), n, hs, St * St, st()))
        objs, li = [], 0
        for level in reversed(self.layers):
            for layer in reversed(level):
                h, cur_obj, _ = layer.down_train(h, noise[2 * li + 1], autotune=autotune)
                objs.append(cur_obj)
                li += 1
        h_last = h
        objs = torch.stack(objs)
        kl_obj = torch.empty(n, **f32)
        _capi.check(lib.iaf_colsum(_ptr(objs), _ptr(kl_obj), li, n, st()))
        lo, hi = -0.5 + 1 / 512., 0.5 - 1 / 512.
        x_out = torch.empty((n, 3, S, S), **f32)
        _capi.check(lib.iaf_deconvk_forward(_ptr(h_last), _ptr(self._w_dec), _ptr(p["x_dec/b"]), _ptr(x_out), n, hs, S // 2, S // 2, 3, 5, 5,
                                            2, 1, lo, hi, st()))
        log_pxz = discretized_logistic(x_out, p["dec_log_stdv"], sample=xf)
        obj = torch.empty(1, **f32)
        _capi.check(lib.iaf_sum_axpy(_ptr(kl_obj), _ptr(log_pxz), -1.0, _ptr(obj), n, st()))
        # ---- backward of the top end: obj = sum(kl_obj - log_pxz)  (tf_train.py:206-211)
        grads = {} if grads is None else grads
        gs = lambda nm: grads.setdefault(nm, torch.empty_like(p[nm]))
        d_xout = torch.empty_like(x_out)
        dls_rows = torch.empty(n, **f32)
        logscale = p["dec_log_stdv"].reshape(1).contiguous()
        _capi.check(lib.iaf_discretized_logistic_backward(_ptr(x_out), _ptr(logscale), _ptr(xf), -1.0, lo, hi, _ptr(d_xout), _ptr(dls_rows), n,
                                                          3 * S * S, 1 / 256.0, st()))
        _capi.check(lib.iaf_sum_axpy(_ptr(dls_rows), None, 0.0, _ptr(gs("dec_log_stdv")), n, st()))
        dW = torch.empty_like(p["x_dec/V"])
        _capi.check(lib.iaf_convk_wgrad(_ptr(d_xout), _ptr(h_last), _ptr(dW), n, 3, S, S, hs, 5, 5, 2, 0, 1, st()))
        gs("x_dec/V"), gs("x_dec/g"), gs("x_dec/b")
        scratch = torch.empty(hs * 3, **f32)
        _capi.check(lib.iaf_convk_weightnorm_backward(_ptr(p["x_dec/V"]), _ptr(p["x_dec/g"]), _ptr(dW), _ptr(grads["x_dec/V"]),
                                                      _ptr(grads["x_dec/g"]), _ptr(scratch), 5, 5, hs, 3, 1, st()))
        _capi.check(lib.iaf_channel_sum(_ptr(d_xout), _ptr(grads["x_dec/b"]), n, 3, S * S, st()))
        # d h_last = elu'(h_last) * (the strided conv of d x_out with x_dec's filter: the adjoint of the transposed conv)
        zero_b = torch.zeros(hs, **f32)
        t = torch.empty_like(h_last)
        _capi.check(lib.iaf_convk_forward(_ptr(d_xout), _ptr(self._w_dec), _ptr(zero_b), _ptr(t), n, 3, S, S, hs, 5, 5, 2, 0, st()))
        d = torch.empty_like(h_last)
        _capi.check(lib.iaf_mul_elu_grad(_ptr(t), _ptr(h_last), _ptr(d), t.numel(), st()))
        lgrads = {}
        for i, level in enumerate(self.layers):
            for j, layer in enumerate(level):
                pre = "IAF_%d_%d/" % (i, j)
                lgrads[(i, j)] = {k[len(pre):]: v for k, v in grads.items() if k.startswith(pre)}
        self._fb = dict(x_out=x_out, obj=obj, grads=grads, lgrads=lgrads, d=d, xf=xf, n=n, S=S, St=St, autotune=autotune,
                        dko=torch.ones(n, **f32), keep=(dW, scratch, zero_b, t, d_xout, dls_rows, logscale, log_pxz, kl_obj, objs))
        return self._fb

    def fb_segment(self, si):
        """forward_backward, segment si (in order, after fb_begin): the backward of its layers and the deferred mask + weight-norm backward
        of its convs -- on return every gradient of bucket si is complete (enqueued on the current stream)."""
        lib, p, hs, F, sg = _capi.lib(), self.params, self.h_size, self._fb, self._segs[si]
        grads, lgrads, n, S, St, autotune, st = F["grads"], F["lgrads"], F["n"], F["S"], F["St"], F["autotune"], _stream
        gs = lambda nm: grads.setdefault(nm, torch.empty_like(p[nm]))
        # ---- the layer stack: the top-down pass backwards (= in up-pass order), then the bottom-up pass backwards
        for (i, j) in sg["down"]:
            F["d"] = self.layers[i][j].down_backward(F["d"], F["dko"], self._lparams[(i, j)], lgrads[(i, j)], autotune=autotune)
        if sg["h_top"]:
            _capi.check(lib.iaf_channel_sum(_ptr(F["d"]), _ptr(gs("h_top")), n, hs, St * St, st()))        # adjoint of the tile (:190-192)
            F["d"] = torch.zeros((n, hs, St, St), dtype=torch.float32, device=F["d"].device)   # the up pass's last output is not used (h_top replaces it)
        for (i, j) in sg["up"]:
            F["d"] = self.layers[i][j].up_backward(F["d"], self._lparams[(i, j)], lgrads[(i, j)], autotune=autotune)
        tup = lambda dct, nm: (dct[nm + "/V"], dct[nm + "/g"], dct[nm + "/b"])
        sg["wn"].run(stack_params=[IAFLayer.stack_params(self._lparams[ij]) for ij in sg["stack_ids"]],
                     stack_grads=[IAFLayer.stack_params(lgrads[ij]) for ij in sg["stack_ids"]],
                     conv_params=[tup(self._lparams[ij], nm) for ij, nm in sg["conv_ids"]],
                     conv_grads=[tup(lgrads[ij], nm) for ij, nm in sg["conv_ids"]])
        for (i, j) in set(sg["down"]) | set(sg["up"]):
            for k, v in lgrads[(i, j)].items():
                grads["IAF_%d_%d/%s" % (i, j, k)] = v
        if sg["x_enc"]:                                                                                    # :183
            d = F["d"]
            dW = torch.empty_like(p["x_enc/V"])
            _capi.check(lib.iaf_convk_wgrad(_ptr(F["xf"]), _ptr(d), _ptr(dW), n, 3, S, S, hs, 5, 5, 2, 0, 0, st()))
            gs("x_enc/V"), gs("x_enc/g"), gs("x_enc/b")
            _capi.check(lib.iaf_convk_weightnorm_backward(_ptr(p["x_enc/V"]), _ptr(p["x_enc/g"]), _ptr(dW), _ptr(grads["x_enc/V"]),
                                                          _ptr(grads["x_enc/g"]), None, 5, 5, 3, hs, 0, st()))
            _capi.check(lib.iaf_channel_sum(_ptr(d), _ptr(grads["x_enc/b"]), n, hs, (S // 2) ** 2, st()))
            F["keep"] = F["keep"] + (dW,)

    def forward_backward(self, x, noise, grads=None, autotune=False, on_bucket=None):
        """One tower's forward and backward in mode "train", k = 1: returns (x_out, obj [1], grads) with grads[name] = d obj / d params[name]
        for every variable (written into the tensors of `grads` where it has them -- e.g. the views of parallel.FlatParams.g).
        autotune=True: the first call at a new batch size searches the launch shapes of the plain convs and their data gradients (what
        cuDNN's algorithm search does for the reference).  The layer stack's backward is IAFLayer.down_backward / .up_backward chained through the model (the down pass
        in up-pass order, then the up pass in reverse); the two ends -- likelihood, clip, x_dec, h_top, x_enc -- are the launches of
        csrc/iaf_model_edge.hpp.  noise as in forward().  on_bucket(i): called as soon as the gradients of bucket i (set_grad_buckets) are
        complete on the current stream -- where a data-parallel step issues that bucket's all-reduce (OverlappedGradReduce.reduce)."""
        F = self.fb_begin(x, noise, grads, autotune)
        for si in range(len(self._segs)):
            self.fb_segment(si)
            if on_bucket is not None:
                on_bucket(si)
        return F["x_out"], F["obj"], F["grads"]

    def exchange_errors(self):
        """bounded waits of the hand-overs inside the one-launch IAF steps that gave up, over all layers (0 = never; it synchronises the
        device).  A NaN loss with a non-zero count means the exchange, not the numerics: re-arm with ARStack.set_halo_exchange(True) and,
        if the step is replayed from a hipGraph, capture it again (include/iaf_hip.h)."""
        return sum(layer.posterior.stack.exchange_errors() for level in self.layers for layer in level)

    def bits_per_dim(self, loss, batch_size):
        """tf_train.py:133 for one tower: loss / (log 2 * num_pixels * batch_size)"""
        return float(loss) / (math.log(2.) * 3 * self.image_size ** 2 * batch_size)
