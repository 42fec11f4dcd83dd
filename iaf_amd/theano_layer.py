"""The IAF part of the reference's Theano layer, models.cvae_layer (models.py:14-345), for the two posteriors BASELINE names:
'down_iaf2_nl' (configs 0-2, 4) and 'up_iaf2_nl' (config 3), prior 'diag'.

Boundary: the plain Theano convs around the IAF step (up_conv1/2, down_conv1/2: graphy/nodes/conv.py) are out
of scope; this wrapper takes THEIR OUTPUTS in the reference's channel order and returns what the next conv consumes:

    up_conv1 output    [h_det (n_h) | qz_mean (n_z) | qz_logsd (n_z) | context (n_h)]            models.py:139-143, 164/181
    down_conv1 output  [h_det (n_h) | pz_mean (n_z) | pz_logsd (n_z) || rz_mean | rz_logsd | down_context (n_h)]  :273-279, 296-297
    next conv's input  concat([h_det, z])  (the TF path concatenates [z, h_det])                  :180, 318

The masked-AR convs run on the GPU through the Theano statement of the operator (IAF_VARIANT_THEANO[_FLIPMASK]); the free
bits follow models.py:454-466 (a SCALAR per layer: sum over channels of max(kl_min, batch-mean of the per-channel KL))."""
import ctypes

import torch

from . import _capi
from .distributions import gaussian_diag_logps_logsd
from .iaf_layer import gaussian_sample
from .layers import ARStack, kl_free_bits, split, _ptr, _stream


class CVAELayerIAF(object):
    def __init__(self, name, n_h, n_z, depth_ar, posterior="down_iaf2_nl", flipmask=False, kl_min=0.0):
        if posterior not in ("down_iaf2_nl", "up_iaf2_nl"):
            raise Exception("Unknown posterior " + posterior)                     # models.py:115
        self.name, self.n_h, self.n_z, self.depth_ar = name, int(n_h), int(n_z), int(depth_ar)
        self.posterior, self.kl_min = posterior, float(kl_min)
        self.stack = ARStack(self.n_z, [self.n_h] * self.depth_ar,
                             variant=_capi.IAF_VARIANT_THEANO_FLIPMASK if flipmask else _capi.IAF_VARIANT_THEANO)
        self._st = None
        self.training = False
        self._rel = None

    def set_training(self, on=True):
        """keep what backward() needs in up() / down_q() (call load() again afterwards: the data-gradient weight packs are
        written by the next prepare)"""
        self.stack.set_training(on)
        self.training = bool(on)

    def load(self, w):
        """w: the reference's parameter dict; reads w[name + '_posterior_conv1_<i>_w|_b|_s'] and '..._out_<i>_...'"""
        pre = self.name + "_posterior_conv1_"
        self._rel = {k[len(pre):]: v for k, v in w.items() if k.startswith(pre)}
        self.stack.prepare(self._rel)

    def up(self, h, eps=None):
        """h: output of up_conv1 [B, 2 n_h + 2 n_z, H, W].  Returns the input of up_conv2 (before its nonlinearity):
        h_det for 'down_iaf2_nl' (models.py:180-182), concat([h_det, z]) for 'up_iaf2_nl' (:168-176; needs eps)."""
        n_h, n_z = self.n_h, self.n_z
        h_det, qz_mean, qz_logsd, context = split(h, 1, [n_h, n_z, n_z, n_h])
        st = dict(qz_mean=qz_mean, qz_logsd=qz_logsd, context=context)
        if self.posterior == "up_iaf2_nl":
            if eps is None:
                raise ValueError("'up_iaf2_nl' samples the posterior in the up pass: pass eps")
            z0 = gaussian_sample(qz_mean, qz_logsd, eps)                                          # rand.py:81-83
            logq0 = gaussian_diag_logps_logsd(qz_mean, qz_logsd, z0)                              # rand.py:85-86
            z, logdet = (self.stack.iaf_step_train if self.training else self.stack.iaf_step)(z0, context)   # models.py:170-173
            st.update(z=z, logq0=logq0, logdet=logdet, z0=z0, eps=eps)
            out = torch.cat([h_det, z], dim=1)                                                    # :176
        else:
            out = h_det
        self._st = st
        return out

    def down_q(self, h, eps=None):
        """h: output of down_conv1 [B, 2 n_h + 4 n_z, H, W] ('down_iaf2_nl') or [B, n_h + 2 n_z, H, W] ('up_iaf2_nl').
        Returns dict(h = concat([h_det, z]) for down_conv2, kl [B, n_z, H, W] = logqs - logps (:328), kl_sum [B] (:455),
        obj_kl (:458-466: scalar tensor when kl_min > 0, else kl_sum))."""
        n_h, n_z, st = self.n_h, self.n_z, self._st
        if self.posterior == "down_iaf2_nl":
            h_det, pz_mean, pz_logsd, rz_mean, rz_logsd, down_context = split(h, 1, [n_h, n_z, n_z, n_z, n_z, n_h])
            if self.training:
                blk = self.stack.posterior_block_train(st["qz_mean"], st["qz_logsd"], rz_mean, rz_logsd, pz_mean, pz_logsd,
                                                       st["context"], down_context, eps, self.kl_min)
                st.update(rz_mean=rz_mean, rz_logsd=rz_logsd, pz_mean=pz_mean, pz_logsd=pz_logsd, eps=eps, z=blk["z"])
                obj_kl = blk["kl_obj"][0] if self.kl_min > 0 else blk["kl_cost"]
                return dict(h=torch.cat([h_det, blk["z"]], dim=1), z=blk["z"], kl=None, kl_sum=blk["kl_cost"], obj_kl=obj_kl)
            blk = self.stack.posterior_block(st["qz_mean"], st["qz_logsd"], rz_mean, rz_logsd, pz_mean, pz_logsd, st["context"],
                                             down_context, eps, self.kl_min, want_kl_elem=True)   # :272-285, 296-298
            z, kl, kl_sum, kl_obj = blk["z"], blk["kl_elem"], blk["kl_cost"], blk["kl_obj"]
        else:
            h_det, pz_mean, pz_logsd = split(h, 1, [n_h, n_z, n_z])
            z = st["z"]                                                                           # :216-217
            logp = gaussian_diag_logps_logsd(pz_mean, pz_logsd, z)                                # :298
            kl = torch.empty_like(z)
            _capi.check(_capi.lib().iaf_kl_combine(_ptr(st["logq0"]), _ptr(st["logdet"]), _ptr(logp), _ptr(kl), kl.numel(),
                                                   _stream()))
            if self.training and self.kl_min > 0:        # the gate of the free bits is what backward() multiplies with
                kl_sum, kl_obj, gate = kl_free_bits(kl, self.kl_min, want_gate=True)
                st["gate"] = gate
            else:
                kl_sum, kl_obj = kl_free_bits(kl, self.kl_min)
            st.update(pz_mean=pz_mean, pz_logsd=pz_logsd, kl=kl)
        # TF: kl_obj[b] = sum_c max(mean_b sum_hw kl, kl_min) for every b (tf_train.py:79-82); Theano adds that SCALAR
        # once per layer (models.py:460-461)
        obj_kl = kl_obj[0] if self.kl_min > 0 else kl_sum
        return dict(h=torch.cat([h_det, z], dim=1), z=z, kl=kl, kl_sum=kl_sum, obj_kl=obj_kl)


    def backward(self, d_h, d_obj=None, d_up=None):
        """T.grad (graphy/misc/optim.py:102) of  <d_up, up()> + <d_h, down_q()['h']> + <d_obj, obj_kl>  through the lines
        models.py:139-146,168-176,272-298,454-466.  Must follow up() and down_q() in training mode on the same inputs.
        d_h: gradient of concat([h_det, z]) [B, n_h + n_z, H, W];  d_up: gradient of what up() returned (None = zeros);
        d_obj: gradient of obj_kl (a scalar with free bits, [B] without; None = ones).  Returns dict(d_up_conv1, d_down_conv1
        -- gradients of the two conv outputs in the reference's channel order -- and grads keyed like the reference's w)."""
        if not self.training or self._st is None or "z" not in self._st:
            raise RuntimeError("backward() follows up() and down_q() in training mode")
        n_h, n_z, st = self.n_h, self.n_z, self._st
        z = st["z"]
        B = z.shape[0]
        d_hdet, dz = d_h[:, :n_h], d_h[:, n_h:].contiguous()
        if self.kl_min > 0:      # obj_kl is ONE scalar per layer (models.py:460-461): d obj / d kl[b,c,:,:] = gate[c] / B
            dko = torch.zeros(B, dtype=torch.float32, device=z.device)
            dko[0] = 1.0 if d_obj is None else float(d_obj)
        else:
            dko = torch.ones(B, dtype=torch.float32, device=z.device) if d_obj is None else d_obj.contiguous()
        pre = self.name + "_posterior_conv1_"
        if self.posterior == "down_iaf2_nl":
            bw = self.stack.posterior_block_backward(st["qz_mean"], st["qz_logsd"], st["rz_mean"], st["rz_logsd"], st["pz_mean"],
                                                     st["pz_logsd"], st["eps"], self.kl_min, z, dz, dko, self._rel)
            d_up_hdet = d_up if d_up is not None else torch.zeros_like(d_hdet)
            d_uc1 = torch.cat([d_up_hdet, bw["dmean"], bw["dlogsd"], bw["dcontext"]], dim=1)                     # :141-143
            d_dc1 = torch.cat([d_hdet, bw["dpz_mean"], bw["dpz_logsd"], bw["dmean"], bw["dlogsd"], bw["dcontext"]], dim=1)
            grads = bw["grads"]
        else:
            # kl = logq0 + logdet - logp(z): gate the per-element gradient G as the free bits prescribe, push it through
            # logp (prior side and z), the IAF step, and the reparametrised sample z0 = qz_mean + exp(qz_logsd) eps -- two
            # elementwise launches around iaf_step_backward (iaf_up_iaf2_backward_pre / _post); tensors are storage only
            H, W = int(z.shape[2]), int(z.shape[3])
            d_h = d_h.contiguous()
            d_up_c = d_up.contiguous() if d_up is not None else None
            dz_tot, G = torch.empty_like(z), torch.empty_like(z)
            d_dc1 = torch.empty(B, n_h + 2 * n_z, H, W, dtype=torch.float32, device=z.device)
            free_bits = self.kl_min > 0
            if free_bits and "gate" not in st:
                raise RuntimeError("backward() with free bits follows down_q() in training mode")
            gscale = ((1.0 if d_obj is None else float(d_obj)) / B) if free_bits else 0.0
            _capi.check(_capi.lib().iaf_up_iaf2_backward_pre(
                _ptr(z), _ptr(st["pz_mean"]), _ptr(st["pz_logsd"]), _ptr(d_h), _ptr(d_up_c), _ptr(st["gate"]) if free_bits else None,
                gscale, None if free_bits else _ptr(dko), _ptr(dz_tot), _ptr(G), _ptr(d_dc1), B, n_h, n_z, H * W, _stream()))
            dz0, dctx, grads = self.stack.iaf_step_backward(st["z0"], st["context"], z, st["logdet"], dz_tot, G, self._rel)
            d_uc1 = torch.empty(B, 2 * n_h + 2 * n_z, H, W, dtype=torch.float32, device=z.device)
            _capi.check(_capi.lib().iaf_up_iaf2_backward_post(
                _ptr(dz0), _ptr(st["z0"]), _ptr(st["qz_mean"]), _ptr(G), _ptr(dctx), _ptr(d_up_c), _ptr(d_uc1), B, n_h, n_z, H * W,
                _stream()))
        return dict(d_up_conv1=d_uc1, d_down_conv1=d_dc1, grads={pre + k: v for k, v in grads.items()})


