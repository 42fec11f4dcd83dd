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
from .distributions import gaussian_diag_logps
from .iaf_layer import gaussian_sample
from .layers import ARStack, split, _ptr, _stream


class CVAELayerIAF(object):
    def __init__(self, name, n_h, n_z, depth_ar, posterior="down_iaf2_nl", flipmask=False, kl_min=0.0):
        if posterior not in ("down_iaf2_nl", "up_iaf2_nl"):
            raise Exception("Unknown posterior " + posterior)                     # models.py:115
        self.name, self.n_h, self.n_z, self.depth_ar = name, int(n_h), int(n_z), int(depth_ar)
        self.posterior, self.kl_min = posterior, float(kl_min)
        self.stack = ARStack(self.n_z, [self.n_h] * self.depth_ar,
                             variant=_capi.IAF_VARIANT_THEANO_FLIPMASK if flipmask else _capi.IAF_VARIANT_THEANO)
        self._st = None

    def load(self, w):
        """w: the reference's parameter dict; reads w[name + '_posterior_conv1_<i>_w|_b|_s'] and '..._out_<i>_...'"""
        pre = self.name + "_posterior_conv1_"
        self.stack.prepare({k[len(pre):]: v for k, v in w.items() if k.startswith(pre)})

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
            logq0 = gaussian_diag_logps(qz_mean, qz_logsd * 2.0, z0)                              # rand.py:85-86
            z, logdet = self.stack.iaf_step(z0, context)                                          # models.py:170-173
            st.update(z=z, logq0=logq0, logdet=logdet)
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
            blk = self.stack.posterior_block(st["qz_mean"], st["qz_logsd"], rz_mean, rz_logsd, pz_mean, pz_logsd, st["context"],
                                             down_context, eps, self.kl_min, want_kl_elem=True)   # :272-285, 296-298
            z, kl, kl_sum, kl_obj = blk["z"], blk["kl_elem"], blk["kl_cost"], blk["kl_obj"]
        else:
            h_det, pz_mean, pz_logsd = split(h, 1, [n_h, n_z, n_z])
            z = st["z"]                                                                           # :216-217
            logp = gaussian_diag_logps(pz_mean, pz_logsd * 2.0, z)                                # :298
            kl = torch.empty_like(z)
            _capi.check(_capi.lib().iaf_kl_combine(_ptr(st["logq0"]), _ptr(st["logdet"]), _ptr(logp), _ptr(kl), kl.numel(),
                                                   _stream()))
            kl_sum, kl_obj = kl_reduce(kl, self.kl_min)
        # TF: kl_obj[b] = sum_c max(mean_b sum_hw kl, kl_min) for every b (tf_train.py:79-82); Theano adds that SCALAR
        # once per layer (models.py:460-461)
        obj_kl = kl_obj[0] if self.kl_min > 0 else kl_sum
        return dict(h=torch.cat([h_det, z], dim=1), z=z, kl=kl, kl_sum=kl_sum, obj_kl=obj_kl)


def kl_reduce(kl, kl_min):
    """[B, C, H, W] KL elements -> (kl_sum [B], kl_obj [B]) with the engine's free-bits reduction kernels"""
    from .layers import kl_free_bits
    return kl_free_bits(kl, kl_min)
