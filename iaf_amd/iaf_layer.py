"""The IAF part of tf_train.IAFLayer.down (tf_train.py:56-85) as one engine call.

The two convolutions around it (down_conv1 / down_conv2, tf_train.py:53,93) are outside this
path; the boundary is the channel split of down_conv1's output
(tf_train.py:54) and the stored up-pass tensors (tf_train.py:38)."""
import ctypes

import torch

from . import _capi
from .layers import ARStack, WNConv2d, resample2, _ptr, _stream


class IAFPosterior(object):
    """Holds the ar_multiconv2d variables of one IAFLayer and evaluates
    posterior sample -> IAF step -> log-det -> KL / free bits."""

    def __init__(self, z_size, h_size, depth_ar=2, kl_min=0.25):
        self.z_size, self.h_size, self.kl_min = z_size, h_size, kl_min
        self.stack = ARStack(z_size, [h_size] * depth_ar)
        # set by up() in the reference (tf_train.py:38)
        self.qz_mean = self.qz_logsd = self.up_context = None

    def load(self, params):
        self.stack.prepare(params)

    def set_up_state(self, qz_mean, qz_logsd, up_context):
        self.qz_mean, self.qz_logsd, self.up_context = qz_mean, qz_logsd, up_context

    def down(self, pz_mean, pz_logsd, rz_mean, rz_logsd, down_context, eps, want_kl_elem=False):
        """Returns dict(z, kl_obj, kl_cost): the values tf_train.py:85-87 hand to the rest of down()."""
        return self.stack.posterior_block(self.qz_mean, self.qz_logsd, rz_mean, rz_logsd, pz_mean, pz_logsd,
                                          self.up_context, down_context, eps, self.kl_min, want_kl_elem)


def gaussian_sample(mean, logsd, eps):
    """DiagonalGaussian(mean, 2*logsd).sample with the given noise (distributions.py:21-24; tf_train.py:56,61)"""
    out = torch.empty_like(mean)
    _capi.check(_capi.lib().iaf_gaussian_sample_logsd(_ptr(mean), _ptr(logsd), _ptr(eps), _ptr(out), mean.numel(), _stream()))
    return out


class IAFLayer(object):
    """tf_train.IAFLayer (tf_train.py:23-95), entirely on the GPU; modes "train" / "init" / "sample" (tf_train.py:60-66),
    with or without downsampling (tf_train.py:33,42-43,89-91):

        up():   elu -> up_conv1 -> split(qz_mean, qz_logsd, up_context, h) -> elu -> up_conv3 -> input + 0.1*h
        down(): elu -> down_conv1 -> split(pz_mean, pz_logsd, rz_mean, rz_logsd, down_context, h_det)
                -> posterior sample -> IAF step -> KL / free bits -> elu(concat(z, h_det)) -> down_conv2 -> input + 0.1*h

    Six launches per down() (down_conv1, depth_ar+1 masked convs, 2 KL reductions, down_conv2) and two per up(); every
    ELU / split / concat / residual of the reference graph is fused into a conv's staging or store.  The sampling
    noise `eps` is an input (the reference draws it inside DiagonalGaussian.sample, distributions.py:21-24)."""

    def __init__(self, z_size, h_size, depth_ar=2, kl_min=0.25, downsample=False, mode="train"):
        """downsample=True: the first layer of a coarser level (tf_train.py:196) -- up() halves the resolution (stride-2
        up_conv1, residual resize 0.5), down() doubles it (down_deconv2 instead of down_conv2, residual resize 2).
        Forward: the strided forms of the bf16x3 conv kernel (csrc/iaf_conv_bf3.hpp, S2: nine taps per pixel of the coarser
        grid; WNConv2d.stride2 / .deconv).  Backward: the stride-1 backward kernels between the adjoint resamplings
        (full-resolution conv + subsample; zero-inserted input + rotated filter, csrc/iaf_kernels_resample.hpp; the deconv's
        weight norm has its own backward launches)."""
        if mode not in ("train", "init", "sample"):
            raise ValueError("mode must be 'train', 'init' or 'sample' (tf_train.py:60-66), got %r" % (mode,))
        self.z_size, self.h_size, self.kl_min = int(z_size), int(h_size), float(kl_min)
        self.downsample, self.mode = bool(downsample), mode
        zs, hs = self.z_size, self.h_size
        self.up_conv1 = WNConv2d(hs, 2 * zs + 2 * hs)      # tf_train.py:36
        self.up_conv3 = WNConv2d(hs, hs)                   # :41
        self.down_conv1 = WNConv2d(hs, 4 * zs + 2 * hs)    # :53
        self.down_conv2 = WNConv2d(hs + zs, hs)            # :93 (down_deconv2, :91, when downsampling)
        self.posterior = IAFPosterior(zs, hs, depth_ar, kl_min)

    @property
    def last_conv_name(self):
        return "down_deconv2" if self.downsample else "down_conv2"

    def load(self, params):
        """params: {"up_conv1/V": ..., "ar_multiconv2d/layer_0/V": ..., "down_conv2/b": ...} device fp32 tensors
        ("down_deconv2/..." with V [3,3,h_size,h_size+z_size] for a downsampling layer, tf_train.py:91)."""
        for nm in ("up_conv1", "up_conv3", "down_conv1"):
            getattr(self, nm).prepare(params[nm + "/V"], params[nm + "/g"], params[nm + "/b"])
        nm = self.last_conv_name
        if self.downsample:
            self.down_conv2.prepare_deconv(params[nm + "/V"], params[nm + "/g"], params[nm + "/b"])
        else:
            self.down_conv2.prepare(params[nm + "/V"], params[nm + "/g"], params[nm + "/b"])
        pre = "ar_multiconv2d/"
        self.posterior.load({k[len(pre):]: v for k, v in params.items() if k.startswith(pre)})

    CONVS = ("up_conv1", "up_conv3", "down_conv1", "down_conv2")

    def convs(self):
        return [getattr(self, nm) for nm in self.CONVS]

    def trim_packs(self, B, H, W):
        """inference at ONE size ((B, H, W) = this layer's input; a downsampling layer's convs at H/2 x W/2): every plain conv keeps only the
        weight pack its launch reads (WNConv2d.trim_packs: 4-6 instead of 14 bytes written per weight by every prep launch) and the stack
        only split packs.  Not for training layers (they keep every pack); returns {conv name: pack}."""
        h, w = (H // 2, W // 2) if self.downsample else (H, W)
        kept = {"up_conv1": self.up_conv1.trim_packs(B, H, W, strided=self.downsample),
                "up_conv3": self.up_conv3.trim_packs(B, h, w), "down_conv1": self.down_conv1.trim_packs(B, h, w)}
        if not self.downsample:                     # (the deconv's packs derive from its fp32 pack: all kept)
            kept["down_conv2"] = self.down_conv2.trim_packs(B, h, w)
        st = self.posterior.stack
        if st.step_is_f16(B, h, w):                 # the one-launch step on two fp16 planes reads nothing else
            st.set_packs(f32=False, bf16x3=False, f16x2=True)
            kept["ar_multiconv2d"] = "f16x2"
        elif st.step_is_fused(B, h, w):
            st.set_packs(f32=False, bf16x3=True, f16x2=False)
            kept["ar_multiconv2d"] = "bf16x3"
        return kept

    @staticmethod
    def conv_params(params):
        """the (V, g, b) tuples of one layer's plain convs in `convs()` order (for ConvPrepBatch.run)"""
        return [(params[nm + "/V"], params[nm + "/g"], params[nm + "/b"]) for nm in IAFLayer.CONVS]

    # The public forward methods.  A launch on two fp16 planes (the default arithmetic of the step and of the plain convs, round 6) that met an
    # operand beyond 65504 wrote inf / NaN -- the caller's NaN check sees that step (tf_train.py:283-285) -- and the NEXT call on that stack /
    # conv says so once (_capi.RangeError) while the object goes back to bf16 planes.  At layer level the call is simply repeated (the
    # object now computes in fp32's exponent range) behind a RuntimeWarning; the C ABI and the per-object Python classes raise.
    def _range_retry(self, fn, *a):
        import warnings
        for _ in range(len(self.CONVS) + 1):                  # (every conv and the stack say it once, each for itself)
            try:
                return fn(*a)
            except _capi.RangeError as e:
                warnings.warn("IAFLayer: %s -- the call is repeated on bf16 planes" % (e,), RuntimeWarning, stacklevel=3)
        return fn(*a)

    def up(self, inp, autotune=False):
        return self._range_retry(self._up, inp, autotune)

    def down(self, inp, eps, autotune=False, eps_prior=None):
        return self._range_retry(self._down, inp, eps, autotune, eps_prior)

    def up_train(self, inp, autotune=False):
        return self._range_retry(self._up_train, inp, autotune)

    def down_train(self, inp, eps, autotune=False):
        return self._range_retry(self._down_train, inp, eps, autotune)

    @staticmethod
    def stack_params(params):
        pre = "ar_multiconv2d/"
        return {k[len(pre):]: v for k, v in params.items() if k.startswith(pre)}

    def _up(self, inp, autotune=False):
        zs, hs = self.z_size, self.h_size
        if self.downsample:
            # stride [2,2], SAME (:33,36): the strided kernel (output (i,j) = output (2i+1, 2j+1) of the stride-1 conv)
            parts = self.up_conv1.stride2(inp, elu_input=True, split=[zs, zs, hs, hs])
            inp = resample2(inp, "down_even")                                                             # :42-43
        else:
            parts = self.up_conv1(inp, elu_input=True, split=[zs, zs, hs, hs], autotune=autotune)         # :35-37
        qz_mean, qz_logsd, up_context, h = parts
        self.posterior.set_up_state(qz_mean, qz_logsd, up_context)                                        # :38
        return self.up_conv3(h, elu_input=True, residual=inp, autotune=autotune)[0]                       # :40-44

    def _down(self, inp, eps, autotune=False, eps_prior=None):
        """Returns (output, kl_obj, kl_cost) like tf_train.py:95.  `eps` is the posterior noise (mode "train"),
        `eps_prior` the prior noise modes "init" / "sample" draw instead (tf_train.py:60-61).  autotune=True: the first
        call at a new (B,H,W) searches the launch shapes of the plain convs (cuDNN's algorithm search in the reference)."""
        zs, hs = self.z_size, self.h_size
        pz_mean, pz_logsd, rz_mean, rz_logsd, down_context, h_det = self.down_conv1(
            inp, elu_input=True, split=[zs] * 4 + [hs] * 2, autotune=autotune)                            # :52-54
        po = self.posterior
        if self.mode == "train":
            blk = po.down(pz_mean, pz_logsd, rz_mean, rz_logsd, down_context, eps)                        # :56-85
        else:
            if eps_prior is None:
                raise ValueError("modes 'init' and 'sample' sample the PRIOR (tf_train.py:60-61): pass eps_prior")
            z0 = gaussian_sample(pz_mean, pz_logsd, eps_prior)                                            # :56,61
            if self.mode == "sample":                                                                     # :65-66
                zero = torch.zeros(z0.shape[0], dtype=torch.float32, device=z0.device)
                blk = dict(z=z0, kl_obj=zero, kl_cost=zero.clone())
            else:
                # "init": logqs, the IAF step and the KL run on the prior sample (:67-85) -- the fused posterior block
                # with the posterior noise that reproduces z0
                eps_eq = torch.empty_like(z0)
                _capi.check(_capi.lib().iaf_noise_from_sample(_ptr(z0), _ptr(po.qz_mean), _ptr(po.qz_logsd), _ptr(rz_mean),
                                                              _ptr(rz_logsd), _ptr(eps_eq), z0.numel(), _stream()))
                blk = po.down(pz_mean, pz_logsd, rz_mean, rz_logsd, down_context, eps_eq)
        if self.downsample:
            # h = deconv2d(elu(concat(z, h_det))) (:87-91), input = resize_nearest_neighbor(input, 2) the residual (:90,94): the
            # four output phases as convs of the low-resolution tensors (WNConv2d.deconv)
            out = self.down_conv2.deconv(blk["z"], x2=h_det, elu_input=True, residual=inp)
        else:
            out = self.down_conv2(blk["z"], x2=h_det, elu_input=True, residual=inp, autotune=autotune)[0]  # :87-94
        self.last_block = blk
        return out, blk["kl_obj"], blk["kl_cost"]

    # -- training: forward that keeps what the backward needs, and the backward (tf_train.py:138 for this layer) -----
    def set_training(self, on=True):
        for c in self.convs():
            c.set_training(on)
        self.posterior.stack.set_training(on)

    def _up_train(self, inp, autotune=False):
        zs, hs = self.z_size, self.h_size
        res = inp
        if self.downsample:                                       # as in up()
            parts = self.up_conv1.stride2(inp, elu_input=True, split=[zs, zs, hs, hs])
            res = resample2(inp, "down_even")
        else:
            parts = self.up_conv1(inp, elu_input=True, split=[zs, zs, hs, hs], autotune=autotune)
        qz_mean, qz_logsd, up_context, h = parts
        self.posterior.set_up_state(qz_mean, qz_logsd, up_context)
        out = self.up_conv3(h, elu_input=True, residual=res, autotune=autotune)[0]
        self._up_saved = dict(inp=inp, h=h)
        return out

    def _down_train(self, inp, eps, autotune=False):
        zs, hs = self.z_size, self.h_size
        pz_mean, pz_logsd, rz_mean, rz_logsd, down_context, h_det = self.down_conv1(
            inp, elu_input=True, split=[zs] * 4 + [hs] * 2, autotune=autotune)
        po = self.posterior
        blk = po.stack.posterior_block_train(po.qz_mean, po.qz_logsd, rz_mean, rz_logsd, pz_mean, pz_logsd, po.up_context,
                                             down_context, eps, self.kl_min)
        if self.downsample:                                       # as in down(); the backward builds the zero-inserted inputs it needs
            zu, hu = None, None
            out = self.down_conv2.deconv(blk["z"], x2=h_det, elu_input=True, residual=inp)
        else:
            zu, hu = blk["z"], h_det
            out = self.down_conv2(zu, x2=hu, elu_input=True, residual=inp, autotune=autotune)[0]
        self._down_saved = dict(inp=inp, eps=eps, pz_mean=pz_mean, pz_logsd=pz_logsd, rz_mean=rz_mean, rz_logsd=rz_logsd,
                                h_det=h_det, z=blk["z"], conv2_x=zu, conv2_x2=hu)
        return out, blk["kl_obj"], blk["kl_cost"]

    def down_backward(self, d_out, d_kl_obj, params, grads=None, autotune=False):
        """Backward of down_train.  d_out: gradient of `output`; d_kl_obj [B]: gradient of kl_obj (kl_cost is reporting
        only, tf_train.py:204-206).  Returns d_input; the gradients flowing to the up pass (d qz_mean, d qz_logsd,
        d up_context) are kept for up_backward.  Parameter gradients land in `grads` (name -> tensor, allocated if absent)."""
        grads = {} if grads is None else grads
        sv, po = self._down_saved, self.posterior
        zs, hs = self.z_size, self.h_size

        def gslot(nm):
            return tuple(grads.setdefault(nm + "/" + k, torch.empty_like(params[nm + "/" + k])) for k in ("V", "g", "b"))

        # output = input + 0.1*down_conv2(elu(concat(z, h_det)))                                    (tf_train.py:87-94)
        # downsampling (:89-91): the same conv backward on the zero-inserted inputs -- what reaches the inserted zeros is
        # dropped ("down_odd" is the adjoint of "up_zero_odd") -- and d input = the 2x2 block sums of d_out (adjoint of
        # resize_nearest_neighbor(input, 2)); the deconv's own weight norm is differentiated inside conv.backward
        cn = self.last_conv_name
        if self.downsample:
            sv["conv2_x"], sv["conv2_x2"] = resample2(sv["z"], "up_zero_odd"), resample2(sv["h_det"], "up_zero_odd")
        (d_z, d_h_det), _, _, _ = self.down_conv2.backward(
            sv["conv2_x"], [d_out], params[cn + "/V"], params[cn + "/g"], x2=sv["conv2_x2"], elu_input=True, dy_scale=0.1,
            grads_out=gslot(cn), autotune=autotune)
        if self.downsample:
            d_z, d_h_det = resample2(d_z, "down_odd"), resample2(d_h_det, "down_odd")
            d_out = resample2(d_out, "down_sum4")
        # the IAF posterior block                                                                   (tf_train.py:56-85)
        pre = "ar_multiconv2d/"
        sp = IAFLayer.stack_params(params)
        sg = {k: grads.setdefault(pre + k, torch.empty_like(v)) for k, v in sp.items()}
        pb = po.stack.posterior_block_backward(po.qz_mean, po.qz_logsd, sv["rz_mean"], sv["rz_logsd"], sv["pz_mean"],
                                               sv["pz_logsd"], sv["eps"], self.kl_min, sv["z"], d_z, d_kl_obj, sp,
                                               grads_out=sg)
        self._to_up = dict(d_qz_mean=pb["dmean"], d_qz_logsd=pb["dlogsd"], d_up_context=pb["dcontext"])
        # x = down_conv1(elu(input)) split six ways; d input = d_out + elu'(input) * W^T dY          (tf_train.py:52-54, 94)
        (d_inp,), _, _, _ = self.down_conv1.backward(
            sv["inp"], [pb["dpz_mean"], pb["dpz_logsd"], pb["dmean"], pb["dlogsd"], pb["dcontext"], d_h_det],
            params["down_conv1/V"], params["down_conv1/g"], elu_input=True, dx_residual=d_out, grads_out=gslot("down_conv1"),
            autotune=autotune)
        return d_inp

    def up_backward(self, d_out, params, grads=None, autotune=False):
        """Backward of up_train (after down_backward of the same layer).  Returns d_input."""
        grads = {} if grads is None else grads
        sv, tu = self._up_saved, self._to_up

        def gslot(nm):
            return tuple(grads.setdefault(nm + "/" + k, torch.empty_like(params[nm + "/" + k])) for k in ("V", "g", "b"))

        (d_h,), _, _, _ = self.up_conv3.backward(sv["h"], [d_out], params["up_conv3/V"], params["up_conv3/g"], elu_input=True,
                                                 dy_scale=0.1, grads_out=gslot("up_conv3"), autotune=autotune)   # :40-44
        dys = [tu["d_qz_mean"], tu["d_qz_logsd"], tu["d_up_context"], d_h]
        if self.downsample:
            # stride-2 conv = the stride-1 conv subsampled at the odd positions (:33,36): its gradient is the stride-1 conv's
            # backward on the zero-inserted dY; the residual resize_nearest_neighbor(input, 0.5) (:43) hands d_out to the even ones
            dys = [resample2(t, "up_zero_odd") for t in dys]
            d_out = resample2(d_out, "up_zero_even")
        (d_inp,), _, _, _ = self.up_conv1.backward(
            sv["inp"], dys, params["up_conv1/V"], params["up_conv1/g"], elu_input=True, dx_residual=d_out,
            grads_out=gslot("up_conv1"), autotune=autotune)                                                        # :35-38
        return d_inp
