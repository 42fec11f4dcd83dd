"""GRADIENT ORACLE for the IAF step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference never writes its backward pass: TensorFlow derives it (`opt.compute_gradients(obj)`,
tf_train.py:138; Theano `T.grad`, graphy/misc/optim.py:102).  The oracle therefore restates the FORWARD of the
path in PyTorch-CPU float64 (same lines as oracle/iaf_oracle.py, which tests hold equal to this) and lets autograd
derive the gradients -- i.e. it computes exactly what TF's autodiff of the reference graph computes, including the
mask on dV (the mask multiplies V in the graph, layers.py:57) and the weight-norm chain (layers.py:60).
tests/test_grad_oracle.py checks forward == NumPy oracle and gradients == central finite differences."""
import numpy as np
import torch
import torch.nn.functional as F

from . import iaf_oracle as O

LOG2PI = float(np.log(2 * np.pi))


def _t(a, grad=False):
    t = torch.tensor(np.asarray(a, dtype=np.float64))
    t.requires_grad_(grad)
    return t


def ar_conv2d(x, V, g, b, zerodiagonal):
    """tf_utils/layers.py:144-154 -> 52-64 (mask, weight-norm, SAME cross-correlation, bias)."""
    kh, kw, n_in, n_out = V.shape
    mask = torch.from_numpy(O.get_conv_ar_mask(kh, kw, n_in, n_out, zerodiagonal)).to(torch.float64)
    v = mask * V
    w = torch.exp(g).reshape(1, 1, 1, -1) * v / torch.sqrt(torch.clamp((v * v).sum(dim=(0, 1, 2), keepdim=True), min=1e-12))
    return F.conv2d(x, w.permute(3, 2, 0, 1), b, padding=(kh // 2, kw // 2))


def ar_multiconv2d(x, context, params, n_h):
    """layers.py:158-166"""
    for i in range(len(n_h)):
        p = "layer_%d/" % i
        x = ar_conv2d(x, params[p + "V"], params[p + "g"], params[p + "b"], False)
        if i == 0:
            x = x + context
        x = F.elu(x)
    return [ar_conv2d(x, params["layer_out_%d/V" % i], params["layer_out_%d/g" % i], params["layer_out_%d/b" % i], True)
            for i in range(2)]


def iaf_step(z, context, params, n_h):
    """tf_train.py:69-72"""
    m_raw, s_raw = ar_multiconv2d(z, context, params, n_h)
    m, s = 0.1 * m_raw, 0.1 * s_raw
    return (z - m) / torch.exp(s), s


def iaf_step_grads(z, context, params, n_h, dz_new, dlogsd):
    """Gradients of  L = <dz_new, z_new> + <dlogsd, logsd>  w.r.t. z, context and every V/g/b."""
    zt, ct = _t(z, True), _t(context, True)
    pt = {k: _t(v, True) for k, v in params.items()}
    z_new, logsd = iaf_step(zt, ct, pt, n_h)
    loss = (z_new * _t(dz_new)).sum() + (logsd * _t(dlogsd)).sum()
    loss.backward()
    out = {"z": zt.grad.numpy(), "context": ct.grad.numpy() if ct.grad is not None else np.zeros_like(context)}
    for k, v in pt.items():
        out[k] = v.grad.numpy()
    return out, z_new.detach().numpy(), logsd.detach().numpy()


def gaussian_diag_logps(mean, logvar, sample):
    return -0.5 * (LOG2PI + logvar + (sample - mean) ** 2 / torch.exp(logvar))


def posterior_block(qm, ql, rm, rl, pm, pl, uc, dc, eps, params, n_h, kl_min):
    """tf_train.py:56-85 (mode train)."""
    mean, logvar = rm + qm, 2 * (rl + ql)
    z0 = mean + torch.exp(0.5 * logvar) * eps
    logqs = gaussian_diag_logps(mean, logvar, z0)
    z, s = iaf_step(z0, uc + dc, params, n_h)
    logqs = logqs + s
    logps = gaussian_diag_logps(pm, 2 * pl, z)
    kl = logqs - logps
    n = z.shape[0]
    if kl_min > 0:
        kl_ave = torch.clamp(kl.sum(dim=(2, 3)).mean(dim=0, keepdim=True), min=kl_min)
        kl_obj = kl_ave.repeat(n, 1).sum(dim=1)
    else:
        kl_obj = kl.sum(dim=(1, 2, 3))
    return z, kl_obj, kl.sum(dim=(1, 2, 3))


def posterior_block_grads(inputs, params, n_h, kl_min, dz, dkl_obj):
    """inputs: dict(qm, ql, rm, rl, pm, pl, uc, dc, eps).  L = <dz, z> + <dkl_obj, kl_obj>."""
    it = {k: _t(v, k != "eps") for k, v in inputs.items()}
    pt = {k: _t(v, True) for k, v in params.items()}
    z, kl_obj, kl_cost = posterior_block(it["qm"], it["ql"], it["rm"], it["rl"], it["pm"], it["pl"], it["uc"], it["dc"],
                                         it["eps"], pt, n_h, kl_min)
    loss = (z * _t(dz)).sum() + (kl_obj * _t(dkl_obj)).sum()
    loss.backward()
    out = {k: v.grad.numpy() for k, v in it.items() if k != "eps"}
    for k, v in pt.items():
        out[k] = v.grad.numpy()
    return out, z.detach().numpy(), kl_obj.detach().numpy(), kl_cost.detach().numpy()


# --------------------------------------------------------------------------------------
# whole IAFLayer (tf_train.py:23-95), with and without downsampling, for the backward of the plain convs
# --------------------------------------------------------------------------------------
def conv2d(x, V, g, b):
    """tf_utils/layers.py:52-64, mask=None."""
    w = torch.exp(g).reshape(1, 1, 1, -1) * V / torch.sqrt(torch.clamp((V * V).sum(dim=(0, 1, 2), keepdim=True), min=1e-12))
    return F.conv2d(x, w.permute(3, 2, 0, 1), b, padding=(1, 1))


def conv2d_stride2(x, V, g, b):
    """layers.py:52-64 with stride [2,2], SAME: for even sizes TF pads one row / column at the END only."""
    w = torch.exp(g).reshape(1, 1, 1, -1) * V / torch.sqrt(torch.clamp((V * V).sum(dim=(0, 1, 2), keepdim=True), min=1e-12))
    return F.conv2d(F.pad(x, (0, 1, 0, 1)), w.permute(3, 2, 0, 1), b, stride=2)


def deconv2d(x, V, g, b):
    """layers.py:83-112 as oracle/iaf_oracle.py:deconv2d states it: V [3,3,n_out,n_in], norm over (kh, kw, n_out) per INPUT
    channel, conv2d_transpose(SAME, stride 2): the first 2H x 2W of the full transposed conv."""
    w = torch.exp(g).reshape(1, 1, -1, 1) * V / torch.sqrt(torch.clamp((V * V).sum(dim=(0, 1, 2), keepdim=True), min=1e-12))
    n, c, hh, ww = x.shape
    full = F.conv_transpose2d(x, w.permute(3, 2, 0, 1), stride=2)
    return full[:, :, :2 * hh, :2 * ww] + b.reshape(1, -1, 1, 1)


def iaf_layer(up_inp, down_inp, eps, params, z_size, h_size, kl_min, downsample=False):
    """up (tf_train.py:29-44) then down (46-95), mode train; downsample: stride-2 up_conv1 + resize 0.5 on the way up
    (:33,42-43), down_deconv2 + resize 2 on the way down (:89-91)."""
    zs, hs = z_size, h_size
    if downsample:
        x = conv2d_stride2(F.elu(up_inp), params["up_conv1/V"], params["up_conv1/g"], params["up_conv1/b"])
    else:
        x = conv2d(F.elu(up_inp), params["up_conv1/V"], params["up_conv1/g"], params["up_conv1/b"])
    qz_mean, qz_logsd, up_context, h = torch.split(x, [zs, zs, hs, hs], dim=1)
    h = conv2d(F.elu(h), params["up_conv3/V"], params["up_conv3/g"], params["up_conv3/b"])
    up_out = (up_inp[:, :, ::2, ::2] if downsample else up_inp) + 0.1 * h
    x = conv2d(F.elu(down_inp), params["down_conv1/V"], params["down_conv1/g"], params["down_conv1/b"])
    pz_mean, pz_logsd, rz_mean, rz_logsd, down_context, h_det = torch.split(x, [zs] * 4 + [hs] * 2, dim=1)
    sp = {k[len("ar_multiconv2d/"):]: v for k, v in params.items() if k.startswith("ar_multiconv2d/")}
    n_h = [hs] * sum(1 for k in sp if k.startswith("layer_") and not k.startswith("layer_out") and k.endswith("/g"))
    z, kl_obj, kl_cost = posterior_block(qz_mean, qz_logsd, rz_mean, rz_logsd, pz_mean, pz_logsd, up_context, down_context,
                                         eps, sp, n_h, kl_min)
    if downsample:
        h = deconv2d(F.elu(torch.cat([z, h_det], dim=1)), params["down_deconv2/V"], params["down_deconv2/g"], params["down_deconv2/b"])
        return up_out, down_inp.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3) + 0.1 * h, kl_obj, kl_cost
    h = conv2d(F.elu(torch.cat([z, h_det], dim=1)), params["down_conv2/V"], params["down_conv2/g"], params["down_conv2/b"])
    return up_out, down_inp + 0.1 * h, kl_obj, kl_cost


def iaf_layer_grads(up_inp, down_inp, eps, params, z_size, h_size, kl_min, d_up_out, d_down_out, d_kl_obj, downsample=False):
    """L = <d_up_out, up_out> + <d_down_out, output> + <d_kl_obj, kl_obj>; gradients w.r.t. both inputs and every variable."""
    ut, dt = _t(up_inp, True), _t(down_inp, True)
    pt = {k: _t(v, True) for k, v in params.items()}
    up_out, out, kl_obj, kl_cost = iaf_layer(ut, dt, _t(eps), pt, z_size, h_size, kl_min, downsample)
    loss = (up_out * _t(d_up_out)).sum() + (out * _t(d_down_out)).sum() + (kl_obj * _t(d_kl_obj)).sum()
    loss.backward()
    grads = {k: v.grad.numpy() for k, v in pt.items()}
    fw = dict(up_out=up_out.detach().numpy(), output=out.detach().numpy(), kl_obj=kl_obj.detach().numpy(),
              kl_cost=kl_cost.detach().numpy())
    return dict(up_inp=ut.grad.numpy(), down_inp=dt.grad.numpy(), params=grads), fw


# --------------------------------------------------------------------------------------
# the whole model, CVAE1._forward (tf_train.py:150-218), for `opt.compute_gradients(obj)` (tf_train.py:128)
# --------------------------------------------------------------------------------------
def _sub(params, prefix):
    return {k[len(prefix):]: v for k, v in params.items() if k.startswith(prefix)}


def _layer_up(inp, p, zs, hs, downsample):
    """tf_train.py:29-44 (the lines of iaf_layer above, up pass only)"""
    conv1 = conv2d_stride2 if downsample else conv2d
    x = conv1(F.elu(inp), p["up_conv1/V"], p["up_conv1/g"], p["up_conv1/b"])
    qz_mean, qz_logsd, up_context, h = torch.split(x, [zs, zs, hs, hs], dim=1)
    h = conv2d(F.elu(h), p["up_conv3/V"], p["up_conv3/g"], p["up_conv3/b"])
    return (inp[:, :, ::2, ::2] if downsample else inp) + 0.1 * h, qz_mean, qz_logsd, up_context


def _layer_down(inp, p, qz_mean, qz_logsd, up_context, eps, zs, hs, kl_min, downsample):
    """tf_train.py:46-95, mode train (the lines of iaf_layer above, down pass only)"""
    x = conv2d(F.elu(inp), p["down_conv1/V"], p["down_conv1/g"], p["down_conv1/b"])
    pz_mean, pz_logsd, rz_mean, rz_logsd, down_context, h_det = torch.split(x, [zs] * 4 + [hs] * 2, dim=1)
    sp = _sub(p, "ar_multiconv2d/")
    n_h = [hs] * sum(1 for k in sp if k.startswith("layer_") and not k.startswith("layer_out") and k.endswith("/g"))
    z, kl_obj, kl_cost = posterior_block(qz_mean, qz_logsd, rz_mean, rz_logsd, pz_mean, pz_logsd, up_context, down_context, eps, sp,
                                         n_h, kl_min)
    hh = F.elu(torch.cat([z, h_det], dim=1))
    if downsample:
        h = deconv2d(hh, p["down_deconv2/V"], p["down_deconv2/g"], p["down_deconv2/b"])
        return inp.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3) + 0.1 * h, kl_obj, kl_cost
    return inp + 0.1 * conv2d(hh, p["down_conv2/V"], p["down_conv2/g"], p["down_conv2/b"]), kl_obj, kl_cost


def cvae1_obj(x_uint8, params, z_size, h_size, depth, num_blocks, kl_min, noise):
    """tf_train.py:150-211 for k = 1, mode train, on torch fp64 tensors (params: dict of tensors).  Returns (x_out, obj)."""
    x = torch.clamp((_t(np.asarray(x_uint8, dtype=np.float64)) + 0.5) / 256.0, 0.0, 1.0) - 0.5                 # :153-154
    V, g, b = params["x_enc/V"], params["x_enc/g"], params["x_enc/b"]
    w = torch.exp(g).reshape(1, 1, 1, -1) * V / torch.sqrt(torch.clamp((V * V).sum(dim=(0, 1, 2), keepdim=True), min=1e-12))
    h = F.conv2d(F.pad(x, (1, 2, 1, 2)), w.permute(3, 2, 0, 1), b, stride=2)                                    # :183 (5x5, SAME, even sizes)
    ups = {}
    for i in range(depth):
        for j in range(num_blocks):
            h, qm, ql, uc = _layer_up(h, _sub(params, "IAF_%d_%d/" % (i, j)), z_size, h_size, i > 0 and j == 0)
            ups[(i, j)] = (qm, ql, uc)
    n, hw = x.shape[0], x.shape[2] // 2 ** depth
    h = params["h_top"].reshape(1, -1, 1, 1).repeat(n, 1, hw, hw)                                                # :189-192
    kl_obj = torch.zeros(n, dtype=torch.float64)
    it = iter(noise)
    for i in reversed(range(depth)):
        for j in reversed(range(num_blocks)):
            next(it)                                                                                            # the prior's draw (unused in train mode)
            qm, ql, uc = ups[(i, j)]
            h, cur_obj, _ = _layer_down(h, _sub(params, "IAF_%d_%d/" % (i, j)), qm, ql, uc, _t(next(it)), z_size, h_size, kl_min,
                                        i > 0 and j == 0)
            kl_obj = kl_obj + cur_obj
    V, g, b = params["x_dec/V"], params["x_dec/g"], params["x_dec/b"]
    w = torch.exp(g).reshape(1, 1, -1, 1) * V / torch.sqrt(torch.clamp((V * V).sum(dim=(0, 1, 2), keepdim=True), min=1e-12))
    hh = F.elu(h)
    full = F.conv_transpose2d(hh, w.permute(3, 2, 0, 1), stride=2)                                              # [n,3,2H+3,2W+3]; SAME crops 1 in front
    xo = full[:, :, 1:1 + 2 * hh.shape[2], 1:1 + 2 * hh.shape[3]] + b.reshape(1, -1, 1, 1)                      # :206-207
    xo = torch.clamp(xo, -0.5 + 1 / 512., 0.5 - 1 / 512.)                                                       # :208
    scale = torch.exp(params["dec_log_stdv"])
    binsize = 1 / 256.0
    s = (torch.floor(x / binsize) * binsize - xo) / scale
    logp = torch.log(torch.sigmoid(s + binsize / scale) - torch.sigmoid(s) + 1e-7)                              # distributions.py:28-32
    log_pxz = logp.sum(dim=(1, 2, 3))
    return xo, (kl_obj - log_pxz).sum()                                                                         # :211


def cvae1_grads(x_uint8, params, z_size, h_size, depth, num_blocks, kl_min, noise):
    """d obj / d every variable (what opt.compute_gradients(obj) hands the optimizer, tf_train.py:128).  Returns (grads, x_out, obj)."""
    pt = {k: _t(v, True) for k, v in params.items()}
    xo, obj = cvae1_obj(x_uint8, pt, z_size, h_size, depth, num_blocks, kl_min, noise)
    obj.backward()
    # (a variable obj does not depend on -- the top layer's up_conv3: its output is replaced by h_top, tf_train.py:189-192 -- has gradient 0)
    return {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in pt.items()}, xo.detach().numpy(), float(obj.detach())


# --------------------------------------------------------------------------------------
# the Theano statement of the same operator (graphy/nodes/ar.py, graphy/nodes/conv.py), for T.grad
# (graphy/misc/optim.py:102) of the lines models.py:168-176 / 272-285
# --------------------------------------------------------------------------------------
def theano_ar_conv2d(h, w_, b_, s_, n_in, n_out, zerodiagonal=True, flipmask=False):
    """graphy/nodes/ar.py:304-330 as oracle/iaf_oracle.py:theano_ar_conv2d states it: mask, centre rows zeroed
    (set_subtensor, :268-276), kerns / (norm + 1e-8) * exp(3 s), border-indicator channel, true convolution 'valid'."""
    mask = torch.from_numpy(np.ascontiguousarray(O.theano_ar_mask(n_in, n_out, 3, zerodiagonal, flipmask, True))).to(torch.float64)
    n, c, H, W = h.shape
    hp = torch.zeros((n, c + 1, H + 2, W + 2), dtype=torch.float64)
    hp[:, c] = 1.0
    hp[:, c, 1:-1, 1:-1] = 0.0
    hp = hp.clone()
    hp[:, :c, 1:-1, 1:-1] = h                                       # conv.py:71-83
    kerns = mask * w_
    if zerodiagonal:
        keep = torch.ones_like(kerns)
        keep[:(n_out // n_in if n_out >= n_in else 1), :, 1, 1] = 0.0
        kerns = kerns * keep
    norm = torch.sqrt((kerns ** 2).sum(dim=(1, 2, 3), keepdim=True)) + 1e-8
    kerns = kerns / norm * torch.exp(3.0 * s_).reshape(-1, 1, 1, 1)
    return F.conv2d(hp, torch.flip(kerns, dims=(2, 3)), b_)        # conv_mode='conv': the kernel is flipped


def theano_multiconv2d(h, context, w, name, n_in, n_h, n_out, flipmask=False):
    """graphy/nodes/ar.py:378-416, nl='elu'"""
    sizes = [n_in] + list(n_h)
    for i in range(len(n_h)):
        p = "%s_%d" % (name, i)
        h = theano_ar_conv2d(h, w[p + "_w"], w[p + "_b"], w[p + "_s"], sizes[i], sizes[i + 1], False, flipmask)
        if i == 0:
            h = h + context
        h = F.elu(h)
    return [theano_ar_conv2d(h, w["%s_out_%d_w" % (name, i)], w["%s_out_%d_b" % (name, i)], w["%s_out_%d_s" % (name, i)],
                             sizes[-1], n_out[i], True, flipmask) for i in range(len(n_out))]


def theano_iaf2_nl(z, context, w, name, n_z, n_h, flipmask=False):
    """models.py:168-175 / 281-285"""
    m_raw, s_raw = theano_multiconv2d(z, context, w, name, n_z, n_h, [n_z, n_z], flipmask)
    m, s = 0.1 * m_raw, 0.1 * s_raw
    return (z - m) / torch.exp(s), s


def theano_iaf2_nl_grads(z, context, w, name, n_z, n_h, dz_new, dlogsd, flipmask=False):
    """Gradients of  L = <dz_new, z_new> + <dlogsd, logsd>  w.r.t. z, context and every _w/_s/_b."""
    zt, ct = _t(z, True), _t(context, True)
    wt = {k: _t(v, True) for k, v in w.items()}
    z_new, logsd = theano_iaf2_nl(zt, ct, wt, name, n_z, n_h, flipmask)
    loss = (z_new * _t(dz_new)).sum() + (logsd * _t(dlogsd)).sum()
    loss.backward()
    out = {"z": zt.grad.numpy(), "context": ct.grad.numpy()}
    for k, v in wt.items():
        out[k] = v.grad.numpy()
    return out, z_new.detach().numpy(), logsd.detach().numpy()


def theano_cvae_iaf(posterior, h_up, h_dn, eps, w, name, n_h, n_z, depth_ar, kl_min, flipmask=False):
    """The part of models.cvae_layer between its plain convs (oracle/iaf_oracle.py:theano_cvae_layer restates the whole
    layer): h_up / h_dn are the outputs of up_conv1 / down_conv1 in the reference's channel order (models.py:141-143,
    273-279, 296-297).  Returns (what up_conv2 reads, what down_conv2 reads = concat([h_det, z]), kl, obj_kl (:454-466))."""
    pc = name + "_posterior_conv1"
    h_det_u, qm, ql, ctx = torch.split(h_up, [n_h, n_z, n_z, n_h], dim=1)
    if posterior == "up_iaf2_nl":                                                      # :168-176
        z0 = qm + torch.exp(ql) * eps
        logqs = gaussian_diag_logps(qm, 2 * ql, z0)
        z, s = theano_iaf2_nl(z0, ctx, w, pc, n_z, [n_h] * depth_ar, flipmask)
        logqs = logqs + s
        up_out = torch.cat([h_det_u, z], dim=1)
        h_det, pm, pl = torch.split(h_dn, [n_h, n_z, n_z], dim=1)
    else:                                                                              # :272-285
        up_out = h_det_u
        h_det, pm, pl, rm, rl, dctx = torch.split(h_dn, [n_h, n_z, n_z, n_z, n_z, n_h], dim=1)
        mean, logvar = qm + rm, 2 * ql + 2 * rl
        z0 = mean + torch.exp(0.5 * logvar) * eps
        logqs = gaussian_diag_logps(mean, logvar, z0)
        z, s = theano_iaf2_nl(z0, ctx + dctx, w, pc, n_z, [n_h] * depth_ar, flipmask)
        logqs = logqs + s
    kl = logqs - gaussian_diag_logps(pm, 2 * pl, z)                                     # :298, 328
    kl_sum = kl.sum(dim=(1, 2, 3))
    obj = torch.clamp(kl.sum(dim=(2, 3)).mean(dim=0), min=kl_min).sum() if kl_min > 0 else kl_sum   # :458-466
    return up_out, torch.cat([h_det, z], dim=1), kl, obj


def theano_cvae_iaf_grads(posterior, h_up, h_dn, eps, w, name, n_h, n_z, depth_ar, kl_min, d_up, d_h, d_obj, flipmask=False):
    """L = <d_up, up_out> + <d_h, h_out> + <d_obj, obj_kl>; gradients w.r.t. both conv outputs and every stack weight."""
    ut, dt = _t(h_up, True), _t(h_dn, True)
    wt = {k: _t(v, True) for k, v in w.items()}
    up_out, h_out, kl, obj = theano_cvae_iaf(posterior, ut, dt, _t(eps), wt, name, n_h, n_z, depth_ar, kl_min, flipmask)
    loss = (up_out * _t(d_up)).sum() + (h_out * _t(d_h)).sum() + (obj * _t(d_obj)).sum()
    loss.backward()
    fw = dict(up_out=up_out.detach().numpy(), h=h_out.detach().numpy(), kl=kl.detach().numpy(), obj_kl=obj.detach().numpy())
    return dict(h_up=ut.grad.numpy(), h_dn=dt.grad.numpy(), w={k: v.grad.numpy() for k, v in wt.items()}), fw
