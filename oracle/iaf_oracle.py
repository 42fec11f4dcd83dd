"""CPU ORACLE for the IAF posterior hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy restatement (dtype-following; fp64 by default) of the reference's algorithm for
the down_iaf2_nl / up_iaf2_nl masked-autoregressive transform and its log-det-Jacobian
accumulation.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; the product (iaf_amd/) never does and fails loudly without its HIP
library.

Parity pinning: the reference cannot be installed here (TensorFlow/Theano absent), but its
own Python for this path IS executed, unmodified, on a NumPy stand-in for the TF leaf ops
(tests/golden/tf_shim.py + make_golden.py).  tests/test_oracle_golden.py checks every
function below against those reference outputs and against the reference's known-answer
tests (tf_utils/distributions_test.py:7-38).  The Theano statement (graphy/nodes/ar.py, conv.py:pad2dwithchannel,
nodes/__init__.py:nonlinearity, rand.py:gaussian_diag) is pinned the same way: its Python-2 source goes through lib2to3
in memory and runs on tests/golden/theano_shim.py (make_golden_theano.py); tf_utils/adamax.py runs unmodified on variable
stubs (make_golden_adamax.py).  Still restated without a pin: the three affine lines of models.py:170-175 / 281-285
(models.py is a 600-line Python-2 graph builder that needs all of Theano) and the EMA update (a TF library class).

Every function cites the reference file:line (relative to /root/reference) it follows.
"""
import math

import numpy as np


# --------------------------------------------------------------------------------------
# a1 / a2  MADE masks
# --------------------------------------------------------------------------------------
def get_linear_ar_mask(n_in, n_out, zerodiagonal=False):
    """tf_utils/layers.py:115-131 (Python-2 integer division at 120,126)."""
    assert n_in % n_out == 0 or n_out % n_in == 0, "%d - %d" % (n_in, n_out)
    mask = np.ones([n_in, n_out], dtype=np.float32)
    if n_out >= n_in:
        k = n_out // n_in
        for i in range(n_in):
            mask[i + 1:, i * k:(i + 1) * k] = 0
            if zerodiagonal:
                mask[i:i + 1, i * k:(i + 1) * k] = 0
    else:
        k = n_in // n_out
        for i in range(n_out):
            mask[(i + 1) * k:, i:i + 1] = 0
            if zerodiagonal:
                mask[i * k:(i + 1) * k, i:i + 1] = 0
    return mask


def get_conv_ar_mask(h, w, n_in, n_out, zerodiagonal=False):
    """tf_utils/layers.py:134-141.  HWIO; rows above centre dead, centre row left of centre dead,
    centre tap = linear MADE mask."""
    l = (h - 1) // 2
    m = (w - 1) // 2
    mask = np.ones([h, w, n_in, n_out], dtype=np.float32)
    mask[:l, :, :, :] = 0
    mask[l, :m, :, :] = 0
    mask[l, m, :, :] = get_linear_ar_mask(n_in, n_out, zerodiagonal)
    return mask


# --------------------------------------------------------------------------------------
# a3  weight-normed (optionally masked) convolution
# --------------------------------------------------------------------------------------
def l2_normalize(v, axes, epsilon=1e-12):
    """tf.nn.l2_normalize as used at layers.py:45,60: v * rsqrt(max(sum v^2, eps))."""
    ss = np.sum(np.square(v), axis=axes, keepdims=True)
    return v / np.sqrt(np.maximum(ss, epsilon))


def weightnorm_weights(V, g, mask=None):
    """layers.py:56-60: v = mask*V; w = exp(g)[o] * l2_normalize(v, [0,1,2])."""
    v = V if mask is None else mask * V
    return np.exp(g).reshape([1, 1, 1, -1]) * l2_normalize(v, (0, 1, 2))


def _same_pad(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return out, tot // 2, tot - tot // 2


def conv2d_same_nchw(x, w, stride=(1, 1)):
    """tf.nn.conv2d(x, w, [1,1,sh,sw], "SAME", data_format="NCHW") (layers.py:46,64):
    cross-correlation, w HWIO, zero padding."""
    n, c, hh, ww = x.shape
    kh, kw, ci, co = w.shape
    assert ci == c
    sh, sw = stride
    oh, pt, pb = _same_pad(hh, kh, sh)
    ow, pl, pr = _same_pad(ww, kw, sw)
    xp = np.zeros((n, c, hh + pt + pb, ww + pl + pr), dtype=np.result_type(x, w))
    xp[:, :, pt:pt + hh, pl:pl + ww] = x
    y = np.zeros((n, co, oh, ow), dtype=xp.dtype)
    for a in range(kh):
        for b in range(kw):
            patch = xp[:, :, a:a + (oh - 1) * sh + 1:sh, b:b + (ow - 1) * sw + 1:sw]
            # tensordot over channels: [n,c,h,w] x [c,o] -> [n,h,w,o]
            y += np.moveaxis(np.tensordot(patch, w[a, b], axes=([1], [0])), 3, 1)
    return y


def conv2d(x, V, g, b, stride=(1, 1), mask=None):
    """layers.py:52-64 (non-init branch)."""
    w = weightnorm_weights(V, g, mask)
    return conv2d_same_nchw(x, w, stride) + b.reshape([1, -1, 1, 1])


def conv2d_init(x, V0, stride=(1, 1), init_scale=0.1, mask=None):
    """layers.py:38-51 (data-dependent init branch).  Returns (y, g, b)."""
    v = V0 if mask is None else mask * V0
    v_norm = l2_normalize(v, (0, 1, 2))
    x_init = conv2d_same_nchw(x, v_norm, stride)
    m_init = x_init.mean(axis=(0, 2, 3))
    v_init = x_init.var(axis=(0, 2, 3))
    scale_init = init_scale / np.sqrt(v_init + 1e-10)
    g = np.log(scale_init) / 3.0                   # layers.py:49 (sic: applied later as exp(g))
    b = -m_init * scale_init
    y = scale_init.reshape([1, -1, 1, 1]) * (x_init - m_init.reshape([1, -1, 1, 1]))
    return y, g, b


def deconv2d(x, V, g, b, stride=(2, 2)):
    """tf_utils/layers.py:83-112 (non-init branch) + my_deconv2d (67-80): V is [kh, kw, n_out, n_in]; the weight norm is
    `exp(g)[o] * l2_normalize(V, [0, 1, 2])`, i.e. the norm runs over (kh, kw, n_OUT) per INPUT channel (the reference
    normalises the deconv filter over its first three axes like the conv filter, whose third axis is n_in); then
    tf.nn.conv2d_transpose(SAME, stride 2, output = 2x input).  Restated as a gather: output (y, x) collects input
    (i, j) through tap (a, b) where y = 2i + a - pad_t (pad_t = 0 for k = 3, s = 2, SAME)."""
    kh, kw, co, ci = V.shape
    w = np.exp(g).reshape([1, 1, co, 1]) * l2_normalize(V, (0, 1, 2))
    n, c, hh, ww = x.shape
    assert c == ci
    sh, sw = stride
    oh, ow = hh * sh, ww * sw
    _, pt, _ = _same_pad(oh, kh, sh)
    _, pl, _ = _same_pad(ow, kw, sw)
    y = np.zeros((n, co, oh, ow), dtype=np.result_type(x, w))
    for a in range(kh):
        for bb in range(kw):
            # rows y with (y + pt - a) divisible by sh and the quotient inside the input
            ys = [yy for yy in range(oh) if (yy + pt - a) % sh == 0 and 0 <= (yy + pt - a) // sh < hh]
            xs = [xx for xx in range(ow) if (xx + pl - bb) % sw == 0 and 0 <= (xx + pl - bb) // sw < ww]
            if not ys or not xs:
                continue
            iy = [(yy + pt - a) // sh for yy in ys]
            ix = [(xx + pl - bb) // sw for xx in xs]
            patch = x[:, :, iy][:, :, :, ix]                                       # [n, ci, len(ys), len(xs)]
            y[np.ix_(range(n), range(co), ys, xs)] += np.einsum("nchw,oc->nohw", patch, w[a, bb])
    return y + b.reshape([1, -1, 1, 1])


def resize_nearest_neighbor(x, scale):
    """tf_utils/layers.py:169-175 (tf.image.resize_nearest_neighbor, align_corners=False): out[y] = in[floor(y / scale)]"""
    n, c, hh, ww = x.shape
    oh, ow = int(hh * scale), int(ww * scale)
    iy = np.minimum((np.arange(oh) * (hh / float(oh))).astype(int), hh - 1)
    ix = np.minimum((np.arange(ow) * (ww / float(ow))).astype(int), ww - 1)
    return x[:, :, iy][:, :, :, ix]


def elu(x):
    """tf.nn.elu (layers.py:159 default nl)."""
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


# --------------------------------------------------------------------------------------
# a4 / a5  masked AR conv stack
# --------------------------------------------------------------------------------------
def ar_conv2d(x, V, g, b, zerodiagonal=True):
    """layers.py:144-154: 3x3 (filter shape taken from V), stride 1, SAME, mask = a2."""
    kh, kw, n_in, n_out = V.shape
    mask = get_conv_ar_mask(kh, kw, n_in, n_out, zerodiagonal)
    return conv2d(x, V, g, b, mask=mask)


def ar_conv2d_init(x, V0, init_scale=1.0, zerodiagonal=True):
    """ar_conv2d under arg_scope(init=True): layers.py:152-154 -> 38-51 (init_scale default 1.)."""
    kh, kw, n_in, n_out = V0.shape
    mask = get_conv_ar_mask(kh, kw, n_in, n_out, zerodiagonal)
    return conv2d_init(x, V0, init_scale=init_scale, mask=mask)


def ar_multiconv2d(x, context, params, n_h, n_out, nl=elu):
    """layers.py:158-166.  params: {"layer_%d/V|g|b", "layer_out_%d/V|g|b"}.
    Returns the list of n_out raw outputs."""
    for i, size in enumerate(n_h):
        p = "layer_%d/" % i
        assert params[p + "V"].shape[3] == size
        x = ar_conv2d(x, params[p + "V"], params[p + "g"], params[p + "b"], zerodiagonal=False)
        if i == 0:
            x = x + context
        x = nl(x)
    outs = []
    for i, size in enumerate(n_out):
        p = "layer_out_%d/" % i
        assert params[p + "V"].shape[3] == size
        outs.append(ar_conv2d(x, params[p + "V"], params[p + "g"], params[p + "b"], zerodiagonal=True))
    return outs


def iaf_step(z, context, params, n_h):
    """The core unit: tf_train.py:69-72.
    m,s = 0.1*ar_multiconv2d(...); z' = (z-m)/exp(s); log-det increment = s (logqs += s).
    Returns (z_new, arw_logsd)."""
    n_z = z.shape[1]
    m_raw, s_raw = ar_multiconv2d(z, context, params, n_h, [n_z, n_z])
    arw_mean, arw_logsd = m_raw * 0.1, s_raw * 0.1
    z_new = (z - arw_mean) / np.exp(arw_logsd)
    return z_new, arw_logsd


# --------------------------------------------------------------------------------------
# a6  diagonal Gaussian
# --------------------------------------------------------------------------------------
def gaussian_diag_sample(mean, logvar, noise):
    """tf_utils/distributions.py:7-8,20-21 with the noise made an explicit input."""
    return mean + np.exp(0.5 * logvar) * noise


def gaussian_diag_logps(mean, logvar, sample):
    """tf_utils/distributions.py:10."""
    return -0.5 * (np.log(2 * np.pi) + logvar + np.square(sample - mean) / np.exp(logvar))


# --------------------------------------------------------------------------------------
# a7 / a8  IAFLayer
# --------------------------------------------------------------------------------------
def split_channels(x, sizes):
    """tf_utils/common.py:21-36 on split_dim=1."""
    assert x.shape[1] == int(np.sum(sizes))
    ids = np.cumsum([0] + list(sizes))
    return [x[:, ids[i]:ids[i + 1]] for i in range(len(sizes))]


def _sub(params, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in params.items() if k.startswith(prefix)}


def iaf_layer_up(inp, params, z_size, h_size, downsample=False):
    """tf_train.py:29-44.  Returns (output, qz_mean, qz_logsd, up_context)."""
    p = _sub(params, "up_conv1/")
    x = conv2d(elu(inp), p["V"], p["g"], p["b"], stride=(2, 2) if downsample else (1, 1))      # :33-36
    qz_mean, qz_logsd, up_context, h = split_channels(x, [z_size, z_size, h_size, h_size])
    p = _sub(params, "up_conv3/")
    h = conv2d(elu(h), p["V"], p["g"], p["b"])
    if downsample:
        inp = resize_nearest_neighbor(inp, 0.5)                                                # :42-43
    return inp + 0.1 * h, qz_mean, qz_logsd, up_context


def posterior_block(qz_mean, qz_logsd, rz_mean, rz_logsd, pz_mean, pz_logsd, up_context, down_context,
                    eps, ar_params, n_h, kl_min):
    """tf_train.py:56-85 (mode == "train"): the IAF step with its log-det / KL / free-bits
    accumulation, i.e. everything between the two out-of-scope convolutions.
    Returns dict(z, logqs, logps, kl_obj[B], kl_cost[B], z0, arw_logsd)."""
    prior_mean, prior_logvar = pz_mean, 2 * pz_logsd                           # :56
    post_mean, post_logvar = rz_mean + qz_mean, 2 * (rz_logsd + qz_logsd)      # :57
    context = up_context + down_context                                         # :58
    z0 = gaussian_diag_sample(post_mean, post_logvar, eps)                      # :63
    logqs = gaussian_diag_logps(post_mean, post_logvar, z0)                     # :68
    z, arw_logsd = iaf_step(z0, context, ar_params, n_h)                        # :69-71
    logqs = logqs + arw_logsd                                                   # :72
    logps = gaussian_diag_logps(prior_mean, prior_logvar, z)                    # :73
    kl = logqs - logps                                                          # :75
    n = z.shape[0]
    if kl_min > 0:                                                              # :77-82
        kl_ave = np.mean(np.sum(kl, axis=(2, 3)), axis=0, keepdims=True)
        kl_ave = np.maximum(kl_ave, kl_min)
        kl_ave = np.tile(kl_ave, [n, 1])
        kl_obj = np.sum(kl_ave, axis=1)
    else:                                                                       # :84
        kl_obj = np.sum(kl, axis=(1, 2, 3))
    kl_cost = np.sum(kl, axis=(1, 2, 3))                                        # :85
    return dict(z=z, logqs=logqs, logps=logps, kl_obj=kl_obj, kl_cost=kl_cost, z0=z0, arw_logsd=arw_logsd)


def iaf_layer_down(inp, params, qz_mean, qz_logsd, up_context, eps, z_size, h_size, kl_min, mode="train",
                   downsample=False, eps_prior=None):
    """tf_train.py:46-95.  mode "train": z0 = posterior sample (eps); "init": z0 = PRIOR sample (eps_prior), the IAF step
    and the KL run on it (:60-61, 67-85); "sample": z = prior sample, no IAF, kl = 0 (:60-61, 65-66).
    Returns (output, kl_obj, kl_cost, block)."""
    p = _sub(params, "down_conv1/")
    x = conv2d(elu(inp), p["V"], p["g"], p["b"])                                               # :52-53
    pz_mean, pz_logsd, rz_mean, rz_logsd, down_context, h_det = split_channels(
        x, [z_size] * 4 + [h_size] * 2)                                                        # :54
    if mode == "train":
        blk = posterior_block(qz_mean, qz_logsd, rz_mean, rz_logsd, pz_mean, pz_logsd, up_context, down_context,
                              eps, _sub(params, "ar_multiconv2d/"), [h_size, h_size], kl_min)
    else:
        z0 = gaussian_diag_sample(pz_mean, 2 * pz_logsd, eps_prior)                            # :60-61
        if mode == "sample":
            n = z0.shape[0]
            blk = dict(z=z0, kl_obj=np.zeros(n), kl_cost=np.zeros(n))                          # :65-66
        else:
            # the same lines as "train" with z0 from the prior: express it as the posterior noise that yields z0
            post_mean, post_logsd = rz_mean + qz_mean, rz_logsd + qz_logsd
            eps_eq = (z0 - post_mean) / np.exp(post_logsd)
            blk = posterior_block(qz_mean, qz_logsd, rz_mean, rz_logsd, pz_mean, pz_logsd, up_context, down_context,
                                  eps_eq, _sub(params, "ar_multiconv2d/"), [h_size, h_size], kl_min)
    h = elu(np.concatenate([blk["z"], h_det], axis=1))                                         # :87-88
    if downsample:
        inp = resize_nearest_neighbor(inp, 2)                                                  # :90
        p = _sub(params, "down_deconv2/")
        h = deconv2d(h, p["V"], p["g"], p["b"])                                                # :91
    else:
        p = _sub(params, "down_conv2/")
        h = conv2d(h, p["V"], p["g"], p["b"])                                                  # :93
    return inp + 0.1 * h, blk["kl_obj"], blk["kl_cost"], blk                                   # :94-95


# --------------------------------------------------------------------------------------
# a9  k-sample importance-weighted bound
# --------------------------------------------------------------------------------------
def logsumexp(x):
    """tf_utils/distributions.py:35-37 (over axis 1)."""
    x_max = np.max(x, axis=1, keepdims=True)
    return x_max.reshape([-1]) + np.log(np.sum(np.exp(x - x_max), axis=1))


def repeat(x, n):
    """tf_utils/distributions.py:40-52 == np.repeat(x, n, axis=0)."""
    if n == 1:
        return x
    idx = np.tile(np.arange(x.shape[0]).reshape([-1, 1]), [1, n]).reshape([-1])
    return x[idx]


def compute_lowerbound(log_pxz, sum_kl_costs, k=1):
    """tf_utils/distributions.py:55-62."""
    if k == 1:
        return sum_kl_costs - log_pxz
    log_pxz = np.reshape(log_pxz, [-1, k])
    sum_kl_costs = np.reshape(sum_kl_costs, [-1, k])
    return -(-np.log(float(k)) + logsumexp(log_pxz - sum_kl_costs))


def streaming_lowerbound(chunks, k):
    """Same quantity as compute_lowerbound(k>1) but consuming the k importance weights of each
    image in chunks (online max / rescaled sum), never materialising [n, k].
    chunks: iterable of (log_pxz - sum_kl) arrays of shape [n, k_chunk]; sum of k_chunk == k.
    This is OUR formulation for BASELINE config 5 (k = 10^4); it must equal the reference formula."""
    run_max = None
    run_sum = None
    seen = 0
    for c in chunks:
        c = np.asarray(c)
        m = np.max(c, axis=1)
        if run_max is None:
            run_max, run_sum = m, np.sum(np.exp(c - m[:, None]), axis=1)
        else:
            new_max = np.maximum(run_max, m)
            run_sum = run_sum * np.exp(run_max - new_max) + np.sum(np.exp(c - new_max[:, None]), axis=1)
            run_max = new_max
        seen += c.shape[1]
    assert seen == k
    return -(-np.log(float(k)) + run_max + np.log(run_sum))


def discretized_logistic(mean, logscale, sample, binsize=1 / 256.0):
    """tf_utils/distributions.py:28-32 (adjacent to the path)."""
    scale = np.exp(logscale)
    s = (np.floor(sample / binsize) * binsize - mean) / scale
    sig = lambda t: 1.0 / (1.0 + np.exp(-t))
    logp = np.log(sig(s + binsize / scale) - sig(s) + 1e-7)
    return np.sum(logp, axis=(1, 2, 3))


# --------------------------------------------------------------------------------------
# the caller of the path: the whole model's forward pass
# --------------------------------------------------------------------------------------
def cvae1_forward(x_uint8, params, z_size, h_size, depth, num_blocks, kl_min, k, noise, mode="train"):
    """tf_train.py:150-215, CVAE1._forward for ONE tower: x [B,3,S,S] uint8 -> (x_out, obj, loss).  `params` carries the TF names
    (x_enc/{V,g,b}, IAF_i_j/..., h_top, x_dec/{V,g,b}, dec_log_stdv); `noise` the draws of the DiagonalGaussians in graph
    order: top-down, per layer the prior's then the posterior's (distributions.py:15-24 draws at construction).
    Pinned to the reference's own _forward executed on the TF shim: tests/golden/cvae1_forward.npz."""
    x = np.clip((x_uint8.astype(np.float64) + 0.5) / 256.0, 0.0, 1.0) - 0.5                    # :153-154
    x = repeat(x, k)                                                                           # :159
    orig_x = x
    p = _sub(params, "x_enc/")
    h = conv2d(x, p["V"], p["g"], p["b"], stride=(2, 2))                                       # :183
    ups = {}
    for i in range(depth):                                                                     # :184-187
        for j in range(num_blocks):
            ds = i > 0 and j == 0                                                              # :180
            h, qm, ql, uc = iaf_layer_up(h, _sub(params, "IAF_%d_%d/" % (i, j)), z_size, h_size, downsample=ds)
            ups[(i, j)] = (qm, ql, uc)
    n = x.shape[0]
    hw = x.shape[2] // 2 ** depth                                                              # :192 (image_size / 2**len(layers))
    h = np.tile(np.asarray(params["h_top"]).reshape([1, -1, 1, 1]), [n, 1, hw, hw])            # :190-192
    kl_cost = np.zeros(n)
    kl_obj = np.zeros(n)
    it = iter(noise)
    for i in reversed(range(depth)):                                                           # :195-200
        for j in reversed(range(num_blocks)):
            eps_prior, eps_post = next(it), next(it)
            qm, ql, uc = ups[(i, j)]
            h, cur_obj, cur_cost, _ = iaf_layer_down(h, _sub(params, "IAF_%d_%d/" % (i, j)), qm, ql, uc, eps_post, z_size, h_size,
                                                     kl_min, mode=mode, downsample=(i > 0 and j == 0), eps_prior=eps_prior)
            kl_obj = kl_obj + cur_obj
            kl_cost = kl_cost + cur_cost
    p = _sub(params, "x_dec/")
    xo = deconv2d(elu(h), p["V"], p["g"], p["b"])                                              # :206-207
    xo = np.clip(xo, -0.5 + 1 / 512., 0.5 - 1 / 512.)                                          # :208
    log_pxz = discretized_logistic(xo, params["dec_log_stdv"], orig_x)                         # :210
    obj = np.sum(kl_obj - log_pxz)                                                             # :211
    loss = np.sum(compute_lowerbound(log_pxz, kl_cost, k))                                     # :218
    return xo, obj, loss


# --------------------------------------------------------------------------------------
# a13  data-parallel gradient averaging + Adamax
# --------------------------------------------------------------------------------------
def average_grads(tower_grads):
    """tf_utils/common.py:78-115, dense branch: per variable sum over towers then / N.
    tower_grads: list over towers of lists of arrays."""
    out = []
    for per_var in zip(*tower_grads):
        if len(per_var) == 1:
            out.append(per_var[0])
            continue
        g = np.array(per_var[0], copy=True)
        for t in per_var[1:]:
            g = g + t
        out.append(g / len(per_var))
    return out


def adamax_step(var, grad, slot_m, slot_v, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf_utils/adamax.py:40-56.  NB the reference's slot naming: "v" is the FIRST moment,
    "m" the infinity norm.  Returns (var, slot_m, slot_v).  Pinned bit-for-bit against the reference file itself
    (tests/golden/adamax.npz, tests/test_oracle_golden.py::test_adamax_matches_reference)."""
    v_t = beta1 * slot_v + (1.0 - beta1) * grad                 # :50
    m_t = np.maximum(beta2 * slot_m + eps, np.abs(grad))        # :52
    return var - lr * (v_t / m_t), m_t, v_t                     # :53-55


def ema_step(shadow, var, decay=0.999):
    """tf.train.ExponentialMovingAverage(decay=0.999).apply (tf_train.py:157-158), no num_updates:
    shadow -= (1 - decay) * (shadow - var)."""
    return shadow - (1.0 - decay) * (shadow - var)


# --------------------------------------------------------------------------------------
# a10-a12  Theano statement of the same operator.  Pinned against the reference's own ar.py / conv.py run on a NumPy
# Theano stand-in (tests/golden/theano_ar.npz); only the cuDNN conv primitive is supplied by the stand-in.
# --------------------------------------------------------------------------------------
def theano_ar_mask(n_in, n_out, ksize=3, zerodiagonal=True, flipmask=False, pad_channel=True):
    """graphy/nodes/ar.py:243-264.  OIHW with the optional border-indicator input channel."""
    _n_in = n_in + (1 if pad_channel else 0)
    l = (ksize - 1) // 2
    m = (ksize - 1) // 2
    mask = np.ones((n_out, _n_in, ksize, ksize), dtype=np.float32)
    mask[:, :, :l, :] = 0
    mask[:, :, l, :m] = 0
    if n_out >= n_in:
        assert n_out % n_in == 0
        k = n_out // n_in
        for i in range(n_in):
            mask[i * k:(i + 1) * k, i + 1:, l, m] = 0
            if zerodiagonal:
                mask[i * k:(i + 1) * k, i:i + 1, l, m] = 0
    else:
        assert n_in % n_out == 0
        k = n_in // n_out
        for i in range(n_out):
            mask[i:i + 1, (i + 1) * k:, l, m] = 0
            if zerodiagonal:
                mask[i:i + 1, i * k:(i + 1) * k, l, m] = 0
    if flipmask:
        mask = mask[::-1, ::-1, ::-1, ::-1]
    return mask


def theano_pad2dwithchannel(x, ksize=3):
    """graphy/nodes/conv.py:71-83: zero-pad H,W by (k-1)/2 and append a channel that is 1 on
    the padding ring and 0 inside."""
    a = (ksize - 1) // 2
    n, c, h, w = x.shape
    r = np.zeros((n, c + 1, h + 2 * a, w + 2 * a), dtype=x.dtype)
    r[:, c, :, :] = 1.0
    r[:, c, a:-a, a:-a] = 0.0
    r[:, :c, a:-a, a:-a] = x
    return r


def theano_ar_conv2d(h, w_, b_, s_, n_in, n_out, zerodiagonal=True, flipmask=False):
    """graphy/nodes/ar.py:304-330 (l2norm=True, logscale=True, pad_channel=True, 'valid').
    w_ OIHW [n_out, n_in+1, 3, 3].  dnn_conv default conv_mode='conv' => kernel FLIPPED."""
    ksize = w_.shape[2]
    mask = theano_ar_mask(n_in, n_out, ksize, zerodiagonal, flipmask, True)
    hp = theano_pad2dwithchannel(h, ksize)                                   # :309-310
    kerns = mask * w_                                                        # :312
    l = (ksize - 1) // 2
    if zerodiagonal:                                                         # :268-276
        kerns = kerns.copy()
        if n_out >= n_in:
            kerns[:n_out // n_in, :, l, l] = 0.0
        else:
            kerns[:1, :, l, l] = 0.0
    norm = np.sqrt(np.sum(kerns ** 2, axis=(1, 2, 3), keepdims=True)) + 1e-8  # :279-281
    kerns = kerns / norm
    kerns = kerns * np.exp(3.0 * s_).reshape([-1, 1, 1, 1])                  # :316-317 (logscale_scale=3)
    # true convolution, 'valid':  y[o,i,j] = sum_{c,a,b} hp[c, i+a, j+b] * kerns[o,c,K-1-a,K-1-b]
    kf = kerns[:, :, ::-1, ::-1]
    n, c, H, W = hp.shape
    oh, ow = H - ksize + 1, W - ksize + 1
    y = np.zeros((n, n_out, oh, ow), dtype=np.result_type(hp, kf))
    for a in range(ksize):
        for b in range(ksize):
            y += np.moveaxis(np.tensordot(hp[:, :, a:a + oh, b:b + ow], kf[:, :, a, b], axes=([1], [1])), 3, 1)
    return y + b_.reshape([1, -1, 1, 1])                                     # :329


def theano_multiconv2d(h, context, w, name, n_in, n_h, n_out, flipmask=False):
    """graphy/nodes/ar.py:378-416 with nl='elu' (graphy/nodes/__init__.py:174-175)."""
    sizes = [n_in] + list(n_h)
    for i in range(len(n_h)):
        p = "%s_%d" % (name, i)
        h = theano_ar_conv2d(h, w[p + "_w"], w[p + "_b"], w[p + "_s"], sizes[i], sizes[i + 1], False, flipmask)
        if i == 0:
            h = h + context
        h = np.where(h < 0, np.exp(np.minimum(h, 0)) - 1, h)
    out = []
    for i in range(len(n_out)):
        p = "%s_out_%d" % (name, i)
        out.append(theano_ar_conv2d(h, w[p + "_w"], w[p + "_b"], w[p + "_s"], sizes[-1], n_out[i], True, flipmask))
    return out


def theano_iaf2_nl(z, context, w, name, n_z, n_h, flipmask=False):
    """models.py:168-175 / 281-285: arw_mean*=.1; arw_logsd*=.1; z=(z-m)/exp(s); logps += s."""
    arw_mean, arw_logsd = theano_multiconv2d(z, context, w, name, n_z, n_h, [n_z, n_z], flipmask)
    arw_mean, arw_logsd = 0.1 * arw_mean, 0.1 * arw_logsd
    return (z - arw_mean) / np.exp(arw_logsd), arw_logsd


def theano_free_bits(kl, kl_min):
    """models.py:455-466: per layer kl [B,Z,H,W] -> scalar objective term."""
    kl_sum = kl.sum(axis=(1, 2, 3))
    if kl_min > 0:
        k = kl.sum(axis=(2, 3)).mean(axis=0)
        return np.maximum(kl_min, k).sum(), kl_sum
    return kl_sum, kl_sum


LOG2PI = math.log(2 * math.pi)


def theano_conv2d(h, w_, b_, s_):
    """graphy/nodes/conv.py:122-260 (the plain conv around the IAF step: pad_channel=True, 'valid', l2norm=True,
    logscale=True, no down/upsampling): kerns = w / sqrt(sum_{i,h,w} w^2) (NO epsilon, :163-165) * exp(3 s); true
    convolution on the zero-padded input with the border-indicator channel; + b."""
    ksize = w_.shape[2]
    hp = theano_pad2dwithchannel(h, ksize)
    kerns = w_ / np.sqrt(np.sum(w_ ** 2, axis=(1, 2, 3), keepdims=True))
    kerns = kerns * np.exp(3.0 * s_).reshape([-1, 1, 1, 1])
    kf = kerns[:, :, ::-1, ::-1]
    n, c, H, W = hp.shape
    oh, ow = H - ksize + 1, W - ksize + 1
    y = np.zeros((n, w_.shape[0], oh, ow), dtype=np.result_type(hp, kf))
    for a in range(ksize):
        for b in range(ksize):
            y += np.moveaxis(np.tensordot(hp[:, :, a:a + oh, b:b + ow], kf[:, :, a, b], axes=([1], [1])), 3, 1)
    return y + b_.reshape([1, -1, 1, 1])


def _theano_elu(h):
    return np.where(h < 0, np.exp(np.minimum(h, 0)) - 1, h)


def theano_cvae_layer(name, posterior, w, n_h, n_z, depth_ar, up_input, down_input, eps_up, eps_down, flipmask=False):
    """models.cvae_layer (models.py:14-345) with prior 'diag', nl 'elu', no downsampling, for the two posteriors BASELINE
    names: 'down_iaf2_nl' (:138-146 channel order, :201-210, 272-285, 295-298, 317-328) and 'up_iaf2_nl' (:168-176).
    Channel order of up_conv1: [h_det, qz_mean, qz_logsd, context]; of down_conv1: [h_det, pz_mean, pz_logsd | rz_mean,
    rz_logsd, down_context]; concat order [h_det, z] (:180, 318).  Returns dict(up_out, down_out, kl, z)."""
    cw = lambda nm: (w[name + nm + "_w"], w[name + nm + "_b"], w[name + nm + "_s"])
    pc = name + "_posterior_conv1"
    h = theano_conv2d(_theano_elu(up_input), *cw("_up_conv1_1"))                                   # :139
    h_det, qz_mean, qz_logsd = h[:, :n_h], h[:, n_h:n_h + n_z], h[:, n_h + n_z:n_h + 2 * n_z]        # :141-143
    context = h[:, n_h + 2 * n_z:n_h + 2 * n_z + n_h]
    res = {}
    if posterior == "up_iaf2_nl":                                                                  # :168-176
        z0 = qz_mean + np.exp(qz_logsd) * eps_up
        logqs = gaussian_diag_logps(qz_mean, 2 * qz_logsd, z0)
        z, arw_logsd = theano_iaf2_nl(z0, context, w, pc, n_z, [n_h] * depth_ar, flipmask)
        logqs = logqs + arw_logsd
        hu = np.concatenate([h_det, z], axis=1)
    else:                                                                                          # :180-182
        hu = h_det
    res["up_out"] = up_input + 0.1 * theano_conv2d(_theano_elu(hu), *cw("_up_conv2"))             # :192
    h = theano_conv2d(_theano_elu(down_input), *cw("_down_conv1"))                                 # :204-206
    ncp = n_h + 2 * n_z                                                                            # n_conv_down_prior
    if posterior == "down_iaf2_nl":                                                                # :272-285
        rz_mean, rz_logsd = h[:, ncp:ncp + n_z], h[:, ncp + n_z:ncp + 2 * n_z]
        post_mean, post_logvar = qz_mean + rz_mean, 2 * qz_logsd + 2 * rz_logsd
        z0 = post_mean + np.exp(0.5 * post_logvar) * eps_down
        logqs = gaussian_diag_logps(post_mean, post_logvar, z0)
        down_context = h[:, ncp + 2 * n_z:ncp + 2 * n_z + n_h]
        z, arw_logsd = theano_iaf2_nl(z0, context + down_context, w, pc, n_z, [n_h] * depth_ar, flipmask)
        logqs = logqs + arw_logsd
    pz_mean, pz_logsd = h[:, n_h:n_h + n_z], h[:, n_h + n_z:n_h + 2 * n_z]                          # :296-297
    logps = gaussian_diag_logps(pz_mean, 2 * pz_logsd, z)                                          # :298
    hd = np.concatenate([h[:, :n_h], z], axis=1)                                                   # :317-318
    res["down_out"] = down_input + 0.1 * theano_conv2d(_theano_elu(hd), *cw("_down_conv2_1"))      # :325
    res["kl"], res["z"] = logqs - logps, z
    res["up_conv1"], res["down_conv1"] = theano_conv2d(_theano_elu(up_input), *cw("_up_conv1_1")), h
    return res
