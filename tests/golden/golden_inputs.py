"""Deterministic synthetic inputs shared by make_golden.py (build container, runs the
reference through tf_shim) and by the tests (which rebuild the SAME inputs and compare the
oracle / HIP outputs with the stored reference outputs).  Only outputs are stored in the
.npz fixtures; inputs are regenerated from (case name -> seed) with NumPy's legacy
RandomState, whose stream is frozen across NumPy versions.

TEST INFRASTRUCTURE ONLY.
"""
import zlib

import numpy as np


def case_seed(name):
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def conv_params(rng, n_in, n_out, ksize=3, g_std=0.3, b_std=0.1):
    """Weight-normed conv variables in the reference's layout (tf_utils/layers.py:53-55):
    V HWIO [k,k,n_in,n_out] ~ N(0,0.05^2) (layers.py:40 init std), g [n_out], b [n_out]."""
    V = 0.05 * rng.standard_normal((ksize, ksize, n_in, n_out))
    g = g_std * rng.standard_normal((n_out,))
    b = b_std * rng.standard_normal((n_out,))
    return {"V": V, "g": g, "b": b}


def ar_multiconv2d_params(rng, n_z, n_h, n_out):
    """Variables of ar_multiconv2d (layers.py:159-166): layer_{i} then layer_out_{i}."""
    p = {}
    sizes = [n_z] + list(n_h)
    for i in range(len(n_h)):
        for k, v in conv_params(rng, sizes[i], sizes[i + 1]).items():
            p["layer_%d/%s" % (i, k)] = v
    for i, size in enumerate(n_out):
        for k, v in conv_params(rng, sizes[-1], size).items():
            p["layer_out_%d/%s" % (i, k)] = v
    return p


AR_CASES = {
    # name: (B, n_z, n_h list, H, W)
    "ar_tiny":      (2, 4, [8, 8], 5, 5),
    "ar_k_down":    (2, 8, [4, 4], 4, 6),       # n_h < n_z exercises the n_out<n_in mask branch
    "ar_depth1":    (3, 8, [16], 4, 4),          # depth_ar=1 (BASELINE config 1 structure)
    "ar_depth4":    (1, 8, [8, 8, 8, 8], 4, 4),  # depth_ar=4, n_h == n_z (config 4 structure)
    "ar_cfg2_8x8":  (1, 32, [160, 160], 8, 8),   # BASELINE config 2 channel sizes, level 1
    "ar_cfg1_4x4":  (2, 32, [64], 4, 4),         # BASELINE config 1 channel sizes, level 2
}


def ar_case_inputs(name):
    B, n_z, n_h, H, W = AR_CASES[name]
    rng = np.random.RandomState(case_seed(name))
    params = ar_multiconv2d_params(rng, n_z, n_h, [n_z, n_z])
    z = rng.standard_normal((B, n_z, H, W))
    context = rng.standard_normal((B, n_h[0], H, W))
    return dict(B=B, n_z=n_z, n_h=n_h, H=H, W=W, params=params, z=z, context=context)


LAYER_CASES = {
    # name: (B, z_size, h_size, H, W, kl_min, k)
    "layer_tiny_fb":   (3, 4, 8, 6, 6, 0.25, 1),
    "layer_tiny_nofb": (3, 4, 8, 6, 6, 0.0, 1),
    "layer_k2":        (2, 4, 8, 4, 4, 0.1, 2),
    "layer_cfg2_8x8":  (2, 32, 160, 8, 8, 0.25, 1),
}


def layer_case_inputs(name):
    """Variables + inputs for one IAFLayer (tf_train.py:23-95), downsample=False."""
    B, zs, hs, H, W, kl_min, k = LAYER_CASES[name]
    rng = np.random.RandomState(case_seed(name))
    p = {}
    for kk, v in conv_params(rng, hs, 2 * zs + 2 * hs).items():
        p["up_conv1/" + kk] = v
    for kk, v in conv_params(rng, hs, hs).items():
        p["up_conv3/" + kk] = v
    for kk, v in conv_params(rng, hs, 4 * zs + 2 * hs).items():
        p["down_conv1/" + kk] = v
    for kk, v in ar_multiconv2d_params(rng, zs, [hs, hs], [zs, zs]).items():
        p["ar_multiconv2d/" + kk] = v
    for kk, v in conv_params(rng, hs + zs, hs).items():
        p["down_conv2/" + kk] = v
    n = B * k
    up_input = rng.standard_normal((n, hs, H, W))
    down_input = rng.standard_normal((n, hs, H, W))
    eps_prior = rng.standard_normal((n, zs, H, W))       # drawn first (tf_train.py:56), unused in train mode
    eps_post = rng.standard_normal((n, zs, H, W))        # drawn second (tf_train.py:57)
    return dict(B=B, k=k, z_size=zs, h_size=hs, H=H, W=W, kl_min=kl_min, params=p,
                up_input=up_input, down_input=down_input, eps_prior=eps_prior, eps_post=eps_post)


# downsampling IAFLayer (tf_train.py:33,42-43,89-91) and the init / sample modes (tf_train.py:60-66)
LAYER_DS_CASES = {
    # name: (B, z_size, h_size, H, W of the up-pass INPUT, kl_min, k, mode, downsample)
    "layer_ds_tiny":     (2, 4, 8, 8, 8, 0.25, 1, "train", True),
    "layer_ds_cfg2":     (2, 32, 160, 16, 16, 0.25, 1, "train", True),
    "layer_tiny_init":   (3, 4, 8, 6, 6, 0.25, 1, "init", False),
    "layer_tiny_sample": (3, 4, 8, 6, 6, 0.25, 1, "sample", False),
    "layer_ds_init":     (2, 16, 32, 8, 8, 0.1, 1, "init", True),
    "layer_cfg2_init":   (2, 32, 160, 8, 8, 0.25, 1, "init", False),
}


def deconv_params(rng, n_in, n_out, k=3):
    """deconv2d variables (layers.py:83-112): V is [k, k, n_out, n_in]"""
    return {"V": 0.05 * rng.standard_normal((k, k, n_out, n_in)), "g": 0.1 * rng.standard_normal(n_out),
            "b": 0.1 * rng.standard_normal(n_out)}


def layer_ds_case_inputs(name):
    B, zs, hs, H, W, kl_min, k, mode, ds = LAYER_DS_CASES[name]
    rng = np.random.RandomState(case_seed(name))
    p = {}
    for kk, v in conv_params(rng, hs, 2 * zs + 2 * hs).items():
        p["up_conv1/" + kk] = v
    for kk, v in conv_params(rng, hs, hs).items():
        p["up_conv3/" + kk] = v
    for kk, v in conv_params(rng, hs, 4 * zs + 2 * hs).items():
        p["down_conv1/" + kk] = v
    for kk, v in ar_multiconv2d_params(rng, zs, [hs, hs], [zs, zs]).items():
        p["ar_multiconv2d/" + kk] = v
    if ds:
        for kk, v in deconv_params(rng, hs + zs, hs).items():
            p["down_deconv2/" + kk] = v
    else:
        for kk, v in conv_params(rng, hs + zs, hs).items():
            p["down_conv2/" + kk] = v
    n = B * k
    Hl, Wl = (H // 2, W // 2) if ds else (H, W)          # resolution of the latent / of the down-pass input
    up_input = rng.standard_normal((n, hs, H, W))
    down_input = rng.standard_normal((n, hs, Hl, Wl))
    eps_prior = rng.standard_normal((n, zs, Hl, Wl))
    eps_post = rng.standard_normal((n, zs, Hl, Wl))
    return dict(B=B, k=k, z_size=zs, h_size=hs, H=H, W=W, kl_min=kl_min, mode=mode, downsample=ds, params=p,
                up_input=up_input, down_input=down_input, eps_prior=eps_prior, eps_post=eps_post)


MASK_CASES = [
    (32, 160, False), (160, 160, False), (160, 32, True), (32, 64, False), (64, 32, True),
    (64, 64, False), (64, 64, True), (64, 128, False), (128, 64, True), (64, 192, False),
    (192, 64, True), (4, 8, False), (8, 4, True), (8, 8, False), (8, 8, True), (4, 4, True),
    (8, 16, True), (16, 8, False),
]


# ---------------------------------------------------------------- Theano statement (graphy/nodes/ar.py)
THEANO_CASES = {
    # name: (B, n_z, n_h list, H, W, flipmask)
    "th_cfg2_8x8": (2, 32, [160, 160], 8, 8, False),
    "th_cfg1_4x4": (3, 32, [64], 4, 4, False),
    "th_tiny": (2, 4, [8, 8], 5, 3, False),
    "th_deep": (2, 16, [16, 16, 16], 4, 4, False),
    "th_flip_cfg2_8x8": (2, 32, [160, 160], 8, 8, True),       # flipmask=True (ar.py:263-264)
    "th_flip_tiny": (2, 4, [8, 8], 5, 3, True),
    "th_flip_16_32": (3, 16, [32], 6, 5, True),
}
# the reference's cvae_layer (models.py:14-345) executed as a whole: (posterior, B, n_h, n_z, depth_ar, H, W, kl_min)
CVAE_CASES = {
    "cvae_down_iaf2_nl": ("down_iaf2_nl", 3, 32, 16, 2, 6, 5, 0.25),
    "cvae_down_iaf2_nl_64": ("down_iaf2_nl", 2, 64, 32, 2, 8, 8, 0.25),
    "cvae_up_iaf2_nl": ("up_iaf2_nl", 3, 32, 16, 2, 6, 5, 0.1),
    "cvae_up_iaf2_nl_d4": ("up_iaf2_nl", 2, 64, 64, 4, 4, 4, 0.0),     # BASELINE configs[3]: n_z=64, depth_ar=4
}
THEANO_NAME = "1_posterior_conv1"


def cvae_case_inputs(cname, shapes):
    """weights (in sorted key order, shapes as the reference's constructors created them), inputs and noise of one
    cvae_layer fixture -- regenerated from the case seed, like every other fixture's inputs"""
    posterior, B, n_h, n_z, depth_ar, H, W, kl_min = CVAE_CASES[cname]
    rng = np.random.RandomState(case_seed(cname))
    w = {}
    for k in sorted(shapes):
        w[k] = (0.05 if k.endswith("_w") else 0.1) * rng.standard_normal(tuple(int(v) for v in shapes[k]))
    up_input, down_input = rng.standard_normal((B, n_h, H, W)), rng.standard_normal((B, n_h, H, W))
    eps_up, eps_down = rng.standard_normal((B, n_z, H, W)), rng.standard_normal((B, n_z, H, W))
    return dict(w=w, up_input=up_input, down_input=down_input, eps_up=eps_up, eps_down=eps_down)


def theano_case_inputs(cname):
    """weights in the reference's own layout (ar.py:288-296): <name>_<i>_w [n_out, n_in+1, 3, 3], _b, _s; z; context"""
    B, n_z, n_h_list, H, W, _ = THEANO_CASES[cname]
    name = THEANO_NAME
    rng = np.random.RandomState(case_seed(cname))
    sizes = [n_z] + list(n_h_list)
    w = {}
    for i in range(len(n_h_list)):
        w["%s_%d_w" % (name, i)] = 0.05 * rng.standard_normal((sizes[i + 1], sizes[i] + 1, 3, 3))
        w["%s_%d_b" % (name, i)] = 0.1 * rng.standard_normal(sizes[i + 1])
        w["%s_%d_s" % (name, i)] = 0.1 * rng.standard_normal(sizes[i + 1])
    for i in range(2):
        w["%s_out_%d_w" % (name, i)] = 0.05 * rng.standard_normal((n_z, sizes[-1] + 1, 3, 3))
        w["%s_out_%d_b" % (name, i)] = 0.1 * rng.standard_normal(n_z)
        w["%s_out_%d_s" % (name, i)] = 0.1 * rng.standard_normal(n_z)
    z = rng.standard_normal((B, n_z, H, W))
    ctx = rng.standard_normal((B, sizes[-1] if n_h_list else n_z, H, W))
    return w, z, ctx


# ---------------------------------------------------------------- the whole model: CVAE1._forward (tf_train.py:150-215)
MODEL_CASES = {
    # name: (batch_size, k, z_size, h_size, depth, num_blocks, image_size, kl_min)
    "model_tiny": (2, 1, 4, 8, 2, 2, 16, 0.25),
    "model_k2":   (2, 2, 4, 8, 2, 1, 8, 0.1),
    "model_cfg":  (2, 1, 32, 160, 2, 2, 32, 0.25),     # BASELINE geometry (z 32, h 160, 32x32 images -> 16x16 -> 8x8), shallow
    "model_sample": (3, 1, 16, 32, 2, 2, 16, 0.25, "sample"),   # mode "sample" (tf_train.py:60-66): prior samples, kl = 0
}


# the data-dependent init pass: CVAE1(hps, "init") (tf_train.py:175: arg_scope(init=True), every layer in mode "init")
MODEL_INIT_CASES = {
    "model_init": (3, 1, 16, 32, 2, 2, 16, 0.25, "init"),
}


def model_case_inputs(name):
    """Variables (TF names, tf_train.py:175-215), the uint8 image batch and the noise every DiagonalGaussian draws, in the order
    the reference's graph construction draws it: top-down, per layer the prior's noise then the posterior's."""
    case = MODEL_CASES[name] if name in MODEL_CASES else MODEL_INIT_CASES[name]
    B, k, zs, hs, depth, nb, img, kl_min = case[:8]
    mode = case[8] if len(case) > 8 else "train"
    rng = np.random.RandomState(case_seed(name))
    p = {}
    for kk, v in conv_params(rng, 3, hs, ksize=5).items():
        p["x_enc/" + kk] = v
    for kk, v in deconv_params(rng, hs, 3, k=5).items():
        p["x_dec/" + kk] = v
    p["h_top"] = 0.3 * rng.standard_normal(hs)
    p["dec_log_stdv"] = np.array(-0.7 + 0.1 * rng.standard_normal())
    for i in range(depth):
        for j in range(nb):
            pre = "IAF_%d_%d/" % (i, j)
            ds = i > 0 and j == 0
            for kk, v in conv_params(rng, hs, 2 * zs + 2 * hs).items():
                p[pre + "up_conv1/" + kk] = v
            for kk, v in conv_params(rng, hs, hs).items():
                p[pre + "up_conv3/" + kk] = v
            for kk, v in conv_params(rng, hs, 4 * zs + 2 * hs).items():
                p[pre + "down_conv1/" + kk] = v
            for kk, v in ar_multiconv2d_params(rng, zs, [hs, hs], [zs, zs]).items():
                p[pre + "ar_multiconv2d/" + kk] = v
            last = deconv_params(rng, hs + zs, hs) if ds else conv_params(rng, hs + zs, hs)
            for kk, v in last.items():
                p[pre + ("down_deconv2/" if ds else "down_conv2/") + kk] = v
    x = rng.randint(0, 256, size=(B, 3, img, img)).astype(np.uint8)
    noise = []
    for i in reversed(range(depth)):
        H = img // 2 // (2 ** i)
        for j in reversed(range(nb)):
            noise.append(rng.standard_normal((B * k, zs, H, H)))      # prior (drawn, unused in mode "train")
            noise.append(rng.standard_normal((B * k, zs, H, H)))      # posterior
    return dict(B=B, k=k, z_size=zs, h_size=hs, depth=depth, num_blocks=nb, image_size=img, kl_min=kl_min, params=p, x=x, noise=noise,
                mode=mode)
