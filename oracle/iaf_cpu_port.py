"""CPU PORT of the IAF step used ONLY as the timed cpu_baseline leg of bench.py and checked
against the oracle in tests -- TEST/BENCH INFRASTRUCTURE, NOT PRODUCT CODE.

PyTorch-CPU fp32 (oneDNN, multi-threaded) restatement of tf_utils/layers.py:31-64,115-166 +
tf_train.py:70-72.  The reference's own CPU path does not exist (the Theano convs are
cuDNN-only, TensorFlow is not installable here), so bench.py reports this with kind="port"."""
import numpy as np
import torch
import torch.nn.functional as F

from . import iaf_oracle as O


def prepare_weights(params, n_z, n_h):
    """layers.py:56-60 for every conv of the stack; returns OIHW fp32 weights + biases."""
    out = []
    names = ["layer_%d" % i for i in range(len(n_h))] + ["layer_out_0", "layer_out_1"]
    for nm in names:
        V = params[nm + "/V"]
        g = params[nm + "/g"]
        zerodiag = nm.startswith("layer_out")
        mask = torch.from_numpy(O.get_conv_ar_mask(3, 3, V.shape[2], V.shape[3], zerodiag))
        v = mask * V
        w = torch.exp(g).reshape(1, 1, 1, -1) * v / torch.sqrt(torch.clamp((v * v).sum(dim=(0, 1, 2), keepdim=True), min=1e-12))
        out.append((w.permute(3, 2, 0, 1).contiguous(), params[nm + "/b"]))
    return out


def iaf_step(z, context, weights, depth_ar):
    """layers.py:158-166 + tf_train.py:70-72 on prepared weights. Returns (z_new, logsd)."""
    x = z
    for i in range(depth_ar):
        w, b = weights[i]
        x = F.conv2d(x, w, b, padding=1)           # SAME, stride 1, cross-correlation
        if i == 0:
            x = x + context
        x = F.elu(x)
    (wm, bm), (ws, bs) = weights[depth_ar], weights[depth_ar + 1]
    m = F.conv2d(x, wm, bm, padding=1) * 0.1
    s = F.conv2d(x, ws, bs, padding=1) * 0.1
    return (z - m) / torch.exp(s), s


def as_torch(params):
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in params.items()}
