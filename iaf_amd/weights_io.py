"""Weight-file import for the engine ("weight file formats as import fixtures").

Theano side: graphy/ndict.py:205-236 stores a parameter dict as a gzipped tar with two members, in this order:
`arrays.npz` (positional `np.savez(*values)`, i.e. keys arr_0, arr_1, ...) and `names.txt` (one key per line), the
dict ordered by key (`ndict.ordered`, ndict.py:9-10).  `np_loadz` / `np_savez` read and write exactly that; the
helpers below pick one multiconv2d's parameters out of such a dict and hand them to `ARStack(variant="theano")`.

TF side: variables are addressed by scope name (`model/IAF_{i}_{j}/ar_multiconv2d/layer_0/V`, tf_train.py:157-159,
197); `tf_layer_params` filters a flat {name: array} dict (however it was exported from a checkpoint) down to one
IAFLayer's variables under the names `IAFLayer.load` expects.  The TF Saver's binary checkpoint format itself is not
parsed here (it needs TensorFlow)."""
import collections
import io
import os
import tarfile

import numpy as np


def np_savez(d, filename, addext=True):
    """graphy/ndict.py:208-228."""
    if addext:
        filename = filename + ".ndict.tar.gz"
    keys = sorted(d.keys())
    buf = io.BytesIO()
    np.savez(buf, *[np.asarray(d[k]) for k in keys])
    names = "".join("%s\n" % k for k in keys).encode()
    with tarfile.open(filename, "w:gz") as tar:
        for member, payload in (("arrays.npz", buf.getvalue()), ("names.txt", names)):
            info = tarfile.TarInfo(member)
            info.size = len(payload)
            tar.addfile(info, io.BytesIO(payload))
    return filename


def np_loadz(filename):
    """graphy/ndict.py:231-238: members[0] = arrays, members[1] = names; result ordered by key."""
    with tarfile.open(filename, "r:gz") as tar:
        members = tar.getmembers()
        if len(members) < 2:
            raise ValueError("%s: expected arrays.npz and names.txt" % filename)
        arrays = np.load(io.BytesIO(tar.extractfile(members[0]).read()))
        names = tar.extractfile(members[1]).read().decode().splitlines()
        if len(names) != len(arrays.files):
            raise ValueError("%s: %d names for %d arrays" % (filename, len(names), len(arrays.files)))
        result = {names[i]: arrays["arr_" + str(i)] for i in range(len(names))}
    return collections.OrderedDict(sorted(result.items()))


def theano_multiconv2d_params(w, name, depth_ar, n_out=2):
    """The entries of one N.ar.multiconv2d(name, ...) in a Theano parameter dict (graphy/nodes/ar.py:288-296, 388, 394):
    '<name>_<i>_w|_b|_s' and '<name>_out_<i>_w|_b|_s', re-keyed the way ARStack(variant="theano").prepare takes them."""
    out = {}
    for i in range(depth_ar):
        for suffix in ("_w", "_b", "_s"):
            out["%d%s" % (i, suffix)] = np.asarray(w["%s_%d%s" % (name, i, suffix)], dtype=np.float32)
    for i in range(n_out):
        for suffix in ("_w", "_b", "_s"):
            out["out_%d%s" % (i, suffix)] = np.asarray(w["%s_out_%d%s" % (name, i, suffix)], dtype=np.float32)
    return out


def tf_layer_params(variables, scope, ema=False):
    """{'<scope>/up_conv1/V': ..} -> {'up_conv1/V': ..} for one IAFLayer scope such as 'model/IAF_0_3'.
    ema=True picks the shadow variables '<var>/ExponentialMovingAverage' (tf_train.py:157-159) instead."""
    prefix = scope.rstrip("/") + "/"
    tail = "/ExponentialMovingAverage"
    out = {}
    for k, v in variables.items():
        if not k.startswith(prefix):
            continue
        k = k[len(prefix):]
        if k.endswith(":0"):
            k = k[:-2]
        if ema != k.endswith(tail):
            continue
        if ema:
            k = k[:-len(tail)]
        out[k] = np.asarray(v, dtype=np.float32)
    return out


def exists(filename):
    return os.path.exists(filename) or os.path.exists(filename + ".ndict.tar.gz")
