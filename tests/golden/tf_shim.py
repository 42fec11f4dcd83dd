"""NumPy-backed stand-in for the pre-1.0 TensorFlow API subset that the reference's
IAF hot path touches.  TEST INFRASTRUCTURE ONLY (used by make_golden.py in the build
container, never on the GPU box and never by the product).

Why this exists: TensorFlow is not installable here (no network), but the reference's
hot-path *Python* is tiny and pure control flow over a handful of TF primitives.  By
installing this module as ``tensorflow`` in ``sys.modules`` the reference's own files

    /root/reference/tf_utils/layers.py         (masks, weight-norm conv, ar_multiconv2d)
    /root/reference/tf_utils/distributions.py  (DiagonalGaussian, logsumexp, compute_lowerbound)
    /root/reference/tf_utils/common.py         (split, average_grads)
    /root/reference/tf_train.py                (IAFLayer.up/.down)

are imported and executed UNMODIFIED; only the leaf primitives below are ours.  Each
primitive states the TF semantics it restates.  Dtype follows the inputs (feed float64
arrays -> float64 goldens), so goldens are the reference's control flow evaluated in
fp64.

Python-2 semantics the reference relies on are emulated without touching its source:
  * integer ``/``  -> ``Py2Int`` (an int subclass whose true-division floors), injected
    as the module-global ``int`` of layers.py and into decorated-function defaults;
  * list-returning ``map`` -> injected as a module-global ``map``.
"""
import builtins
import contextlib
import sys
import types
import unittest

import numpy as np


# --------------------------------------------------------------------------------------
# Python-2 integer semantics
# --------------------------------------------------------------------------------------
class Py2Int(int):
    """int whose ``/`` is Python-2 integer division when both operands are integral."""

    def _wrap(self, v):
        return Py2Int(v) if isinstance(v, int) and not isinstance(v, bool) else v

    def __truediv__(self, o):
        if isinstance(o, (int, np.integer)):
            return Py2Int(int(self) // int(o))
        return int(self) / o

    def __rtruediv__(self, o):
        if isinstance(o, (int, np.integer)):
            return Py2Int(int(o) // int(self))
        return o / int(self)

    def __add__(self, o): return self._wrap(int.__add__(self, o))
    def __radd__(self, o): return self._wrap(int.__radd__(self, o))
    def __sub__(self, o): return self._wrap(int.__sub__(self, o))
    def __rsub__(self, o): return self._wrap(int.__rsub__(self, o))
    def __mul__(self, o): return self._wrap(int.__mul__(self, o))
    def __rmul__(self, o): return self._wrap(int.__rmul__(self, o))
    def __floordiv__(self, o): return self._wrap(int.__floordiv__(self, o))
    def __rfloordiv__(self, o): return self._wrap(int.__rfloordiv__(self, o))
    def __mod__(self, o): return self._wrap(int.__mod__(self, o))
    def __rmod__(self, o): return self._wrap(int.__rmod__(self, o))
    def __neg__(self): return Py2Int(-int(self))


def py2ify(v):
    if isinstance(v, bool):
        return v
    if isinstance(v, int):
        return Py2Int(v)
    if isinstance(v, tuple):
        return tuple(py2ify(e) for e in v)
    if isinstance(v, list):
        return [py2ify(e) for e in v]
    return v


def py2_map(f, *seqs):
    return list(builtins.map(f, *seqs))


# --------------------------------------------------------------------------------------
# Tensor: ndarray with the few TF methods the reference calls
# --------------------------------------------------------------------------------------
class _Shape(object):
    def __init__(self, shp):
        self._s = tuple(int(d) for d in shp)

    def __getitem__(self, i):
        return self._s[i]

    def __iter__(self):
        return iter(self._s)

    def __len__(self):
        return len(self._s)

    def as_list(self):
        return list(self._s)


class Tensor(np.ndarray):
    def get_shape(self):
        return _Shape(self.shape)

    def set_shape(self, shape):
        assert tuple(int(s) for s in shape) == tuple(self.shape), (shape, self.shape)

    def eval(self, *a, **k):
        return np.asarray(self)

    def initialized_value(self):
        return self


def T(x, dtype=None):
    a = np.asarray(x, dtype=dtype)
    return a.view(Tensor)


# --------------------------------------------------------------------------------------
# variable store / scopes  (tf.get_variable, tf.variable_scope)
# --------------------------------------------------------------------------------------
class VariableStore(object):
    def __init__(self):
        self.vars = {}
        self.scope = []
        self.rng = np.random.RandomState(1234)
        self.noise_log = []          # every tf.random_normal draw, in call order
        self.noise_queue = []        # if non-empty, tf.random_normal pops from here
        self.default_dtype = np.float64

    def full(self, name):
        return "/".join(self.scope + [name])


STORE = VariableStore()


@contextlib.contextmanager
def variable_scope(name, *a, **k):
    STORE.scope.append(name)
    try:
        yield
    finally:
        STORE.scope.pop()


@contextlib.contextmanager
def name_scope(name, *a, **k):
    yield


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True):
    full = STORE.full(name)
    if full in STORE.vars:
        v = STORE.vars[full]
        if shape is not None:
            assert tuple(int(s) for s in shape) == tuple(v.shape), (full, shape, v.shape)
        return T(v)
    if initializer is None:
        raise KeyError("shim variable store has no %r (seed it before the call)" % full)
    if callable(initializer):
        val = initializer([int(s) for s in shape])
    else:
        val = np.asarray(initializer)
    STORE.vars[full] = np.array(val, dtype=STORE.default_dtype)
    return T(STORE.vars[full])


def random_normal_initializer(mean=0.0, stddev=1.0, dtype=None, seed=None):
    def init(shape):
        return mean + stddev * STORE.rng.standard_normal(shape)
    return init


def zeros_initializer(shape, dtype=None):
    return np.zeros(shape)


def constant(value, dtype=None, shape=None, name=None):
    return T(value)


# --------------------------------------------------------------------------------------
# leaf math (each restates documented TF semantics)
# --------------------------------------------------------------------------------------
def _same_pad(n, k, s):
    # TF "SAME": out = ceil(n/s); pad_total = max((out-1)*s + k - n, 0); before = total//2
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return out, tot // 2, tot - tot // 2


def nn_conv2d(x, w, strides, padding, data_format="NHWC", name=None):
    """tf.nn.conv2d: cross-correlation (no kernel flip), filter HWIO, here NCHW only."""
    assert data_format == "NCHW" and padding == "SAME"
    x = np.asarray(x)
    w = np.asarray(w)
    sh, sw = int(strides[2]), int(strides[3])
    n, c, hh, ww = x.shape
    kh, kw, ci, co = w.shape
    assert ci == c
    oh, pt, pb = _same_pad(hh, kh, sh)
    ow, pl, pr = _same_pad(ww, kw, sw)
    xp = np.zeros((n, c, hh + pt + pb, ww + pl + pr), dtype=np.result_type(x, w))
    xp[:, :, pt:pt + hh, pl:pl + ww] = x
    y = np.zeros((n, co, oh, ow), dtype=xp.dtype)
    for a in range(kh):
        for b in range(kw):
            patch = xp[:, :, a:a + (oh - 1) * sh + 1:sh, b:b + (ow - 1) * sw + 1:sw]
            y += np.einsum("nchw,co->nohw", patch, w[a, b])
    return T(y)


def nn_conv2d_transpose(x, filters, output_shape, strides, padding="SAME", data_format="NHWC", name=None):
    """tf.nn.conv2d_transpose = the gradient of tf.nn.conv2d with respect to its input: x NHWC [n,h,w,c_in], filters
    [kh,kw,c_out,c_in], result NHWC `output_shape`.  Written as the adjoint (scatter) of the forward conv whose input is
    the result: forward output (i,j) reads input (i*s + a - pad_t, j*s + b - pad_l), so x[i,j]*f[a,b] lands there."""
    assert data_format == "NHWC" and padding == "SAME"
    x = np.asarray(x)
    f = np.asarray(filters)
    n, hh, ww, ci = x.shape
    kh, kw, co, ci2 = f.shape
    assert ci2 == ci
    sh, sw = int(strides[1]), int(strides[2])
    on, oh, ow, oc = [int(v) for v in output_shape]
    assert on == n and oc == co
    fh, pt, _ = _same_pad(oh, kh, sh)
    fw, pl, _ = _same_pad(ow, kw, sw)
    assert (fh, fw) == (hh, ww), "output_shape inconsistent with the forward conv"
    y = np.zeros((n, oh, ow, co), dtype=np.result_type(x, f))
    for i in range(hh):
        for j in range(ww):
            for a in range(kh):
                for b in range(kw):
                    yy, xx = i * sh + a - pt, j * sw + b - pl
                    if 0 <= yy < oh and 0 <= xx < ow:
                        y[:, yy, xx, :] += x[:, i, j, :] @ f[a, b].T
    return T(y)


def image_resize_nearest_neighbor(images, size, align_corners=False, name=None):
    """tf.image.resize_nearest_neighbor (align_corners=False): out[y,x] = in[min(floor(y*in_h/out_h), in_h-1), ...], NHWC"""
    assert not align_corners
    x = np.asarray(images)
    n, hh, ww, c = x.shape
    oh, ow = int(size[0]), int(size[1])
    iy = np.minimum(np.floor(np.arange(oh) * (hh / float(oh))).astype(int), hh - 1)
    ix = np.minimum(np.floor(np.arange(ow) * (ww / float(ow))).astype(int), ww - 1)
    return T(x[:, iy][:, :, ix])


def nn_l2_normalize(x, dim, epsilon=1e-12, name=None):
    """tf.nn.l2_normalize: x * rsqrt(max(sum(x**2, dim, keepdims), epsilon))."""
    x = np.asarray(x)
    ss = np.sum(np.square(x), axis=tuple(int(d) for d in dim), keepdims=True)
    return T(x / np.sqrt(np.maximum(ss, epsilon)))


def nn_elu(x, name=None):
    x = np.asarray(x)
    return T(np.where(x > 0, x, np.expm1(np.minimum(x, 0))))


def nn_moments(x, axes, name=None, keep_dims=False):
    x = np.asarray(x)
    ax = tuple(int(a) for a in axes)
    m = x.mean(axis=ax, keepdims=keep_dims)
    v = x.var(axis=ax, keepdims=keep_dims)
    return T(m), T(v)


def _axes(a):
    if a is None:
        return None
    if isinstance(a, (list, tuple)):
        return tuple(int(i) for i in a)
    return int(a)


def reduce_sum(x, reduction_indices=None, keep_dims=False, name=None):
    return T(np.sum(np.asarray(x), axis=_axes(reduction_indices), keepdims=keep_dims))


def reduce_mean(x, reduction_indices=None, keep_dims=False, name=None):
    return T(np.mean(np.asarray(x), axis=_axes(reduction_indices), keepdims=keep_dims))


def reduce_max(x, reduction_indices=None, keep_dims=False, name=None):
    return T(np.max(np.asarray(x), axis=_axes(reduction_indices), keepdims=keep_dims))


def random_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):
    shape = tuple(int(s) for s in np.asarray(shape).reshape(-1))
    if STORE.noise_queue:
        eps = np.asarray(STORE.noise_queue.pop(0))
        assert eps.shape == shape, (eps.shape, shape)
    else:
        eps = STORE.rng.standard_normal(shape)
    STORE.noise_log.append(np.array(eps))
    return T(mean + stddev * eps)


def concat(concat_dim, values, name=None):
    """pre-1.0 signature: tf.concat(dim, values)."""
    return T(np.concatenate([np.asarray(v) for v in values], axis=int(concat_dim)))


def split(split_dim, num_split, value, name=None):
    """pre-1.0 signature: tf.split(dim, num, value)."""
    return [T(a) for a in np.split(np.asarray(value), int(num_split), axis=int(split_dim))]


def slice_(x, begin, size, name=None):
    x = np.asarray(x)
    idx = []
    for b, s, n in zip(begin, size, x.shape):
        b = int(b)
        s = int(s)
        idx.append(slice(b, n if s == -1 else b + s))
    return T(x[tuple(idx)])


def tile(x, multiples, name=None):
    return T(np.tile(np.asarray(x), tuple(int(m) for m in multiples)))


def gather(x, idx, name=None):
    return T(np.asarray(x)[np.asarray(idx)])


class IndexedSlices(object):
    def __init__(self, values, indices, dense_shape=None):
        self.values, self.indices, self.dense_shape = values, indices, dense_shape


class NodeDef(object):
    pass


class _Flags(object):
    class _F(object):
        pass

    def __init__(self):
        self.FLAGS = self._F()

    def _def(self, name, default, doc=""):
        setattr(self.FLAGS, name, default)

    DEFINE_string = DEFINE_integer = DEFINE_boolean = DEFINE_float = _def


class TensorFlowTestCase(unittest.TestCase):
    @contextlib.contextmanager
    def test_session(self, *a, **k):
        yield None

    def assertAllClose(self, a, b, rtol=1e-6, atol=1e-6):
        np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


# --------------------------------------------------------------------------------------
# arg_scope (tensorflow.contrib.framework): scoped default kwargs for decorated functions
# --------------------------------------------------------------------------------------
_ARG_STACK = [{}]


@contextlib.contextmanager
def arg_scope(list_ops, **kwargs):
    cur = {k: dict(v) for k, v in _ARG_STACK[-1].items()}
    for op in list_ops:
        key = getattr(op, "_arg_scope_key", op)
        cur.setdefault(key, {}).update(kwargs)
    _ARG_STACK.append(cur)
    try:
        yield cur
    finally:
        _ARG_STACK.pop()


def add_arg_scope(fn):
    # Python-2 integer semantics for literal defaults such as filter_size=(3, 3)
    if fn.__defaults__:
        fn.__defaults__ = tuple(py2ify(d) for d in fn.__defaults__)

    def wrapper(*args, **kwargs):
        merged = dict(_ARG_STACK[-1].get(wrapper, {}))
        merged.update(kwargs)
        return fn(*args, **merged)

    wrapper._arg_scope_key = wrapper
    wrapper.__wrapped__ = fn
    wrapper.__name__ = fn.__name__
    return wrapper


# --------------------------------------------------------------------------------------
# module assembly
# --------------------------------------------------------------------------------------
def _unary(f):
    return lambda x, name=None: T(f(np.asarray(x)))


def build_modules():
    tf = types.ModuleType("tensorflow")
    tf.float32, tf.float16, tf.int32, tf.uint8 = np.float32, np.float16, np.int32, np.uint8
    tf.get_variable = get_variable
    tf.variable_scope = variable_scope
    tf.name_scope = name_scope
    tf.random_normal_initializer = random_normal_initializer
    tf.zeros_initializer = zeros_initializer
    tf.constant = constant
    tf.exp = _unary(np.exp)
    tf.log = _unary(np.log)
    tf.sqrt = _unary(np.sqrt)
    tf.square = _unary(np.square)
    tf.abs = _unary(np.abs)
    tf.floor = _unary(np.floor)
    tf.sigmoid = _unary(lambda x: 1.0 / (1.0 + np.exp(-x)))
    tf.to_float = _unary(lambda x: x.astype(STORE.default_dtype))
    tf.maximum = lambda a, b, name=None: T(np.maximum(np.asarray(a), np.asarray(b)))
    tf.clip_by_value = lambda x, lo, hi, name=None: T(np.clip(np.asarray(x), lo, hi))
    tf.reshape = lambda x, shape, name=None: T(np.reshape(np.asarray(x), [int(s) for s in shape]))
    tf.transpose = lambda x, perm=None, name=None: T(np.transpose(np.asarray(x), perm))
    tf.shape = lambda x, name=None: T(np.array(np.asarray(x).shape, dtype=np.int32))
    tf.range = lambda *a, **k: T(np.arange(*[int(np.asarray(v)) for v in a]))
    tf.zeros = lambda shape, dtype=None, name=None: T(np.zeros([int(s) for s in shape], STORE.default_dtype))
    tf.add_n = lambda xs, name=None: T(sum(np.asarray(x) for x in xs))
    tf.reduce_sum, tf.reduce_mean, tf.reduce_max = reduce_sum, reduce_mean, reduce_max
    tf.random_normal = random_normal
    tf.concat, tf.split, tf.slice, tf.tile, tf.gather = concat, split, slice_, tile, gather
    tf.IndexedSlices, tf.NodeDef = IndexedSlices, NodeDef
    tf.flags = _Flags()
    tf.no_op = lambda *a, **k: None
    tf.placeholder = lambda dtype, shape=None, name=None: None

    nn = types.ModuleType("tensorflow.nn")
    nn.conv2d, nn.l2_normalize, nn.elu, nn.moments = nn_conv2d, nn_l2_normalize, nn_elu, nn_moments
    nn.conv2d_transpose = nn_conv2d_transpose
    tf.nn = nn
    image = types.ModuleType("tensorflow.image")
    image.resize_nearest_neighbor = image_resize_nearest_neighbor
    tf.image = image

    test = types.ModuleType("tensorflow.test")
    test_util = types.ModuleType("tensorflow.test.test_util")
    test_util.TensorFlowTestCase = TensorFlowTestCase
    test.test_util = test_util
    test.main = unittest.main
    tf.test = test

    train = types.ModuleType("tensorflow.train")
    train.Supervisor = object          # only subclassed at import time (common.py:173)
    tf.train = train
    app = types.ModuleType("tensorflow.app")
    app.run = lambda *a, **k: None
    tf.app = app

    mods = {"tensorflow": tf, "tensorflow.nn": nn, "tensorflow.test": test, "tensorflow.train": train}
    # tensorflow.contrib.framework.python.ops  (arg_scope, add_arg_scope)
    chain = ["tensorflow.contrib", "tensorflow.contrib.framework", "tensorflow.contrib.framework.python",
             "tensorflow.contrib.framework.python.ops"]
    parent = tf
    for full in chain:
        m = types.ModuleType(full)
        setattr(parent, full.rsplit(".", 1)[1], m)
        mods[full] = m
        parent = m
    parent.arg_scope, parent.add_arg_scope = arg_scope, add_arg_scope
    return mods


def install():
    """Put the fake tensorflow tree into sys.modules.  Returns the fake ``tf``."""
    mods = build_modules()
    sys.modules.update(mods)
    return mods["tensorflow"]
