"""A NumPy stand-in for the handful of Theano leaf operations the reference's masked-AR conv code touches
(graphy/nodes/ar.py, graphy/nodes/conv.py: pad2dwithchannel, graphy/nodes/__init__.py: nonlinearity,
graphy/nodes/rand.py: gaussian_diag).  TEST INFRASTRUCTURE ONLY, used by make_golden_theano.py in the build container
to EXECUTE THE REFERENCE'S OWN SOURCE eagerly (after an in-memory lib2to3 pass: the files are Python-2 syntax) and
record its outputs as fixtures.  Everything is float64 and eager: a "symbolic" tensor is a value.

What is provided here rather than by the reference: `dnn_conv` (cuDNN: conv_mode='conv' = true convolution,
border_mode 'valid', graphy/nodes/conv.py:9-13), `theano.shared`, the three helpers of graphy/__init__.py that need no
Theano machinery (floatX, sharedf :24-28, Struct :35-39), and elementwise / reduction leaves."""
import sys
import types

import numpy as np


class Py2Int(int):
    """Python-2 integer `/` (ar.py:241-242, 250-258; conv.py:75-76)"""
    def __truediv__(self, o):
        return Py2Int(int(self) // int(o)) if isinstance(o, int) else int(self) / o

    def __rtruediv__(self, o):
        return Py2Int(int(o) // int(self)) if isinstance(o, int) else o / int(self)

    def __sub__(self, o):
        return Py2Int(int(self) - int(o)) if isinstance(o, int) else int(self) - o

    def __add__(self, o):
        return Py2Int(int(self) + int(o)) if isinstance(o, int) else int(self) + o


class _Tag(object):
    def __init__(self, owner):
        self._o = owner

    @property
    def test_value(self):
        return self._o.a


def _v(x):
    return x.a if isinstance(x, TT) else x


class TT(object):
    """eager tensor"""
    __array_ufunc__ = None      # numpy defers to our reflected operators

    def __init__(self, a, base=None):
        self.a = np.asarray(a) if np.asarray(a).dtype == bool else np.asarray(a, dtype=np.float64)
        self._base = base

    tag = property(lambda self: _Tag(self))
    shape = property(lambda self: tuple(self.a.shape))
    ndim = property(lambda self: self.a.ndim)

    def __getitem__(self, idx):
        return TT(self.a[idx], base=(self, idx))

    # arithmetic
    def __add__(self, o): return TT(self.a + _v(o))
    __radd__ = __add__
    def __iadd__(self, o): return TT(self.a + _v(o))
    def __sub__(self, o): return TT(self.a - _v(o))
    def __rsub__(self, o): return TT(_v(o) - self.a)
    def __isub__(self, o): return TT(self.a - _v(o))
    def __mul__(self, o): return TT(self.a * _v(o))
    __rmul__ = __mul__
    def __imul__(self, o): return TT(self.a * _v(o))
    def __truediv__(self, o): return TT(self.a / _v(o))
    def __rtruediv__(self, o): return TT(_v(o) / self.a)
    def __itruediv__(self, o): return TT(self.a / _v(o))
    def __pow__(self, o): return TT(self.a ** _v(o))
    def __neg__(self): return TT(-self.a)
    def __abs__(self): return TT(np.abs(self.a))
    def __lt__(self, o): return TT(self.a < _v(o))
    def __le__(self, o): return TT(self.a <= _v(o))
    def __gt__(self, o): return TT(self.a > _v(o))
    def __ge__(self, o): return TT(self.a >= _v(o))

    # reductions / shape ops
    def sum(self, axis=None, keepdims=False): return TT(self.a.sum(axis=axis, keepdims=keepdims))
    def mean(self, axis=None, keepdims=False): return TT(self.a.mean(axis=axis, keepdims=keepdims))
    def std(self, axis=None, keepdims=False): return TT(self.a.std(axis=axis, keepdims=keepdims))
    def max(self, axis=None, keepdims=False): return TT(self.a.max(axis=axis, keepdims=keepdims))

    def dimshuffle(self, *pattern):
        if len(pattern) == 1 and isinstance(pattern[0], (list, tuple)):
            pattern = tuple(pattern[0])
        perm = [p for p in pattern if p != 'x']
        a = np.transpose(self.a, perm)
        idx = tuple(None if p == 'x' else slice(None) for p in pattern)
        return TT(a[idx])

    def flatten(self, ndim=1):
        return TT(self.a.reshape(self.a.shape[:ndim - 1] + (-1,)))

    # shared-variable face (theano.shared)
    def get_value(self, borrow=False):
        return self.a

    def set_value(self, v, borrow=False):
        self.a = np.asarray(v, dtype=np.float64)


def dnn_conv(img, kerns, border_mode='valid', subsample=(1, 1), conv_mode='conv', **_):
    """cuDNN convolution as the reference calls it: TRUE convolution (flipped kernel), 'valid' border, stride 1."""
    assert border_mode == 'valid' and tuple(subsample) == (1, 1) and conv_mode == 'conv'
    x, k = _v(img), _v(kerns)
    B, C, H, W = x.shape
    O, C2, kh, kw = k.shape
    assert C == C2
    kf = k[:, :, ::-1, ::-1]
    out = np.zeros((B, O, H - kh + 1, W - kw + 1))
    for u in range(kh):
        for v in range(kw):
            out += np.einsum('bchw,oc->bohw', x[:, :, u:u + H - kh + 1, v:v + W - kw + 1], kf[:, :, u, v])
    return TT(out)


def install():
    """registers fake `theano`, `theano.tensor`, `theano.sandbox.cuda.dnn` and `graphy` modules; returns (theano, graphy)"""
    theano = types.ModuleType("theano")
    theano.config = types.SimpleNamespace(device="gpu", floatX="float64")
    theano.shared = lambda x, **kw: TT(x)
    T = types.ModuleType("theano.tensor")
    T.sqrt = lambda x: TT(np.sqrt(_v(x)))
    T.exp = lambda x: TT(np.exp(_v(x)))
    T.log = lambda x: TT(np.log(_v(x)))
    T.tanh = lambda x: TT(np.tanh(_v(x)))
    T.maximum = lambda a, b: TT(np.maximum(_v(a), _v(b)))
    T.zeros = lambda shape, dtype=None: TT(np.zeros(tuple(int(s) for s in shape)))
    T.constant = lambda x: TT(x)
    T.switch = lambda c, a, b: TT(np.where(_v(c), _v(a), _v(b)))
    T.concatenate = lambda xs, axis=0: TT(np.concatenate([_v(x) for x in xs], axis=axis))

    def set_subtensor(sub, val):
        base, idx = sub._base
        new = base.a.copy()
        new[idx] = _v(val)
        return TT(new)
    T.set_subtensor = set_subtensor
    T.nnet = types.SimpleNamespace(softplus=lambda x: TT(np.logaddexp(0, _v(x))))
    theano.tensor = T
    sandbox = types.ModuleType("theano.sandbox")
    cuda = types.ModuleType("theano.sandbox.cuda")
    dnn = types.ModuleType("theano.sandbox.cuda.dnn")
    dnn.dnn_conv = dnn_conv
    dnn.dnn_pool = None
    sandbox.cuda = cuda
    cuda.dnn = dnn
    theano.sandbox = sandbox
    for name, m in (("theano", theano), ("theano.tensor", T), ("theano.sandbox", sandbox), ("theano.sandbox.cuda", cuda),
                    ("theano.sandbox.cuda.dnn", dnn)):
        sys.modules[name] = m

    G = types.ModuleType("graphy")
    G.__path__ = []
    G.floatX = "float64"
    G.sharedf = lambda x, target=None, name=None, borrow=False, broadcastable=None: TT(np.asarray(x, dtype=np.float64))

    class Struct(object):
        """graphy/__init__.py:35-39 (a Python-2 classic class: an instance attribute __call__ makes it callable)"""
        def __init__(self, **entries):
            self.__dict__.update(entries)

        def __call__(self, *a, **k):
            return self.__dict__["__call__"](*a, **k)
    G.Struct = Struct
    sys.modules["graphy"] = G
    return theano, G
