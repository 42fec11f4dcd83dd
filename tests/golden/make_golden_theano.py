#!/usr/bin/env python3
"""Generate tests/golden/theano_ar.npz by EXECUTING THE REFERENCE'S OWN THEANO-SIDE SOURCE for the masked-AR conv
(read-only, from /root/reference): graphy/nodes/ar.py (conv2d, multiconv2d), graphy/nodes/conv.py (pad2dwithchannel),
graphy/nodes/__init__.py (nonlinearity), graphy/nodes/rand.py (gaussian_diag).  The files are Python-2 syntax, so each
is passed through lib2to3 IN MEMORY (nothing is written or copied) and exec'd on theano_shim (NumPy leaves, cuDNN conv).
Python-2 integer division is supplied through Py2Int arguments.  Build container only:
    python tests/golden/make_golden_theano.py
TEST INFRASTRUCTURE ONLY."""
import contextlib
import io
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = os.environ.get("IAF_REFERENCE", "/root/reference")

import theano_shim as S            # noqa: E402
import golden_inputs as gi         # noqa: E402

P = S.Py2Int
theano, G = S.install()


def load_py2(modname, relpath, package_path=None):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor
        tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
        src = open(os.path.join(REF, relpath)).read()
        code = str(tool.refactor_string(src + "\n", relpath))
    m = types.ModuleType(modname)
    m.__file__ = os.path.join(REF, relpath)
    if package_path is not None:
        m.__path__ = package_path
    sys.modules[modname] = m
    exec(compile(code, m.__file__, "exec"), m.__dict__)
    return m


with contextlib.redirect_stdout(io.StringIO()):
    N = load_py2("graphy.nodes", "graphy/nodes/__init__.py", package_path=[])
    G.nodes = N
    N.conv = load_py2("graphy.nodes.conv", "graphy/nodes/conv.py")
    N.rand = load_py2("graphy.nodes.rand", "graphy/nodes/rand.py")
    N.ar = load_py2("graphy.nodes.ar", "graphy/nodes/ar.py")


class _NoiseQueue(object):
    """stands in for G.rng_curand (graphy/__init__.py:17-21): hands out the fixture's noise in call order"""
    def __init__(self):
        self.q = []

    def normal(self, size=None, **kw):
        a = self.q.pop(0)
        assert tuple(a.shape) == tuple(int(v) for v in size), (a.shape, size)
        return S.TT(a)


NOISE = _NoiseQueue()
G.rng_curand = NOISE


def gen_cvae_layers():
    """models.cvae_layer (models.py:14-345) itself -- up(), down_q() with posterior down_iaf2_nl / up_iaf2_nl -- on random
    weights; plus the free-bits lines of cvae1 (models.py:454-466, restated here: they sit inside the model function)."""
    with contextlib.redirect_stdout(io.StringIO()):
        models = load_py2("models", "models.py")
    out = {}
    for cname, (posterior, B, n_h, n_z, depth_ar, H, W, kl_min) in gi.CVAE_CASES.items():
        w = {}
        with contextlib.redirect_stdout(io.StringIO()):
            layer = models.cvae_layer("1", "diag", posterior, P(n_h), P(n_h), P(n_z), P(depth_ar), False, "elu", (P(3), P(3)),
                                      False, "nn", w)
        shapes = {k: w[k].a.shape for k in w}
        c = gi.cvae_case_inputs(cname, shapes)
        for k in sorted(w):
            w[k].set_value(c["w"][k])
            out["%s/w_shape/%s" % (cname, k)] = np.asarray(shapes[k], dtype=np.int64)
        up_input, down_input, eps_up, eps_down = c["up_input"], c["down_input"], c["eps_up"], c["eps_down"]
        NOISE.q[:] = [eps_up] + ([eps_down] if posterior.startswith("down") else [])
        with contextlib.redirect_stdout(io.StringIO()):
            up_out = layer.up(S.TT(up_input), w)
            down_out, kl = layer.down_q(S.TT(down_input), True, w)
        assert not NOISE.q
        kl = kl.a
        kl_sum = kl.sum(axis=(1, 2, 3))                                        # models.py:455
        if kl_min > 0:                                                         # :458-461
            obj_kl = np.maximum(np.asarray(kl_min), kl.sum(axis=(2, 3)).mean(axis=0)).sum()
        else:
            obj_kl = kl_sum                                                    # :466
        out.update({cname + "/up_out": up_out.a, cname + "/down_out": down_out.a,
                    cname + "/kl": kl, cname + "/kl_sum": kl_sum, cname + "/obj_kl": np.asarray(obj_kl)})
    path = os.path.join(HERE, "theano_cvae_layer.npz")
    np.savez_compressed(path, **out)
    print("wrote theano_cvae_layer.npz %.1f KiB, %d arrays" % (os.path.getsize(path) / 1024.0, len(out)))


def main():
    out = {}
    for cname, (B, n_z, n_h, H, W, flip) in gi.THEANO_CASES.items():
        name = gi.THEANO_NAME
        wvals, z, ctx = gi.theano_case_inputs(cname)
        w = {}
        with contextlib.redirect_stdout(io.StringIO()):
            # the reference call, models.py:92: N.ar.multiconv2d(name, n_z, depth_ar*[n_h2], [n_z,n_z], kernel, False, nl=nl, w=w)
            f = N.ar.multiconv2d(name, P(n_z), [P(v) for v in n_h], [P(n_z), P(n_z)], (P(3), P(3)), flip, nl="elu", w=w)
            for k, v in wvals.items():          # overwrite the random initial values with the fixture's
                assert w[k].a.shape == v.shape, (k, w[k].a.shape, v.shape)
                w[k].set_value(v)
            m_raw, s_raw = f(S.TT(z), S.TT(ctx), w)      # models.py:170, 281
        out[cname + "/m_raw"] = m_raw.a
        out[cname + "/s_raw"] = s_raw.a
    # one conv on its own, both mask variants x flipmask (ar.py:200-375), n_out > n_in and n_out < n_in
    for zd, flip, n_in, n_out in ((False, False, 8, 16), (True, False, 8, 16), (False, True, 16, 32), (True, True, 16, 32),
                                  (True, True, 32, 16), (False, False, 32, 16)):
        rng = np.random.RandomState(77 + zd + 2 * flip + n_in)
        B, H, W = 2, 4, 5
        w = {}
        with contextlib.redirect_stdout(io.StringIO()):
            f = N.ar.conv2d("c", P(n_in), P(n_out), (P(3), P(3)), zd, flip, w=w)
            wv = 0.05 * rng.standard_normal((n_out, n_in + 1, 3, 3))
            bv, sv = 0.1 * rng.standard_normal(n_out), 0.1 * rng.standard_normal(n_out)
            w["c_w"].set_value(wv); w["c_b"].set_value(bv); w["c_s"].set_value(sv)
            x = rng.standard_normal((B, n_in, H, W))
            y = f(S.TT(x), w)
        key = "conv_zd%d" % int(zd) if (not flip and n_in == 8) else "conv_zd%d_flip%d_%d_%d" % (int(zd), int(flip), n_in, n_out)
        out[key + "/w"], out[key + "/b"], out[key + "/s"], out[key + "/x"], out[key + "/y"] = wv, bv, sv, x, y.a
    # pad2dwithchannel (conv.py:71-83)
    x = np.arange(2 * 3 * 2 * 3, dtype=np.float64).reshape(2, 3, 2, 3)
    out["pad/x"] = x
    out["pad/y"] = N.conv.pad2dwithchannel(S.TT(x), (P(3), P(3))).a
    # gaussian_diag log-density (rand.py:78-87)
    rng = np.random.RandomState(5)
    mean, logvar, sample = rng.standard_normal((2, 3, 4)), 0.3 * rng.standard_normal((2, 3, 4)), rng.standard_normal((2, 3, 4))
    rv = N.rand.gaussian_diag(S.TT(mean), S.TT(logvar), S.TT(sample))
    out["gauss/mean"], out["gauss/logvar"], out["gauss/sample"], out["gauss/logps"] = mean, logvar, sample, rv.logps.a
    path = os.path.join(HERE, "theano_ar.npz")
    np.savez_compressed(path, **out)
    print("wrote theano_ar.npz %.1f KiB, %d arrays" % (os.path.getsize(path) / 1024.0, len(out)))


if __name__ == "__main__":
    main()
    gen_cvae_layers()
