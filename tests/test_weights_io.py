"""Weight-file import: the Theano ndict .tar.gz container (graphy/ndict.py:205-238) and TF scope-name
filtering.  The reference's ndict.py is Python-2 source and cannot be imported here, so the container layout is
checked structurally against what ndict.py:208-238 writes and reads (member order, positional arr_i keys, names.txt)."""
import io
import tarfile

import numpy as np

import golden_inputs as gi
from iaf_amd import weights_io
from oracle import iaf_oracle as O


def test_ndict_container_layout_and_round_trip(tmp_path):
    rng = np.random.RandomState(0)
    d = {"b_w": rng.standard_normal((3, 4)).astype(np.float32), "a_s": rng.standard_normal(5).astype(np.float32),
         "c/x": np.arange(6, dtype=np.int64).reshape(2, 3)}
    fn = weights_io.np_savez(d, str(tmp_path / "w"))
    assert fn.endswith(".ndict.tar.gz")                                   # ndict.py:210-211
    with tarfile.open(fn, "r:gz") as tar:
        members = tar.getmembers()
        assert [m.name for m in members] == ["arrays.npz", "names.txt"]   # ndict.py:212-213, 223-225; read by index :234-235
        names = tar.extractfile(members[1]).read().decode().splitlines()
        arrays = np.load(io.BytesIO(tar.extractfile(members[0]).read()))
    assert names == sorted(d)                                             # ordered(d), ndict.py:214
    assert arrays.files == ["arr_0", "arr_1", "arr_2"]                    # positional np.savez, ndict.py:216
    back = weights_io.np_loadz(fn)
    assert list(back) == sorted(d)
    for k in d:
        np.testing.assert_array_equal(back[k], d[k])
        assert back[k].dtype == d[k].dtype


def test_theano_stack_params_from_ndict_feed_the_oracle(tmp_path):
    """a Theano-named parameter dict survives the container and evaluates to the same multiconv2d output"""
    rng = np.random.RandomState(4)
    n_z, n_h, depth = 4, 8, 2
    w, sizes = {}, [n_z] + [n_h] * depth
    for i in range(depth):
        w["q_%d_w" % i] = (0.05 * rng.standard_normal((sizes[i + 1], sizes[i] + 1, 3, 3))).astype(np.float32)
        w["q_%d_b" % i] = (0.1 * rng.standard_normal(sizes[i + 1])).astype(np.float32)
        w["q_%d_s" % i] = (0.1 * rng.standard_normal(sizes[i + 1])).astype(np.float32)
    for i in range(2):
        w["q_out_%d_w" % i] = (0.05 * rng.standard_normal((n_z, n_h + 1, 3, 3))).astype(np.float32)
        w["q_out_%d_b" % i] = (0.1 * rng.standard_normal(n_z)).astype(np.float32)
        w["q_out_%d_s" % i] = (0.1 * rng.standard_normal(n_z)).astype(np.float32)
    w["unrelated"] = np.zeros(3, np.float32)
    fn = weights_io.np_savez(w, str(tmp_path / "model"))
    p = weights_io.theano_multiconv2d_params(weights_io.np_loadz(fn), "q", depth)
    assert sorted(p) == sorted(["0_w", "0_b", "0_s", "1_w", "1_b", "1_s", "out_0_w", "out_0_b", "out_0_s", "out_1_w",
                                "out_1_b", "out_1_s"])
    z, ctx = rng.standard_normal((2, n_z, 5, 5)), rng.standard_normal((2, n_h, 5, 5))
    f64 = lambda d: {k: v.astype(np.float64) for k, v in d.items()}
    a = O.theano_multiconv2d(z, ctx, f64({"q_" + k: v for k, v in p.items()}), "q", n_z, [n_h] * depth, [n_z, n_z])
    b = O.theano_multiconv2d(z, ctx, f64(w), "q", n_z, [n_h] * depth, [n_z, n_z])
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_tf_scope_filter_and_ema():
    c = gi.layer_case_inputs("layer_tiny_fb")
    flat = {}
    for k, v in c["params"].items():
        flat["model/IAF_0_3/" + k + ":0"] = v
        flat["model/IAF_0_3/" + k + "/ExponentialMovingAverage:0"] = v * 0.5
        flat["model/IAF_0_4/" + k + ":0"] = v * 2
    raw = weights_io.tf_layer_params(flat, "model/IAF_0_3")
    ema = weights_io.tf_layer_params(flat, "model/IAF_0_3", ema=True)
    assert sorted(raw) == sorted(c["params"]) == sorted(ema)
    for k, v in c["params"].items():
        np.testing.assert_array_equal(raw[k], v.astype(np.float32))
        np.testing.assert_array_equal(ema[k], (v * 0.5).astype(np.float32))


def test_split_mirror_of_common_split():
    """tf_utils/common.py:21-36 on host tensors: slices in order, assert on sizes that do not add up"""
    import pytest
    import torch
    import iaf_amd
    x = torch.arange(2 * 12 * 3, dtype=torch.float32).reshape(2, 12, 3)
    parts = iaf_amd.split(x, 1, [4, 4, 2, 2])
    assert [tuple(p.shape) for p in parts] == [(2, 4, 3), (2, 4, 3), (2, 2, 3), (2, 2, 3)]
    assert all(p.is_contiguous() for p in parts)
    assert torch.equal(torch.cat(parts, dim=1), x)
    with pytest.raises(AssertionError):
        iaf_amd.split(x, 1, [4, 4, 2])
