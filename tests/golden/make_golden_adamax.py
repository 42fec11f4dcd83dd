#!/usr/bin/env python3
"""Generate tests/golden/adamax.npz by EXECUTING THE REFERENCE'S OWN tf_utils/adamax.py (read-only, from
/root/reference): AdamaxOptimizer._prepare / _create_slots / _apply_dense run for a few steps on NumPy-backed
variables.  TensorFlow is absent; the handful of TF names the file touches are stubbed below (identity casts, a
variable with .assign, a slot dictionary in the Optimizer base).  Build container only.  TEST INFRASTRUCTURE ONLY."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("IAF_REFERENCE", "/root/reference")


class Var(object):
    """tf.Variable stand-in: value + assign (returns the new value, like the assign op's output)"""
    class _DT(object):
        base_dtype = "float64"
    dtype = _DT()

    def __init__(self, a):
        self.a = np.array(a, dtype=np.float64)

    def assign(self, v):
        self.a = np.array(v, dtype=np.float64)
        return self.a

    def __mul__(self, o): return self.a * o
    __rmul__ = __mul__
    def __add__(self, o): return self.a + o
    __radd__ = __add__


def install_stubs():
    tf = types.ModuleType("tensorflow")
    tf.float16 = "float16"
    tf.maximum = np.maximum
    tf.abs = np.abs
    python = types.ModuleType("tensorflow.python")
    ops_pkg = types.ModuleType("tensorflow.python.ops")
    cfo = types.ModuleType("tensorflow.python.ops.control_flow_ops")
    cfo.group = lambda *a: a
    mo = types.ModuleType("tensorflow.python.ops.math_ops")
    mo.cast = lambda x, dtype: x
    so = types.ModuleType("tensorflow.python.ops.state_ops")
    so.assign_sub = lambda var, delta: var.assign(var.a - delta)
    fw = types.ModuleType("tensorflow.python.framework")
    fops = types.ModuleType("tensorflow.python.framework.ops")
    fops.convert_to_tensor = lambda x, name=None: x
    tr = types.ModuleType("tensorflow.python.training")
    opt = types.ModuleType("tensorflow.python.training.optimizer")

    class Optimizer(object):
        """the three base-class services adamax.py uses: name, zero slots, slot lookup"""
        def __init__(self, use_locking, name):
            self._name, self._slots = name, {}

        def _zeros_slot(self, var, slot_name, op_name):
            self._slots.setdefault(slot_name, {})[id(var)] = Var(np.zeros_like(var.a))

        def get_slot(self, var, name):
            return self._slots[name][id(var)]
    opt.Optimizer = Optimizer
    for name, m in (("tensorflow", tf), ("tensorflow.python", python), ("tensorflow.python.ops", ops_pkg),
                    ("tensorflow.python.ops.control_flow_ops", cfo), ("tensorflow.python.ops.math_ops", mo),
                    ("tensorflow.python.ops.state_ops", so), ("tensorflow.python.framework", fw),
                    ("tensorflow.python.framework.ops", fops), ("tensorflow.python.training", tr),
                    ("tensorflow.python.training.optimizer", opt)):
        sys.modules[name] = m


def main():
    install_stubs()
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_adamax", os.path.join(REF, "tf_utils", "adamax.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                      # the reference file, unmodified
    rng = np.random.RandomState(20240)
    var0 = rng.standard_normal(257)
    grads = rng.standard_normal((6, 257)) * np.array([1.0, 0.5, 2.0, 1e-9, 0.0, 3.0])[:, None]   # incl. tiny and zero steps
    lr = 0.01
    o = mod.AdamaxOptimizer(learning_rate=lr)          # tf_train.py:146: AdamaxOptimizer(lr)
    var = Var(var0)
    o._prepare()
    o._create_slots([var])
    out = {"var0": var0, "grads": grads, "lr": np.float64(lr)}
    for t in range(grads.shape[0]):
        o._apply_dense(grads[t], var)
        out["var_%d" % t] = var.a.copy()
        out["m_%d" % t] = o.get_slot(var, "m").a.copy()
        out["v_%d" % t] = o.get_slot(var, "v").a.copy()
    path = os.path.join(HERE, "adamax.npz")
    np.savez_compressed(path, **out)
    print("wrote adamax.npz %.1f KiB" % (os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
