#!/usr/bin/env python3
"""Generate tests/golden/cvae1_forward.npz by EXECUTING THE REFERENCE'S OWN CVAE1._forward (tf_train.py:150-215, read-only from
/root/reference) on tf_shim, like make_golden.py does for the layers: python tests/golden/make_golden_model.py
The method is called unbound on a stand-in `self` carrying what it reads (hps, mode, dec_log_stdv); the constructor around it
(placeholders, towers, optimizer) is not on the path.  num_gpus = 2 with gpu = 0 keeps the summary branch (tf_train.py:204-208,
215-218: last tower only) out.  TEST INFRASTRUCTURE ONLY."""
import types

import numpy as np

import make_golden as MG          # installs the shim and imports the reference (its __main__ block does not run)
import golden_inputs as gi

tf_shim, TT, L, P, STORE = MG.tf_shim, MG.TT, MG.L, MG.P, MG.STORE


def gen_model():
    out = {}
    for name in gi.MODEL_CASES:
        c = gi.model_case_inputs(name)
        hps = TT.HParams(batch_size=P(c["B"]), k=P(c["k"]), z_size=P(c["z_size"]), h_size=P(c["h_size"]), kl_min=c["kl_min"],
                         depth=P(c["depth"]), num_blocks=P(c["num_blocks"]), image_size=P(c["image_size"]), num_gpus=P(2))
        MG.seed_store("", c["params"])
        STORE.noise_log[:] = []
        STORE.noise_queue[:] = list(c["noise"])
        me = types.SimpleNamespace(hps=hps, mode=c["mode"], dec_log_stdv=tf_shim.T(np.float64(c["params"]["dec_log_stdv"])))
        x_out, obj, loss = TT.CVAE1._forward(me, tf_shim.T(c["x"]), 0)
        assert not STORE.noise_queue and len(STORE.noise_log) == len(c["noise"])
        num_pixels = 3 * c["image_size"] ** 2
        out[name + "/x_out"], out[name + "/obj"], out[name + "/loss"] = x_out, obj, loss
        out[name + "/h_top_out"] = me.h_top
        # tf_train.py:133: bits_per_dim of ONE tower's loss
        out[name + "/bits_per_dim"] = np.asarray(loss) / (np.log(2.) * num_pixels * c["B"])
        print(name, "obj", float(np.asarray(obj)), "loss", float(np.asarray(loss)), "bits/dim", float(out[name + "/bits_per_dim"]))
    MG.save("cvae1_forward", **out)


def gen_model_init():
    """the init pass: only the V's (and h_top, dec_log_stdv) are seeded; every g and b is created by the reference's init branches"""
    out = {}
    for name in gi.MODEL_INIT_CASES:
        c = gi.model_case_inputs(name)
        hps = TT.HParams(batch_size=P(c["B"]), k=P(c["k"]), z_size=P(c["z_size"]), h_size=P(c["h_size"]), kl_min=c["kl_min"],
                         depth=P(c["depth"]), num_blocks=P(c["num_blocks"]), image_size=P(c["image_size"]), num_gpus=P(2))
        MG.seed_store("", {k: v for k, v in c["params"].items() if not (k.endswith("/g") or k.endswith("/b"))})
        STORE.noise_log[:] = []
        STORE.noise_queue[:] = list(c["noise"])
        me = types.SimpleNamespace(hps=hps, mode="init", dec_log_stdv=tf_shim.T(np.float64(c["params"]["dec_log_stdv"])))
        x_out, obj, loss = TT.CVAE1._forward(me, tf_shim.T(c["x"]), 0)
        assert not STORE.noise_queue
        out[name + "/x_out"] = x_out
        n = 0
        for k, v in STORE.vars.items():
            if k.endswith("/g") or k.endswith("/b"):
                out[name + "/var/" + k] = v
                n += 1
        print(name, "initialised", n, "g / b variables; x_out range", float(np.min(x_out)), float(np.max(x_out)))
    MG.save("cvae1_init", **out)


if __name__ == "__main__":
    gen_model()
    gen_model_init()
