#!/usr/bin/env python3
"""GPU dev tool (round 6): the two-plane fp16 step kernels ("f16x2") against the fp64 oracle, next to bf16x3 and exact fp32, on the
BASELINE geometries and on the dynamic-range cases of tests/test_hip_dynamic_range.py; and their launch time (relaunched hot).
python tools/f16x2_check.py [--quick]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_inputs as gi  # noqa: E402
import iaf_amd  # noqa: E402
from oracle import iaf_oracle as O  # noqa: E402  (dev tool: the checker)

N_Z, N_H, D = 32, 160, 2
EPS32 = 2.0 ** -24
ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true")
a = ap.parse_args()


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy().astype(np.float64)


def f32(x):
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def oracle_raw(z, ctx, params, chunk=8):
    p32 = {k: f32(v) for k, v in params.items()}
    em, es = [], []
    for b0 in range(0, z.shape[0], chunk):
        m_, s_ = O.ar_multiconv2d(f32(z[b0:b0 + chunk]), f32(ctx[b0:b0 + chunk]), p32, [N_H] * D, [N_Z, N_Z])
        em.append(m_); es.append(s_)
    return np.concatenate(em), np.concatenate(es)


def stacks(params):
    dp = {k: dev(v) for k, v in params.items()}
    out = []
    for prec in ("f32", "bf16x3", "f16x2"):
        st = iaf_amd.ARStack(N_Z, [N_H] * D)
        st.set_precision(prec)
        st.prepare(dp)
        out.append((prec, st, dp))
    return out


def case(label, params, z, ctx, H):
    em, es = oracle_raw(z, ctx, params)
    scale = max(np.abs(em).max(), np.abs(es).max())
    errs = {}
    for prec, st, dp in stacks(params):
        m, s = st.ar_multiconv2d(dev(z), dev(ctx))
        hm, hs = host(m), host(s)
        errs[prec] = max(np.abs(hm - em).max(), np.abs(hs - es).max()) if np.isfinite(hm).all() and np.isfinite(hs).all() else float("inf")
        if prec == "f16x2":
            errs["range_word"] = st.range_errors()
    bound = 2.0 * errs["f32"] + 4.0 * EPS32 * scale
    print("%-46s scale %9.3g  f32 %9.3g  bf16x3 %9.3g  f16x2 %9.3g (x%.2f of f32; bound %9.3g %s) range word %d" % (
        label, scale, errs["f32"], errs["bf16x3"], errs["f16x2"], errs["f16x2"] / max(errs["f32"], 1e-30), bound,
        "ok" if errs["f16x2"] <= bound else "FAIL", errs["range_word"]), flush=True)
    return errs


def main():
    print(torch.cuda.get_device_name(0))
    for H, B in ((16, 32), (8, 32)):
        rng = np.random.RandomState(0)
        params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
        z, ctx = rng.standard_normal((B, N_Z, H, H)), rng.standard_normal((B, N_H, H, H))
        case("baseline %dx%d B=%d" % (H, H, B), params, z, ctx, H)
    if not a.quick:
        for H in (16, 8):
            for scale in (1e-3, 1e3, 1e5):
                rng = np.random.RandomState(900 + H)
                params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
                B = 32 if H == 16 else 8
                z, ctx = scale * rng.standard_normal((B, N_Z, H, H)), scale * rng.standard_normal((B, N_H, H, H))
                case("activations x%g %dx%d" % (scale, H, H), params, z, ctx, H)
        for gshift in (-3.0, 3.0):
            rng = np.random.RandomState(77)
            params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
            for k in params:
                if k.endswith("/g"):
                    params[k] = params[k] + gshift
            z, ctx = rng.standard_normal((8, N_Z, 16, 16)), rng.standard_normal((8, N_H, 16, 16))
            case("exp(g) x%.3g" % np.exp(gshift), params, z, ctx, 16)
        rng = np.random.RandomState(5)
        params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
        p32 = {k: f32(v) for k, v in params.items()}
        z = rng.standard_normal((8, N_Z, 16, 16))
        h0 = O.ar_conv2d(f32(z), p32["layer_0/V"], p32["layer_0/g"], p32["layer_0/b"], zerodiagonal=False)
        ctx = -h0 + 1e-3 * rng.standard_normal(h0.shape)
        case("cancellation", params, z, ctx, 16)
    # time: the whole step relaunched hot (iaf_step_time_layer, layer -2)
    rng = np.random.RandomState(0)
    params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
    for H in (16, 8):
        z, ctx = dev(rng.standard_normal((32, N_Z, H, H))), dev(rng.standard_normal((32, N_H, H, H)))
        line = "step %dx%d B=32 relaunched hot:" % (H, H)
        for prec, st, dp in stacks(params):
            if prec == "f32":
                continue
            ts = [st.time_layer(-2, z, ctx, reps=200) for _ in range(5)]
            line += "  %s %.2f us" % (prec, 1e3 * float(np.median(ts)))
        print(line, flush=True)


if __name__ == "__main__":
    main()
