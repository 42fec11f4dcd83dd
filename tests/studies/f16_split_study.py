#!/usr/bin/env python
"""CPU numerics study (no GPU): could the IAF stack's convs run on TWO fp16 planes per operand (three part-products on
v_mfma_f32_16x16x32_f16) instead of three bf16 planes (six part-products on ..._bf16) and still be held to
tests/test_hip_dynamic_range.py's criterion

    err(split) <= 2 * err(exact fp32) + 4 * 2^-24 * |output scale|

against the fp64 chain on the same fp32-rounded inputs?  Half the matrix-core instructions and two thirds of the pack and LDS
bytes would be the prize (iaf_step_fused.hpp's K loops run at the MFMA issue floor).  The study emulates the arithmetic
exactly where it matters: the planes are rounded as the hardware would round them (torch's fp16 / bf16 casts, subnormals kept),
a part-product of two planes is exact in fp32 (11 + 11 or 8 + 8 significand bits), and the K sum accumulates in fp32 (torch's
fp32 conv of the plane-valued operands).  Forms:

    f32     the exact-fp32 chain (the yardstick of the criterion)
    bf16x3  what ships: x = x0 + x1 + x2 in bf16, six part-products (a0b0 a0b1 a1b0 a0b2 a1b1 a2b0), one accumulator
    f16x2   x = hi + lo in fp16, three part-products (hh hl lh), one accumulator
    f16x2s  the same with lo kept as lo * 2^11 (so it stays in fp16's NORMAL range wherever hi does): the two cross products go
            to an accumulator of their own that is scaled by 2^-11 when the sums meet

Cases = the dynamic-range test's: inputs scaled by 1e-3 / 1 / 1e+3, exp(g) moved by e^-3 / e^+3 on every conv, and the
cancellation case (context = -(first conv) +- 1e-3).  Reference operator: tf_utils/layers.py:56-64,158-166."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_inputs as gi
from oracle import iaf_oracle as O

torch.set_num_threads(max(1, os.cpu_count() or 1))
N_Z, N_H, D, B, H = 32, 160, 2, 8, 16
EPS32 = 2.0 ** -24
F16_MAX = 65504.0


def planes(t, dtype, n, lo_scale=1.0):
    out, r = [], t.clone()
    for i in range(n):
        p = (r * (lo_scale if i else 1.0)).to(dtype).to(torch.float32) / (lo_scale if i else 1.0)
        out.append(p)
        r = r - p
    return out


def conv32(x, w):
    return F.conv2d(x, w, padding=1)


def conv_form(x, w, form):
    if form == "f32":
        return conv32(x, w)
    if form == "bf16x3":
        a, b = planes(x, torch.bfloat16, 3), planes(w, torch.bfloat16, 3)
        acc = torch.zeros_like(conv32(a[0], b[0]))
        for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):      # small terms first, as one accumulator would see them
            acc = acc + conv32(a[i], b[j])
        return acc
    if form == "f16x2":
        a, b = planes(x, torch.float16, 2), planes(w, torch.float16, 2)
        return (conv32(a[1], b[0]) + conv32(a[0], b[1])) + conv32(a[0], b[0])
    if form == "f16x2s":
        a, b = planes(x, torch.float16, 2, 2048.0), planes(w, torch.float16, 2, 2048.0)
        cross = conv32(a[1] * 2048.0, b[0]) + conv32(a[0], b[1] * 2048.0)  # (the planes as stored: lo * 2^11)
        return cross * (1.0 / 2048.0) + conv32(a[0], b[0])
    raise ValueError(form)


def eff_weight(params, name, zerodiag):
    V, g = params[name + "/V"], params[name + "/g"]
    kh, kw, ni, no = V.shape
    w = O.weightnorm_weights(V, g, O.get_conv_ar_mask(kh, kw, ni, no, zerodiag))     # HWIO
    return torch.tensor(np.ascontiguousarray(w.transpose(3, 2, 0, 1)).astype(np.float32))


def stack(params, z, ctx, form):
    f64 = form == "f64"
    cast = (lambda t: t.double()) if f64 else (lambda t: t)
    x, c = cast(torch.tensor(z.astype(np.float32))), cast(torch.tensor(ctx.astype(np.float32)))
    peak = 0.0

    def conv(h, name, zd):
        nonlocal peak
        w = eff_weight(params, name, zd)
        b = torch.tensor(params[name + "/b"].astype(np.float32)).view(1, -1, 1, 1)
        peak = max(peak, float(h.abs().max()), float(w.abs().max()))
        if f64:
            return F.conv2d(h, w.double(), padding=1) + b.double()
        return conv_form(h, w, form) + b

    h = x
    for i in range(D):
        h = conv(h, "layer_%d" % i, False)
        if i == 0:
            h = h + c
        h = F.elu(h)
    return conv(h, "layer_out_0", True), conv(h, "layer_out_1", True), peak


def run_case(label, params, z, ctx):
    rm, rs, peak = stack(params, z, ctx, "f64")
    scale = max(float(rm.abs().max()), float(rs.abs().max()))
    errs = {}
    for form in ("f32", "bf16x3", "f16x2", "f16x2s"):
        m, s, _ = stack(params, z, ctx, form)
        e = max(float((m.double() - rm).abs().max()), float((s.double() - rs).abs().max()))
        errs[form] = e if np.isfinite(e) else float("inf")
    bound = 2.0 * errs["f32"] + 4.0 * EPS32 * scale
    line = "%-44s out scale %8.3g  largest operand %8.3g  bound %8.3g |" % (label, scale, peak, bound)
    for form in ("f32", "bf16x3", "f16x2", "f16x2s"):
        line += "  %s %8.3g%s" % (form, errs[form], "" if form == "f32" else (" ok  " if errs[form] <= bound else " FAIL"))
    print(line)
    return {k: v <= bound for k, v in errs.items()}


def main():
    verdict = {"bf16x3": True, "f16x2": True, "f16x2s": True}
    def note(r):
        for k in verdict:
            verdict[k] &= r[k]
    for scale in (1e-3, 1.0, 1e3, 1e5):
        rng = np.random.RandomState(900 + H)
        params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
        z, ctx = scale * rng.standard_normal((B, N_Z, H, H)), scale * rng.standard_normal((B, N_H, H, H))
        r = run_case("activations x%g" % scale, params, z, ctx)
        if scale <= 1e3:
            note(r)
    for gshift in (-3.0, 3.0):
        rng = np.random.RandomState(77)
        params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
        for k in params:
            if k.endswith("/g"):
                params[k] = params[k] + gshift
        z, ctx = rng.standard_normal((B, N_Z, H, H)), rng.standard_normal((B, N_H, H, H))
        note(run_case("exp(g) x%.3g on every conv" % np.exp(gshift), params, z, ctx))
    rng = np.random.RandomState(5)
    params = gi.ar_multiconv2d_params(rng, N_Z, [N_H] * D, [N_Z, N_Z])
    p32 = {k: np.asarray(v, dtype=np.float32).astype(np.float64) for k, v in params.items()}
    z = rng.standard_normal((B, N_Z, H, H))
    h0 = O.ar_conv2d(z.astype(np.float32).astype(np.float64), p32["layer_0/V"], p32["layer_0/g"], p32["layer_0/b"], zerodiagonal=False)
    ctx = -h0 + 1e-3 * rng.standard_normal(h0.shape)
    note(run_case("cancellation (context = -conv +- 1e-3)", params, z, ctx))
    print()
    print("criterion over the dynamic-range test's cases (x1e-3 .. x1e+3, exp(g) e^-3 / e^+3, cancellation): " +
          ", ".join("%s %s" % (k, "holds" if v else "FAILS") for k, v in verdict.items()))
    print("(x1e+5 is beyond the test's cases: it shows where fp16's largest finite number, %g, ends the two-plane forms)" % F16_MAX)


if __name__ == "__main__":
    main()
