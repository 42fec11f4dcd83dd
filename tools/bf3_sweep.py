#!/usr/bin/env python3
"""GPU dev tool: per-layer time of the IAF step in both precisions (exact-fp32 MFMA vs bf16x3), automatic launch
shapes, then every compiled bf16x3 shape.  50 back-to-back launches per HIP event pair (iaf_step_time_layer).
    python tools/bf3_sweep.py [--batch 32] [--hw 16 8] [--n-z 32 --n-h 160 --depth-ar 2] [--sweep]"""
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

BF3_SHAPES = [(4, 1, 4, 1), (2, 1, 4, 1), (1, 1, 4, 1), (1, 4, 1, 1), (2, 1, 4, 2), (1, 1, 4, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--hw", type=int, nargs="+", default=[16, 8])
    ap.add_argument("--n-z", type=int, default=32)
    ap.add_argument("--n-h", type=int, default=160)
    ap.add_argument("--depth-ar", type=int, default=2)
    ap.add_argument("--sweep", action="store_true")
    a = ap.parse_args()
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    rng = np.random.RandomState(0)
    params = gi.ar_multiconv2d_params(rng, a.n_z, [a.n_h] * a.depth_ar, [a.n_z, a.n_z])
    dp = {k: dev(v) for k, v in params.items()}
    for H in a.hw:
        z = dev(rng.standard_normal((a.batch, a.n_z, H, H)))
        ctx = dev(rng.standard_normal((a.batch, a.n_h, H, H)))
        tot = {}
        for prec in ("f32", "bf16x3"):
            st = iaf_amd.ARStack(a.n_z, [a.n_h] * a.depth_ar)
            st.set_precision(prec)
            st.prepare(dp)
            st.iaf_step(z, ctx)
            tot[prec] = 0.0
            for layer in range(a.depth_ar + 1):
                us = 1e3 * min(st.time_layer(layer, z, ctx, reps=50) for _ in range(3))
                w = st.layer_work(layer, a.batch, H, H)
                tot[prec] += us
                print("B=%d %dx%d layer %d %-7s (runs %s) %8.2f us  %6.1f TF live" %
                      (a.batch, H, H, layer, prec, st.layer_precision(layer, a.batch, H, H), us, w["live_flops"] / us / 1e6), flush=True)
        print("B=%d %dx%d IAF step (sum of layers): f32 %.1f us, bf16x3 %.1f us  (x%.2f)" %
              (a.batch, H, H, tot["f32"], tot["bf16x3"], tot["f32"] / tot["bf16x3"]), flush=True)
        if not a.sweep:
            continue
        st = iaf_amd.ARStack(a.n_z, [a.n_h] * a.depth_ar)
        st.prepare(dp)
        for layer in range(a.depth_ar + 1):
            is_out = layer == a.depth_ar
            ncot = (2 * a.n_z if is_out else a.n_h) // 16
            res = []
            for nt in ((4, 2) if is_out else (5, 4, 2)):
                if ncot % nt:
                    continue
                for (ppw, pxt, ks, wco) in BF3_SHAPES:
                    if ncot % (nt * wco):
                        continue
                    try:
                        st.set_tuning_bf3(layer, nt, ppw, pxt, ks, wco)
                        st.iaf_step(z, ctx)
                        us = 1e3 * min(st.time_layer(layer, z, ctx, reps=50) for _ in range(2))
                    except ValueError as e:
                        continue
                    res.append((us, nt, ppw, pxt, ks, wco))
            res.sort()
            for us, nt, ppw, pxt, ks, wco in res:
                print("   layer %d  %8.2f us  nt=%d ppw=%d pxt=%d ks=%d wco=%d" % (layer, us, nt, ppw, pxt, ks, wco), flush=True)
            st.set_tuning_bf3(layer, 0, 0, 0, 0)


if __name__ == "__main__":
    main()
