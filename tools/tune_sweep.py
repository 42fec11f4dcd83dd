#!/usr/bin/env python3
"""GPU dev tool: time every compiled launch shape for each GEMM layer of an AR stack with the
engine's per-launch HIP events.  python tools/tune_sweep.py [--batch 32] [--hw 16]"""
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

SHAPES = [(4, 1, 1), (4, 1, 2), (2, 2, 1), (2, 2, 2), (2, 1, 2), (2, 1, 4), (1, 1, 4), (1, 2, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--hw", type=int, default=16)
    ap.add_argument("--n-z", type=int, default=32)
    ap.add_argument("--n-h", type=int, default=160)
    ap.add_argument("--depth-ar", type=int, default=2)
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    rng = np.random.RandomState(0)
    params = gi.ar_multiconv2d_params(rng, a.n_z, [a.n_h] * a.depth_ar, [a.n_z, a.n_z])
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    z = dev(rng.standard_normal((a.batch, a.n_z, a.hw, a.hw)))
    ctx = dev(rng.standard_normal((a.batch, a.n_h, a.hw, a.hw)))
    st = iaf_amd.ARStack(a.n_z, [a.n_h] * a.depth_ar)
    st.prepare({k: dev(v) for k, v in params.items()})
    nlayers = a.depth_ar + 1
    for layer in range(nlayers):
        is_out = layer == a.depth_ar
        ncot = (2 * a.n_z if is_out else a.n_h) // 16
        lw = st.layer_work(layer, a.batch, a.hw, a.hw)
        res = []
        for nt in ([2, 4] if is_out else [5, 4, 3, 2, 1]):
            for (pxt, wco, ks) in SHAPES:
                if ncot % (nt * wco) != 0:
                    continue
                try:
                    st.set_tuning(layer, nt, pxt, wco, ks)
                except ValueError:
                    continue
                st.profile_enable(layer, a.reps + 8)
                try:
                    for _ in range(5):
                        st.iaf_step(z, ctx)
                    torch.cuda.synchronize()
                    st.profile_read()
                    for _ in range(a.reps):
                        st.iaf_step(z, ctx)
                    torch.cuda.synchronize()
                    ms = st.profile_read()
                except Exception as e:  # unsupported LDS size etc.
                    print("layer %d nt=%d shape=%s: %s" % (layer, nt, (pxt, wco, ks), e))
                    continue
                us = 1e3 * float(np.median(ms))
                res.append((us, nt, pxt, wco, ks))
        res.sort()
        print("layer %d (%s) B=%d %dx%d live %.3f GFLOP  floor@157.3TF %.2f us" %
              (layer, "out" if is_out else "hidden", a.batch, a.hw, a.hw, lw["live_flops"] / 1e9,
               lw["live_flops"] / 157.3e12 * 1e6))
        for us, nt, pxt, wco, ks in res:
            print("   %8.2f us  %6.1f TF  nt=%d pxt=%d wco=%d ks=%d" % (us, lw["live_flops"] / us / 1e6, nt, pxt, wco, ks))
        if res:
            st.set_tuning(layer, *res[0][1:])
    st.profile_enable(-1, 0)


if __name__ == "__main__":
    main()
