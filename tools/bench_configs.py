#!/usr/bin/env python3
"""Times one IAF step (iaf_step_forward) and one posterior block (iaf_posterior_block_forward) for every
BASELINE.json config on one GPU: hipGraph replay of 20 calls, wall clock around 50 replays.  These are the
parity-test configs; bench.py's headline line is config 2 only.  python tools/bench_configs.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_inputs as gi  # noqa: E402
import iaf_amd  # noqa: E402

CONFIGS = [
    # name, B, n_z, n_h, depth_ar, levels
    ("config1 n_z=32 n_h=64 depth_ar=1 bs=16", 16, 32, 64, 1, [16, 8, 4]),
    ("config2 n_z=32 n_h=160 depth_ar=2 bs=32", 32, 32, 160, 2, [16, 8]),
    ("config4 n_z=64 n_h=64 depth_ar=4 bs=32", 32, 64, 64, 4, [16, 8, 4]),
    ("config4 n_z=64 n_h=128 depth_ar=4 bs=32", 32, 64, 128, 4, [16, 8, 4]),
    ("config4 n_z=64 n_h=192 depth_ar=4 bs=32", 32, 64, 192, 4, [16, 8, 4]),
    ("config5 n_z=32 n_h=160 depth_ar=2 bs=256 (IW eval rows)", 256, 32, 160, 2, [16, 8]),
]


def timed(fn, inner=20, reps=30):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(inner):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (reps * inner)


def main():
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    print("| config | latent | IAF step us | samples/s | live GFLOP | TFLOP/s | posterior block us | kernels (autotuned per layer) |")
    print("|---|---|---|---|---|---|---|---|")
    for name, B, n_z, n_h, d, levels in CONFIGS:
        rng = np.random.RandomState(0)
        params = gi.ar_multiconv2d_params(rng, n_z, [n_h] * d, [n_z, n_z])
        st = iaf_amd.ARStack(n_z, [n_h] * d)
        st.prepare({k: dev(v) for k, v in params.items()})
        for hw in levels:
            f = lambda c: dev(rng.standard_normal((B, c, hw, hw)))
            z, ctx = f(n_z), f(n_h)
            out = (torch.empty_like(z), torch.empty_like(z))
            picks = st.autotune(z, ctx, reps=10)
            t_step = timed(lambda: st.iaf_step(z, ctx, out=out))
            qm, ql, rm, rl, pm, pl, eps, dc = f(n_z), 0.25 * f(n_z), f(n_z), 0.25 * f(n_z), f(n_z), 0.25 * f(n_z), f(n_z), f(n_h)
            t_blk = timed(lambda: st.posterior_block(qm, ql, rm, rl, pm, pl, ctx, dc, eps, 0.25), inner=10, reps=20)
            w = st.step_work(B, hw, hw)
            print("| %s | [%d,%d,%d,%d] | %.1f | %.0f | %.3f | %.1f | %.1f | %s |" %
                  (name, B, n_z, hw, hw, 1e6 * t_step, B / t_step, w["live_flops"] / 1e9, w["live_flops"] / t_step / 1e12,
                   1e6 * t_blk, " ".join(c for c, _ in picks)))


if __name__ == "__main__":
    main()
