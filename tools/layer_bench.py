#!/usr/bin/env python
"""GPU dev tool: per-conv timing of one non-downsampling IAFLayer (up + down) at BASELINE configs[1] sizes.
Back-to-back launches bracketed by events on torch's current stream (the stream the engine launches on)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
import golden_inputs as gi  # noqa: E402
import iaf_amd  # noqa: E402


def timeit(fn, reps=50, rounds=5):
    fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b) * 1e3 / reps)
    return float(np.median(best))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--hw", type=int, default=16)
    ap.add_argument("--tune", type=str, default="")
    args = ap.parse_args()
    B, zs, hs, H = args.batch, 32, 160, args.hw
    c = gi.layer_case_inputs("layer_cfg2_8x8")
    params = {k: torch.from_numpy(np.asarray(v, np.float32)).cuda() for k, v in c["params"].items()}
    layer = iaf_amd.IAFLayer(zs, hs, depth_ar=2, kl_min=0.25)
    layer.load(params)
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda ch: torch.randn((B, ch, H, H), device="cuda", generator=g)
    up_in, down_in, eps = rn(hs), rn(hs), rn(zs)
    layer.up(up_in)
    layer.down(down_in, eps)
    convs = [("up_conv1", layer.up_conv1, lambda cv: cv(up_in, elu_input=True, split=[zs, zs, hs, hs])),
             ("up_conv3", layer.up_conv3, lambda cv: cv(up_in, elu_input=True, residual=up_in)),
             ("down_conv1", layer.down_conv1, lambda cv: cv(down_in, elu_input=True, split=[zs] * 4 + [hs] * 2)),
             ("down_conv2", layer.down_conv2, lambda cv: cv(eps, x2=down_in, elu_input=True, residual=down_in))]
    shapes = [(5, 4, 1, 1), (4, 4, 1, 1), (2, 4, 1, 1), (5, 2, 2, 1), (4, 2, 2, 1), (2, 2, 2, 1), (1, 2, 2, 1), (5, 4, 1, 2),
              (4, 4, 1, 2), (2, 4, 1, 2), (1, 4, 1, 2), (5, 2, 2, 2), (2, 2, 2, 2), (1, 2, 2, 2), (5, 2, 1, 2), (4, 2, 1, 2),
              (2, 2, 1, 2), (1, 2, 1, 2), (2, 1, 2, 2), (1, 1, 2, 2), (5, 2, 1, 4), (2, 2, 1, 4), (1, 2, 1, 4), (2, 1, 1, 4), (1, 1, 1, 4)]
    for name, cv, call in convs:
        fl, by = cv.work(B, H, H)
        t = timeit(lambda: call(cv))
        print("%-10s %3d->%3d auto      %7.2f us  %6.1f TF  (min bytes %.1f MB -> %.0f GB/s)" %
              (name, cv.n_in, cv.n_out, t, fl / t / 1e6, by / 1e6, by / t / 1e3))
        if args.tune:
            res = []
            for sh in shapes:
                try:
                    cv.set_tuning(*sh)
                    res.append((timeit(lambda: call(cv), reps=20, rounds=3), sh))
                except ValueError:
                    pass
            cv.set_tuning(0, 0, 0, 0)
            res.sort()
            print("    best shapes:", ", ".join("%s %.2f" % (s, t) for t, s in res[:5]))
    layer.up(up_in)
    t_up = timeit(lambda: layer.up(up_in), reps=20)
    t_down = timeit(lambda: layer.down(down_in, eps), reps=20)
    print("IAFLayer.up %.1f us   IAFLayer.down %.1f us (weights cached, model-chosen launch shapes)" % (t_up, t_down))
    layer.up(up_in, autotune=True)
    layer.down(down_in, eps, autotune=True)
    t_up = timeit(lambda: layer.up(up_in), reps=20)
    t_down = timeit(lambda: layer.down(down_in, eps), reps=20)
    print("IAFLayer.up %.1f us   IAFLayer.down %.1f us (weights cached, autotuned: %s)" %
          (t_up, t_down, {n: getattr(layer, n)._tuned[(B, H, H)] for n in ("up_conv1", "up_conv3", "down_conv1", "down_conv2")}))
    fl = sum(getattr(layer, n).work(B, H, H)[0] for n in ("up_conv1", "up_conv3", "down_conv1", "down_conv2"))
    fl += layer.posterior.stack.step_work(B, H, H)["live_flops"]
    print("whole layer (up+down): %.2f GFLOP live -> %.1f TF; %.0f samples/s per layer" %
          (fl / 1e9, fl / (t_up + t_down) / 1e6, B / (t_up + t_down) * 1e6))


if __name__ == "__main__":
    main()
