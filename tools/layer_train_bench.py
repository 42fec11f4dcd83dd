#!/usr/bin/env python
"""GPU dev tool: forward+backward of one IAFLayer at BASELINE configs[1] sizes (for rocprofv3 --kernel-trace --stats)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
import golden_inputs as gi  # noqa: E402
import iaf_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--hw", type=int, default=16)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    B, zs, hs, H = args.batch, 32, 160, args.hw
    c = gi.layer_case_inputs("layer_cfg2_8x8")
    params = {k: torch.from_numpy(np.asarray(v, np.float32)).cuda() for k, v in c["params"].items()}
    layer = iaf_amd.IAFLayer(zs, hs, depth_ar=2, kl_min=0.25)
    layer.set_training(True)
    layer.load(params)
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda ch: torch.randn((B, ch, H, H), device="cuda", generator=g)
    up_in, down_in, eps, dU, dD = rn(hs), rn(hs), rn(zs), rn(hs), rn(hs)
    dK = torch.ones(B, device="cuda")
    grads = {}

    def fwd():
        layer.up_train(up_in)
        return layer.down_train(down_in, eps)

    def bwd():
        layer.down_backward(dD, dK, params, grads)
        layer.up_backward(dU, params, grads)

    def timeit(fn, reps):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    fwd(); bwd()
    tf = timeit(fwd, args.steps)
    tb = timeit(bwd, args.steps)
    print("IAFLayer %dx%d B=%d: forward(train) %.1f us, backward %.1f us, total %.1f us -> %.0f samples/s per layer"
          % (H, H, B, tf, tb, tf + tb, B / (tf + tb) * 1e6))


if __name__ == "__main__":
    main()
