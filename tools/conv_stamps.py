#!/usr/bin/env python3
"""GPU dev tool: where a plain 9-tap bf16x3 conv launch spends its cycles -- per-workgroup stamps (iaf_conv3x3_set_debug) of
the four plain convs of an IAFLayer at B=32 16x16 (or --hw 8), in their autotuned or given launch shape."""
import argparse, ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_inputs as gi, iaf_amd
from iaf_amd import _capi
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--hw", type=int, default=16)
ap.add_argument("--shape", type=str, default="", help="nt,ppw,wco,ks (bf16x3 shape) instead of the autotuned one")
a = ap.parse_args()
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
rng = np.random.RandomState(0)
B, H = a.batch, a.hw
for name, ci, co, split in (("up_conv1", 160, 384, [32, 32, 160, 160]), ("up_conv3", 160, 160, [160]),
                            ("down_conv1", 160, 448, [32] * 4 + [160] * 2), ("down_conv2", 192, 160, [160])):
    p = gi.conv_params(rng, ci, co)
    cv = iaf_amd.WNConv2d(ci, co)
    cv.prepare(dev(p["V"]), dev(p["g"]), dev(p["b"]))
    x = dev(rng.standard_normal((B, ci, H, H)))
    if a.shape:
        nt, ppw, wco, ks = (int(v) for v in a.shape.split(","))
        try:
            cv.set_tuning(-nt, ppw, wco, ks)
        except Exception as e:
            print(name, "shape not available:", e); continue
        sh = (-nt, ppw, wco, ks)
    else:
        cv(x, elu_input=True, split=split, autotune=True)
        sh = cv._tuned[(B, H, H)]
    for _ in range(5): cv(x, elu_input=True, split=split)
    torch.cuda.synchronize()
    buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
    _capi.check(_capi.lib().iaf_conv3x3_set_debug(cv._h, ctypes.c_void_p(buf.data_ptr()), buf.numel() * 8))
    cv(x, elu_input=True, split=split)
    torch.cuda.synchronize()
    _capi.check(_capi.lib().iaf_conv3x3_set_debug(cv._h, None, 0))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): cv(x, elu_input=True, split=split)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    st = buf.cpu().numpy().reshape(-1, 8)
    st = st[st[:, 0] != 0]
    d = lambda i, j: float(np.mean(st[:, i] - st[:, j]))
    t0, t1 = st[:, 0].min(), st[:, 5].max()
    print("%-10s shape %s: %.1f us; %d workgroups; per workgroup (cycles): ring primed %.0f, tile staged +%.0f, K loop +%.0f, "
          "exchange +%.0f, epilogue +%.0f = %.0f" % (
              name, sh, us, len(st), d(1, 0), d(2, 1), d(3, 2), d(4, 3), d(5, 4), d(5, 0)), flush=True)
