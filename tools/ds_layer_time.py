#!/usr/bin/env python3
"""GPU dev tool: forward time (up + down, weights cached, graph replay) of ONE IAFLayer in three forms -- a plain layer at
16x16, a plain layer at 8x8, and the downsampling layer between them (tf_train.py:196: stride-2 up_conv1, deconv2d) -- and
of the downsampling layer's two strided convs on their own.  VERDICT r03 #7's yardstick: downsampling layer <= 1.3x a plain one."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_inputs as gi, iaf_amd
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--rounds", type=int, default=2)
a = ap.parse_args()
zs, hs, B = 32, 160, a.batch
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
rng, wrng = np.random.RandomState(1), np.random.RandomState(5)


def make(ds):
    p = {}
    for nm, (ci, co) in (("up_conv1", (hs, 2 * zs + 2 * hs)), ("up_conv3", (hs, hs)), ("down_conv1", (hs, 4 * zs + 2 * hs))):
        for k, v in gi.conv_params(wrng, ci, co).items():
            p[nm + "/" + k] = dev(v)
    last = gi.deconv_params(wrng, hs + zs, hs) if ds else gi.conv_params(wrng, hs + zs, hs)
    for k, v in last.items():
        p[("down_deconv2/" if ds else "down_conv2/") + k] = dev(v)
    for k, v in gi.ar_multiconv2d_params(wrng, zs, [hs, hs], [zs, zs]).items():
        p["ar_multiconv2d/" + k] = dev(v)
    lay = iaf_amd.IAFLayer(zs, hs, depth_ar=2, kl_min=0.25, downsample=ds)
    lay.load(p)
    return lay


stream = torch.cuda.Stream()
cases = {}
for name, ds, Hin in (("plain 16x16", False, 16), ("plain 8x8", False, 8), ("downsampling 16->8", True, 16)):
    lay = make(ds)
    Hlow = Hin // 2 if ds else Hin
    up_in = dev(0.5 * rng.standard_normal((B, hs, Hin, Hin)))
    down_in = dev(0.5 * rng.standard_normal((B, hs, Hlow, Hlow)))
    eps = dev(rng.standard_normal((B, zs, Hlow, Hlow)))
    parts = {}

    def up(lay=lay, up_in=up_in):
        return lay.up(up_in)

    def down(lay=lay, down_in=down_in, eps=eps):
        return lay.down(down_in, eps)

    for pn, fn in (("up", up), ("down", down)):
        with torch.cuda.stream(stream):
            up(); fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            keep = fn()
        parts[pn] = (g, keep)
    cases[name] = (lay, parts)
with torch.cuda.stream(stream):
    for _ in range(100):
        for lay, parts in cases.values():
            for g, _ in parts.values(): g.replay()
torch.cuda.synchronize()
for rnd in range(a.rounds):
    tot = {}
    for name, (lay, parts) in cases.items():
        t = {}
        for pn, (g, _) in parts.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                e0.record(stream)
                for _ in range(a.reps): g.replay()
                e1.record(stream)
            torch.cuda.synchronize()
            t[pn] = e0.elapsed_time(e1) / a.reps * 1e3
        tot[name] = t["up"] + t["down"]
        print("round %d %-20s up %7.1f us  down %7.1f us  total %7.1f us" % (rnd, name, t["up"], t["down"], tot[name]), flush=True)
    print("round %d downsampling / plain 16x16 = %.2f, / plain 8x8 = %.2f" % (
        rnd, tot["downsampling 16->8"] / tot["plain 16x16"], tot["downsampling 16->8"] / tot["plain 8x8"]), flush=True)
