#!/usr/bin/env python3
"""GPU dev tool: CVAE1.forward_backward (the whole model's objective and all 1004 gradients, BASELINE geometry: z 32, h 160, 2 levels x
10 layers, 32x32 images, B = 32, weights prepared once) timed as a hipGraph replay, with the share of the two ends (rocprof-free: the
edge launches are timed on their own)."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_inputs as gi, iaf_amd
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--blocks", type=int, default=10); ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
gi.MODEL_CASES["bench"] = (a.batch, 1, 32, 160, 2, a.blocks, 32, 0.25)
c = gi.model_case_inputs("bench")
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
model = iaf_amd.CVAE1(z_size=32, h_size=160, kl_min=0.25, depth=2, num_blocks=a.blocks, k=1, image_size=32)
model.set_training(True)
model.load({k: dev(v) for k, v in c["params"].items()})
x = torch.from_numpy(c["x"]).cuda()
noise = [dev(e) for e in c["noise"]]
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for _ in range(3):
        x_out, obj, grads = model.forward_backward(x, noise)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=stream):
    keep = model.forward_backward(x, noise)
with torch.cuda.stream(stream):
    for _ in range(5): g.replay()
torch.cuda.synchronize()
for rnd in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record(stream)
        for _ in range(a.reps): g.replay()
        e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print("round %d: forward + backward of the whole model %.3f ms per %d-image step (%d parameters tensors, obj %.1f, bits/dim %.3f)" % (
        rnd, ms, a.batch, len(keep[2]), float(keep[1]), float(keep[1]) / (np.log(2.) * 3072 * a.batch)), flush=True)
