#!/usr/bin/env python3
"""GPU dev tool: the 16x16 one-launch step inside a graph of NS launches of NS different stacks (every launch meets packs
that are cold in the XCDs' L2s, as in the bench line), timed per exchange debug knob, alternating runs on one box:
   0 production | 4 never through L2 | 1 lists ignore the placement | -1 halo rows recomputed (no exchange)"""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_inputs as gi, iaf_amd
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--hw", type=int, default=16)
ap.add_argument("--stacks", type=int, default=10); ap.add_argument("--reps", type=int, default=300)
ap.add_argument("--knobs", default="0,4,1,-1"); ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--posterior", action="store_true")
a = ap.parse_args()
rng = np.random.RandomState(0)
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
B, HW = a.batch, a.hw
z = dev(rng.standard_normal((B, 32, HW, HW))); ctx = dev(rng.standard_normal((B, 160, HW, HW)))
stream = torch.cuda.Stream()
graphs = {}
for knob in [int(k) for k in a.knobs.split(",")]:
    stacks = []
    for i in range(a.stacks):
        st = iaf_amd.ARStack(32, [160, 160])
        st.prepare({k: dev(v) for k, v in gi.ar_multiconv2d_params(np.random.RandomState(i), 32, [160, 160], [32, 32]).items()})
        if knob < 0: st.set_halo_exchange(False)
        else: st.set_halo_exchange_debug(knob)
        stacks.append(st)
    outs = [(torch.empty_like(z), torch.empty_like(z)) for _ in stacks]
    def step():
        cur = z
        for st, o in zip(stacks, outs):
            st.iaf_step(cur, ctx, out=o); cur = o[0]
    with torch.cuda.stream(stream):
        step(); step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        step()
    graphs[knob] = (g, stacks, outs)
with torch.cuda.stream(stream):
    for _ in range(300):
        for g, _, _ in graphs.values(): g.replay()
torch.cuda.synchronize()
ref = None
for rnd in range(a.rounds):
    for knob, (g, stacks, outs) in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            e0.record(stream)
            for _ in range(a.reps): g.replay()
            e1.record(stream)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.reps / a.stacks * 1e3
        paths = stacks[0].exchange_paths() if knob >= 0 else (0, 0)
        errs = sum(st.exchange_errors() for st in stacks)
        fin = bool(torch.isfinite(outs[-1][0]).all())
        print("round %d knob %2d: %.2f us per launch (graph of %d, %dx%d B=%d)  rows via L2 / memory of stack 0: %d / %d  errors %d finite %s"
              % (rnd, knob, us, a.stacks, HW, HW, B, paths[0], paths[1], errs, fin), flush=True)
