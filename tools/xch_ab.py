#!/usr/bin/env python3
"""GPU dev tool: the one-launch step inside a graph of NS launches of NS different stacks (every launch meets packs that are
cold in the XCDs' L2s, as in the bench line), timed per variant, alternating runs on one box.  A variant is the exchange debug
knob of include/iaf_hip.h (0 production | 1 lists ignore the placement | 3 ... and tickets out of order | -1 halo rows
recomputed, no exchange), with suffix s for ONE stack launched NS times (its packs stay in the L2s)."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_inputs as gi, iaf_amd
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--hw", type=int, default=16)
ap.add_argument("--stacks", type=int, default=10); ap.add_argument("--reps", type=int, default=300)
ap.add_argument("--knobs", default="0,1,-1,0s"); ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
rng = np.random.RandomState(0)
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
B, HW = a.batch, a.hw
z = dev(rng.standard_normal((B, 32, HW, HW))); ctx = dev(rng.standard_normal((B, 160, HW, HW)))
stream = torch.cuda.Stream()
graphs = {}
for name in a.knobs.split(","):
    knob, same = int(name.rstrip("s")), name.endswith("s")
    stacks = []
    for i in range(a.stacks):
        if same and i: stacks.append(stacks[0]); continue
        st = iaf_amd.ARStack(32, [160, 160])
        st.prepare({k: dev(v) for k, v in gi.ar_multiconv2d_params(np.random.RandomState(i), 32, [160, 160], [32, 32]).items()})
        if knob < 0: st.set_halo_exchange(False)
        else: st.set_halo_exchange_debug(knob)
        stacks.append(st)
    outs = [(torch.empty_like(z), torch.empty_like(z)) for _ in stacks]
    def step(stacks=stacks, outs=outs, same=same):
        cur = z
        for st, o in zip(stacks, outs):
            st.iaf_step(cur, ctx, out=o)
            if not same: cur = o[0]
    with torch.cuda.stream(stream):
        step(); step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        step()
    graphs[name] = (g, stacks, outs)
with torch.cuda.stream(stream):
    for _ in range(300):
        for g, _, _ in graphs.values(): g.replay()
torch.cuda.synchronize()
for rnd in range(a.rounds):
    for name, (g, stacks, outs) in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            e0.record(stream)
            for _ in range(a.reps): g.replay()
            e1.record(stream)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.reps / a.stacks * 1e3
        errs = sum(st.exchange_errors() for st in set(stacks))
        fin = bool(torch.isfinite(outs[-1][0]).all())
        print("round %d variant %4s: %.2f us per launch (graph of %d, %dx%d B=%d)  errors %d finite %s"
              % (rnd, name, us, a.stacks, HW, HW, B, errs, fin), flush=True)
