#!/usr/bin/env python3
"""GPU dev tool: the batched weight-prep launch of the bench's 20 stacks, us per launch.  argv[1]: f16 (the headline's packs: two-plane fp16
only; default) | bf3 (bf16x3 only) | both (bf16x3 + fp16).  IAF_PREP_DBG: bit 0 tiles in blockIdx order (not paired per XCD), bit 1 the round-5 tile function."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_inputs as gi, iaf_amd
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
stacks, plist = [], []
for i in range(20):
    st = iaf_amd.ARStack(32, [160, 160])
    p = {k: dev(v) for k, v in gi.ar_multiconv2d_params(np.random.RandomState(i), 32, [160, 160], [32, 32]).items()}
    which = sys.argv[1] if len(sys.argv) > 1 else "f16"
    st.set_packs(f32=False, bf16x3=which != "f16", f16x2=which != "bf3"); st.prepare(p); stacks.append(st); plist.append(p)
prep = iaf_amd.PrepBatch(stacks)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for _ in range(5): prep.run(plist)
    stream.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        for _ in range(10): prep.run(plist)
    for _ in range(20): g.replay()
    stream.synchronize()
    for rnd in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(50): g.replay()
        b.record(stream); b.synchronize()
        print("prep of 20 stacks: %.2f us per launch incl. gap (graph of 10; packs %s; IAF_PREP_DBG=%s)" % (a.elapsed_time(b) / 500 * 1e3, which, os.environ.get("IAF_PREP_DBG", "0")))
