#!/usr/bin/env python3
"""GPU dev tool: run iaf_step N times on one stack (for rocprofv3 --pmc / --kernel-trace).
python tools/run_step.py --hw 16 --reps 20 [--tune "1:5,4,1,1"]"""
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

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--hw", type=int, default=16)
ap.add_argument("--n-z", type=int, default=32)
ap.add_argument("--n-h", type=int, default=160)
ap.add_argument("--depth-ar", type=int, default=2)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--prep", action="store_true", help="re-run weight prep every rep")
ap.add_argument("--tune", type=str, default="")
ap.add_argument("--precision", type=str, default="bf16x3")
ap.add_argument("--tune-bf3", type=str, default="", help="layer:nt,ppw,pxt,ks;... ('auto' = iaf_stack_autotune first)")
a = ap.parse_args()
rng = np.random.RandomState(0)
params = gi.ar_multiconv2d_params(rng, a.n_z, [a.n_h] * a.depth_ar, [a.n_z, a.n_z])
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
z = dev(rng.standard_normal((a.batch, a.n_z, a.hw, a.hw)))
ctx = dev(rng.standard_normal((a.batch, a.n_h, a.hw, a.hw)))
st = iaf_amd.ARStack(a.n_z, [a.n_h] * a.depth_ar)
st.set_precision(a.precision)
dp = {k: dev(v) for k, v in params.items()}
st.prepare(dp)
if a.tune_bf3 == "auto":
    print("autotune:", st.autotune(z, ctx))
elif a.tune_bf3:
    for item in a.tune_bf3.split(";"):
        lay, shp = item.split(":")
        st.set_tuning_bf3(int(lay), *[int(v) for v in shp.split(",")])
if a.tune:
    for item in a.tune.split(";"):
        lay, shp = item.split(":")
        st.set_tuning(int(lay), *[int(v) for v in shp.split(",")])
for _ in range(a.reps):
    if a.prep:
        st.prepare(dp, force=True)
    st.iaf_step(z, ctx)
torch.cuda.synchronize()
print("done")
