#!/usr/bin/env python3
"""GPU dev tool: per-workgroup s_memtime phase breakdown of one conv layer."""
import argparse, ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_inputs as gi, iaf_amd
from iaf_amd import _capi
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--hw", type=int, default=16)
ap.add_argument("--n-z", type=int, default=32); ap.add_argument("--n-h", type=int, default=160)
ap.add_argument("--depth-ar", type=int, default=2); ap.add_argument("--tune", type=str, default="")
ap.add_argument("--precision", type=str, default="f32"); ap.add_argument("--tune-bf3", type=str, default="")
ap.add_argument("--fuse", type=str, default="never")
a = ap.parse_args()
rng = np.random.RandomState(0)
params = gi.ar_multiconv2d_params(rng, a.n_z, [a.n_h] * a.depth_ar, [a.n_z, a.n_z])
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
z = dev(rng.standard_normal((a.batch, a.n_z, a.hw, a.hw))); ctx = dev(rng.standard_normal((a.batch, a.n_h, a.hw, a.hw)))
st = iaf_amd.ARStack(a.n_z, [a.n_h] * a.depth_ar); st.set_precision(a.precision); st.set_fuse_first(a.fuse); st.prepare({k: dev(v) for k, v in params.items()})
if a.tune:
    for item in a.tune.split(";"):
        lay, shp = item.split(":"); st.set_tuning(int(lay), *[int(v) for v in shp.split(",")])
if a.tune_bf3:
    for item in a.tune_bf3.split(";"):
        lay, shp = item.split(":"); st.set_tuning_bf3(int(lay), *[int(v) for v in shp.split(",")])
names = ["start->loads issued", "loads issued->tile staged(barrier)", "staged->steady loop done", "steady->K loop done", "K loop done->end"]
if a.precision == "bf16x3":
    names = ["start->loads issued", "loads issued->tile staged(barrier)", "staged->K loop done", "K done->exchange done", "exchange->end"]
for layer in range(a.depth_ar + 1):
    buf = torch.zeros(8 * 65536, dtype=torch.int64, device="cuda")
    for _ in range(3):
        st.iaf_step(z, ctx)
    _capi.check(_capi.lib().iaf_stack_set_debug(st._h, layer, ctypes.c_void_p(buf.data_ptr())))
    st.iaf_step(z, ctx); torch.cuda.synchronize()
    _capi.check(_capi.lib().iaf_stack_set_debug(st._h, -1, None))
    t = buf.cpu().numpy().reshape(-1, 8); t = t[t[:, 0] != 0]
    if len(t) == 0:
        print("layer %d: no launch of its own (fused into the next)" % layer); continue
    d = np.diff(t[:, :6], axis=1).astype(np.float64)
    span = (t[:, 5].max() - t[:, 0].min())
    print("layer %d: %d WGs; kernel span %.0f ticks; per-WG total median %.0f; first start->last start %.0f" %
          (layer, len(t), span, np.median(t[:, 5] - t[:, 0]), t[:, 0].max() - t[:, 0].min()))
    for i, n in enumerate(names):
        print("    %-36s median %8.0f   max %8.0f ticks" % (n, np.median(d[:, i]), d[:, i].max()))
    if (t[:, 7] != 0).any():
        print("    %-36s median %8.0f   (inside 'tile staged')" % ("fused first layer", np.median(t[:, 7] - t[:, 6])))
