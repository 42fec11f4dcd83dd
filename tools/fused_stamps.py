#!/usr/bin/env python3
"""GPU dev tool: per-workgroup cycle stamps of the one-launch IAF step (iaf_step_fused.hpp)."""
import argparse, ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_inputs as gi, iaf_amd
from iaf_amd import _capi
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--hw", type=int, default=16)
ap.add_argument("--n-z", type=int, default=32); ap.add_argument("--n-h", type=int, default=160)
ap.add_argument("--depth-ar", type=int, default=2); ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--knob", type=int, default=0, help="halo-exchange debug knob (include/iaf_hip.h); -1: halo rows recomputed")
a = ap.parse_args()
rng = np.random.RandomState(0)
params = gi.ar_multiconv2d_params(rng, a.n_z, [a.n_h] * a.depth_ar, [a.n_z, a.n_z])
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
z = dev(rng.standard_normal((a.batch, a.n_z, a.hw, a.hw))); ctx = dev(rng.standard_normal((a.batch, a.n_h, a.hw, a.hw)))
st = iaf_amd.ARStack(a.n_z, [a.n_h] * a.depth_ar); st.prepare({k: dev(v) for k, v in params.items()})
if a.knob < 0: st.set_halo_exchange(False)
elif a.knob: st.set_halo_exchange_debug(a.knob)
out = (torch.empty_like(z), torch.empty_like(z))
for _ in range(5):
    st.iaf_step(z, ctx, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    st.iaf_step(z, ctx, out=out)
e1.record(); torch.cuda.synchronize()
print("iaf_step %dx%d B=%d knob %d: %.2f us per call (back to back, eager); errors %d" % (
    a.hw, a.hw, a.batch, a.knob, e0.elapsed_time(e1) / a.reps * 1e3, st.exchange_errors()))
buf = torch.zeros(32 * 65536, dtype=torch.int64, device="cuda")
_capi.check(_capi.lib().iaf_stack_set_debug(st._h, -2, ctypes.c_void_p(buf.data_ptr())))
st.iaf_step(z, ctx, out=out); torch.cuda.synchronize()
_capi.check(_capi.lib().iaf_stack_set_debug(st._h, -1, None))
t = buf.cpu().numpy().reshape(-1, 32); t = t[t[:, 0] != 0]
if len(t) == 0:
    print("the step did not run as one launch at this size"); sys.exit(0)
names = ["start -> z staged (barrier)", "first conv + epilogue", "second conv + epilogue", "output conv", "exchange + affine + stores"]
d = np.diff(t[:, :6], axis=1).astype(np.float64)
print("%d WGs; kernel span %.0f ticks; per-WG total median %.0f; first start->last start %.0f" %
      (len(t), t[:, 5].max() - t[:, 0].min(), np.median(t[:, 5] - t[:, 0]), t[:, 0].max() - t[:, 0].min()))
for i, n in enumerate(names):
    print("    %-32s median %8.0f   max %8.0f ticks" % (n, np.median(d[:, i]), d[:, i].max()))
if (t[:, 8] != 0).all():      # finer stamps (wave 0): prologue and the two epilogues
    m = lambda a, b_: np.median(t[:, a] - t[:, b_])
    print("    prologue: loads issued %.0f | zero fill + z arrives + z -> LDS %.0f | barrier %.0f" % (m(8, 0), m(9, 8), m(1, 9)))
    print("    first epilogue: context -> LDS + next weights requested %.0f | barrier %.0f | bias+ctx+ELU+split -> LDS %.0f (incl. end barrier)" % (
        m(10, 6), m(11, 10), m(2, 11)))
    if (t[:, 12] != 0).all():
        print("    second epilogue: %.0f + barrier %.0f; output pair: K loop %.0f, to the exchange buffer + barrier %.0f, transform + stores %.0f" % (
            m(12, 7), m(3, 12), m(4, 3), m(13, 4), m(5, 13)))
if (t[:, 30] != 0).any():
    v = t[:, 30] - 1; lst = (v >> 32).astype(np.int64); tk = (v & 0xffffffff).astype(np.int64); blk = np.arange(len(t))
    print("    tickets: %d of %d workgroups took theirs from one of the four lists of XCD blockIdx %% 8; lists used %d; tickets per list max %d" % (
        int((lst // 4 == blk % 8).sum()), len(t), len(set(lst.tolist())), int(np.bincount(lst).max())))
if (t[:, 28] != 0).any():
    k = t[(t[:, 28] != 0) & (t[:, 29] != 0) & (t[:, 14] != 0) & (t[:, 18] != 0)]
    if len(k):
        md = lambda a, b_: np.median(k[:, a] - k[:, b_])
        print("    compute waves (wave 0): second conv own taps %.0f | wait for the helpers' row %.0f | taps below %.0f;  output pair own taps %.0f | wait %.0f | taps below %.0f" % (
            md(28, 2), md(15, 28), md(7, 15), md(29, 3), md(21, 29), md(4, 21)))
        print("    helpers (first helper wave): import 1 asks %+.0f after the first epilogue's barrier, row complete +%.0f later, re-arm + LDS stores +%.0f;  import 2: asks %+.0f after the second epilogue's barrier, complete +%.0f, stores +%.0f" % (
            md(14, 2), md(16, 14), md(17, 16), md(18, 3), md(19, 18), md(20, 19)))
if (t[:, 28] == 0).all() and (t[:, 29] != 0).any() and (t[:, 19] != 0).any():
    # the pair form (8-pixel rows): two workgroups per (image, two rows); dbg[26] = 2 * item + half + 1
    k = t[(t[:, 29] != 0) & (t[:, 19] != 0)]
    md = lambda a, b_: np.median(k[:, a] - k[:, b_])
    print("    PAIR form: %d workgroups.  compute waves (wave 0): output pair, own half's channels %.0f | wait for the partner's half %.0f | the rest %.0f" % (
        len(k), md(29, 3), md(21, 29), md(4, 21)))
    print("    helpers (first helper wave): export issued %+.0f after the last hidden epilogue's barrier; import asks %+.0f, complete +%.0f later, re-arm + LDS stores +%.0f; sweeps %s (1, 2, 3, more)" % (
        md(27, 3), md(18, 3), md(19, 18), md(20, 19), [int((k[:, 25] - 1 == n).sum()) for n in (0, 1, 2)] + [int((k[:, 25] - 1 > 2).sum())]))
    byid = {int(r[26]) - 1: r for r in k if r[26] != 0}
    d1, d2, d3 = [], [], []
    for i, r in byid.items():
        pr = byid.get(i ^ 1)
        if pr is None or pr[27] == 0: continue
        d1.append(float(r[18]) - float(pr[27])); d2.append(float(r[19]) - float(pr[27])); d3.append(float(r[0]) - float(pr[0]))
    if d1:
        print("    partner's export issued -> this one starts asking: median %.0f ticks; -> has the half: median %.0f (min %.0f max %.0f); start - partner's start: median |%.0f| max |%.0f|" % (
            np.median(d1), np.median(d2), min(d2), max(d2), np.median(np.abs(d3)), max(np.abs(d3))))
elif (t[:, 26] != 0).any():
    k = t[t[:, 24] != 0]
    print("    sweeps until the row was complete: import 1 %s, import 2 %s (histogram over WGs: 1, 2, 3, more)" % (
        [int((k[:, 24] - 1 == n).sum()) for n in (0, 1, 2)] + [int((k[:, 24] - 1 > 2).sum())],
        [int((k[:, 25] - 1 == n).sum()) for n in (0, 1, 2)] + [int((k[:, 25] - 1 > 2).sum())]))
    byslot = {int(r[26]) - 1: r for r in t if r[26] != 0}
    d1, d2, d3 = [], [], []
    for sl, r in byslot.items():
        pr = byslot.get(sl + 1)
        if pr is None or r[14] == 0 or pr[27] == 0: continue
        d1.append(float(r[14]) - float(pr[27])); d2.append(float(r[16]) - float(pr[27])); d3.append(float(r[0]) - float(pr[0]))
    if d1:
        print("    producer's export issued -> consumer starts asking: median %.0f (min %.0f max %.0f) ticks; -> consumer has the row: median %.0f (min %.0f max %.0f); consumer's start - producer's start: median %.0f (min %.0f max %.0f)" % (
            np.median(d1), min(d1), max(d1), np.median(d2), min(d2), max(d2), np.median(d3), min(d3), max(d3)))
if (t[:, 6] != 0).all():
    print("    of which: first conv K loop %.0f, its epilogue + barrier %.0f; second conv K loop %.0f, its epilogue + barrier %.0f (wave 0)" % (
        np.median(t[:, 6] - t[:, 1]), np.median(t[:, 2] - t[:, 6]), np.median(t[:, 7] - t[:, 2]), np.median(t[:, 3] - t[:, 7])))
