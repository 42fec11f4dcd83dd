#!/usr/bin/env python3
"""GPU dev tool: the halo hand-over under uneven load (MI355X_MICROARCH.md: "test every hand-off under UNEVEN load ... checking every
word").  Stream A: the exchanging step and the recomputing step on fresh inputs every iteration, outputs compared on the device;
stream B: 256 MB copies and a stream of small launches the whole time; every few hundred iterations the batch size changes (other
grids, other row-buffer layouts, workgroups in several rounds of the chip)."""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_inputs as gi, iaf_amd
ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=3000); ap.add_argument("--knob", type=int, default=0)
a = ap.parse_args()
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
params = {k: dev(v) for k, v in gi.ar_multiconv2d_params(np.random.RandomState(7), 32, [160, 160], [32, 32]).items()}
xs, rc = iaf_amd.ARStack(32, [160, 160]), iaf_amd.ARStack(32, [160, 160])
rc.set_halo_exchange(False); xs.set_halo_exchange_debug(a.knob)
xs.prepare(params); rc.prepare(params)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
big = torch.empty(64 << 20, dtype=torch.float32, device="cuda"); big2 = torch.empty_like(big); small = torch.zeros(1024, device="cuda")
g = torch.Generator(device="cuda").manual_seed(1)
worst = torch.zeros((), device="cuda"); bad = torch.zeros((), device="cuda")
t0 = time.time()
sizes = [32, 5, 64, 17, 128, 32, 300, 8]
for it in range(a.iters):
    B = sizes[(it // 250) % len(sizes)]
    with torch.cuda.stream(sb):
        if it % 3 == 0: big2.copy_(big, non_blocking=True)
        for _ in range(4): small.add_(1.0)
    with torch.cuda.stream(sa):
        z = torch.randn(B, 32, 16, 16, device="cuda", generator=g); ctx = torch.randn(B, 160, 16, 16, device="cuda", generator=g)
        zx, sx = xs.iaf_step(z, ctx); zr, sr = rc.iaf_step(z, ctx)
        d = torch.maximum((sx - sr).abs().max(), ((zx - zr).abs() / (1.0 + zr.abs())).max())
        worst = torch.maximum(worst, torch.nan_to_num(d, nan=1e9)); bad = bad + (~torch.isfinite(zx)).any().float()
    if it % 500 == 499:
        torch.cuda.synchronize()
        print("iter %5d  B=%3d  worst |diff| so far %.3g  non-finite outputs %d  errors %d  (%.0f s)" % (it + 1, B, float(worst), int(bad), xs.exchange_errors(), time.time() - t0), flush=True)
torch.cuda.synchronize()
ok = float(worst) < 5e-5 and int(bad) == 0 and xs.exchange_errors() == 0
print("XCH STRESS", "CLEAN" if ok else "FAILED", "worst %.3g" % float(worst))
sys.exit(0 if ok else 1)
