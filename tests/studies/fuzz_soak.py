#!/usr/bin/env python
"""Extended randomised soak (GPU): many random (channels, depth, batch, geometry) problems through the masked stack,
the inverse, the plain / masked single convs with fused options, and backward of the plain convs, each against the
oracle.  Not collected by pytest (minutes of runtime); python tests/studies/fuzz_soak.py [n_cases] [seed]."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_inputs as gi  # noqa: E402
import iaf_amd  # noqa: E402
from oracle import iaf_oracle as O  # noqa: E402
from oracle import iaf_grad_oracle as G  # noqa: E402

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
host = lambda t: t.detach().cpu().numpy().astype(np.float64)
f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.RandomState(seed)
    worst = 0.0
    for it in range(n):
        kind = rng.randint(3)
        B = int(rng.choice([1, 2, 3, 4, 7]))
        H, W = int(rng.randint(1, 13)), int(rng.randint(1, 21))
        if kind == 0:      # masked stack + inverse
            n_z = int(rng.choice([16, 32, 48, 64, 4, 6]))
            mult = int(rng.choice([1, 2, 3, 5]))
            n_h = n_z * mult if rng.randint(2) or n_z < 16 else max(16, n_z // int(rng.choice([1, 2])))
            if not (n_h % n_z == 0 or n_z % n_h == 0) or n_h > 256:
                n_h = n_z
            d = int(rng.randint(0, 4))
            params = gi.ar_multiconv2d_params(rng, n_z, [n_h] * d, [n_z, n_z])
            z, ctx = rng.standard_normal((B, n_z, H, W)), rng.standard_normal((B, n_h, H, W))
            st = iaf_amd.ARStack(n_z, [n_h] * d)
            st.prepare({k: dev(v) for k, v in params.items()})
            zn, ls = st.iaf_step(dev(z), dev(ctx) if d else None)
            ez, es = O.iaf_step(f32(z), f32(ctx), {k: f32(v) for k, v in params.items()}, [n_h] * d)
            e = max(np.abs(host(zn) - ez).max(), np.abs(host(ls) - es).max())
            if d:
                back, _, _, _ = st.iaf_step_inverse(zn, dev(ctx), max_sweeps=H * W * n_z + 2, tol=1e-6, check_every=1)
                e = max(e, np.abs(host(back) - f32(z)).max())
            desc = "stack z%d h%d d%d B%d %dx%d" % (n_z, n_h, d, B, H, W)
        else:               # plain / masked single conv, fused options, (kind 2: + backward on the MFMA path)
            n_in = int(rng.choice([16, 32, 48, 80, 160, 192, 5, 12]))
            n_out = int(rng.choice([16, 32, 64, 96, 160, 7, 20]))
            m = int(rng.randint(3))
            if m and not (n_in % n_out == 0 or n_out % n_in == 0):
                m = 0
            elu, res = bool(rng.randint(2)), bool(rng.randint(2))
            p = gi.conv_params(rng, n_in, n_out)
            x, r = rng.standard_normal((B, n_in, H, W)), rng.standard_normal((B, n_out, H, W))
            cv = iaf_amd.WNConv2d(n_in, n_out, ar_mask=None if m == 0 else (m == 2))
            train = kind == 2 and m == 0 and n_in % 16 == 0 and n_out % 16 == 0
            if train:
                cv.set_training(True)
            V, g, b = dev(p["V"]), dev(p["g"]), dev(p["b"])
            cv.prepare(V, g, b)
            y = cv(dev(x), elu_input=elu, residual=dev(r) if res else None)[0]
            xin = O.elu(f32(x)) if elu else f32(x)
            ey = O.conv2d(xin, f32(p["V"]), f32(p["g"]), f32(p["b"])) if m == 0 else \
                O.ar_conv2d(xin, f32(p["V"]), f32(p["g"]), f32(p["b"]), zerodiagonal=(m == 2))
            if res:
                ey = f32(r) + 0.1 * ey
            e = np.abs(host(y) - ey).max()
            desc = "conv %d->%d m%d elu%d res%d B%d %dx%d%s" % (n_in, n_out, m, elu, res, B, H, W, " +bwd" if train else "")
            if train:
                dy = rng.standard_normal((B, n_out, H, W))
                xt = G._t(f32(x), True)
                pt = {k: G._t(f32(v), True) for k, v in p.items()}
                yy = G.conv2d(torch.nn.functional.elu(xt) if elu else xt, pt["V"], pt["g"], pt["b"])
                (yy * G._t(f32(dy))).sum().backward()
                (dx,), dV, dg, db = cv.backward(dev(x), [dev(dy)], V, g, elu_input=elu)
                for got, want in ((dx, xt.grad), (dV, pt["V"].grad), (dg, pt["g"].grad), (db, pt["b"].grad)):
                    w_ = want.numpy()
                    e = max(e, np.abs(host(got) - w_).max() / max(np.abs(w_).max(), 1e-30))
        worst = max(worst, e)
        if e > 1e-4 or not np.isfinite(e):
            print("FAIL case %d: %s  err %.3g" % (it, desc, e))
            sys.exit(1)
    print("soak ok: %d cases, worst error %.3g (seed %d)" % (n, worst, seed))


if __name__ == "__main__":
    main()
