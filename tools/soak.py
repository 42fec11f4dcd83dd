"""Determinism soak of the IAF step (one autoregressive-property violation in ~700 test runs).

    python tools/soak.py [--iters 20000] [--fresh 1500] [--B 32] [--hw 16 8]

Two phases per latent size, every output compared BIT FOR BIT on the device against the first run's:
  steady  one stack, `iters` x (step(z), step(z_perturbed)), with churn every few iterations: training forward/backward
          on a second stack, allocator traffic, a third stack created / prepared / destroyed (hipMalloc + hipFree while
          work is queued), launch-shape changes on the churn stack;
  fresh   `fresh` x (new stack, prepare, step(z), step(z_perturbed)) -- the pattern of the pytest run that showed the
          violation (first launches right behind hipMalloc + the prep kernel).
Also checks the autoregressive property itself and NaN-freeness (meaningful with the -DIAF_EXP_POISON_LDS build:
IAF_HIP_LIB=iaf_amd/_lib_poison/libiaf_hip.so, where every LDS word a launch did not write reads as NaN).
Exit code 1 on any mismatch."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch

import golden_inputs as gi
import iaf_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def run(B, H, iters, fresh, n_z=32, n_h=160, d=2):
    rng = np.random.RandomState(2024 + H)
    params = gi.ar_multiconv2d_params(rng, n_z, [n_h] * d, [n_z, n_z])
    z, ctx = dev(rng.standard_normal((B, n_z, H, H))), dev(rng.standard_normal((B, n_h, H, H)))
    dp = {k: dev(v) for k, v in params.items()}
    qh, qw, c = H // 2, H // 2 - 1, 11
    z2 = z.clone()
    z2[:, c, qh, qw] += 0.5
    allowed = torch.zeros(z.shape, dtype=torch.bool, device="cuda")
    allowed[:, :, :qh, :] = True
    allowed[:, :, qh, :qw] = True
    allowed_s = allowed.clone()
    allowed_s[:, c + 1:, qh, qw] = True
    allowed_z = allowed_s.clone()
    allowed_z[:, c, qh, qw] = True

    st = iaf_amd.ARStack(n_z, [n_h] * d)
    st.prepare(dp)
    ref = [t.clone() for t in (*st.iaf_step(z, ctx), *st.iaf_step(z2, ctx))]
    torch.cuda.synchronize()
    nan0 = sum(int(torch.isnan(t).sum()) for t in ref)
    viol = int(((ref[2] != ref[0]) & ~allowed_z).sum()) + int(((ref[3] != ref[1]) & ~allowed_s).sum())
    print("B=%d %dx%d reference run: nan=%d autoregressive violations=%d" % (B, H, H, nan0, viol), flush=True)
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    first_bad = [None]

    def check(outs, it, phase):
        nonlocal bad
        m = torch.zeros((), dtype=torch.int64, device="cuda")
        for o, r in zip(outs, ref):
            m += (o != r).sum()              # NaN != NaN also counts
        bad += (m > 0).to(torch.int64)
        if it % 256 == 255 and first_bad[0] is None and int(bad) > 0:    # one read-back per 256 iterations
            first_bad[0] = (phase, it)
            for name, o, r in zip(("z0", "s0", "z1", "s1"), outs, ref):
                w = (o != r).nonzero()
                if len(w):
                    print("  MISMATCH in %s near %s iteration %d: %d elements, first %s got %r ref %r" %
                          (name, phase, it, len(w), w[0].tolist(), float(o[tuple(w[0])]), float(r[tuple(w[0])])), flush=True)

    # ---- steady phase with churn
    st2 = iaf_amd.ARStack(n_z, [n_h] * d)
    st2.set_training(True)
    st2.prepare(dp)
    shapes = [(5, 4, 1, 1), (5, 2, 1, 2), (5, 2, 2, 1), (5, 1, 1, 4)]
    t0 = time.time()
    for it in range(iters):
        if it % 7 == 0:
            a, b = st2.iaf_step_train(z, ctx)
            st2.iaf_step_backward(z, ctx, a, b, torch.randn_like(z), torch.randn_like(z), dp)
        if it % 11 == 0:
            junk = torch.full((1 << (16 + it % 9),), float("nan"), device="cuda")
            del junk
        if it % 53 == 0:
            st3 = iaf_amd.ARStack(n_z, [n_h] * d)
            st3.prepare(dp)
            st3.iaf_step(z2, ctx)
            del st3
        if it % 97 == 0:
            s = shapes[(it // 97) % len(shapes)]
            try:
                st2.set_tuning(1, *s)
            except ValueError:
                pass
        check((*st.iaf_step(z, ctx), *st.iaf_step(z2, ctx)), it, "steady")
    torch.cuda.synchronize()
    n_steady = int(bad)
    print("  steady: %d iterations, %d with a bit difference (%.1f s)" % (iters, n_steady, time.time() - t0), flush=True)

    # ---- fresh-stack phase
    t0 = time.time()
    for it in range(fresh):
        s = iaf_amd.ARStack(n_z, [n_h] * d)
        s.prepare({k: v.clone() for k, v in dp.items()} if it % 3 == 0 else dp)
        check((*s.iaf_step(z, ctx), *s.iaf_step(z2, ctx)), it, "fresh")
        del s
    torch.cuda.synchronize()
    n_fresh = int(bad) - n_steady
    print("  fresh:  %d stacks, %d with a bit difference (%.1f s)" % (fresh, n_fresh, time.time() - t0), flush=True)
    return nan0 + viol + n_steady + n_fresh


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20000)
    ap.add_argument("--fresh", type=int, default=1500)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--hw", type=int, nargs="+", default=[16, 8])
    a = ap.parse_args()
    print("library:", iaf_amd._capi.LIB_PATH, flush=True)
    total = 0
    for H in a.hw:
        total += run(a.B, H, a.iters, a.fresh)
    print("SOAK", "CLEAN" if total == 0 else "FAILED (%d)" % total)
    sys.exit(0 if total == 0 else 1)
