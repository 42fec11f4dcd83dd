#!/usr/bin/env python
"""GPU dev tool: convergence and cost of the Jacobi inverse of the IAF step at BASELINE configs[1] sizes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
import golden_inputs as gi  # noqa: E402
import iaf_amd  # noqa: E402


def main():
    B, n_z, n_h, d = 32, 32, 160, 2
    for H in (16, 8):
        rng = np.random.RandomState(0)
        params = gi.ar_multiconv2d_params(rng, n_z, [n_h] * d, [n_z, n_z])
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
        stack = iaf_amd.ARStack(n_z, [n_h] * d)
        stack.prepare({k: dev(v) for k, v in params.items()})
        z0, ctx = dev(rng.standard_normal((B, n_z, H, H))), dev(rng.standard_normal((B, n_h, H, H)))
        z, _ = stack.iaf_step(z0, ctx)
        errs = []
        for n in range(1, 13):
            back, _, _, _ = stack.iaf_step_inverse(z, ctx, max_sweeps=n, tol=0.0)
            errs.append(float((back - z0).abs().max()))
        print("%dx%d  max|z0_rec - z0| after n sweeps: %s" % (H, H, " ".join("%d:%.1e" % (i + 1, e) for i, e in enumerate(errs))))
        _, _, sweeps, res = stack.iaf_step_inverse(z, ctx, max_sweeps=200, tol=1e-6, check_every=1)
        print("      tol 1e-6 reached after %d sweeps (last update %.2e)" % (sweeps, res))
        # the whole call with the residual tested on the device every 2 sweeps, 16 sweeps queued, early-out behind the converged one
        for _ in range(3):
            stack.iaf_step_inverse_queued(z, ctx, max_sweeps=16, tol=1e-6, check_every=2)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            stack.iaf_step_inverse_queued(z, ctx, max_sweeps=16, tol=1e-6, check_every=2)
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / 20 * 1e3
        print("      tol 1e-6, 16 sweeps queued, test every 2, no host in the loop: %.1f us per inverse -> %.0f samples/s" % (t, B / t * 1e6))
        g = torch.cuda.CUDAGraph()
        out = (torch.empty_like(z), torch.empty_like(z))
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            side.wait_stream(torch.cuda.current_stream())
            stack.iaf_step_inverse_queued(z, ctx, max_sweeps=16, tol=1e-6, check_every=2, out=out)
            side.synchronize()
            with torch.cuda.graph(g, stream=side):
                stack.iaf_step_inverse_queued(z, ctx, max_sweeps=16, tol=1e-6, check_every=2, out=out)
            for _ in range(5):
                g.replay()
            side.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(side)
            for _ in range(20):
                g.replay()
            b.record(side)
            b.synchronize()
        t = a.elapsed_time(b) / 20 * 1e3
        print("      the same as ONE hipGraph replay: %.1f us per inverse -> %.0f samples/s (max |z0_rec - z0| %.1e)" % (
            t, B / t * 1e6, float((out[0] - z0).abs().max())))
        for n in (sweeps, 8):
            stack.iaf_step_inverse(z, ctx, max_sweeps=n, tol=0.0)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                stack.iaf_step_inverse(z, ctx, max_sweeps=n, tol=0.0)
            b.record()
            torch.cuda.synchronize()
            t = a.elapsed_time(b) / 20 * 1e3
            print("      %d sweeps, no sync: %.1f us per inverse (%.1f us per sweep) -> %.0f samples/s" % (n, t, t / n, B / t * 1e6))


if __name__ == "__main__":
    main()
