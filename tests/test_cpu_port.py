"""The torch-CPU port timed as bench.py's cpu_baseline must agree with the oracle. CPU only."""
import numpy as np
import torch

import golden_inputs as gi
from oracle import iaf_cpu_port as Pt
from oracle import iaf_oracle as O


def test_cpu_port_matches_oracle():
    rng = np.random.RandomState(4)
    n_z, n_h, B, H, W = 32, [64, 64], 2, 6, 5
    params = gi.ar_multiconv2d_params(rng, n_z, n_h, [n_z, n_z])
    z = rng.standard_normal((B, n_z, H, W)).astype(np.float32)
    ctx = rng.standard_normal((B, n_h[0], H, W)).astype(np.float32)
    tp = Pt.as_torch(params)
    w = Pt.prepare_weights(tp, n_z, n_h)
    zn, s = Pt.iaf_step(torch.from_numpy(z), torch.from_numpy(ctx), w, len(n_h))
    p32 = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in params.items()}
    ez, es = O.iaf_step(z.astype(np.float64), ctx.astype(np.float64), p32, n_h)
    np.testing.assert_allclose(zn.numpy(), ez, atol=1e-4)
    np.testing.assert_allclose(s.numpy(), es, atol=1e-4)
