"""A/B check of an experiment library (IAF_HIP_LIB=...) on the posterior block: kl_obj / kl_cost against torch reductions of
the block's own KL elements (tf_train.py:77-85), and the time per block.  python tools/kl_fold_check.py [--batch 32]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import golden_inputs as gi  # noqa: E402
import iaf_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    B, n_z, n_h = a.batch, 32, 160
    rng = np.random.RandomState(5)
    params = gi.ar_multiconv2d_params(rng, n_z, [n_h, n_h], [n_z, n_z])
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    st = iaf_amd.ARStack(n_z, [n_h, n_h])
    st.prepare({k: dev(v) for k, v in params.items()})
    print("library:", os.environ.get("IAF_HIP_LIB", "product"))
    for hw in (16, 8):
        f = lambda c, s=1.0: dev(s * rng.standard_normal((B, c, hw, hw)))
        pin = [f(n_z), f(n_z, 0.25), f(n_z), f(n_z, 0.25), f(n_z), f(n_z, 0.25), f(n_h), f(n_h), f(n_z)]
        for kl_min in (0.25, 0.0):
            out = st.posterior_block(*pin, kl_min, want_kl_elem=True)
            torch.cuda.synchronize()
            S = out["kl_elem"].double().sum(dim=(2, 3))                       # [B, Z]
            cost = S.sum(dim=1)
            obj = torch.clamp(S.mean(dim=0), min=kl_min).sum().expand(B) if kl_min > 0 else cost
            e1 = float((out["kl_cost"].double() - cost).abs().max() / cost.abs().max())
            e2 = float((out["kl_obj"].double() - obj).abs().max() / obj.abs().max())
            out2 = st.posterior_block(*pin, kl_min)                            # without the KL tensor
            torch.cuda.synchronize()
            e3 = float((out2["kl_obj"] - out["kl_obj"]).abs().max())
            print("  %2dx%-2d kl_min=%.2f  rel err kl_cost %.1e kl_obj %.1e  (no kl_elem: |d kl_obj| %.1e, one launch: %d)"
                  % (hw, hw, kl_min, e1, e2, e3, st.step_is_fused(B, hw, hw)))
            assert e1 < 1e-5 and e2 < 1e-5 and e3 < 1e-3
        for _ in range(20):
            st.posterior_block(*pin, 0.25)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            st.posterior_block(*pin, 0.25)
        torch.cuda.synchronize()
        print("  %2dx%-2d posterior block %.1f us (eager, back to back)" % (hw, hw, 1e6 * (time.perf_counter() - t0) / 300))


if __name__ == "__main__":
    main()
