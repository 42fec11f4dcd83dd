"""dev probe: where the HOST time of one IAFLayer backward goes (cProfile over eager calls)"""
import cProfile, pstats, os, sys, io
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import golden_inputs as gi, iaf_amd
B, zs, hs, H = 32, 32, 160, 16
c = gi.layer_case_inputs("layer_cfg2_8x8")
params = {k: torch.from_numpy(np.asarray(v, np.float32)).cuda() for k, v in c["params"].items()}
layer = iaf_amd.IAFLayer(zs, hs, depth_ar=2, kl_min=0.25)
layer.set_training(True); layer.load(params)
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda ch: torch.randn((B, ch, H, H), device="cuda", generator=g)
up_in, down_in, eps, dU, dD = rn(hs), rn(hs), rn(zs), rn(hs), rn(hs)
dK = torch.ones(B, device="cuda"); grads = {}
def fwd():
    layer.up_train(up_in); return layer.down_train(down_in, eps)
def bwd():
    layer.down_backward(dD, dK, params, grads); layer.up_backward(dU, params, grads)
for _ in range(3):
    try: fwd()
    except Exception as e: print("fwd raised", type(e).__name__)
bwd(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): bwd()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:6000])
