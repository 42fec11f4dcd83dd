"""debug: which conv of an IAFLayer raises the fp16 range word, and on what input"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden")); sys.path.insert(0, os.path.join(R, "tests"))
import golden_inputs as gi
import iaf_amd as amd
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
B, zs, hs, H, W = 16, 32, 160, 16, 16
c = gi.layer_case_inputs("layer_cfg2_8x8")
rng = np.random.RandomState(3)
params = {k: dev(v) for k, v in c["params"].items()}
layer = amd.IAFLayer(zs, hs, depth_ar=2, kl_min=0.0)
layer.load(params)
convs = dict(up_conv1=layer.up_conv1, up_conv3=layer.up_conv3, down_conv1=layer.down_conv1, down_conv2=layer.down_conv2)
pr = lambda tag: print(tag, {k: v.range_errors() for k, v in convs.items()}, "stack", layer.posterior.stack.range_errors(), flush=True)
pr("after load")
up_in, down_in = rng.standard_normal((B, hs, H, W)), rng.standard_normal((B, hs, H, W))
eps = rng.standard_normal((B, zs, H, W))
layer.up(dev(up_in))
pr("after up")
out, kl_obj, kl_cost = layer.down(dev(down_in), dev(eps))
torch.cuda.synchronize()
pr("after down")
print("z absmax", layer.last_block["z"].abs().max().item(), "out absmax", out.abs().max().item(), torch.isfinite(out).all().item())
