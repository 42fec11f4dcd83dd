import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import numpy as np, torch, golden_inputs as gi, iaf_amd
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
H = int(sys.argv[1]) if len(sys.argv) > 1 else 16
junk = torch.full((int(sys.argv[2]) if len(sys.argv) > 2 else 1,), float("nan"), device="cuda")   # poison some memory first
big = torch.full((64 << 20,), float("nan"), device="cuda"); del big                                  # ... and the allocator cache
rng = np.random.RandomState(2024 + H)
B, n_z, n_h, d = 32, 32, 160, 2
params = gi.ar_multiconv2d_params(rng, n_z, [n_h]*d, [n_z, n_z])
z = dev(rng.standard_normal((B, n_z, H, H))); ctx = dev(rng.standard_normal((B, n_h, H, H)))
st = iaf_amd.ARStack(n_z, [n_h]*d); st.prepare({k: dev(v) for k, v in params.items()})
z0, s0 = st.iaf_step(z, ctx)
qh, qw, c = H // 2, H // 2 - 1, 11
z2 = z.clone(); z2[:, c, qh, qw] += 0.5
z1, s1 = st.iaf_step(z2, ctx)
za, sa = st.iaf_step(z, ctx)
torch.cuda.synchronize()
allowed = torch.zeros(z.shape, dtype=torch.bool, device="cuda")
allowed[:, :, :qh, :] = True; allowed[:, :, qh, :qw] = True
allowed_s = allowed.clone(); allowed_s[:, c + 1:, qh, qw] = True
allowed_z = allowed_s.clone(); allowed_z[:, c, qh, qw] = True
dz, ds = (z1 != z0), (s1 != s0)
vz, vs = (dz & ~allowed_z), (ds & ~allowed_s)
det = torch.equal(za, z0) and torch.equal(sa, s0)
nanz = int(torch.isnan(z0).sum()) + int(torch.isnan(z1).sum())
print("H", H, "viol_z", int(vz.sum()), "viol_s", int(vs.sum()), "first==third", det, "nan", nanz,
      "where", vz.nonzero()[:3].tolist(), (z1 - z0)[vz][:3].tolist() if vz.any() else "")
