import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from depthg_amd import ContrastiveCorrelationLoss
from oracle import depthg_oracle as O
dev = torch.device("cuda:0")
B, C, D, hw, S, N = 2, 64, 32, 14, 11, 2
g = torch.Generator().manual_seed(5)
f, fp = torch.randn(B, C, hw, hw, generator=g), torch.randn(B, C, hw, hw, generator=g)
c, cp = torch.randn(B, D, hw, hw, generator=g), torch.randn(B, D, hw, hw, generator=g)
d = torch.randint(0, 256, (B, 1, 4 * hw, 4 * hw), generator=g).float()
c1 = torch.rand(B, S, S, 2, generator=g) * 2 - 1
c2 = torch.rand(B, S, S, 2, generator=g) * 2 - 1
perms = [O.super_perm(B, g) for _ in range(N)]
cfg = O.default_cfg(feature_samples=S, neg_samples=N, dim=D, dg_outputs="full")
ref = O.forward(cfg, f, fp, c, cp, d, d, coords1=c1, coords2=c2, perms=perms)
out = ContrastiveCorrelationLoss(cfg).forward_with(f.to(dev), fp.to(dev), c.to(dev), cp.to(dev), d.to(dev), c1.to(dev), c2.to(dev), [p.to(dev) for p in perms])
P = S * S
for i, name in ((1, "intra"), (3, "inter"), (5, "neg")):
    e = (out[i].cpu() - ref[i]).abs().reshape(-1, P, P)
    print(name, "max", float(e.max()), "rows with err>2e-5:", sorted(set((e > 2e-5).nonzero()[:, 1].tolist()))[:40], "cols:", sorted(set((e > 2e-5).nonzero()[:, 2].tolist()))[:40])
    n, p, q = [int(v) for v in (e == e.max()).nonzero()[0]]
    print("   worst at image", n, "p", p, "q", q, "ref", float(ref[i].reshape(-1, P, P)[n, p, q]), "got", float(out[i].cpu().reshape(-1, P, P)[n, p, q]))
x = O.norm(O.sample(c, c1)).reshape(B, D, -1).permute(0, 2, 1)
y = O.norm(O.sample(cp, c2)).reshape(B, D, -1).permute(0, 2, 1)
got = out[3].cpu().reshape(-1, P, P)[0, :, 20]
refc = ref[3].reshape(-1, P, P)[0, :, 20]
yh = y[0, 20].half().float()
ylo = ((y[0, 20] - yh) * 2048).half().float() / 2048
xh = x[0].half().float()
xlo = ((x[0] - xh) * 2048).half().float() / 2048
cand = {"y hi only": (x[0].double() @ yh.double()), "no y_lo term (xh*yh + xl*yh)": ((xh + xlo).double() @ yh.double()),
        "no x_lo term": (xh.double() @ (yh + ylo).double()), "hi*hi only": (xh.double() @ yh.double())}
print("got - ref     :", [f"{float(v):+.2e}" for v in (got - refc)[:10]])
for k, v in cand.items():
    print(f"{k:28s}:", [f"{float(u):+.2e}" for u in (v.float() - refc)[:10]], " max|got - cand|", float((got.double() - v).abs().max()))
