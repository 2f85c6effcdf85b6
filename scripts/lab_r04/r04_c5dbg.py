import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthg_amd import ContrastiveCorrelationLoss
from depthg_amd.loss import identity_coords
from oracle import depthg_oracle as O
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(505)
B, C, D, hw, N = 2, 384, 70, int(sys.argv[1]) if len(sys.argv) > 1 else 56, 2
f, fp = torch.randn(B, C, hw, hw, generator=g), torch.randn(B, C, hw, hw, generator=g)
c, cp = torch.randn(B, D, hw, hw, generator=g), torch.randn(B, D, hw, hw, generator=g)
d = torch.randint(0, 256, (B, 1, 448, 448), generator=g).float()
cfg = O.default_cfg(feature_samples=hw, neg_samples=N, dim=D, dg_outputs="reduced")
co = identity_coords(B, hw, dev)
perms = [O.super_perm(B, g).to(dev) for _ in range(N)]
lf = ContrastiveCorrelationLoss(cfg)
cg = c.to(dev).requires_grad_(True)
lf.forward_with(f.to(dev), fp.to(dev), cg, cp.to(dev).requires_grad_(True), d.to(dev), co, co, perms, shared_coords=True, identity_grid=True)
print(os.environ.get("DEPTHG_LIB", "production")[-20:], [f"{v:.6e}" for v in lf.last_scalars.tolist()])
