#!/usr/bin/env python3
"""developer aid (round 6): where the NaN of the exact-mask form comes from - scalars, gradient NaN counts, against the oracle."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from depthg_amd import ContrastiveCorrelationLoss
from oracle import depthg_oracle as O
dev = torch.device("cuda:0")
B, hw = 2, 28
g = torch.Generator().manual_seed(728)
f, fp = torch.randn(B, 384, hw, hw, generator=g), torch.randn(B, 384, hw, hw, generator=g)
c, cp = torch.randn(B, 70, hw, hw, generator=g), torch.randn(B, 70, hw, hw, generator=g)
d = torch.randint(0, 256, (B, 1, 8 * hw, 8 * hw), generator=g).float()
perms = [O.super_perm(B, g) for _ in range(5)]
co = O.identity_coords(B, hw)
for flag in (True, False):
    cfg = O.default_cfg(feature_samples=hw, dg_outputs="reduced", dg_exact_masks=flag)
    cg, cpg = c.to(dev).requires_grad_(True), cp.to(dev).requires_grad_(True)
    loss = ContrastiveCorrelationLoss(cfg)
    out = loss.forward_with(f.to(dev), fp.to(dev), cg, cpg, d.to(dev), co.to(dev), co.to(dev), [p.to(dev) for p in perms], shared_coords=True, identity_grid=True)
    print("exact", flag, "scalars", [float(x) for x in loss.last_scalars])
    loss.total.backward()
    torch.cuda.synchronize()
    for name, t in (("code", cg.grad), ("code_pos", cpg.grad)):
        print("   grad", name, "nan", int(torch.isnan(t).sum()), "inf", int(torch.isinf(t).sum()), "absmax", float(t.nan_to_num(0).abs().max()))
