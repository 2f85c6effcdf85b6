"""developer check of k_corr_small's cd precision: materialised cd tensors against the oracle's fp32 / an fp64 recomputation"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from depthg_amd import ContrastiveCorrelationLoss
from oracle import depthg_oracle as O

dev = torch.device("cuda:0")
for (B, C, D, hw, S, N) in ((2, 64, 32, 14, 11, 2), (2, 2048, 32, 7, 11, 2), (3, 96, 90, 12, 12, 2), (2, 64, 100, 14, 6, 1)):
    g = torch.Generator().manual_seed(5)
    f, fp = torch.randn(B, C, hw, hw, generator=g), torch.randn(B, C, hw, hw, generator=g)
    c, cp = torch.randn(B, D, hw, hw, generator=g), torch.randn(B, D, hw, hw, generator=g)
    d = torch.randint(0, 256, (B, 1, 4 * hw, 4 * hw), generator=g).float()
    c1 = torch.rand(B, S, S, 2, generator=g) * 2 - 1
    c2 = torch.rand(B, S, S, 2, generator=g) * 2 - 1
    perms = [O.super_perm(B, g) for _ in range(N)]
    cfg = O.default_cfg(feature_samples=S, neg_samples=N, dim=D, dg_outputs="full")
    ref = O.forward(cfg, f, fp, c, cp, d, d, coords1=c1, coords2=c2, perms=perms)
    out = ContrastiveCorrelationLoss(cfg).forward_with(f.to(dev), fp.to(dev), c.to(dev), cp.to(dev), d.to(dev), c1.to(dev), c2.to(dev),
                                                       [p.to(dev) for p in perms])
    print(f"B={B} C={C} D={D} hw={hw} S={S}:", " ".join(f"t[{i}] max|d|={float((out[i].cpu() - ref[i]).abs().max()):.2e}" for i in (1, 3, 5, 7)),
          " losses", " ".join(f"{abs(float(out[i].mean()) - float(ref[i].mean())) / abs(float(ref[i].mean())):.1e}" for i in (0, 2, 4, 6)))
