#!/usr/bin/env python3
"""developer aid: gradient error of the HIP path against the oracle at the headline width (B=4 subset) and on small shapes,
with and without zero_clamp: relative L2, cosine, largest element error relative to the largest gradient element."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthg_amd import ContrastiveCorrelationLoss
from oracle import depthg_oracle as O
dev = torch.device("cuda:0")
for (B, C, D, hw, zc, seed) in [(4, 384, 70, 28, True, 4321), (4, 384, 70, 28, False, 4321), (2, 384, 70, 28, True, 7), (4, 384, 64, 28, True, 11),
                                (2, 768, 100, 28, True, 5), (2, 384, 70, 56, True, 9)]:
    g = torch.Generator().manual_seed(seed)
    f, fp = torch.randn(B, C, hw, hw, generator=g), torch.randn(B, C, hw, hw, generator=g)
    c, cp = torch.randn(B, D, hw, hw, generator=g), torch.randn(B, D, hw, hw, generator=g)
    d = torch.randint(0, 256, (B, 1, 8 * hw, 8 * hw), generator=g).float()
    perms = [O.super_perm(B, g) for _ in range(5)]
    cfg = O.default_cfg(feature_samples=hw, dg_outputs="reduced", zero_clamp=zc, dim=D)
    coords = O.identity_coords(B, hw)
    cr, cpr = c.clone().requires_grad_(True), cp.clone().requires_grad_(True)
    ref = O.forward(cfg, f, fp, cr, cpr, d, d, coords1=coords, coords2=coords, perms=perms)
    O.total_loss(cfg, ref).backward()
    cg, cpg = c.to(dev).requires_grad_(True), cp.to(dev).requires_grad_(True)
    out = ContrastiveCorrelationLoss(cfg).forward_with(f.to(dev), fp.to(dev), cg, cpg, d.to(dev), coords.to(dev), coords.to(dev),
                                                       [p.to(dev) for p in perms], shared_coords=True, identity_grid=True)
    O.total_loss(cfg, out).backward()
    for name, got, want in (("code", cg.grad.cpu(), cr.grad), ("code_pos", cpg.grad.cpu(), cpr.grad)):
        rel = float((got - want).norm() / want.norm())
        cos = float((got * want).sum() / (got.norm() * want.norm()))
        mx = float((got - want).abs().max() / want.abs().max())
        print(f"B={B} C={C} D={D} {hw}x{hw} zero_clamp={zc}: d/d{name:8s} rel-L2 {rel:.2e}  1-cos {1-cos:.1e}  max|err|/max|g| {mx:.2e}")
