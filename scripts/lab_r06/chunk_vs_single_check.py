import sys, torch
sys.path.insert(0, '/root/repo')
from depthg_amd import ContrastiveCorrelationLoss, ops
from oracle import depthg_oracle as O
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(77)
B, C, D, hw, N = 3, 768, 70, 20, 2
S = 14
f, fp = torch.randn(B, C, hw, hw, generator=g), torch.randn(B, C, hw, hw, generator=g)
c, cp = torch.randn(B, D, hw, hw, generator=g), torch.randn(B, D, hw, hw, generator=g)
d = torch.randint(0, 256, (B, 1, 80, 80), generator=g).float()
perms = [O.super_perm(B, g) for _ in range(N)]
cfg = O.default_cfg(feature_samples=S, neg_samples=N, dim=D, dg_outputs="reduced")
c1, c2 = (torch.rand(B, S, S, 2, generator=g) * 2.2 - 1.1), (torch.rand(B, S, S, 2, generator=g) * 2.2 - 1.1)
cr, cpr = c.clone().requires_grad_(True), cp.clone().requires_grad_(True)
ref = O.forward(cfg, f, fp, cr, cpr, d, d, coords1=c1, coords2=c2, perms=perms)
O.total_loss(cfg, ref).backward()
for limit in (768, 384):
    ops.BLOB_MAX_C = limit
    cg, cpg = c.to(dev).requires_grad_(True), cp.to(dev).requires_grad_(True)
    out = ContrastiveCorrelationLoss(cfg).forward_with(f.to(dev), fp.to(dev), cg, cpg, d.to(dev), c1.to(dev), c2.to(dev), [p.to(dev) for p in perms])
    O.total_loss(cfg, out).backward()
    print(limit, [ '%.2e' % (abs(float(o.detach().mean()) - float(r.detach().mean())) / (abs(float(r.detach().mean())) + 1e-12)) for o, r in zip(out, ref)],
          'grad rel', '%.2e %.2e' % (float((cg.grad.cpu() - cr.grad).norm() / cr.grad.norm()), float((cpg.grad.cpu() - cpr.grad).norm() / cpr.grad.norm())))
