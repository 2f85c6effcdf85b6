import sys, torch
sys.path.insert(0, '/root/repo')
from depthg_amd import ContrastiveCorrelationLoss
from oracle import depthg_oracle as O
dev = torch.device('cuda:0')
for (B, C, D, hw, S, N, shared, full) in ((2, 1024, 70, 20, 16, 2, False, False), (3, 1536, 24, 18, 14, 1, True, True), (2, 800, 90, 24, 13, 3, False, False)):
    g = torch.Generator().manual_seed(C + hw)
    f, fp = torch.randn(B, C, hw, hw, generator=g), torch.randn(B, C, hw, hw, generator=g)
    c, cp = torch.randn(B, D, hw, hw, generator=g), torch.randn(B, D, hw, hw, generator=g)
    d = torch.randint(0, 256, (B, 1, 4 * hw, 4 * hw), generator=g).float()
    perms = [O.super_perm(B, g) for _ in range(N)]
    cfg = O.default_cfg(feature_samples=S, neg_samples=N, dim=D, dg_outputs="full" if full else "reduced")
    if shared:
        c1 = (torch.rand(1, S, S, 2, generator=g).expand(B, S, S, 2).contiguous()) * 2.2 - 1.1
        c2 = c1
    else:
        c1, c2 = torch.rand(B, S, S, 2, generator=g) * 2.2 - 1.1, torch.rand(B, S, S, 2, generator=g) * 2.2 - 1.1
    cr, cpr = c.clone().requires_grad_(True), cp.clone().requires_grad_(True)
    ref = O.forward(cfg, f, fp, cr, cpr, d, d, coords1=c1, coords2=c2, perms=perms)
    O.total_loss(cfg, ref).backward()
    cg, cpg = c.to(dev).requires_grad_(True), cp.to(dev).requires_grad_(True)
    out = ContrastiveCorrelationLoss(cfg).forward_with(f.to(dev), fp.to(dev), cg, cpg, d.to(dev), c1.to(dev), c2.to(dev), [p.to(dev) for p in perms],
                                                       shared_coords=shared)
    O.total_loss(cfg, out).backward()
    for i in range(len(ref)):
        a, b = float(out[i].detach().mean()), float(ref[i].detach().mean())
        extra = ''
        if full and out[i].dim() > 0:
            extra = ' maxerr %.2e shape %s' % (float((out[i].detach().cpu() - ref[i].detach()).abs().max()), tuple(out[i].shape) == tuple(ref[i].shape))
        print(C, S, i, '%.6e %.6e rel %.2e' % (a, b, abs(a - b) / (abs(b) + 1e-12)), extra)
    for got, want, n in ((cg.grad, cr.grad, 'code'), (cpg.grad, cpr.grad, 'code_pos')):
        print('  grad', n, 'rel L2 %.3e' % float((got.cpu() - want).norm() / want.norm()))
