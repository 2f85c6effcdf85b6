"""developer aid: ProjectionHead.forward_pair against two forward() calls on random shapes (values bit-identical, gradients 1e-5)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from depthg_amd.head import ProjectionHead, draw_keep_masks_pair
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for seed in range(n):
    g = torch.Generator().manual_seed(1000 + seed)
    pick = lambda lo, hi: int(torch.randint(lo, hi + 1, (), generator=g))
    B = pick(1, 9); C = 8 * pick(1, 96); D = pick(1, 128); h, w = pick(3, 30), pick(3, 30)
    proj = ["nonlinear", "nonlinear", "linear"][pick(0, 2)]
    fd = bool(pick(0, 1))
    f, fp = (torch.randn(B, C, h, w, generator=g) * 2).to(dev), (torch.randn(B, C, h, w, generator=g) * 2).to(dev)
    up, upp = torch.randn(B, D, h, w, generator=g).to(dev), torch.randn(B, D, h, w, generator=g).to(dev)
    torch.manual_seed(seed)
    head = ProjectionHead(C, D, proj).to(dev).train()
    keeps = draw_keep_masks_pair(B, C, dev, 0.1, use=(True, proj == "nonlinear", fd))
    ka = tuple(k[:B] if k is not None else None for k in keeps); kb = tuple(k[B:] if k is not None else None for k in keeps)
    c1, f1 = head(f, fd, ka); c2, f2 = head(fp, fd, kb)
    ((c1 * up).sum() + (c2 * upp).sum()).backward()
    want = [p.grad.clone() for p in head.parameters()]
    for p in head.parameters(): p.grad = None
    (pc1, pf1), (pc2, pf2) = head.forward_pair(f, fp, fd, keeps)
    ((pc1 * up).sum() + (pc2 * upp).sum()).backward()
    ok = torch.equal(pc1, c1) and torch.equal(pc2, c2) and torch.equal(pf1, f1) and torch.equal(pf2, f2)
    rel = max(float((p.grad - w_).norm() / (w_.norm() + 1e-30)) for p, w_ in zip(head.parameters(), want))
    if not ok or not rel < 2e-5:
        bad += 1
        print(f"FAIL seed {seed}: B={B} C={C} D={D} {h}x{w} {proj} feats_dropout={fd} values_equal={ok} grad rel {rel:.2e}")
print(f"{n} cases, {bad} failures")
