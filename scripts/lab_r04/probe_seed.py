"""developer aid: one fuzz case (scripts/fuzz_parity.py's generator) with the un-reduced tensors compared element by element."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from depthg_amd import ContrastiveCorrelationLoss
from oracle import depthg_oracle as O
seed = int(sys.argv[1])
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(seed)
pick = lambda lo, hi: int(torch.randint(lo, hi + 1, (), generator=g))
dense = pick(0, 2) == 0
B = pick(1, 9); C = [32, 64, 100, 128, 200, 384, 384, 768][pick(0, 7)]
D = pick(4, 128) if pick(0, 3) else [70, 90, 96, 128][pick(0, 3)]
N = pick(1, 6)
if dense:
    h = w = pick(6, 30); S = h
else:
    h, w = pick(5, 30), pick(5, 30); S = pick(2, min(h, w, 14))
flags = dict(pointwise=bool(pick(0, 3)), zero_clamp=bool(pick(0, 3)), stabalize=pick(0, 4) == 0, depth_feat_correlation_loss=bool(pick(0, 3)))
nograd = pick(0, 3) == 0; full = not dense and pick(0, 2) == 0; line = not dense and pick(0, 4) == 0
print(f"seed {seed}: dense={dense} B={B} C={C} D={D} {h}x{w} S={S} N={N} line={line} {flags}")
f, fp = torch.randn(B, C, h, w, generator=g), torch.randn(B, C, h, w, generator=g)
c, cp = torch.randn(B, D, h, w, generator=g), torch.randn(B, D, h, w, generator=g)
d = torch.randint(0, 256, (B, 1, 3 * h + pick(0, 5), 3 * w + pick(0, 5)), generator=g).float()
d[:, :, :pick(0, 6), :pick(0, 6)] = 0.0
if pick(0, 2) == 0 or B == 1:
    perms = [torch.randint(0, B, (B,), generator=g) for _ in range(N)]
else:
    perms = [O.super_perm(B, g) for _ in range(N)]
assert not dense
shared = pick(0, 2) == 0 and not full
S2 = 1 if line else S
c1 = (torch.rand(1, S, S2, 2, generator=g).expand(B, S, S2, 2).contiguous() if shared else torch.rand(B, S, S2, 2, generator=g)) * 2.2 - 1.1
c2 = c1 if shared else torch.rand(B, S, S2, 2, generator=g) * 2.2 - 1.1
for mode in ("full",) if not shared else ("reduced",):
    cfg = O.default_cfg(feature_samples=S, neg_samples=N, dim=D, dg_outputs=mode, dg_dense_grid=False, **flags)
    ref = O.forward(cfg, f, fp, c, cp, d, d, coords1=c1, coords2=c2, perms=perms)
    out = ContrastiveCorrelationLoss(cfg).forward_with(f.to(dev), fp.to(dev), c.to(dev), cp.to(dev), d.to(dev), c1.to(dev), c2.to(dev),
                                                       [p.to(dev) for p in perms], shared_coords=shared)
    for i in range(len(ref)):
        a, b = out[i].detach().cpu().float(), ref[i].detach().float()
        print(i, "mean", float(a.mean()), float(b.mean()), "shape", tuple(a.shape), tuple(b.shape))
        if a.shape == b.shape and a.dim() > 0:
            e = (a - b).abs()
            k = int(e.argmax())
            print("    max element error", float(e.max()), "at", k, "got", float(a.flatten()[k]), "want", float(b.flatten()[k]))
