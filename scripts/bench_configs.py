#!/usr/bin/env python3
"""developer aid: forward+backward time of the loss for the other SURVEY section 8(d) configurations (one GPU, synthetic inputs).
Not the headline metric (bench.py); numbers go to DESIGN.md."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthg_amd import ContrastiveCorrelationLoss
from oracle.depthg_oracle import default_cfg   # cfg container only (no oracle arithmetic is run here)

CONFIGS = {
    # name: (B, C, D, hw, S, sampling, pointwise, dense)
    "C2 ViT-S Potsdam  B=16 C=384 D=90 S=11 fps":      (16, 384, 90, 28, 11, "fps", True, False),
    "C3 ViT-B Cityscapes B=32 C=768 D=100 S=11 none":   (32, 768, 100, 28, 11, "none", False, False),
    "C4 ViT-B COCO shard B=8 C=768 D=90 S=12 fps":      (8, 768, 90, 28, 12, "fps", True, False),
    "headline general path B=32 C=384 D=70 S=28 rand":  (32, 384, 70, 28, 28, "none", True, False),
    "headline dense path   B=32 C=384 D=70 S=28 ident": (32, 384, 70, 28, 28, "none", True, True),
    "C5 hi-res B=8 C=384 D=70 56x56 S=56 rand":         (8, 384, 70, 56, 56, "none", True, False),
    "ViT-B dense path  B=32 C=768 D=100 S=28 ident":    (32, 768, 100, 28, 28, "none", True, True),
}
dev = torch.device("cuda:0")
flt = os.environ.get("DG_CFG_FILTER", "")
for name, (B, C, D, hw, S, samp, pw, dense) in CONFIGS.items():
    if flt and flt not in name:
        continue
    g = torch.Generator().manual_seed(1)
    f, fp = (torch.randn(B, C, hw, hw, generator=g).to(dev) for _ in range(2))
    c, cp = (torch.randn(B, D, hw, hw, generator=g).to(dev).requires_grad_(True) for _ in range(2))
    d, dp = (torch.randint(0, 256, (B, 1, 8 * hw, 8 * hw), generator=g).float().to(dev) for _ in range(2))
    cfg = default_cfg(feature_samples=S, depth_sampling=samp, pointwise=pw, dg_outputs="reduced", dg_dense_grid=dense)
    loss = ContrastiveCorrelationLoss(cfg)
    def step():
        c.grad = None; cp.grad = None
        loss(f, fp, None, None, c, cp, d, dp)
        loss.total.backward()
    try:
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n): step()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{name:52s} {ms:8.3f} ms/step  {1e3/ms:8.1f} steps/s")
    except Exception as e:   # report, keep going
        print(f"{name:52s} FAILED: {type(e).__name__}: {e}")
