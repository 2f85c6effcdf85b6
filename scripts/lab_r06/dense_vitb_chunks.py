#!/usr/bin/env python3
"""developer aid: the dense 28 x 28 step at ViT-B width (C = 768) as ONE call (k_corr_main holds the 768-channel vectors) against TWO
channel chunks of 384 (each through k_corr2, the headline's kernel): ms per step (forward + backward), eager, events around 30 steps."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from depthg_amd import ContrastiveCorrelationLoss, ops  # noqa: E402
from oracle import depthg_oracle as O  # noqa: E402  (default_cfg / identity_coords helpers only)

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B, C, D, hw, N = 32, 768, 70, 28, 5
f, fp = torch.randn(B, C, hw, hw, generator=g).to(dev), torch.randn(B, C, hw, hw, generator=g).to(dev)
c, cp = torch.randn(B, D, hw, hw, generator=g).to(dev).requires_grad_(True), torch.randn(B, D, hw, hw, generator=g).to(dev).requires_grad_(True)
d = torch.randint(0, 256, (B, 1, 224, 224), generator=g).float().to(dev)
cfg = O.default_cfg(feature_samples=hw, neg_samples=N, dim=D, dg_outputs="reduced", dg_dense_grid=True)
co = O.identity_coords(B, hw).to(dev)
perms = [O.super_perm(B, g).to(dev) for _ in range(N)]
loss = ContrastiveCorrelationLoss(cfg)
res = {}
for name, maxc in (("one call", 768), ("two chunks", 384), ("one call", 768), ("two chunks", 384)):
    ops.BLOB_MAX_C = maxc
    for _ in range(5):
        loss.forward_with(f, fp, c, cp, d, co, co, perms, shared_coords=True, identity_grid=True); loss.total.backward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        out = loss.forward_with(f, fp, c, cp, d, co, co, perms, shared_coords=True, identity_grid=True); loss.total.backward()
    e1.record(); torch.cuda.synchronize()
    print(name, "%.4f ms per step" % (e0.elapsed_time(e1) / 30), "total", float(loss.total), "grad", float(c.grad.norm()))
    c.grad = None; cp.grad = None
