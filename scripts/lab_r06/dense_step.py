#!/usr/bin/env python3
"""developer aid: N dense headline-shaped steps (forward + backward) at batch B, for a kernel trace per batch size:
   rocprofv3 --kernel-trace --stats ... -- python3 scripts/lab_r06/dense_step.py <B> [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from depthg_amd import ContrastiveCorrelationLoss  # noqa: E402
from oracle import depthg_oracle as O  # noqa: E402  (its default_cfg / identity_coords helpers only)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
C, D, hw, N = 384, 70, 28, 5
f, fp = torch.randn(B, C, hw, hw, generator=g).to(dev), torch.randn(B, C, hw, hw, generator=g).to(dev)
c, cp = torch.randn(B, D, hw, hw, generator=g).to(dev).requires_grad_(True), torch.randn(B, D, hw, hw, generator=g).to(dev).requires_grad_(True)
d = torch.randint(0, 256, (B, 1, 224, 224), generator=g).float().to(dev)
cfg = O.default_cfg(feature_samples=hw, neg_samples=N, dim=D, dg_outputs="reduced", dg_dense_grid=True)
co = O.identity_coords(B, hw).to(dev)
perms = [O.super_perm(B, g).to(dev) for _ in range(N)]
loss = ContrastiveCorrelationLoss(cfg)
for _ in range(steps):
    out = loss.forward_with(f, fp, c, cp, d, co, co, perms, shared_coords=True, identity_grid=True)
    loss.total.backward()
torch.cuda.synchronize()
