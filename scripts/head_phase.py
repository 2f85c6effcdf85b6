#!/usr/bin/env python3
"""developer aid: forward-kernel phase stamps of the head with / without the feats output (DG_HEAD_STAMPS + -DDG_DEVTOOLS build)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from depthg_amd.head import ProjectionHead
dev = torch.device("cuda:0")
B, C, D, hw = 32, 384, 70, 28
head = ProjectionHead(C, D).to(dev).train()
f = torch.randn(B, C, hw, hw, device=dev)
for fd in (True, False):
    for _ in range(3):
        with torch.no_grad():
            code, feats = head(f, feats_dropout=fd)
    torch.cuda.synchronize()
    print("feats_dropout", fd)
    print(open(os.environ["DG_HEAD_STAMPS"]).read())
