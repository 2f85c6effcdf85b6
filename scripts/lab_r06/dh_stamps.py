#!/usr/bin/env python3
"""developer aid: phase stamps of two blocks of k_head_dh at the paired headline shape (needs a -DDG_DEVTOOLS build selected with
DEPTHG_LIB and DG_DH_STAMPS=<file>): python scripts/lab_r06/dh_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from depthg_amd.head import ProjectionHead  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, C, D, h = 32, 384, 70, 28
head = ProjectionHead(C, D).to(dev)
f, fp = torch.randn(B, C, h, h, device=dev), torch.randn(B, C, h, h, device=dev)
for it in range(3):
    (code, _), (code_pos, _) = head.forward_pair(f, fp)
    (code.square().mean() + code_pos.square().mean()).backward()
    torch.cuda.synchronize()
print(open(os.environ["DG_DH_STAMPS"]).read())
