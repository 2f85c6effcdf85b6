#!/usr/bin/env python3
"""developer aid: time the head's kernels alone (HIP events), headline shape.  DG_HEAD_STAMPS=<file> with a -DDG_DEVTOOLS build of
dg_head.hip writes the forward kernel's phase stamps of one block."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from depthg_amd.head import ProjectionHead
dev = torch.device("cuda:0")
B, C, D, hw = 32, 384, 70, 28
head = ProjectionHead(C, D).to(dev).train()
f = torch.randn(B, C, hw, hw, device=dev)
up = torch.randn(B, D, hw, hw, device=dev)
for _ in range(5):
    code, feats = head(f)
    (code * up).sum().backward()
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
n = 50
e[0].record()
for _ in range(n):
    code, feats = head(f)
e[1].record()
for _ in range(n):
    code, feats = head(f)
    code.backward(up)
e[2].record()
torch.cuda.synchronize()
fw = e[0].elapsed_time(e[1]) / n * 1e3
print(f"forward {fw:.1f} us   forward+backward {e[1].elapsed_time(e[2]) / n * 1e3:.1f} us")
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        code, feats = head(f)
        code.backward(up)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
