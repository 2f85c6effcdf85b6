#!/usr/bin/env python3
"""developer aid: k_fps_coords time against the number of selected points (what is setup - pooling, point table, output - and what
is the round chain)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from depthg_amd import ops
dev = torch.device("cuda:0")
B = 16
d, dp = torch.rand(B, 1, 224, 224, device=dev) * 255, torch.rand(B, 1, 224, 224, device=dev) * 255
for S in (1, 2, 4, 8, 11, 16):
    for _ in range(5):
        ops.fps_coords_pair(d, dp, (28, 28), S)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.fps_coords_pair(d, dp, (28, 28), S)
    e1.record()
    torch.cuda.synchronize()
    print(f"S={S:2d} ({S * S:3d} picks)  {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per call")
