#!/usr/bin/env python3
"""developer aid: k_fps_coords time against the number of samples (prologue / per-round split).  usage: fps_time.py [B hw dhw]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from depthg_amd import ops
B, hw, dhw = (int(x) for x in (sys.argv[1:4] + ["16", "28", "224"][len(sys.argv) - 1:]))
d = torch.rand(B, 1, dhw, dhw, device="cuda") * 9 + 0.5
for S in (1, 2, 6, 11, 12, 16):
    for _ in range(3): ops.fps_coords(d, (hw, hw), S)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): ops.fps_coords(d, (hw, hw), S)
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} {hw}x{hw} from {dhw}: S={S:2d} ({S*S:3d} samples)  {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per call")
