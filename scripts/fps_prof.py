#!/usr/bin/env python3
"""developer aid: run k_fps_coords a few times at one size (profile with rocprofv3 --kernel-trace --stats).  usage: fps_prof.py B hw dhw S"""
import sys, torch
sys.path.insert(0, "/root/repo")
from depthg_amd import ops
B, hw, dhw, S = (int(x) for x in sys.argv[1:5])
d = torch.rand(B, 1, dhw, dhw, device="cuda") * 9 + 0.5
for _ in range(10): ops.fps_coords(d, (hw, hw), S)
torch.cuda.synchronize()
