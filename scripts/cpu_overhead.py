#!/usr/bin/env python3
"""developer aid: host (Python + launch) time per step vs GPU time per step of the headline loop."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from depthg_amd import ContrastiveCorrelationLoss
dev = torch.device("cuda:0")
cfg = bench.make_cfg()
loss_fn = ContrastiveCorrelationLoss(cfg)
f, fp, c, cp, d, dp = bench.synth_inputs(32, 1234, dev)
c.requires_grad_(True); cp.requires_grad_(True)
def step():
    c.grad = None; cp.grad = None
    loss_fn(f, fp, None, None, c, cp, d, dp)
    loss_fn.total.backward()
for _ in range(10): step()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/n:.3f} ms/step, total {1e3*(t2-t0)/n:.3f} ms/step (GPU-bound if total > enqueue)")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
