#!/usr/bin/env python3
"""developer aid: host (Python + launch) time per step against the step time, for one of bench.py's configurations, and the
functions the host time goes to.  usage: cpu_overhead.py [config]   (headline, C2, C3, C4shard, C5)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from depthg_amd import ContrastiveCorrelationLoss
name = sys.argv[1] if len(sys.argv) > 1 else "headline"
conf = bench.CONFIGS[name]
dev = torch.device("cuda:0")
loss_fn = ContrastiveCorrelationLoss(bench.make_cfg(conf))
f, fp, c, cp, d, dp = bench.synth_inputs(conf["H"]["B"], 1234, dev, conf["H"])
c.requires_grad_(True); cp.requires_grad_(True)
seed_grad = torch.ones((), device=dev)
def step():
    c.grad = None; cp.grad = None
    loss_fn(f, fp, None, None, c, cp, d, dp)
    loss_fn.total.backward(gradient=seed_grad)
for _ in range(10): step()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{name}: host enqueue {1e3*(t1-t0)/n:.3f} ms/step, total {1e3*(t2-t0)/n:.3f} ms/step (GPU-bound if total > enqueue)")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
