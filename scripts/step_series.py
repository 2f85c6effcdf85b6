#!/usr/bin/env python3
"""developer aid: per-step GPU time of the headline step from a cold start (HIP events around every step), to see how long
the chip takes to reach its sustained clock.  python scripts/step_series.py [steps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from depthg_amd import ContrastiveCorrelationLoss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
conf = bench.CONFIGS["headline"]
dev = torch.device("cuda:0")
loss_fn = ContrastiveCorrelationLoss(bench.make_cfg(conf))
f, fp, c, cp, d, dp = bench.synth_inputs(32, 1234, dev)
c.requires_grad_(True); cp.requires_grad_(True)
seed = torch.ones((), device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
torch.cuda.synchronize()
ev[0].record()
for i in range(n):
    c.grad = None; cp.grad = None
    loss_fn(f, fp, None, None, c, cp, d, dp)
    loss_fn.total.backward(gradient=seed)
    ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
cum = 0.0
for i in range(0, n, 10):
    blk = t[i:i + 10]
    cum += sum(blk)
    print(f"steps {i:4d}-{i + 9:4d}: mean {sum(blk) / len(blk):.4f} ms   (cumulative {cum:.1f} ms)")
