#!/usr/bin/env python3
"""developer aid / race detector: the same inputs must give bit-identical scalars and gradients on every run
(no floating-point atomics anywhere on the path; a mismatch means a missing wait or barrier)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from depthg_amd import ContrastiveCorrelationLoss
from depthg_amd.loss import identity_coords
dev = torch.device("cuda:0")
f, fp, c, cp, d, dp = bench.synth_inputs(32, 99, dev)
g = torch.Generator().manual_seed(5)
perms = torch.stack([torch.randperm(32, generator=g) for _ in range(5)]).to(dev)
perms = torch.where(perms == torch.arange(32, device=dev), perms + 1, perms) % 32
bad = 0
for dense in (True, False):
    cfg = bench.make_cfg(bench.CONFIGS["headline"])
    loss = ContrastiveCorrelationLoss(cfg)
    if dense:
        c1 = c2 = identity_coords(32, 28, dev)
    else:
        gg = torch.Generator().manual_seed(11)
        c1 = (torch.rand(32, 28, 28, 2, generator=gg) * 2 - 1).to(dev)
        c2 = (torch.rand(32, 28, 28, 2, generator=gg) * 2 - 1).to(dev)
    ref = None
    for it in range(25):
        cg, cpg = c.clone().requires_grad_(True), cp.clone().requires_grad_(True)
        loss.forward_with(f, fp, cg, cpg, d, c1, c2, perms, shared_coords=dense, identity_grid=dense)
        loss.total.backward()
        cur = (loss.scalars.detach().clone(), cg.grad.clone(), cpg.grad.clone())
        if ref is None:
            ref = cur
        else:
            same = all(torch.equal(a, b) for a, b in zip(ref, cur))
            if not same:
                bad += 1
                print(f"dense={dense} run {it}: MISMATCH", [float((a - b).abs().max()) for a, b in zip(ref, cur)])
    print(f"dense={dense}: 25 runs, total {float(ref[0][8]):.9e}")
print("DETERMINISTIC" if bad == 0 else f"{bad} mismatching runs")
sys.exit(1 if bad else 0)
