#!/usr/bin/env python3
"""Records what the HIP path computes for the two shards of tests/test_dp_gloo.py's problem: per rank, the gradients of the
stand-in head (dW, dbias) with d total / d code coming from the HIP loss on an MI355X.  Run on a GPU box:
    python tests/golden/make_dp_hip_fixture.py gpurun_out/dp_hip_shards.npz      (then copy the file to tests/golden/)
The CPU test test_two_rank_allreduce_of_recorded_hip_gradients all-reduces these over gloo."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_dp_gloo as T
from oracle import depthg_oracle as O            # cfg container and the shard-local permutation draws only
from depthg_amd import ContrastiveCorrelationLoss
from depthg_amd.parallel import shard_range

dev = torch.device("cuda:0")
world = 2
out = {}
for rank in range(world):
    feats, feats_pos, depth, w, bias, coords1, coords2 = T._make_problem()
    lo, hi = shard_range(T.B_GLOBAL, world, rank)
    w = w.to(dev).requires_grad_(True)
    bias = bias.to(dev).requires_grad_(True)
    f, fp, d = feats[lo:hi].to(dev), feats_pos[lo:hi].to(dev), depth[lo:hi].to(dev)
    code = torch.nn.functional.conv2d(f, w, bias)
    code_pos = torch.nn.functional.conv2d(fp, w, bias)
    g = torch.Generator().manual_seed(100 + rank)
    perms = [O.super_perm(hi - lo, g).to(dev) for _ in range(T.N)]
    cfg = O.default_cfg(feature_samples=T.S, neg_samples=T.N)
    loss = ContrastiveCorrelationLoss(cfg)
    res = loss.forward_with(f, fp, code, code_pos, d, coords1[lo:hi].to(dev), coords2[lo:hi].to(dev), perms)
    O.total_loss(cfg, res).backward()
    out[f"dw{rank}"] = w.grad.cpu().numpy()
    out[f"db{rank}"] = bias.grad.cpu().numpy()
np.savez(sys.argv[1], **out)
print({k: float(np.abs(v).max()) for k, v in out.items()})
