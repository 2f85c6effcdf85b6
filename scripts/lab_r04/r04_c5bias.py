#!/usr/bin/env python3
"""developer aid (round 4): signed error of the loss means against the oracle on dense grids of growing size, B = 2, for the library
named by DEPTHG_LIB - where does config 5's 8e-5 come from?   python scripts/r04_c5bias.py [S ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from depthg_amd import ContrastiveCorrelationLoss
from oracle import depthg_oracle as O
dev = torch.device("cuda:0")
torch.set_num_threads(16)
conf = bench.CONFIGS["C5"]
for S in [int(a) for a in sys.argv[1:]] or [28, 40, 56]:
    H = dict(conf["H"], B=2, h=S, w=S, S=S, depth_hw=8 * S)
    f, fp, c, cp, d, dp = bench.synth_inputs(2, 505, "cpu", H)
    g = torch.Generator().manual_seed(506)
    perms = [O.super_perm(2, g) for _ in range(5)]
    cfg = O.default_cfg(feature_samples=S, neg_samples=5, dim=70, pointwise=True, depth_sampling="none", dg_outputs="reduced", **conf["scal"])
    co = O.identity_coords(2, S)
    ref = O.forward(cfg, f, fp, c, cp, d, dp, coords1=co, coords2=co, perms=perms)
    T = lambda t: t.to(dev)
    loss = ContrastiveCorrelationLoss(cfg)
    out = loss.forward_with(T(f), T(fp), T(c).requires_grad_(True), T(cp), T(d), T(co), T(co), [T(p) for p in perms], shared_coords=True, identity_grid=True)
    sig = lambda i: (float(out[i].mean()) - float(ref[i].mean())) / abs(float(ref[i].mean()))
    print(f"S={S} P={S*S} kernel={loss.last_call and __import__('depthg_amd').ops.corr_main_kernel_name(loss.last_call[0])}",
          "loss (got-ref)/|ref|:", [f"{sig(i):+.2e}" for i in (0, 2, 4, 6)], "cd means rel:", [f"{sig(i):+.2e}" for i in (1, 3, 5)],
          "ref cd means", [f"{float(ref[i].mean()):+.3e}" for i in (1, 3, 5)])
    # forward-only call (no gradient pieces: k_corr_main's per-element sums instead of the FOLD dot products)
    with torch.no_grad():
        out2 = loss.forward_with(T(f), T(fp), T(c), T(cp), T(d), T(co), T(co), [T(p) for p in perms], shared_coords=True, identity_grid=True)
    sig2 = lambda i: (float(out2[i].mean()) - float(ref[i].mean())) / abs(float(ref[i].mean()))
    print("      forward-only:", [f"{sig2(i):+.2e}" for i in (0, 2, 4, 6)])
