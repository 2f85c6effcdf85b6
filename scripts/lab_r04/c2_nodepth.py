"""developer aid: the config-2 step with and without the depth term (how much of k_gs_xm is its depth blocks)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from depthg_amd import ContrastiveCorrelationLoss
dev = torch.device("cuda:0")
conf = bench.CONFIGS["C2"]; H = conf["H"]
depth_on = sys.argv[1] == "1"
cfg = bench.make_cfg(conf, depth_feat_correlation_loss=depth_on)
lf = ContrastiveCorrelationLoss(cfg)
f, fp, c, cp, d, dp = bench.synth_inputs(H["B"], 1234, dev, H)
c.requires_grad_(True); cp.requires_grad_(True)
for i in range(300):
    c.grad = None; cp.grad = None
    lf(f, fp, None, None, c, cp, d, dp)
    lf.total.backward()
torch.cuda.synchronize()
print("done", depth_on, float(lf.total))
