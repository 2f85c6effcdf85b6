import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from depthg_amd import ContrastiveCorrelationLoss, ops
dev = torch.device("cuda:0")
cfg = bench.make_cfg()
loss_fn = ContrastiveCorrelationLoss(cfg)
f, fp, c, cp, d, dp = bench.synth_inputs(32, 1234, dev)
c.requires_grad_(True); cp.requires_grad_(True)
def step():
    c.grad = None; cp.grad = None
    loss_fn(f, fp, None, None, c, cp, d, dp)
    loss_fn.total.backward()
for _ in range(20): step()
torch.cuda.synchronize()
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
for _ in range(300): step()
pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22)
print(st.getvalue()[:4500])
