import sys, torch, time
sys.path.insert(0, "/root/repo")
from depthg_amd import ops
dev = torch.device("cuda:0")
d = torch.randint(0, 256, (8, 1, 224, 224)).float().to(dev)
for S in (2, 6, 12, 20, 28):
    for _ in range(3): ops.fps_coords(d, (28, 28), S)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.fps_coords(d, (28, 28), S)
    e1.record(); torch.cuda.synchronize()
    print(f"S={S:2d} rounds={S*S-1:3d}  {e0.elapsed_time(e1)/20*1e3:7.1f} us")
