#!/usr/bin/env python3
"""developer aid: the nearest-neighbour table at cocostuff size (49,629 x 384; src/precompute_knns.py:97-115) - similarity slices on
the fp32 MFMA (dg_knn_similarities) against the library GEMM, with the same dg_topk_rows selection."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from depthg_amd import knn, ops
dev = torch.device("cuda:0")
n, F = 49629, 384
g = torch.Generator().manual_seed(1)
x = torch.nn.functional.normalize(torch.randn(n, F, generator=g), dim=1).to(dev)
step = n // 64
def table(sim):
    out = []
    for i in range(0, n, step):
        out.append(ops.topk_rows(sim(x[i:i + step]), 30))
    return torch.cat(out)
for name, sim in (("dg_knn_similarities", lambda q: ops.knn_similarities(q, x)), ("torch.matmul", lambda q: torch.matmul(q, x.t()))):
    t = table(sim); torch.cuda.synchronize()
    t0 = time.perf_counter(); t = table(sim); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): s = sim(x[:step])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:22s} whole table {dt*1e3:7.1f} ms; one {step} x {n} x {F} slice {ms:6.2f} ms = {2.0*step*n*F/ms/1e9:6.1f} TFLOP/s")
    res = t if name.startswith("dg") else res
    if not name.startswith("dg"):
        print("tables agree on", float((res == t).float().mean()), "of the entries")
