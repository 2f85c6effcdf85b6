#!/usr/bin/env python3
"""developer aid: A/B timing of the fused correlation launch between library builds, interleaved rounds in ONE process
(cdna guide 5.4 rule 24).   python scripts/ab_corr.py tagA tagB ...   (tag 'hip' = the production library, others =
depthg_amd/lib/libdepthg_<tag>.so from scripts/build_variant.sh).  Headline shape; --nodepth drops the depth job."""
import ctypes, os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from depthg_amd import _lib, ops
from depthg_amd.loss import identity_coords

args = [a for a in sys.argv[1:] if not a.startswith("--")]
nodepth = "--nodepth" in sys.argv
libs = {}
for tag in args:
    _lib._lib = None
    _lib.LIB_PATH = os.path.join(ROOT, "depthg_amd", "lib", f"libdepthg_{tag}.so")
    libs[tag] = _lib.load()
dev = torch.device("cuda:0")
B, C, D, hw, N = 32, 384, 70, 28, 5
g = torch.Generator().manual_seed(1234)
f, fp = torch.randn(B, C, hw, hw, generator=g).to(dev), torch.randn(B, C, hw, hw, generator=g).to(dev)
c, cp = torch.randn(B, D, hw, hw, generator=g).to(dev), torch.randn(B, D, hw, hw, generator=g).to(dev)
d = torch.randint(0, 256, (B, 1, 8 * hw, 8 * hw), generator=g).float().to(dev)
coords = identity_coords(B, hw, dev)
perms = torch.stack([torch.randperm(B, generator=g) for _ in range(N)]).to(dev)
desc = ops.make_desc(B, C, D, hw, hw, hw, N, pointwise=True, zero_clamp=True, stabalize=False, depth_term=not nodepth,
                     need_grad=True, shared_coords=True, shifts=(0.08, 0.02, 0.66, 0.03), depth_hw=(8 * hw, 8 * hw),
                     identity_grid=True, weights=(0.67, 0.25, 0.63, 0.19))
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
state = {}
for tag, lib in libs.items():
    nb = lib.dg_corr_workspace_bytes(ctypes.byref(desc))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    out = torch.empty(_lib.DG_OUT_COUNT, dtype=torch.float32, device=dev)
    rc = lib.dg_corr_forward(ctypes.byref(desc), P(f), P(fp), P(c), P(cp), P(None if nodepth else d), P(coords), P(coords), P(perms), P(out), P(ws), nb, stream)
    assert rc == 0, lib.dg_last_error()
    torch.cuda.synchronize()
    state[tag] = (ws, nb, out.cpu().tolist())
    print(tag, "scalars", [round(x, 6) for x in state[tag][2][:4]])
times = {t: [] for t in libs}
# (the GPU leaves its idle power state only after tens of ms of load: a second of launches first, then short interleaved rounds)
for _ in range(6000):
    lib0 = next(iter(libs.values()))
    lib0.dg_corr_relaunch_main(ctypes.byref(desc), P(perms), P(state[next(iter(libs))][0]), state[next(iter(libs))][1], stream)
torch.cuda.synchronize()
for rnd in range(42):
    for tag, lib in libs.items():
        ws, nb, _ = state[tag]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            lib.dg_corr_relaunch_main(ctypes.byref(desc), P(perms), P(ws), nb, stream)
        e1.record()
        torch.cuda.synchronize()
        if rnd >= 2:
            times[tag].append(e0.elapsed_time(e1) / 40 * 1e3)
for tag, v in times.items():
    print(f"{tag:12s} median {statistics.median(v):7.1f} us   min {min(v):7.1f} us")
