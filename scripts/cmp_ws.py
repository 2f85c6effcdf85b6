#!/usr/bin/env python3
"""developer aid: run dg_corr_forward with two library builds on the same inputs and report where the workspaces differ.
   python scripts/cmp_ws.py tagA tagB [B] [dense]      (dense: the identity-grid path, shared coordinates)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from depthg_amd import _lib, ops

tags = sys.argv[1:3]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 3
DENSE = len(sys.argv) > 4 and sys.argv[4] == "dense"
libs = {}
for tag in tags:
    _lib._lib = None
    _lib.LIB_PATH = os.path.join(ROOT, "depthg_amd", "lib", f"libdepthg_{tag}.so")
    libs[tag] = _lib.load()
dev = torch.device("cuda:0")
hw = int(os.environ.get("CMP_HW", "28"))
C, D, N, S = 384, 70, 2, hw
g = torch.Generator().manual_seed(128)
f, fp = torch.randn(B, C, hw, hw, generator=g).to(dev), torch.randn(B, C, hw, hw, generator=g).to(dev)
c, cp = torch.randn(B, D, hw, hw, generator=g).to(dev), torch.randn(B, D, hw, hw, generator=g).to(dev)
d = torch.randint(0, 256, (B, 1, 4 * hw, 4 * hw), generator=g).float()
d[:, :, :9, :7] = 0.0                       # a region of zero depth: indicators that are not all 1
d = d.to(dev)
c1 = (torch.rand(B, S, S, 2, generator=g) * 2 - 1).to(dev)
c2 = (torch.rand(B, S, S, 2, generator=g) * 2 - 1).to(dev)
perms = torch.stack([torch.randperm(B, generator=g) for _ in range(N)]).to(dev)
desc = ops.make_desc(B, C, D, hw, hw, S, N, pointwise=True, zero_clamp=True, stabalize=False, depth_term=True,
                     need_grad=True, shared_coords=DENSE, shifts=(0.08, 0.02, 0.66, 0.03), depth_hw=(4 * hw, 4 * hw),
                     identity_grid=DENSE, weights=(0.67, 0.25, 0.63, 0.19))
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
res = {}
for tag, lib in libs.items():
    nb = lib.dg_corr_workspace_bytes(ctypes.byref(desc))
    ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
    out = torch.zeros(_lib.DG_OUT_COUNT, dtype=torch.float32, device=dev)
    rc = lib.dg_corr_forward(ctypes.byref(desc), P(f), P(fp), P(c), P(cp), P(d), P(c1), P(c2), P(perms), P(out), P(ws), nb, stream)
    assert rc == 0, lib.dg_last_error()
    torch.cuda.synchronize()
    res[tag] = (ws, out.cpu())
    print(tag, out.cpu().tolist())
a, b = res[tags[0]][0], res[tags[1]][0]
diff = (a != b).nonzero().flatten()
print("workspace bytes", a.numel(), "differing bytes", diff.numel())
if diff.numel():
    d64 = diff.cpu().numpy()
    # contiguous-ish regions (gap > 1 MiB starts a new region)
    import numpy as np
    cuts = np.nonzero(np.diff(d64) > (1 << 20))[0]
    starts = np.concatenate([[0], cuts + 1]); ends = np.concatenate([cuts, [len(d64) - 1]])
    for s, e in zip(starts, ends):
        lo, hi = int(d64[s]), int(d64[e])
        n = e - s + 1
        va = a[lo:lo + 64].cpu().view(torch.float16).float().tolist()[:8]
        vb = b[lo:lo + 64].cpu().view(torch.float16).float().tolist()[:8]
        print(f"region [{lo}, {hi}] ({(hi-lo+1)/1e6:.2f} MB span, {n} differing bytes)  f16 A {va}  B {vb}")
# per S-tile comparison of the G tiles (the last T allocations of the workspace: [t][image][S tile][R tile][2048 B])
nt = (S * S + 31) // 32
T = 2 + N
gsz = B * nt * nt * 2048
total = a.numel()
for tj in range(T):
    ga_ = a[total - (T - tj) * gsz: total - (T - tj - 1) * gsz].view(torch.float16).float().view(B, nt, nt, 1024)
    gb_ = b[total - (T - tj) * gsz: total - (T - tj - 1) * gsz].view(torch.float16).float().view(B, nt, nt, 1024)
    dmax = (ga_ - gb_).abs().amax(dim=(0, 3))          # [S tile][R tile]
    print("job", tj, "max |dG| per S tile:", [round(float(x), 3) for x in dmax.amax(dim=1)])
    print("        per R tile:", [round(float(x), 3) for x in dmax.amax(dim=0)])
    if tj == 0:
        # inside the last S tile: which accumulator rows differ?  tile = [2 k-steps][64 lanes][8]
        t_a = ga_[0, 1, 0].view(2, 64, 8); t_b = gb_[0, 1, 0].view(2, 64, 8)
        rows = {}
        for sp in range(2):
            for h in range(2):
                for e in range(8):
                    row = 16 * sp + (e & 3) + 8 * (e >> 2) + 4 * h
                    rows[row] = float((t_a[sp, 32 * h:32 * h + 32, e] - t_b[sp, 32 * h:32 * h + 32, e]).abs().max())
        print("        S tile 1, R tile 0, max |dG| per tile row:", [round(rows[r], 3) for r in range(32)])
        print("        lane 0 A:", [round(float(x), 3) for x in t_a[:, 0, :].flatten()])
        print("        lane 0 B:", [round(float(x), 3) for x in t_b[:, 0, :].flatten()])
        print("        lane 5 A:", [round(float(x), 3) for x in t_a[:, 5, :].flatten()])
        print("        lane 5 B:", [round(float(x), 3) for x in t_b[:, 5, :].flatten()])
        # is B's tile 1 equal to A's tile of another index?
        for tt in range(nt):
            dd = float((ga_[0, tt, 0] - gb_[0, 1, 0]).abs().max())
            if dd < 0.05: print("        B tile 1 matches A tile", tt, dd)
# gradient tiles of the region that differs most (if it has the size of one [B][tiles][DP/32][1024] fp32 buffer): where?
if diff.numel():
    lo = int(d64[0]) // 256 * 256
    ngt = B * nt * 3 * 1024
    for base in (lo, lo - 256, lo - 512):
        if base >= 0 and base + ngt * 4 <= a.numel():
            ta = a[base:base + ngt * 4].view(torch.float32).view(B, nt, 3, 1024)
            tb = b[base:base + ngt * 4].view(torch.float32).view(B, nt, 3, 1024)
            dd = (ta - tb).abs()
            print("gradient-tile view at", base, ": max |d| per R tile:", [round(float(x), 4) for x in dd.amax(dim=(0, 2, 3))])
            print("   per channel group:", [round(float(x), 4) for x in dd.amax(dim=(0, 1, 3))], " scale of values:", float(tb.abs().max()))
            break
# (round 4) which accumulator registers / lanes of the raw gradient tiles differ: tile = [group 3][i>>2 (4)][lane 64][i&3 (4)]
if diff.numel():
    lo = int(os.environ.get("CMP_BASE", int(d64[0]) // 256 * 256))        # CMP_BASE: byte offset of the buffer to view (dRA of a pair-set)
    ngt = B * nt * 3 * 1024
    for base in (lo,):
        if base >= 0 and base + ngt * 4 <= a.numel():
            ta = a[base:base + ngt * 4].view(torch.float32).view(B, nt, 3, 4, 64, 4)
            tb0 = b[base:base + ngt * 4].view(torch.float32).view(B, nt, 3, 4, 64, 4)
            dd0 = (ta - tb0).abs()
            for grp in range(3):
                print(f"group {grp}: max |d| even tiles {float(dd0[:, 0::2, grp].max()):.4f} odd tiles {float(dd0[:, 1::2, grp].max()):.4f}; per register:",
                      [round(float(x), 3) for x in dd0[:, :, grp].amax(dim=(0, 1, 3)).reshape(-1)], "lanes with a difference:",
                      [int(i) for i in (dd0[:, :, grp].amax(dim=(0, 1, 2, 4)) > 0).nonzero().flatten()])
            # which tiles
            print("tiles with a difference (image 0):", [int(i) for i in (dd0[0].amax(dim=(1, 2, 3, 4)) > 0).nonzero().flatten()])
            nzd = (dd0 > 0).nonzero()
            print("differing elements:", nzd.shape[0], "first ten (image, tile, group, i>>2, lane, i&3):", nzd[:10].tolist())
            for idx in nzd[:6].tolist():
                print("   ", idx, "A", float(ta[tuple(idx)]), "B", float(tb0[tuple(idx)]))
            tb = b[base:base + ngt * 4].view(torch.float32).view(B, nt, 3, 4, 64, 4)
            dd = (ta - tb).abs()
            per_reg = dd[:, :, 0].amax(dim=(0, 1, 3)).permute(0, 1).reshape(16)     # [i>>2][i&3]
            print("group 0: max |d| per accumulator register i:", [round(float(x), 4) for x in dd[:, :, 0].amax(dim=(0, 1, 3)).reshape(-1)])
            print("group 0: max |d| per lane:", [round(float(x), 3) for x in dd[:, :, 0].amax(dim=(0, 1, 2, 4))])
            nz = (dd[:, :, 0] > 0).float().mean().item()
            print("group 0: fraction of differing elements (all tiles):", nz, " even tiles only:", (dd[:, 0::2, 0] > 0).float().mean().item())
            rel = (dd[:, 0::2, 0] / (tb[:, 0::2, 0].abs() + 1e-6))
            print("group 0, even tiles: median relative diff", float(rel.median()), "mean", float(rel.mean()))
            t0a, t0b = ta[0, 0, 0], tb[0, 0, 0]
            print("tile 0 lane 3: A", [round(float(x), 4) for x in t0a[:, 3, :].flatten()])
            print("tile 0 lane 3: B", [round(float(x), 4) for x in t0b[:, 3, :].flatten()])
            break
