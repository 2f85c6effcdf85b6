#!/usr/bin/env python3
"""developer aid: per-CU timeline of the fused kernel from DG_BLOCKLOG=<file> (stamp build: make EXTRA=-DDG_STAMP_BUILD).
Every block logs hw id, xcc id and four 100 MHz wall-clock stamps: entry, first tile landed, tile loop done, exit."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16).astype(np.int64)
a = a[a[:, 2] > 0]
hw, xcc = a[:, 0], a[:, 1] & 0xf
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)      # cu, sh, se, xcc
t0 = a[:, 2].min()
T = (a[:, 2:6] - t0) * 10.0 / 1000.0      # us
kind, rb = a[:, 6], a[:, 7]
print(f"{len(a)} blocks on {len(set(cu.tolist()))} CUs; kernel span {T[:, 3].max():.1f} us")
pro, loop, epi = T[:, 1] - T[:, 0], T[:, 2] - T[:, 1], T[:, 3] - T[:, 2]
for k in sorted(set(kind.tolist())):
    for r in sorted(set(rb.tolist())):
        m = (kind == k) & (rb == r)
        if m.any():
            print(f"kind {k} rb {r}: n={m.sum():4d} prologue {pro[m].mean():6.2f} loop {loop[m].mean():7.2f} epilogue {epi[m].mean():5.2f} us")
gaps, busy, first, last = [], [], [], []
for c in sorted(set(cu.tolist())):
    m = np.where(cu == c)[0]
    o = m[np.argsort(T[m, 0])]
    first.append(T[o[0], 0]); last.append(T[o[-1], 3])
    busy.append((T[o, 3] - T[o, 0]).sum())
    gaps += list(T[o[1:], 0] - T[o[:-1], 3])
gaps = np.array(gaps)
print(f"blocks per CU: {len(a) / len(first):.2f}; first start {np.mean(first):.2f} us (max {np.max(first):.2f}); "
      f"last end mean {np.mean(last):.1f} min {np.min(last):.1f} max {np.max(last):.1f} us")
print(f"gap between consecutive blocks on a CU: mean {gaps.mean():.2f} median {np.median(gaps):.2f} p90 {np.percentile(gaps, 90):.2f} us; "
      f"busy per CU mean {np.mean(busy):.1f} us")
# prologue / epilogue by the block's position on its CU (1st, 2nd, 3rd ... block the CU ran): the first round starts on all CUs at once
rank = np.zeros(len(a), dtype=int)
for c in sorted(set(cu.tolist())):
    m = np.where(cu == c)[0]
    o = m[np.argsort(T[m, 0])]
    rank[o] = np.arange(len(o))
for k in range(rank.max() + 1):
    m = (rank == k) & (kind == 0)
    if m.any():
        print(f"block #{k} of its CU (full blocks): n={m.sum():4d} start {T[m, 0].mean():6.1f} us  prologue mean {pro[m].mean():5.2f} median {np.median(pro[m]):5.2f} "
              f"p90 {np.percentile(pro[m], 90):5.2f}  loop {loop[m].mean():6.2f}  epilogue {epi[m].mean():5.2f} us")
# finer stamps (kept in registers, stored at the end): 8 kernel entry, 9 prologue loads issued, 10 first tile landed, 11 FOLD + raw tiles done, 12 behind the block barrier
F = (a[:, 8:13] - t0) * 10.0 / 1000.0
m = (kind == 0)
print("full blocks, mean us: entry->decoded %.2f | ->loads issued %.2f | ->tile 0 landed %.2f | ->loop start %.2f || loop %.2f || ->FOLD+stores %.2f | ->barrier %.2f | ->end %.2f" % (
    (T[m, 0] - F[m, 0]).mean(), (F[m, 1] - T[m, 0]).mean(), (F[m, 2] - F[m, 1]).mean(), (T[m, 1] - F[m, 2]).mean(), (T[m, 2] - T[m, 1]).mean(),
    (F[m, 3] - T[m, 2]).mean(), (F[m, 4] - F[m, 3]).mean(), (T[m, 3] - F[m, 4]).mean()))
