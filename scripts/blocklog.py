#!/usr/bin/env python3
"""developer aid: per-CU timeline of the fused kernel from DG_BLOCKLOG=<file> (stamp build: make EXTRA=-DDG_STAMP_BUILD).
Every block logs hw id, xcc id and four 100 MHz wall-clock stamps: entry, first tile landed, tile loop done, exit."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
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
