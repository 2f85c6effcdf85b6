#!/usr/bin/env python3
"""developer aid: print the phase stamps the fused kernel wrote (DG_STAMPS=<file> python bench.py ...).
stamps per tile: 0 top, 1 after wait+barrier, 2 after DMA issue, 3 after the first phase (waves 0-3: chain, 4-7: post of t-1)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint32).reshape(8, 25, 4).astype(np.int64)
t0 = a[:, 0, 0].min()
print("wave: mean over tiles 4..20 of [wait+barrier, issue, phase A, phase B, iteration] cycles")
for w in range(8):
    s = a[w, 4:21]
    nxt = a[w, 5:22, 0]
    f = lambda x: int(np.mean(x & 0xffffffff))
    print(f"w{w} start {a[w,0,0]-t0:6d} | {f(s[:,1]-s[:,0]):5d} {f(s[:,2]-s[:,1]):5d} {f(s[:,3]-s[:,2]):5d} {f(nxt-s[:,3]):5d} {f(nxt-s[:,0]):5d}")
print("total first->last stamp:", int(a[:, 24, 3].max() - t0))
