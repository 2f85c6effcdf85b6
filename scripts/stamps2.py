#!/usr/bin/env python3
"""developer aid: phase stamps of k_corr2 (build with -DDG_DEVTOOLS -DC2_STAMPS, run with DG_STAMPS=<file>).
stamps per tile: 0 start of phase A, 1 end of A, 2 end of B (before the wait), 3 after wait+barrier, 4 end of C, 5 end of D."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint32).astype(np.int64)
a = raw[:600].reshape(4, 25, 6)
blk = raw[600:616].reshape(4, 4)
print("wave: mean over tiles 3..21 of [A, B, wait+barrier, C, D, loop-back, iteration] cycles (each stamp costs ~40)")
for w in range(4):
    s = a[w, 3:22]
    nxt = a[w, 4:23, 0]
    f = lambda x: int(np.mean(x & 0xffffffff))
    print(f"w{w} | A {f(s[:,1]-s[:,0]):5d}  B {f(s[:,2]-s[:,1]):5d}  wait {f(s[:,3]-s[:,2]):5d}  C {f(s[:,4]-s[:,3]):5d}  D {f(s[:,5]-s[:,4]):5d}  back {f(nxt-s[:,5]):4d} | iter {f(nxt-s[:,0]):5d}")
print("first tile start -> last stamp:", int((a[:, 24, 5].max() - a[:, 0, 0].min()) & 0xffffffff))
print("block-level (wave: entry -> first tile landed -> loop done -> block done), cycles:")
for w in range(4):
    b = blk[w]
    print(f"w{w} | prologue {(b[1]-b[0]) & 0xffffffff:6d}  loop {(b[2]-b[1]) & 0xffffffff:6d}  epilogue {(b[3]-b[2]) & 0xffffffff:6d}  total {(b[3]-b[0]) & 0xffffffff:6d}")
