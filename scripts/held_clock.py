#!/usr/bin/env python3
"""developer aid: the clock k_corr2's CUs HOLD inside the tile loop, from DG_BLOCKLOG=<file> of a block-log build
(scripts/build_variant.sh blog -DDG_DEVTOOLS -DC2_BLOCKLOG): every block logs the shader clock (s_memtime) and the 100-MHz
wall clock at the two ends of its tile loop; cycles / wall time is the clock held.  Full row blocks only (kind 0)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16).astype(np.int64)
a = a[(a[:, 2] > 0) & (a[:, 6] == 0) & (a[:, 14] > a[:, 13])]
wall_us = (a[:, 4] - a[:, 3]) * 0.01
cyc = a[:, 14] - a[:, 13]
ghz = cyc / wall_us / 1e3
print(f"{len(a)} full row blocks: tile loop {wall_us.mean():.1f} us (min {wall_us.min():.1f}, max {wall_us.max():.1f}), "
      f"{cyc.mean() / 1e3:.1f} k shader cycles -> clock held {ghz.mean():.3f} GHz (p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f})")
