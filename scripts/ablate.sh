#!/bin/bash
# developer aid: time the fused kernel with parts switched off (results invalid in all but 0)
# DG_DEBUG bits: 1 no tile DMA, 4 no gradient MFMAs, 8 no G-tile stores, 16 no epilogue VALU; k_gs: 32 no compute, 64 no G loads
for d in ${@:-0 1 4 8 16 29}; do
  echo -n "DG_DEBUG=$d  "
  DG_DEBUG=$d python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
done
