#!/bin/bash
for c in C2 C3; do
  echo "== $c"
  DG_SMALL_DEBUG=1 DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_dev.so python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --eager --clock-warmup-s 0.5 2>&1 | grep "k_corr_small block" | tail -3
done
