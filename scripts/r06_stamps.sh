#!/bin/bash
# round 6: phase stamps of one full block of k_corr2, plain form and exact-mask form (no cd chain)
L=/root/repo/depthg_amd/lib/libdepthg_stamps.so
for m in plain exact; do
  flag=""; [ $m = exact ] && flag="--exact-masks"
  DEPTHG_LIB=$L DG_STAMPS=/tmp/st_$m.bin python bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager --clock-warmup-s 0.2 $flag > /dev/null 2>/tmp/st_$m.err
  echo "== $m"; python scripts/stamps2.py /tmp/st_$m.bin
done
