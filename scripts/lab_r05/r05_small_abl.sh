#!/bin/bash
# developer: k_corr_small's time with parts removed (results invalid), per-kernel averages under rocprofv3
for c in ${CFGS:-C2 C3}; do
  for abl in 0 16 32 64 128 240; do
    out=/tmp/abl_$c_$abl; rm -rf $out; mkdir -p $out
    ( cd /tmp && export TMPDIR=/tmp && DG_SMALL_DEBUG=$abl DEPTHG_LIB=$GRAFT_REPO_ROOT/depthg_amd/lib/libdepthg_dev.so rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --config $c --no-cpu-baseline --steps 50 --clock-warmup-s 0.25 > /dev/null 2>&1 )
    python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_corr_small" in r["Name"] or "k_small_finish" in r["Name"]:
        print("$c abl", $abl >> 4, r["Name"][:40], f'{float(r["AverageNs"])/1e3:8.1f}')
PY
  done
done
