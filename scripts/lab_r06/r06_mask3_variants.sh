#!/bin/bash
# round 6: k_cd_mask3 forms (waves per block W, parts of the S walk) at the headline, kernel-trace averages (developer build of dg_prep)
L=/root/repo/depthg_amd/lib/libdepthg_m3dev.so
for v in "8 1" "5 1" "5 2" "7 1" "6 1" "8 2"; do set -- $v
  out=/root/repo/gpurun_out/kstats_m3_$1_$2; rm -rf $out; mkdir -p $out
  ( cd /tmp && export TMPDIR=/tmp && DEPTHG_LIB=$L DG_MASK3_W=$1 DG_MASK3_SPLIT=$2 DG_SPLIT_MASKS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 /root/repo/bench.py --no-cpu-baseline --steps 50 --clock-warmup-s 0.25 --exact-masks > $out/bench.json 2>$out/err.txt )
  python3 - <<PY
import csv, glob, json
fs = glob.glob("$out/**/*kernel_stats.csv", recursive=True)
for f in fs:
    for r in csv.DictReader(open(f)):
        if "mask3" in r["Name"]:
            print("W=$1 parts=$2", r["Name"][:40], r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
try: print("   ms_per_step", json.loads(open("$out/bench.json").read().strip().splitlines()[-1])["ms_per_step"])
except Exception as e: print("   bench line:", e, open("$out/err.txt").read()[-300:])
PY
  find $out -name "*kernel_trace.csv" -delete
done
