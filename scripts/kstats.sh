#!/bin/bash
# Per-kernel averages of one bench.py configuration under rocprofv3 (run on the GPU box): scripts/kstats.sh C3 [more bench args]
cfg=${1:-headline}; shift
out=/root/repo/gpurun_out/kstats_$cfg${TAG:+_$TAG}; rm -rf $out
mkdir -p $out
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 /root/repo/bench.py --config $cfg --no-cpu-baseline --steps 50 --clock-warmup-s 0.25 "$@" > $out/bench.json 2>/dev/null )
python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.3:
        print(f'{r["Name"][:90]:90s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:8.1f}')
PY
find $out -name "*kernel_trace.csv" -delete
tail -1 $out/bench.json | cut -c1-200
