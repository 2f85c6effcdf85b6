#!/bin/bash
# developer aid: per-kernel average durations of one bench run under rocprofv3 (tag = $1, further arguments go to bench.py)
export TMPDIR=/tmp
tag=$1; shift
out=/root/repo/gpurun_out/prof_$tag
rm -rf $out $out.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $out.json 2>/dev/null )
python3 - <<PY
import csv,glob
f=glob.glob("$out/*/*kernel_stats.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if r["Name"].startswith(("k_","void k_"))]
import json
try: b=json.loads(open("$out.json").read().strip().splitlines()[-1]); print("$tag ms_per_step", b["ms_per_step"], "gpu kernel sum per step (us)", round(sum(float(r["AverageNs"])*int(r["Calls"]) for r in csv.DictReader(open(f)))/1e3/23,1))
except Exception as e: print("bench line unreadable", e)
print("$tag", " | ".join(f'{r["Name"].split("(")[0][-28:]} {float(r["AverageNs"])/1e3:.0f}us x{int(r["Calls"])}' for r in rows))
PY
