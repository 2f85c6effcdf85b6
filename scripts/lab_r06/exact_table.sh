#!/bin/bash
# developer aid: per-kernel table of the headline step with exact clamp masks (kernel-trace averages)
export TMPDIR=/tmp
out=/root/repo/gpurun_out/exact_table
mkdir -p $out
python3 /root/repo/bench.py --exact-masks --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'])"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 /root/repo/bench.py --exact-masks --steps 50 --warmup 5 --clock-warmup-s 0.25 --no-cpu-baseline > /dev/null 2>&1 )
python3 - <<PY
import csv, glob
f = glob.glob("$out/stats/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-70s calls %6s  avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1000))
PY
find $out -name "*kernel_trace.csv" -delete
