#!/bin/bash
# developer aid: per-kernel table of the headline+head step (kernel-trace averages), optionally with another library
# usage (GPU box): scripts/lab_r06/head_table.sh <tag> [lib.so]
tag=${1:-head}
export TMPDIR=/tmp
out=/root/repo/gpurun_out/$tag
mkdir -p $out
[ -n "$2" ] && export DEPTHG_LIB=$2
python3 /root/repo/bench.py --config headline+head --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'], 'held GHz', d['roofline'].get('held_clock_ghz'))" > $out/step.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 /root/repo/bench.py --config headline+head --steps 50 --warmup 5 --clock-warmup-s 0.25 --no-cpu-baseline > /dev/null 2>&1 )
python3 - <<PY
import csv, glob
f = glob.glob("$out/stats/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
with open("$out/kernels.txt", "w") as o:
    for r in rows[:24]:
        o.write("%-70s calls %6s  avg %8.1f us  %5s %%\n" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1000, r["Percentage"]))
PY
cat $out/step.txt $out/kernels.txt
find $out -name "*kernel_trace.csv" -delete
