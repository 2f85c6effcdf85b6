#!/bin/bash
# developer aid: per-kernel averages of scripts/bench_configs.py restricted to configs matching $1
export TMPDIR=/tmp
out=/root/repo/gpurun_out/profcfg
rm -rf $out
( cd /tmp && DG_CFG_FILTER="$1" rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 /root/repo/scripts/bench_configs.py > $out.txt 2>/dev/null )
cat $out.txt
python3 - <<PY
import csv,glob
f=glob.glob("$out/*/*kernel_stats.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
for r in rows[:14]: print(f'{r["Name"].split("(")[0][-50:]:52s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Percentage"]}%')
PY
