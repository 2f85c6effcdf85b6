#!/bin/bash
# developer aid: PMC counters of the fused kernel. usage: scripts/pmc.sh <tag> <counter...>
tag=$1; shift
export TMPDIR=/tmp
out=/root/repo/gpurun_out/pmc_$tag
( cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 )
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/*/*counter_collection.csv")[0]
acc=collections.defaultdict(float); n=0
seen=set()
for r in csv.DictReader(open(f)):
    if "corr_main" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); seen.add(r["Dispatch_Id"])
n=len(seen)
print("$tag launches",n,{k:round(v/n/1e6,2) for k,v in acc.items()}, "(millions per launch)")
PY
