#!/bin/bash
cfg=$1
out=/root/repo/gpurun_out/pmcrw_$cfg; rm -rf $out; mkdir -p $out
for s in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $s --output-format csv -d $out/$s -- python3 /root/repo/bench.py --config $cfg --steps 3 --warmup 1 --clock-warmup-s 0 --no-cpu-baseline > /dev/null 2>&1 )
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    fs = sum(v.get("FETCH_SIZE",[0]))/max(1,len(v.get("FETCH_SIZE",[0])))*2*1024/1e6
    ws = sum(v.get("WRITE_SIZE",[0]))/max(1,len(v.get("WRITE_SIZE",[0])))*1024/1e6
    if fs + ws > 5: print(f"{k:62s} read {fs:8.1f} MB  written {ws:8.1f} MB")
PY
