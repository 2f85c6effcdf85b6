#!/bin/bash
# developer aid: HBM-side bytes (FETCH_SIZE / WRITE_SIZE, separate passes) per launch of the head kernels in the headline+head step
tag=${1:-head_pmc}
export TMPDIR=/tmp
out=/root/repo/gpurun_out/$tag
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 /root/repo/bench.py --config headline+head --steps 3 --warmup 1 --clock-warmup-s 0 --no-cpu-baseline > /dev/null 2>&1 )
done
python3 - <<PY
import csv, glob, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$out/%s/*/*counter_collection.csv" % c)[0]
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in acc: res[k][c] = acc[k] / len(n[k])
with open("$out/pmc.txt", "w") as o:
    for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0))[:16]:
        # KiB units; FETCH_SIZE counts 64 B per 128-B request of a wide stream on gfx950 (x2 for streams of whole lines)
        o.write("%-60s fetch %8.1f MB (x2: %8.1f)  write %8.1f MB\n" % (k[:60], v.get("FETCH_SIZE", 0) / 1024, 2 * v.get("FETCH_SIZE", 0) / 1024, v.get("WRITE_SIZE", 0) / 1024))
PY
cat $out/pmc.txt
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete
