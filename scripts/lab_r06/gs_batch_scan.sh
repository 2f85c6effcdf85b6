#!/bin/bash
# developer aid: k_gs (and the other kernels of the dense step) per image at several batch sizes - does the launch's time follow its
# bytes, or the number of rounds its blocks take on the chip's slots?
export TMPDIR=/tmp
for B in 16 21 24 32 42 48 64; do
  out=/root/repo/gpurun_out/gsscan_$B
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 /root/repo/scripts/lab_r06/dense_step.py $B 30 > /dev/null 2>&1 )
  python3 - <<PY
import csv, glob
f = glob.glob("$out/*/*kernel_stats.csv")[0]
row = {r["Name"].split("<")[0].split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1000 for r in csv.DictReader(open(f))}
print("B=%-3d" % $B, " ".join("%s %.1f (%.2f/img)" % (k, row[k], row[k] / $B) for k in ("k_corr2", "k_gs", "k_combine_out", "k_prep_dense") if k in row))
PY
  find $out -name "*kernel_trace.csv" -delete
done
