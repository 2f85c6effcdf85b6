cd /root/repo; mkdir -p gpurun_out/r04
for m in 1 0; do
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04/c2depth$m -- python3 /root/repo/scripts/lab_r04/c2_nodepth.py $m > /dev/null 2>&1 )
python3 - <<PY
import csv, glob
f = glob.glob("/root/repo/gpurun_out/r04/c2depth$m/**/*kernel_stats.csv", recursive=True)[0]
print("depth term", $m)
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 2.0: print(f'   {r["Name"][:70]:70s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:8.1f}')
PY
find /root/repo/gpurun_out/r04/c2depth$m -name "*kernel_trace.csv" -delete
done
