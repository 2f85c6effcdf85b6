#!/bin/bash
# developer aid: per-kernel SQ counters (per-launch averages, whole chip) of one bench configuration, with the time the VALU
# instruction count alone would take (4 cycles per wave instruction, 1024 SIMDs, 1.9 GHz).  usage: scripts/pmc_all.sh C3
export TMPDIR=/tmp
cfg=${1:-headline}
out=/root/repo/gpurun_out/pmcall_$cfg
rm -rf $out; mkdir -p $out
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM --output-format csv -d $out -- python3 /root/repo/bench.py --config $cfg --steps 2 --warmup 1 --clock-warmup-s 0 --no-cpu-baseline > /dev/null 2>&1 )
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("$out/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
print(f"{'kernel':44s} {'VALU':>9s} {'SALU':>9s} {'LDS':>9s} {'VMEM':>9s} {'bankconf':>9s} {'VALU-bound us':>13s} wait_inst/wave")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    c = {x: y / len(n[k]) for x, y in v.items()}
    print(f"{k[:44]:44s} {c.get('SQ_INSTS_VALU',0):9.3g} {c.get('SQ_INSTS_SALU',0):9.3g} {c.get('SQ_INSTS_LDS',0):9.3g} {c.get('SQ_INSTS_VMEM',0):9.3g} "
          f"{c.get('SQ_LDS_BANK_CONFLICT',0):9.3g} {c.get('SQ_INSTS_VALU',0)*4/1024/1.9e3:13.1f} {c.get('SQ_WAIT_INST_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1):.2f}")
PY
find $out -name "*.csv" -size +1M -delete
