#!/bin/bash
# SQ / memory counters of one kernel of one bench.py configuration (run on the GPU box): scripts/pmc_kernel.sh C3 plane_blend
cfg=${1:-C3}; pat=${2:-plane}
out=/root/repo/gpurun_out/pmck_$cfg; rm -rf $out; mkdir -p $out
sets=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_SALU"
      "FETCH_SIZE" "WRITE_SIZE" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum")
i=0
for s in "${sets[@]}"; do
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $s --output-format csv -d $out/p$i -- python3 /root/repo/bench.py --config $cfg --steps 3 --warmup 1 --clock-warmup-s 0 --no-cpu-baseline > /dev/null 2>&1 )
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$pat" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
