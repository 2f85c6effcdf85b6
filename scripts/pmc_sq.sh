#!/bin/bash
# developer aid: SQ counter passes for the fused kernel (per-launch averages). usage: scripts/pmc_sq.sh [kernel substring]
export TMPDIR=/tmp
kern=${1:-k_corr_main}
out=/root/repo/gpurun_out/pmcsq
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 /root/repo/bench.py --steps 2 --warmup 1 --clock-warmup-s 0 --no-cpu-baseline ${BENCH_ARGS} > /dev/null 2>&1 )
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob("$out/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$kern" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for c in sorted(acc): print(f"{c:32s} {acc[c]/len(n[c]):.4g}")
PY
