#!/bin/bash
out=gpurun_out/r06_run4; mkdir -p $out
python scripts/lab_r06_nan.py 2>&1 | grep -v amdgpu | head -4
python -m pytest tests -x -q -m gpu > $out/gputests.txt 2>&1; echo "pytest rc $?" >> $out/gputests.txt
tail -3 $out/gputests.txt
for i in 1 2; do
python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_plain_$i.json 2> $out/bench_plain_$i.err
DG_SPLIT_MASKS=0 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --exact-masks > $out/bench_exact_seq_$i.json 2> $out/bench_exact_seq_$i.err
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --exact-masks > $out/bench_exact_split_$i.json 2> $out/bench_exact_split_$i.err
done
for v in "8 1" "8 2" "5 1" "7 1" "6 1"; do set -- $v
DEPTHG_LIB=depthg_amd/lib/libdepthg_m3dev.so DG_MASK3_W=$1 DG_MASK3_SPLIT=$2 DG_SPLIT_MASKS=0 TAG=m3_$1_$2 scripts/kstats.sh headline --exact-masks 2>&1 | grep "k_cd_mask3\|ms_per_step" | cut -c1-200 | sed "s/^/W=$1 split=$2: /"
done
for f in $out/bench_*.json; do echo $f; python - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["ms_per_step"], d["loss_total"], r["kernel"], r["kernel_ms"], r["frac"], r.get("held_clock_ghz"), r["algorithmic_gflop_per_launch"])
PY
done
