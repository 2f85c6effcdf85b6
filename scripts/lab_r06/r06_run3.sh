#!/bin/bash
out=gpurun_out/r06_run3; mkdir -p $out
python -m pytest tests/test_gpu_boundary.py -x -q -k "exact_masks_on_the_dense" 2>&1 | grep -v amdgpu | tail -3
python -m pytest tests -x -q -m gpu > $out/gputests.txt 2>&1; echo "pytest rc $?" >> $out/gputests.txt
for i in 1 2; do
python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_plain_$i.json 2> $out/bench_plain_$i.err
DG_SPLIT_MASKS=0 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --exact-masks > $out/bench_exact_seq_$i.json 2> $out/bench_exact_seq_$i.err
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --exact-masks > $out/bench_exact_split_$i.json 2> $out/bench_exact_split_$i.err
done
TAG=exact scripts/kstats.sh headline --exact-masks > $out/kstats_exact.txt 2>&1
TAG=exactseq DG_SPLIT_MASKS=0 scripts/kstats.sh headline --exact-masks > $out/kstats_exact_seq.txt 2>&1
tail -3 $out/gputests.txt
for f in $out/bench_*.json; do echo $f; python - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["ms_per_step"], d["loss_total"], r["kernel"], r["kernel_ms"], r["frac"], r.get("held_clock_ghz"), r["algorithmic_gflop_per_launch"])
PY
done
cat $out/kstats_exact.txt $out/kstats_exact_seq.txt
