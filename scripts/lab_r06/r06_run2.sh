#!/bin/bash
# round 6, second GPU call: the exact-mask form of k_corr2 without its cd chain + FOLD, the mask chain on the side stream
out=gpurun_out/r06_run2; mkdir -p $out
python -m pytest tests -x -q -m gpu > $out/gputests.txt 2>&1; echo "pytest rc $?" >> $out/gputests.txt
for i in 1 2; do
python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_plain_$i.json 2> $out/bench_plain_$i.err
DG_SPLIT_MASKS=0 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --exact-masks > $out/bench_exact_seq_$i.json 2> $out/bench_exact_seq_$i.err
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --exact-masks > $out/bench_exact_split_$i.json 2> $out/bench_exact_split_$i.err
done
TAG=exact scripts/kstats.sh headline --exact-masks > $out/kstats_exact.txt 2>&1
TAG=exactseq DG_SPLIT_MASKS=0 scripts/kstats.sh headline --exact-masks > $out/kstats_exact_seq.txt 2>&1
python scripts/parity_table.py $out/parity.md > $out/parity.log 2>&1
tail -3 $out/gputests.txt
for f in $out/bench_*.json; do echo $f; python - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["ms_per_step"], r["kernel"], r["kernel_ms"], r["frac"], r.get("held_clock_ghz"), r["algorithmic_gflop_per_launch"])
PY
done
cat $out/kstats_exact.txt $out/kstats_exact_seq.txt
