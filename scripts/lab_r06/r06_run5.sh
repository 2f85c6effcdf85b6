#!/bin/bash
# round 6: FPS beside the map copies (C2 / C4 shard), A/B against the plane sampler route
out=gpurun_out/r06_run5; mkdir -p $out
python -m pytest tests -x -q -m gpu > $out/gputests.txt 2>&1; echo "pytest rc $?" >> $out/gputests.txt
tail -3 $out/gputests.txt
for i in 1 2; do for c in C2 C4shard C3; do
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --config $c > $out/bench_${c}_overlap_$i.json 2> $out/bench_${c}_overlap_$i.err
DG_BENCH_NO_FPS_OVERLAP=1 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --config $c > $out/bench_${c}_plain_$i.json 2> $out/bench_${c}_plain_$i.err
done; done
for f in $out/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$f".split("/")[-1], d["ms_per_step"], d["loss_total"], r["bound"], r["frac"])
except Exception as e: print("$f", "ERR", e, open("$f".replace(".json",".err")).read()[-600:])
PY
done
TAG=ov scripts/kstats.sh C2 2>&1 | tail -14
