#!/bin/bash
# configs 2-4 with the fused small-grid kernel and with the multi-launch path of round 4 (DG_SMALL_PATH=0), + per-kernel times
out=gpurun_out/$1; mkdir -p $out
for c in C2 C3 C4shard; do
  python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline > $out/bench_$c.json 2> $out/bench_$c.err
  DG_SMALL_PATH=0 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline > $out/bench_${c}_old.json 2> $out/bench_${c}_old.err
  python - <<PY
import json
for tag in ("", "_old"):
    try:
        l = json.loads(open("$out/bench_$c%s.json" % tag).read().strip().splitlines()[-1])
        print("$c%s" % tag, l["ms_per_step"], "ms  kernel", l["roofline"]["kernel"], l["roofline"]["kernel_ms"], "loss", l["loss_total"])
    except Exception as e:
        print("$c%s" % tag, "FAILED", e); print(open("$out/bench_$c%s.err" % tag).read()[-1500:])
PY
done
cd /tmp; export TMPDIR=/tmp
for c in C2 C3 C4shard; do
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
  f=$(find /tmp/prof_$c -name "*kernel_stats.csv" | head -1)
  echo "== $c"; [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$out/kernel_stats_$c.csv && head -14 $f | cut -c1-150
done
