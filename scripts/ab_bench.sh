#!/bin/bash
# developer aid: whole-step A/B between library builds: bench.py (headline, no CPU baseline) per tag, two interleaved rounds.
#   scripts/ab_bench.sh hip tagA tagB ...     (tag 'hip' = production library, others = depthg_amd/lib/libdepthg_<tag>.so)
cd "$(dirname "$0")/.."
for round in 1 2; do
  for tag in "$@"; do
    DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so python3 bench.py --steps ${STEPS:-300} --warmup 20 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag round $round: ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"
  done
done
