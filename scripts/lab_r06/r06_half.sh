#!/bin/bash
# round 6: fp16 gradient tiles - the GPU suite, then the headline / C5 / exact lines with and without (DG_HALF_TILES=0), and measured gradient errors
out=gpurun_out/r06_half; mkdir -p $out
python -m pytest tests -x -q -m gpu > $out/gputests.txt 2>&1; echo "pytest rc $?" >> $out/gputests.txt; tail -4 $out/gputests.txt
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', 'step', d['ms_per_step'], 'loss', d['loss_total'], 'kernel ms', r['kernel_ms'], 'frac', r['frac'], 'GHz', r['held_clock_ghz'], 'Mcyc', r['kernel_mcycles'])"; }
for i in 1 2; do
  python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | line half
  DG_HALF_TILES=0 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | line fp32
done
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --exact-masks 2>/dev/null | line exact_half
DG_HALF_TILES=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --exact-masks 2>/dev/null | line exact_fp32
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --config C5 2>/dev/null | line C5_half
DG_HALF_TILES=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --config C5 2>/dev/null | line C5_fp32
python scripts/parity_table.py $out/parity_half.md > /dev/null 2>&1; head -12 $out/parity_half.md
