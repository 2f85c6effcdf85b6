#!/bin/bash
# quick check of a k_corr2 change: the dense-grid parity tests, then the headline line twice (+ exact masks once)
python -m pytest tests -x -q -m gpu -k "headline or exact_masks or config5 or dense or fold or walks or golden_forward" 2>&1 | grep -v amdgpu | tail -3
for i in 1 2; do python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('plain', d['ms_per_step'], d['loss_total'], r['kernel_ms'], r['frac'], r['held_clock_ghz'], r['kernel_mcycles'])"; done
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --exact-masks 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('exact', d['ms_per_step'], d['loss_total'], r['kernel_ms'], r['frac'], r['held_clock_ghz'], r['kernel_mcycles'])"
