#!/bin/bash
# round 6: what bounds k_corr2's tile loop - ablations with WRONG results, cycles per launch (bench.py roofline.kernel_mcycles)
#   abl_gst: G stores into one 2 KiB   abl_dma: every tile = tile 0   abl_nodma: no tile fetched in the loop   abl_nogst: no G store issued
for v in ${VARIANTS:-hip abl_gst abl_dma abl_both abl_nodma abl_nogst abl_nodmagst}; do
  for i in 1 2; do DEPTHG_LIB=/root/repo/depthg_amd/lib/libdepthg_$v.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline $BENCHARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', 'step', d['ms_per_step'], 'kernel ms', r['kernel_ms'], 'GHz', r['held_clock_ghz'], 'Mcycles', r['kernel_mcycles'])"; done
done
