#!/bin/bash
# A/B of k_corr2's FOLD (DG_FOLD_INTRA=0 keeps the intra pair-set's G tiles and its k_gs job): headline and config 5, alternating
out=/root/repo/gpurun_out/r05_fold_ab.txt; : > $out
for i in 1 2 3; do
  for f in 1 0; do
    echo "headline DG_FOLD_INTRA=$f: $(DG_FOLD_INTRA=$f python3 /root/repo/bench.py --no-cpu-baseline | python3 -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], "ms/step, k_corr2", d["roofline"]["kernel_ms"], "ms")')" >> $out
  done
done
for f in 1 0; do
  echo "C5 DG_FOLD_INTRA=$f: $(DG_FOLD_INTRA=$f python3 /root/repo/bench.py --no-cpu-baseline --config C5 | python3 -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], "ms/step, k_corr2", d["roofline"]["kernel_ms"], "ms")')" >> $out
done
for f in 1 0; do echo "kernel trace, DG_FOLD_INTRA=$f" >> $out; DG_FOLD_INTRA=$f TAG=fold$f /root/repo/scripts/kstats.sh headline | head -6 >> $out; done
cat $out
