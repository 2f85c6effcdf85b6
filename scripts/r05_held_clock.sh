#!/bin/bash
# the clock k_corr2 holds in its tile loop at 64 / 128 / 256 workgroups (block-log build: scripts/build_variant.sh blog -DDG_DEVTOOLS -DC2_BLOCKLOG)
out=/root/repo/gpurun_out/r05_held_clock.txt; : > $out
export DEPTHG_LIB=/root/repo/depthg_amd/lib/libdepthg_blog.so
for g in 256 128 64 256; do
  DG_C2_GRID=$g DG_BLOCKLOG=/tmp/blog_$g.bin python3 /root/repo/bench.py --no-cpu-baseline --steps 20 --warmup 5 --eager > /tmp/b_$g.json 2>/dev/null
  echo "workgroups $g: step $(python3 -c "import json;print(json.loads(open('/tmp/b_$g.json').readlines()[-1])['ms_per_step'])") ms; $(python3 /root/repo/scripts/held_clock.py /tmp/blog_$g.bin)" >> $out
done
cat $out
