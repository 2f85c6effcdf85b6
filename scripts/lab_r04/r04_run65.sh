cd /root/repo; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -q -x -k "depth or headline or dense or config or exact or golden" 2>&1 | tail -3
for i in 1 2 3; do
for tag in pre hip; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
for tag in pre hip; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  TAG=$tag scripts/kstats.sh headline 2>&1 | grep -E "k_gs|k_corr2|ms_per"
  for c in C2 C3 C4shard C5; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag $c', d['ms_per_step'])"; done
  timeout 300 python bench.py --exact-masks --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag xm', d['ms_per_step'])"
done
