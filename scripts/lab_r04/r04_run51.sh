cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x -k "exact or graph" 2>&1 | tail -3
for i in 1 2; do
for tag in noside hip; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  timeout 300 python bench.py --exact-masks --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag graph', d['ms_per_step'], d['loss_total'])"
  timeout 300 python bench.py --exact-masks --eager --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag eager', d['ms_per_step'], d['loss_total'])"
done; done
