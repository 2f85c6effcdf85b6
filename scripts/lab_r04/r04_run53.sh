cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x -k "headline or dense or graph or config5 or boundary or sweep" 2>&1 | tail -3
for i in 1 2 3; do
for tag in noearly hip; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag graph', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
for tag in noearly hip; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  timeout 300 python bench.py --eager --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag eager', d['ms_per_step'])"
  timeout 300 python bench.py --config C5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag C5', d['ms_per_step'])"
done
