cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3; do
for tag in static hip; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_loop'])"
done; done
for tag in static hip; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  timeout 300 python bench.py --force-dist --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag force-dist', d['ms_per_step'])"
  timeout 300 python bench.py --config C5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag C5', d['ms_per_step'])"
done
