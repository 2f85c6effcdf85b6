cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x -k "headline or dense or exact or config5 or boundary" 2>&1 | tail -2
for i in 1 2 3; do
for tag in gfirst hip; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_loop'])"
done; done
