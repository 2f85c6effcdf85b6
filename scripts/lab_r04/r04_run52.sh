cd /root/repo
timeout 900 python -m pytest tests/test_gpu_head.py tests/test_segmenter.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3; do
for tag in nohs hip; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  timeout 300 python bench.py --config headline+head --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"
done; done
