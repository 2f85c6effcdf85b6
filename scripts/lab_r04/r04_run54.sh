cd /root/repo
for i in 1 2; do
for tag in hip nogst nodma both; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_loop'], d['loss_total'])"
done; done
