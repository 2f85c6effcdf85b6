cd /root/repo
for i in 1 2 3; do
for v in 0 1; do
  if [ $v = 1 ]; then export DG_SCAT8=1; else unset DG_SCAT8; fi
  for c in C2 C3 C4shard; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scat8=$v $c', d['ms_per_step'])"; done
done; done
export DG_SCAT8=1
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
