cd /root/repo
for i in 1 2 3; do
for v in 16 8; do
  export DG_SCAT_SMALL_CG=$v
  for c in C2 C4shard; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cg=$v $c', d['ms_per_step'])"; done
done; done
