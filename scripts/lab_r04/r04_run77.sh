cd /root/repo
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -2
for i in 1 2 3; do
for tag in pre hip; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  for c in C2 C3 C4shard; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag $c', d['ms_per_step'])"; done
done; done
unset DEPTHG_LIB
for c in C2 C3; do scripts/kstats.sh $c 2>&1 | grep -E "gather|cd_mask|ms_per" | cut -c1-200 | sed 's/"host_ms.*//'; done
