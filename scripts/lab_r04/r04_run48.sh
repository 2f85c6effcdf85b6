cd /root/repo; mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3; do
for tag in presym hip; do
  if [ $tag = hip ]; then unset DEPTHG_LIB; else export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_$tag.so; fi
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
unset DEPTHG_LIB
for c in C2 C3 C4shard C5; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'])"; done
python scripts/parity_table.py gpurun_out/r04/parity_sym.md > /dev/null 2>&1; head -12 gpurun_out/r04/parity_sym.md
