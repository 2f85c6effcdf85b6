cd /root/repo
export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_c2dev.so
for i in 1 2 3; do
for g in 256 248 240 232 224; do
  DG_C2_GRID=$g timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid $g', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
for w in static dynamic; do
  DG_C2_WALK=$w timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('walk $w', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
