cd /root/repo; mkdir -p gpurun_out/r04
export DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_blog.so
for g in 256 192 128 64 256; do
  DG_C2_GRID=$g timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid $g', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_loop'])"
done
