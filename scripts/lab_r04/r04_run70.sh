cd /root/repo; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_head.py tests/test_gpu_segmenter.py -q -x 2>&1 | tail -3
for i in 1 2; do
  TAG=fb16 scripts/kstats.sh headline+head 2>&1 | grep -E "k_head|ms_per" | cut -c1-250 | sed 's/"host_ms.*//'
done
for i in 1 2 3; do timeout 300 python bench.py --config headline+head --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline+head', d['ms_per_step'])"; done
