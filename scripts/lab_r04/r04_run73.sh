cd /root/repo; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_head.py tests/test_segmenter.py -q -m gpu 2>&1 | tail -4
for i in 1 2 3; do
for ab in writefeats ""; do
  timeout 300 python bench.py --config headline+head --no-cpu-baseline ${ab:+--ablate $ab} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline+head ${ab:-deferred}', d['ms_per_step'])"
done; done
TAG=defer scripts/kstats.sh headline+head 2>&1 | grep -E "k_head_fwd|k_prep|ms_per" | cut -c1-250 | sed 's/"host_ms.*//'
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['ms_per_step'], d['roofline']['kernel_ms'])"; done
