cd /root/repo; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_head.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --config headline+head --no-cpu-baseline --ablate twocalls 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two calls', d['ms_per_step'])"
timeout 300 python bench.py --config headline+head --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pair     ', d['ms_per_step'])"
done
TAG=pair bash scripts/kstats.sh headline+head 2>&1 | tail -22
