cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x -k "keep_masks or head or bench_line" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --config headline+head --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline+head', d['ms_per_step'])"; done
