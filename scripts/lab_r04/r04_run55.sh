cd /root/repo
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-900
timeout 900 python -m pytest tests -m gpu -q -x -k "bench_line or configs" 2>&1 | tail -2
