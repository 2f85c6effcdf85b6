cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x -k "rand_coords or rng or graph or config3 or sweep" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --config C3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3', d['ms_per_step'])"; done
timeout 300 python bench.py --config C3 --eager --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 eager', d['ms_per_step'])"
