cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x -k "exact" 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --exact-masks --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exact', d['ms_per_step'])"; done
timeout 300 python bench.py --config C5 --exact-masks --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 exact', d['ms_per_step'])"
TAG=xm2 bash scripts/kstats.sh headline --exact-masks 2>&1 | grep -E "cd_mask3|gs" | cut -c1-110
