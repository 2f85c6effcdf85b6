cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x -k "keep_masks or rand_coords" 2>&1 | tail -2
for i in 1 2 3; do
timeout 300 python bench.py --config headline+head --no-cpu-baseline --ablate torchmasks 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('torch masks ', d['ms_per_step'])"
timeout 300 python bench.py --config headline+head --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('device masks', d['ms_per_step'])"
done
timeout 300 python bench.py --config C3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3', d['ms_per_step'])"
