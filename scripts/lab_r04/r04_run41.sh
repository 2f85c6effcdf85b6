cd /root/repo
for i in 1 2 3; do
timeout 300 python bench.py --config headline+head --no-cpu-baseline --ablate torchmasks 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('torch masks ', d['ms_per_step'])"
timeout 300 python bench.py --config headline+head --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('device masks', d['ms_per_step'])"
done
