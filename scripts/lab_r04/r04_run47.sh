cd /root/repo; mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04/gputests_final2.txt 2>&1; tail -3 gpurun_out/r04/gputests_final2.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for c in headline headline+head C3; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'], d['value'])"; done
