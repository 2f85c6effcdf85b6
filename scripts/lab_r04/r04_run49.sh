cd /root/repo; mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for c in headline C3; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'])"; done
python scripts/parity_table.py gpurun_out/r04/parity_sym.md > /dev/null 2>&1; sed -n 5,11p gpurun_out/r04/parity_sym.md
timeout 900 python scripts/fuzz_parity.py 300 60000 2>&1 | tail -3
