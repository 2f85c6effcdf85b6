cd /root/repo; mkdir -p gpurun_out/r04
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -3
DG_POISON=1 timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py 2>/dev/null | tail -1 | cut -c1-400
