cd /root/repo; mkdir -p gpurun_out/r04
DG_POISON=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python scripts/fuzz_parity.py 200 70000 2>&1 | tail -2
