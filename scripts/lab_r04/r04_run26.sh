cd /root/repo
python scripts/r04_c5dbg.py 56 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 600 python scripts/ab_corr.py hip prev 2>&1 | tail -3
