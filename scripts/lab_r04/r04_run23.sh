cd /root/repo; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -q -x -k "headline or dense or exact or config5 or sweep or boundary" 2>&1 | tail -3
timeout 600 python scripts/ab_corr.py hip prev 2>&1 | tail -3
