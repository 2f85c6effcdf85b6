cd /root/repo; mkdir -p gpurun_out/r04
timeout 600 python scripts/ab_corr.py hip prev > gpurun_out/r04/ab_prologue3.txt 2>&1; tail -4 gpurun_out/r04/ab_prologue3.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "headline or dense or exact or config5 or sweep or boundary" 2>&1 | tail -3
