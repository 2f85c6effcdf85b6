cd /root/repo; mkdir -p gpurun_out/r04
timeout 900 python scripts/ab_corr.py hip p1 prev > gpurun_out/r04/ab_persist3.txt 2>&1; tail -4 gpurun_out/r04/ab_persist3.txt
timeout 900 python scripts/ab_corr.py prev p1 hip 2>&1 | tail -3
