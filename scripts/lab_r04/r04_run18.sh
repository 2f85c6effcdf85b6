cd /root/repo; mkdir -p gpurun_out/r04
TAG=def bash scripts/kstats.sh headline 2>&1 | tail -12
TAG=tgo bash scripts/kstats.sh headline --total-grad-only 2>&1 | tail -12
