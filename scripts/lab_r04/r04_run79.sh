cd /root/repo; mkdir -p gpurun_out/r04
for c in C2 C3 C4shard; do python3 bench.py --config $c > gpurun_out/r04/r04_bench_$c.json 2>/dev/null; python3 bench.py --config $c --eager --no-cpu-baseline > gpurun_out/r04/r04_bench_${c}_eager.json 2>/dev/null; done
for c in C2 C3 C4shard; do tail -1 gpurun_out/r04/r04_bench_$c.json | cut -c1-200; done
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -2
DG_POISON=1 timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -2
