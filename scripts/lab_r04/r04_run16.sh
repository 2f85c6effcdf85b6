cd /root/repo; mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests -m gpu -q -k "fps or multi_gpu_schedule" 2>&1 | tail -3
timeout 300 python scripts/fuzz_parity.py 1 20078 2>&1 | tail -3
timeout 300 python scripts/fuzz_parity.py 1 20256 2>&1 | tail -3
timeout 600 python scripts/ab_corr.py hip xnop > gpurun_out/r04/ab_xnop.txt 2>&1; tail -6 gpurun_out/r04/ab_xnop.txt
timeout 300 python bench.py --config C2 --no-cpu-baseline > gpurun_out/r04/bench_C2_pair.json 2>gpurun_out/r04/bench_C2_pair.err; cat gpurun_out/r04/bench_C2_pair.json
timeout 300 python bench.py --config C4shard --no-cpu-baseline > gpurun_out/r04/bench_C4_pair.json 2>gpurun_out/r04/bench_C4_pair.err; cat gpurun_out/r04/bench_C4_pair.json
