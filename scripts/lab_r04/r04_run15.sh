cd /root/repo; mkdir -p gpurun_out/r04
DG_POISON=1 timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04/gputests_poison.txt 2>&1; tail -3 gpurun_out/r04/gputests_poison.txt
timeout 1500 python scripts/fuzz_parity.py 500 20000 > gpurun_out/r04/fuzz_500.txt 2>&1; tail -4 gpurun_out/r04/fuzz_500.txt
timeout 1500 python scripts/fuzz_parity.py 250 30000 edge > gpurun_out/r04/fuzz_edge_250.txt 2>&1; tail -4 gpurun_out/r04/fuzz_edge_250.txt
timeout 900 python scripts/fuzz_samplers.py 200 > gpurun_out/r04/fuzz_samplers.txt 2>&1; tail -4 gpurun_out/r04/fuzz_samplers.txt
