cd /root/repo; mkdir -p gpurun_out/r04
DG_POISON=1 timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04/gputests_poison_final.txt 2>&1; tail -3 gpurun_out/r04/gputests_poison_final.txt
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04/gputests_final.txt 2>&1; tail -3 gpurun_out/r04/gputests_final.txt
timeout 1500 python scripts/fuzz_parity.py 500 40000 > gpurun_out/r04/fuzz_500_final.txt 2>&1; tail -4 gpurun_out/r04/fuzz_500_final.txt
timeout 1500 python scripts/fuzz_parity.py 250 50000 edge > gpurun_out/r04/fuzz_edge_250_final.txt 2>&1; tail -3 gpurun_out/r04/fuzz_edge_250_final.txt
timeout 900 python scripts/fuzz_samplers.py 200 > gpurun_out/r04/fuzz_samplers_final.txt 2>&1; tail -2 gpurun_out/r04/fuzz_samplers_final.txt
