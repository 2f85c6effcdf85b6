cd /root/repo; mkdir -p gpurun_out/r04
timeout 2400 python scripts/fuzz_parity.py 1000 80000 > gpurun_out/r04/fuzz_1000_head.txt 2>&1; tail -6 gpurun_out/r04/fuzz_1000_head.txt
timeout 1500 python scripts/fuzz_parity.py 300 90000 edge > gpurun_out/r04/fuzz_edge_300_head.txt 2>&1; tail -4 gpurun_out/r04/fuzz_edge_300_head.txt
