cd /root/repo
mkdir -p gpurun_out/r04
python scripts/r04_c5bias.py 26 38 42 2>&1 | grep "^S=\|forward-only" > gpurun_out/r04/c5bias_sweep2.txt
cat gpurun_out/r04/c5bias_sweep2.txt | cut -c1-200
