cd /root/repo; mkdir -p gpurun_out/r04
python scripts/r04_c5bias.py 28 32 40 56 2>&1 | grep "^S=\|forward-only" | cut -c1-160 > gpurun_out/r04/c5bias_fixed.txt
cat gpurun_out/r04/c5bias_fixed.txt
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "exact_masks" > gpurun_out/r04/gputests_xm.txt 2>&1
grep -n "exact masks\|passed\|failed\|Error\|error" gpurun_out/r04/gputests_xm.txt | head -30
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04/gputests9.txt 2>&1
tail -5 gpurun_out/r04/gputests9.txt
python bench.py --no-cpu-baseline > gpurun_out/r04/bench9.json 2> gpurun_out/r04/bench9.err; tail -c 500 gpurun_out/r04/bench9.json
python bench.py --no-cpu-baseline --exact-masks > gpurun_out/r04/bench9_xm.json 2> gpurun_out/r04/bench9_xm.err; tail -c 700 gpurun_out/r04/bench9_xm.json; tail -3 gpurun_out/r04/bench9_xm.err
