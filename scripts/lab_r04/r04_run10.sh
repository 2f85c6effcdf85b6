cd /root/repo; mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "exact_masks" > gpurun_out/r04/gputests_xm2.txt 2>&1
grep -n "exact masks\|passed\|failed\|^E  " gpurun_out/r04/gputests_xm2.txt | cut -c1-300 | head -30
python bench.py --no-cpu-baseline --exact-masks > gpurun_out/r04/bench10_xm.json 2> gpurun_out/r04/bench10_xm.err; tail -c 600 gpurun_out/r04/bench10_xm.json
python bench.py --no-cpu-baseline --config C5 > gpurun_out/r04/bench10_C5.json 2>/dev/null; tail -c 500 gpurun_out/r04/bench10_C5.json
python bench.py --no-cpu-baseline --config C5 --exact-masks > gpurun_out/r04/bench10_C5_xm.json 2>/dev/null; tail -c 500 gpurun_out/r04/bench10_C5_xm.json
scripts/kstats.sh headline --exact-masks 2>&1 | tail -14
