cd /root/repo; mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04/gputests11.txt 2>&1
tail -4 gpurun_out/r04/gputests11.txt
python bench.py --no-cpu-baseline --exact-masks > gpurun_out/r04/bench11_xm.json 2> gpurun_out/r04/bench11_xm.err; tail -c 300 gpurun_out/r04/bench11_xm.json
scripts/kstats.sh headline --exact-masks 2>&1 | tail -12 | cut -c1-120
python bench.py --no-cpu-baseline > gpurun_out/r04/bench11.json 2>/dev/null; tail -c 400 gpurun_out/r04/bench11.json
