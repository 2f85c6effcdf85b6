set -x
cd /root/repo
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04/gputests3.txt 2>&1
tail -15 gpurun_out/r04/gputests3.txt
python scripts/parity_table.py gpurun_out/r04/parity3.md > gpurun_out/r04/parity3.log 2>&1
head -12 gpurun_out/r04/parity3.md
python bench.py --force-dist --no-cpu-baseline 2> gpurun_out/r04/bench_fd.err | grep "^{" > gpurun_out/r04/bench_fd.json
tail -c 1200 gpurun_out/r04/bench_fd.json
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench3.json 2> gpurun_out/r04/bench3.err
tail -c 1500 gpurun_out/r04/bench3.json
