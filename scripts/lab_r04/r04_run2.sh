set -x
cd /root/repo
mkdir -p gpurun_out/r04
python scripts/ab_corr.py r03 hip > gpurun_out/r04/ab1.txt 2>&1
DG_STAMPS=$PWD/gpurun_out/r04/stamps2.bin DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_stamps.so python bench.py --eager --steps 3 --warmup 2 --clock-warmup-s 1 --no-cpu-baseline > gpurun_out/r04/stamps2_bench.json 2> gpurun_out/r04/stamps2_bench.err
python scripts/stamps2.py gpurun_out/r04/stamps2.bin > gpurun_out/r04/stamps2.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputests2.txt 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r04/bench2.json 2> gpurun_out/r04/bench2.err
cat gpurun_out/r04/ab1.txt gpurun_out/r04/stamps2.txt; tail -15 gpurun_out/r04/gputests2.txt
tail -c 900 gpurun_out/r04/bench2.json
