set -x
cd /root/repo
mkdir -p gpurun_out/r04
./scripts/micro/mfma_shape_sustained > gpurun_out/r04/micro_shape.txt 2>&1
ZERO=1 ./scripts/micro/mfma_shape_sustained 40000 0.6 > gpurun_out/r04/micro_shape_zero.txt 2>&1
DG_STAMPS=$PWD/gpurun_out/r04/stamps.bin DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_stamps.so python bench.py --eager --steps 3 --warmup 2 --clock-warmup-s 1 --no-cpu-baseline > gpurun_out/r04/stamps_bench.json 2> gpurun_out/r04/stamps_bench.err
python scripts/stamps2.py gpurun_out/r04/stamps.bin > gpurun_out/r04/stamps.txt 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r04/bench0.json 2> gpurun_out/r04/bench0.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench0_driver.json 2>> gpurun_out/r04/bench0.err
cat gpurun_out/r04/micro_shape.txt gpurun_out/r04/micro_shape_zero.txt gpurun_out/r04/stamps.txt
tail -c 600 gpurun_out/r04/bench0.json; tail -c 400 gpurun_out/r04/bench0_driver.json
