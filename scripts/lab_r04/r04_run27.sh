cd /root/repo; mkdir -p gpurun_out/r04
python scripts/r04_c5dbg.py 56 2>&1 | tail -1
timeout 600 python scripts/ab_corr.py hip prev > gpurun_out/r04/ab_persist1.txt 2>&1; tail -3 gpurun_out/r04/ab_persist1.txt
DG_BLOCKLOG=$PWD/gpurun_out/r04/blocklog3.bin DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_blog.so timeout 300 python bench.py --eager --steps 3 --warmup 2 --clock-warmup-s 1 --no-cpu-baseline > gpurun_out/r04/blog_bench.json 2> gpurun_out/r04/blog_bench.err
python scripts/blocklog.py gpurun_out/r04/blocklog3.bin | tee gpurun_out/r04/blocklog3.txt | tail -8
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
