cd /root/repo; mkdir -p gpurun_out/r04
python scripts/r04_c5dbg.py 56 2>&1 | tail -1
python scripts/r04_c5dbg.py 28 2>&1 | tail -1
timeout 600 python scripts/ab_corr.py hip prev > gpurun_out/r04/ab_persist2.txt 2>&1; tail -4 gpurun_out/r04/ab_persist2.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
DG_BLOCKLOG=$PWD/gpurun_out/r04/blocklog4.bin DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_blog.so timeout 300 python bench.py --eager --steps 3 --warmup 2 --clock-warmup-s 1 --no-cpu-baseline > gpurun_out/r04/blog_bench.json 2> gpurun_out/r04/blog_bench.err
python scripts/blocklog.py gpurun_out/r04/blocklog4.bin | tee gpurun_out/r04/blocklog4.txt | tail -8
timeout 300 python bench.py --no-cpu-baseline | tail -1 | cut -c1-250
