cd /root/repo; mkdir -p gpurun_out/r04
timeout 600 python scripts/ab_corr.py hip prev > gpurun_out/r04/ab_prologue2.txt 2>&1; tail -4 gpurun_out/r04/ab_prologue2.txt
DG_BLOCKLOG=$PWD/gpurun_out/r04/blocklog2.bin DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_blog.so timeout 300 python bench.py --eager --steps 3 --warmup 2 --clock-warmup-s 1 --no-cpu-baseline > gpurun_out/r04/blog_bench.json 2> gpurun_out/r04/blog_bench.err
python scripts/blocklog.py gpurun_out/r04/blocklog2.bin | tee gpurun_out/r04/blocklog2.txt | tail -7
timeout 900 python -m pytest tests -m gpu -q -x -k "headline or dense or exact or config5 or sweep or boundary" 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline | tail -1 | cut -c1-300
