cd /root/repo; mkdir -p gpurun_out/r04
DG_BLOCKLOG=$PWD/gpurun_out/r04/blocklog.bin DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_blog.so timeout 300 python bench.py --eager --steps 3 --warmup 2 --clock-warmup-s 1 --no-cpu-baseline > gpurun_out/r04/blog_bench.json 2> gpurun_out/r04/blog_bench.err
python scripts/blocklog.py gpurun_out/r04/blocklog.bin | tee gpurun_out/r04/blocklog.txt
