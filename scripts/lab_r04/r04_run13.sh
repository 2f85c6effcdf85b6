cd /root/repo; mkdir -p gpurun_out/r04
python scripts/knn_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/knn_time2.txt
timeout 900 python -m pytest tests -m gpu -x -q -s -k "knn or exact_masks or headline_width or correlated" > gpurun_out/r04/gputests13.txt 2>&1; grep -n "passed\|failed\|exact masks\|hl28\|correlated" gpurun_out/r04/gputests13.txt | cut -c1-260
scripts/kstats.sh headline --exact-masks 2>&1 | tail -11 | cut -c1-110
