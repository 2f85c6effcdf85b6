cd /root/repo; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -x -q -k "knn or topk" > gpurun_out/r04/gputests_knn.txt 2>&1; tail -3 gpurun_out/r04/gputests_knn.txt
python scripts/knn_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/knn_time.txt
