cd /root/repo; mkdir -p gpurun_out/r04
python scripts/knn_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/knn_time3.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "knn or topk" 2>&1 | tail -2
