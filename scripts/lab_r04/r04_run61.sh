cd /root/repo
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "walks or config5" 2>&1 | tail -3
