cd /root/repo
timeout 900 python -m pytest tests/test_gpu_head.py tests/test_segmenter.py -q -m gpu 2>&1 | tail -2
for i in 1 2; do TAG=red scripts/kstats.sh headline+head 2>&1 | grep -E "k_head_reduce|ms_per" | cut -c1-250 | sed 's/"host_ms.*//'; done
