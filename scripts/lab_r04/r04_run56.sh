cd /root/repo; timeout 900 python scripts/lab_r04/fuzz_head_pair.py 60 2>&1 | tail -8
