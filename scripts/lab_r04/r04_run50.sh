cd /root/repo; python scripts/lab_r04/probe_seed.py 60221 2>&1 | tail -24
