cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x -k "segment or step or head" 2>&1 | tail -3
