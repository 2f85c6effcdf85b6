cd /root/repo
for i in 1 2 3; do
timeout 300 python -m pytest tests -m gpu -q -x -k "config5_hires_56_vs" 2>&1 | tail -1
DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_oldred.so timeout 300 python -m pytest tests -m gpu -q -x -k "config5_hires_56_vs" 2>&1 | tail -1
done
