cd /root/repo; mkdir -p gpurun_out/r04
CMP_BASE=19719424 CMP_HW=40 python scripts/cmp_ws.py hip nocorr2 2 dense 2>&1 | grep "^group\|^tiles with\|differing el\|^    \[" | cut -c1-900 > gpurun_out/r04/cmpws40b.txt
cat gpurun_out/r04/cmpws40b.txt
