cd /root/repo
mkdir -p gpurun_out/r04
CMP_HW=40 python scripts/cmp_ws.py hip nocorr2 2 dense > gpurun_out/r04/cmpws40.txt 2>&1
cat gpurun_out/r04/cmpws40.txt | cut -c1-400
