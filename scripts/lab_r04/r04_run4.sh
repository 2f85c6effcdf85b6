cd /root/repo
mkdir -p gpurun_out/r04
python scripts/r04_c5bias.py 28 40 56 > gpurun_out/r04/c5bias_hip.txt 2>&1
DEPTHG_LIB=$PWD/depthg_amd/lib/libdepthg_nocorr2.so python scripts/r04_c5bias.py 28 40 56 > gpurun_out/r04/c5bias_nocorr2.txt 2>&1
cat gpurun_out/r04/c5bias_hip.txt gpurun_out/r04/c5bias_nocorr2.txt
